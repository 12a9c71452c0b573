#!/usr/bin/env python3
"""Monte-Carlo of one traversal wavefront under the cost model the SQ counters give: the kernel is VALU-issue bound, so a
launch costs (wave-level VALU instructions issued), whatever the lanes do.  64 lanes; a ray is a geometric-length sequence of items, each an inner-node
step or (with probability 4.7 / 36.3) a triangle test; a whole-wave inner step costs CI instructions and advances the lanes that hold an inner node, a
leaf step CL, a refill CR.  Policies: the product's fixed rounds (k inner steps, then the leaf phase; refill once `refill_at` lanes are idle), greedy
(whichever step advances more lanes per instruction), leaf phase only once >= k lanes hold a leaf.  Output: instructions per ray relative to the product's
setting -- it reproduces the measured A/B sweep (profiles/r3/ab_scheduling_knobs_C3.txt) within about 1 % and shows there is nothing to gain from scheduling.
  python tools/round_schedule_model.py > profiles/r3/round_schedule_model.txt"""
import random
random.seed(1)
CI, CL, CR = 138.0, 78.0, 100.0
P_LEAF, STEPS_PER_RAY = 4.7 / 36.3, 36.3


def sim(policy, nrays=200000, refill_at=12, k=2):
    lanes = [None] * 64; cost = 0.0; pool = nrays; step_i = 0

    def advance(l):
        l[0] -= 1
        if l[0] <= 0:
            return None
        l[1] = 'L' if random.random() < P_LEAF else 'I'
        return l
    while True:
        idle = [i for i, l in enumerate(lanes) if l is None]
        if pool > 0 and len(idle) >= refill_at:
            for i in idle:
                if pool > 0:
                    lanes[i] = [max(1, int(random.expovariate(1 / STEPS_PER_RAY))), 'I']; pool -= 1
            cost += CR
        nI = sum(1 for l in lanes if l and l[1] == 'I'); nL = sum(1 for l in lanes if l and l[1] == 'L')
        if nI == 0 and nL == 0:
            if pool == 0:
                break
            continue
        if policy == 'fixed':
            phase = step_i % (k + 1); step_i += 1
            do = 'I' if phase < k else 'L'
            if (do == 'I' and nI == 0) or (do == 'L' and nL == 0):
                continue
        elif policy == 'greedy':
            do = 'I' if nI / CI >= nL / CL else 'L'
        else:
            do = 'L' if (nL >= k or nI == 0) else 'I'
        cost += CI if do == 'I' else CL
        for i, l in enumerate(lanes):
            if l and l[1] == do:
                lanes[i] = advance(l)
    return cost / nrays


base = sim('fixed', k=2)
print(f"product (2 inner steps per round, refill at 12 idle lanes): {base:.1f} wave-level VALU instructions per ray (measured: 118)")
for k in (1, 3, 4):
    print(f"{k} inner step(s) per round: x {sim('fixed', k=k) / base:.4f}")
print(f"greedy step choice: x {sim('greedy') / base:.4f}")
for k in (8, 12, 16, 20, 24, 32):
    print(f"leaf phase once >= {k} lanes hold a leaf: x {sim('thresh', k=k) / base:.4f}")
for r in (4, 8, 16, 24):
    print(f"refill at {r} idle lanes: x {sim('fixed', refill_at=r) / base:.4f}")

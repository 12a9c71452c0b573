import random
random.seed(1)
CI, CL, CR = 118.0, 63.0, 100.0
P_LEAF, STEPS_PER_RAY = 4.7 / 36.3, 36.3
def gen_ray():
    n = max(1, int(random.expovariate(1 / STEPS_PER_RAY)))
    return ['L' if random.random() < P_LEAF else 'I' for _ in range(n)]
def sim_park(policy, nrays=60000, refill_at=12, k=2, thresh=24, depth=1, extra=0.15, cpark=4.0, alpha=1.0, bl=12):
    lanes = [None]*64; parked = [0]*64; cost = 0.0; pool = nrays
    util_i = []; util_l = []
    def settle(i):
        l = lanes[i]
        while l and l[-1]=='L' and parked[i] < depth:
            l.pop(); parked[i] += 1
            if extra and random.random() < extra: l.insert(0, 'I')
        if l is not None and not l and parked[i]==0: lanes[i] = None
    while True:
        idle = [i for i,l in enumerate(lanes) if l is None]
        if pool > 0 and len(idle) >= refill_at:
            for i in idle:
                if pool > 0: lanes[i] = gen_ray(); parked[i]=0; pool -= 1; settle(i)
            cost += CR
        nI = sum(1 for l in lanes if l and l[-1]=='I')
        nP = sum(1 for p in parked if p)
        nB = sum(1 for i,l in enumerate(lanes) if l is not None and (not l or l[-1]=='L') and parked[i]>=depth)
        if nI == 0 and nP == 0:
            if pool == 0: break
            continue
        if policy == 'thresh': leaf = nP >= thresh or nI == 0
        elif policy == 'thresh_bl': leaf = nP >= thresh or nI == 0 or nB >= bl
        elif policy == 'greedy': leaf = nI == 0 or nP * CI >= alpha * nI * CL
        if leaf:
            cost += CL; util_l.append(nP)
            for i in range(64):
                if parked[i]:
                    parked[i] -= 1; settle(i)
            continue
        for _ in range(k):
            nI = sum(1 for l in lanes if l and l[-1]=='I')
            if nI == 0: break
            cost += CI + cpark; util_i.append(nI)
            for i,l in enumerate(lanes):
                if l and l[-1]=='I':
                    l.pop(); settle(i)
    return cost/nrays, sum(util_i)/len(util_i)/64, sum(util_l)/len(util_l)/64
def sim_fixed(nrays=60000, refill_at=12, k=2):
    lanes = [None]*64; cost = 0.0; pool = nrays; step_i = 0
    while True:
        idle = [i for i,l in enumerate(lanes) if l is None]
        if pool > 0 and len(idle) >= refill_at:
            for i in idle:
                if pool > 0: lanes[i] = gen_ray(); pool -= 1
            cost += CR
        nI = sum(1 for l in lanes if l and l[-1]=='I'); nL = sum(1 for l in lanes if l and l[-1]=='L')
        if nI == 0 and nL == 0:
            if pool == 0: break
            continue
        phase = step_i % (k+1); step_i += 1
        do = 'I' if phase < k else 'L'
        if (do=='I' and nI==0) or (do=='L' and nL==0): continue
        cost += CI if do=='I' else CL
        for i,l in enumerate(lanes):
            if l and l[-1]==do:
                l.pop()
                if not l: lanes[i] = None
    return cost/nrays
base = sim_fixed()
print("fixed", base)
for th in (16, 20, 24, 28):
    for bl in (8, 12, 16, 64):
        c,ui,ul = sim_park('thresh_bl', thresh=th, bl=bl)
        print("thresh %d bl %d k=2: x %.4f  inner util %.3f leaf util %.3f" % (th, bl, c/base, ui, ul))
for k in (1,2,3):
    c,ui,ul = sim_park('thresh_bl', thresh=24, bl=12, k=k)
    print("k=%d thresh 24 bl 12: x %.4f util %.3f %.3f" % (k, c/base, ui, ul))
for a in (0.6, 0.8, 1.0, 1.3):
    c,ui,ul = sim_park('greedy', alpha=a, k=1)
    print("greedy alpha %.1f k=1: x %.4f util %.3f %.3f" % (a, c/base, ui, ul))

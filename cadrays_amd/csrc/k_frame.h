// k_frame.h -- part of kernels.hip (ONE translation unit: included there inside namespace crh::(anonymous), after k_shade).  The frame kernel: a small batch --
// one Redraw() = +1 sample per pixel of one frame (reference AppViewer.cxx:1045-1047) -- in ONE launch instead of 1 + 2-3 per bounce.
// ================================================================== the frame kernel
// Why.  The wavefront schedule (kernels.hip: k_raygen -> [k_trace_nearest -> k_shade -> k_trace_any] x depth) ends every launch with the longest rays of a few
// wavefronts while the rest of the chip idles; a 512-sample batch hides that (25 ms per launch), a lone 1080p frame of 2 M paths is twenty such drains: 4.3 ms
// for 1.2 ms of work at the batch rate, and the application restarts with a lone frame on every camera move (AppViewer.cxx:979-984).
// What.  Every workgroup -- ONE of 16 wavefronts per compute unit -- is a small streaming path tracer over its own paths: it claims path slots in chunks from ONE
// global cursor, generates their camera rays, and keeps two rings in LDS -- rays to trace (nearest-hit and shadow rays mixed) and hit records to shade.  Thirteen
// wavefronts trace with the persistent engine of k_traversal.h (lanes refill from the ring as rays finish; when the ring is dry the idle lanes take over parts of
// the long rays' stacks, DON); the last three only shade (survivors and shadow rays go back into the ray ring) and claim the next chunk when the ray ring runs low.
// No stage ever waits for another workgroup, and a path's bounces follow each other without a launch boundary: the only drain is the end of the frame.
// Per path nothing changes: the same camera_ray / trace steps / shade_path in the same order, the shadow ray of bounce b is resolved BEFORE the ray of bounce
// b + 1 is traced (the lane that finishes the shadow ray goes on with the continuation itself), so the radiance record receives its terms in the order of the
// wavefront schedule and every frame is bit-identical to it.  A path lives at the position of its slot for its whole life (no compaction: a queue entry is the
// slot; the state is rewritten in place by the one lane that holds the path).
constexpr uint32_t kFrameEmpty = 0xFFFFFFFFu;   // a ring slot nobody has written yet
constexpr uint32_t kFrameAny   = 0x80000000u;   // ray-ring entry: the path's shadow ray (else its camera-path ray)
constexpr uint32_t kFrameBounceShift = 26;      // ray_d.w of a frame-kernel path: bounce << 26 | slot << 1 | inside-a-medium (slots < 2^25: small batches only)
constexpr int      kFrameMats  = 32;            // materials staged in LDS (4 KB)
#ifndef CRH_FRAME_BLOCK
#define CRH_FRAME_BLOCK 1024                    // threads per workgroup of the frame kernel: ONE workgroup of 16 wavefronts per compute unit shares the rings, so that the
#endif                                          // few rays of the late bounces gather in a few full wavefronts instead of trickling through all of them
#ifndef CRH_FRAME_MINWAVES
#define CRH_FRAME_MINWAVES 4                    // wavefronts per SIMD the register allocation must allow (4: 128 VGPRs, what the shading code needs)
#endif
#ifndef CRH_FRAME_DON
#define CRH_FRAME_DON 1                         // the frame engine's idle lanes take over parts of the long rays' stacks once the ray ring is dry
#endif
constexpr int      kFrameBlock = CRH_FRAME_BLOCK;
constexpr uint32_t frame_pow2(uint32_t x) { uint32_t p = 1; while (p < x) p <<= 1; return p; }
#ifndef CRH_FRAME_MISS_RING
#define CRH_FRAME_MISS_RING 1                   // round 6 experiment: rays that hit nothing wait in a ring of their own, so that a shading batch is either all misses (environment /
#endif                                          // implicit light, the path ends: a few hundred instructions) or all surface hits (the BSDF code) -- profiles/r6/lone_frame.md
#ifndef CRH_FRAME_RING_MUL
#define CRH_FRAME_RING_MUL 4
#endif
constexpr uint32_t kFrameRing  = frame_pow2((uint32_t)CRH_FRAME_RING_MUL * kFrameBlock);      // entries of each ring (a power of two) = the most paths a workgroup may have alive (every live path is in at most one ring)

// the three spin loops of the kernel end on two invariants (the live count never undercounts; a workgroup never holds more paths than ring entries).
// Should a future change break one, a loop that has spun ~0.5 s (legitimate waits are microseconds) raises the workgroup's error flag: the workgroup leaves, the
// code lands in the context's device error word (crh_sync / crh_get_stats / the read-backs report CRH_E_DEVICE "frame kernel gave up: ...") -- no hung GPU.
constexpr uint32_t kFrameSpinLimit = 1u << 23;
constexpr uint32_t kFrameErrTake = 1u, kFrameErrPush = 2u, kFrameErrIdle = 4u;
struct FrameArgs {
  const uint32_t* tile_ids; uint32_t n_tiles; const uint32_t* n_tiles_dev;      // as k_raygen's
  const uint32_t* seeds; uint32_t n_samples; int seed_per_tile;
  uint32_t seed_vals[16];     // seeds == nullptr: the frame seeds of the batch's <= 16 samples by value (a frame's tracing then reads nothing the host uploaded for it)
  uint32_t* ctl;              // [0] slot cursor, [1] workgroups finished (both zero at launch; the last workgroup to leave zeroes them again)
  uint32_t max_live;          // paths a workgroup keeps alive at most (<= kFrameRing)
  uint32_t gen_chunk;         // path slots a wavefront claims at a time (a multiple of 64)
  uint32_t low_water;         // a feeder wavefront claims the next chunk once fewer rays than this wait in the ring
  uint32_t n_feed;            // wavefronts of a workgroup that only shade and generate (the last ones)
  uint32_t starve;            // a feeder shades fewer than a wavefront's worth of hits only while fewer rays than this wait in the ring (the tracers are about to starve)
  uint32_t claim_step;        // tracer wavefront w takes rays from the ring only while >= w * claim_step wait there: scarce rays go to the first wavefronts
  uint32_t help;              // a tracer wavefront shades a batch itself once this many hit records wait (the feeders have fallen behind)
  uint32_t help_low;          // ... and prefers a FULL shading batch to tracing while fewer rays than this wait in the ring (0: a tracer traces whatever is there)
  uint32_t* err;              // the context's device error word (never touched in a healthy run)
};

// ---- rings: multi-producer / multi-consumer inside one workgroup.  head / tail are tickets; a slot holds kFrameEmpty until its producer has written it and is
// emptied again by its consumer, so neither side ever sees the other's half-done work.  A workgroup never has more entries than live paths <= kFrameRing.
__device__ __forceinline__ uint32_t ring_claim(uint32_t* head, uint32_t* tail, uint32_t want, uint32_t& base)      // whole wavefront; returns the number claimed
{
  uint32_t n = 0, b = 0;
  if (lane_id() == 0) {
    for (;;) {
      const uint32_t h = __hip_atomic_load(head, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP), t = __hip_atomic_load(tail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      const uint32_t avail = t - h;
      if (avail == 0u || avail > 0x7FFFFFFFu) break;
      n = min(avail, want);
      if (atomicCAS(head, h, h + n) == h) { b = h; break; }
      n = 0;
    }
  }
  base = __shfl(b, 0);
  return __shfl(n, 0);
}
__device__ __forceinline__ uint32_t ring_take(uint32_t* slots, uint32_t ticket, uint32_t* wg_err)
{
  uint32_t* p = slots + (ticket & (kFrameRing - 1u));
  uint32_t v, spins = 0;
  while ((v = atomicExch(p, kFrameEmpty)) == kFrameEmpty) {                                  // its producer holds the ticket and is about to write
    __builtin_amdgcn_s_sleep(1);
    if (++spins >= kFrameSpinLimit || __hip_atomic_load(wg_err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0u) { atomicOr(wg_err, kFrameErrTake); return 0u; }
  }
  return v;
}
__device__ __forceinline__ void ring_push(uint32_t* slots, uint32_t* tail, bool pred, uint32_t value, uint32_t* wg_err)      // whole wavefront
{
  const unsigned long long m = __ballot(pred);
  if (m == 0ull) return;
  const uint32_t lane = lane_id(), leader = (uint32_t)__ffsll((long long)m) - 1u;
  uint32_t base = 0;
  if (lane == leader) base = atomicAdd(tail, (uint32_t)__popcll(m));
  base = __shfl(base, leader);
  if (pred) {
    uint32_t* p = slots + ((base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))) & (kFrameRing - 1u));
    uint32_t spins = 0;
    while (atomicCAS(p, kFrameEmpty, value) != kFrameEmpty) {                                             // the consumer of the entry a lap ago is about to empty it
      __builtin_amdgcn_s_sleep(1);
      if (++spins >= kFrameSpinLimit || __hip_atomic_load(wg_err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0u) { atomicOr(wg_err, kFrameErrPush); break; }
    }
  }
}
__device__ __forceinline__ uint32_t ring_count(uint32_t* head, uint32_t* tail)
{
  const uint32_t d = __hip_atomic_load(tail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) - __hip_atomic_load(head, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  return d > 0x7FFFFFFFu ? 0u : d;
}
__device__ __forceinline__ void wave_sub(uint32_t* word, bool pred)          // *word -= number of lanes with pred, one LDS atomic per wavefront
{
  const unsigned long long m = __ballot(pred);
  if (m != 0ull && lane_id() == (uint32_t)__ffsll((long long)m) - 1u) atomicSub(word, (uint32_t)__popcll(m));
}

template <bool TWO>
__global__ __launch_bounds__(kFrameBlock, CRH_FRAME_MINWAVES) void k_frame(DScene S, DPaths P, FrameArgs A, DCounters* C)
{
  constexpr int kBlock = kFrameBlock;
  __shared__ uint32_t stk[kLdsStack * kBlock];
  __shared__ uint32_t s_bound[CRH_FRAME_DON ? kBlock : 1];
  __shared__ float4 s_mats[kFrameMats * 8];
  __shared__ uint32_t s_rq[kFrameRing], s_sq[kFrameRing];
#if CRH_FRAME_MISS_RING
  __shared__ uint32_t s_mq[kFrameRing];      // hit records that are misses
#endif
  __shared__ uint32_t s_ctl[12];     // [0] ray head, [1] ray tail, [2] shade head, [3] shade tail, [4] live paths, [5] the slot cursor has run out, [6] error, [8] miss head, [9] miss tail
  uint32_t* const rq_head = &s_ctl[0]; uint32_t* const rq_tail = &s_ctl[1]; uint32_t* const sq_head = &s_ctl[2]; uint32_t* const sq_tail = &s_ctl[3];
  uint32_t* const live = &s_ctl[4]; uint32_t* const cursor_out = &s_ctl[5]; uint32_t* const wg_err = &s_ctl[6];      // [6]: a spin loop of this workgroup gave up
  for (uint32_t i = threadIdx.x; i < kFrameRing; i += (uint32_t)kBlock) { s_rq[i] = kFrameEmpty; s_sq[i] = kFrameEmpty; }
#if CRH_FRAME_MISS_RING
  for (uint32_t i = threadIdx.x; i < kFrameRing; i += (uint32_t)kBlock) s_mq[i] = kFrameEmpty;
  uint32_t* const mq_head = &s_ctl[8]; uint32_t* const mq_tail = &s_ctl[9];
#endif
  if (threadIdx.x < 12u) s_ctl[threadIdx.x] = 0u;
  const bool mats_in_lds = S.n_mats <= (uint32_t)kFrameMats;
  if (mats_in_lds) for (uint32_t i = threadIdx.x; i < S.n_mats * 8u; i += (uint32_t)kBlock) s_mats[i] = S.mats[i];
  __syncthreads();

  const uint32_t n_tiles = A.n_tiles_dev ? *A.n_tiles_dev : A.n_tiles;
  const uint32_t total = n_tiles * S.tile_size * S.tile_size * A.n_samples;
  const uint32_t lane = lane_id();
  float4* const ray_o = P.ray_o[0]; float4* const ray_d = P.ray_d[0]; float4* const thr = P.thr[0];
  uint32_t n_near = 0, n_any = 0, n_shaded = 0;      // this wavefront's share of crh_stats
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
  const Top2 t2 = top2_of(S);

  // Who does what.  Shading and ray generation are what keeps the ray ring from running dry, and a wavefront can only turn to them when none of its lanes holds a
  // ray -- so the LAST n_feed wavefronts of the workgroup (the "feeders") never trace: they shade as soon as hits wait and claim the next chunk of slots when
  // the ray ring runs low; asleep they cost no issue slot.  The others trace.  (claim_step > 0: tracer w takes rays from the ring only while >= w * claim_step
  // of them wait, so that scarce rays gather in the first wavefronts -- measured, it does not pay: sixteen wavefronts sharing one ring gather them already.)
  const uint32_t wave = threadIdx.x >> 6, n_waves = (uint32_t)kBlock >> 6;
  const bool feeder = wave + A.n_feed >= n_waves;
  const uint32_t claim_min = feeder ? 0u : wave * A.claim_step;
  auto load_u = [](uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); };
  auto may_generate = [&]() { return load_u(cursor_out) == 0u && load_u(live) + A.gen_chunk <= A.max_live; };

  auto shade_some = [&](bool misses) {
    uint32_t base = 0;
#if CRH_FRAME_MISS_RING
    uint32_t* const q_slots = misses ? s_mq : s_sq;
    const uint32_t n = ring_claim(misses ? mq_head : sq_head, misses ? mq_tail : sq_tail, 64u, base);
#else
    uint32_t* const q_slots = s_sq; (void)misses;
    const uint32_t n = ring_claim(sq_head, sq_tail, 64u, base);
#endif
    if (n == 0u) return;
    const bool mine = lane < n;
#if CRH_FRAME_STATS
    const unsigned long long fs_t0 = (unsigned long long)clock64();
    if (lane == 0) { atomicAdd(&g_frame_stats[7], 1ull); atomicAdd(&g_frame_stats[8], (unsigned long long)n); }
#endif
    uint32_t pos = 0;
    if (mine) pos = ring_take(q_slots, base + lane, wg_err);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");          // the hit record and the path state were written by another wavefront of this workgroup
    bool cont = false, shadow = false;
    float4 n_o = zero4, n_d = zero4, n_t = zero4, s_o = zero4, s_d = zero4, s_c = zero4;
#if CRH_FRAME_STATS
    {   // lanes of the batch in its largest class: miss (environment) / material 0 / material 1 / any other material
      int cls = -1;
      if (mine) { const int hk = __float_as_int(P.hit[pos].w); cls = hk < 0 ? 0 : 1 + min(__float_as_int(S.shade[4u * (uint32_t)hk].w), 2); }
      uint32_t big = 0;
      for (int c = 0; c < 4; ++c) big = max(big, (uint32_t)__popcll(__ballot(cls == c)));
      if (lane == 0) atomicAdd(&g_frame_stats[15], (unsigned long long)big);
    }
#endif
    if (mine) {
      const float4 o4 = ray_o[pos], d4 = ray_d[pos], h = P.hit[pos];
      const uint32_t dw = __float_as_uint(d4.w), bounce = dw >> kFrameBounceShift;
      const bool first = bounce == 0u, last = bounce + 1u >= S.max_depth;
      const float4 t4 = first ? make_float4(1.0f, 1.0f, 1.0f, CRH_MAXFLOAT) : thr[pos];
      shade_path<false>(S, P, s_mats, mats_in_lds, bounce, first, last, o4, d4, t4, h, pos, cont, shadow, n_o, n_d, n_t, s_o, s_d, s_c, n_shaded);
      if (cont) {
        n_d.w = __uint_as_float(((bounce + 1u) << kFrameBounceShift) | (__float_as_uint(n_d.w) & ((1u << kFrameBounceShift) - 1u)));
        ray_o[pos] = n_o; ray_d[pos] = n_d; thr[pos] = n_t;
      }
      if (shadow) { s_c.w = cont ? 1.0f : 0.f; P.sh_o[pos] = s_o; P.sh_d[pos] = s_d; P.sh_c[pos] = s_c; }      // .w: a camera-path ray waits behind this shadow ray
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    ring_push(s_rq, rq_tail, shadow, pos | kFrameAny, wg_err);
    ring_push(s_rq, rq_tail, cont && !shadow, pos, wg_err);
    wave_sub(live, mine && !cont && !shadow);                        // the path ends here
    n_any += (uint32_t)__popcll(__ballot(shadow)); n_near += (uint32_t)__popcll(__ballot(cont));
#if CRH_FRAME_STATS
    if (lane == 0) atomicAdd(&g_frame_stats[21], (unsigned long long)clock64() - fs_t0);
#endif
  };

  auto generate = [&]() -> bool {
#if CRH_FRAME_STATS
    const unsigned long long fs_t0 = (unsigned long long)clock64();
#endif
    uint32_t ok = 0, cbase = 0;
    if (lane == 0 && load_u(cursor_out) == 0u) {
      const uint32_t before = atomicAdd(live, A.gen_chunk);             // counted BEFORE the slots are claimed: nobody sees "no paths, no slots" in between
      if (before + A.gen_chunk > A.max_live) atomicSub(live, A.gen_chunk);
      else {
        cbase = atomicAdd(A.ctl, A.gen_chunk);
        if (cbase >= total) {
          atomicSub(live, A.gen_chunk); __hip_atomic_store(cursor_out, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#if CRH_FRAME_STATS || CRH_FRAME_TIMELINE
          { const unsigned long long t = (unsigned long long)wall_clock64(); atomicMax(&g_frame_stats[27], ~t); atomicMax(&g_frame_stats[28], t); }
#endif
        }
        else ok = 1u;
      }
    }
    ok = __shfl(ok, 0); cbase = __shfl(cbase, 0);
    if (!ok) return false;
    uint32_t made = 0;
    for (uint32_t it = 0; it < A.gen_chunk; it += 64u) {
      const uint32_t pid = cbase + it + lane;
      bool valid = pid < total;
      uint32_t px = 0, py = 0, s = 0, local = 0;
      if (valid) { slot_to_pixel_sample(pid, A.n_samples, local, s); valid = slot_pixel(S, A.tile_ids, local, px, py); }
      if (valid) {
        v3 o, d; uint32_t rng;
        const uint32_t fseed = A.seeds ? (A.seed_per_tile ? A.seeds[local / (S.tile_size * S.tile_size)] : A.seeds[s]) : A.seed_vals[s & 15u];
        camera_ray(S, fseed, px, py, o, d, rng);
        ray_o[pid] = mk4(o, __uint_as_float(rng));
        ray_d[pid] = mk4(d, __uint_as_float(pid << 1));               // bounce 0, outside
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      ring_push(s_rq, rq_tail, valid, pid, wg_err);
      made += (uint32_t)__popcll(__ballot(valid));
    }
    if (lane == 0 && made != A.gen_chunk) atomicSub(live, A.gen_chunk - made);      // slots past the end or outside the image (edge tiles are partial)
    n_near += made;
#if CRH_FRAME_STATS
    if (lane == 0) atomicAdd(&g_frame_stats[22], (unsigned long long)clock64() - fs_t0);
#endif
    return true;
  };

  auto trace_some = [&]() {
    uint32_t nn = 0, nt = 0;
    trace_engine<false, false, TWO, CRH_FRAME_DON != 0, true, kFrameBlock>(S.nodes, S.tris, S.inst_leaf, S.root, S.guard_box, t2, nullptr, 0u, &stk[threadIdx.x],
      [&](uint32_t ticket, v3& o, v3& d, float& tmax, uint32_t& tag, bool& any_l) {
        tag = ring_take(s_rq, ticket, wg_err);
        any_l = (tag & kFrameAny) != 0u; tag &= ~kFrameAny;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        const float4 o4 = any_l ? P.sh_o[tag] : ray_o[tag], d4 = any_l ? P.sh_d[tag] : ray_d[tag];
        o = xyz(o4); d = xyz(d4); tmax = any_l ? o4.w : CRH_MAXFLOAT;
      },
      [&](bool fin, uint32_t tag, float4 h, bool f, bool any_l, v3& o, v3& d, float& tmax) -> bool {
        bool go_on = false;
        if (fin && any_l) {
          const float4 c = P.sh_c[tag];
          if (!f) {                                                   // unoccluded: the pending contribution joins the path's radiance
            float4 r = P.rad[tag];
            if (__float_as_uint(r.w) != P.stamp) r = zero4;
            r.x += c.x; r.y += c.y; r.z += c.z; r.w = __uint_as_float(P.stamp);
            P.rad[tag] = r;
          }
          go_on = c.w != 0.f;
          if (go_on) { const float4 o4 = ray_o[tag], d4 = ray_d[tag]; o = xyz(o4); d = xyz(d4); tmax = CRH_MAXFLOAT; }
        }
        const bool to_shade = fin && !any_l;
        if (to_shade) P.hit[tag] = h;
        if (__ballot(to_shade) != 0ull) {
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
#if CRH_FRAME_MISS_RING
          const bool miss = __float_as_int(h.w) < 0;
          ring_push(s_sq, sq_tail, to_shade && !miss, tag, wg_err);
          ring_push(s_mq, mq_tail, to_shade && miss, tag, wg_err);
#else
          ring_push(s_sq, sq_tail, to_shade, tag, wg_err);
#endif
        }
        wave_sub(live, fin && any_l && !go_on);                      // a shadow ray with nothing behind it: the path is done
        return go_on;
      }, nn, nt, CRH_FRAME_DON ? &s_bound[threadIdx.x & ~63u] : nullptr,
      [&](uint32_t want, uint32_t& base) -> uint32_t {
        if (ring_count(rq_head, rq_tail) < max(claim_min, 1u)) return 0u;      // scarce rays are left to the wavefronts before this one
        return ring_claim(rq_head, rq_tail, want, base);
      });
  };

#if CRH_FRAME_STATS
  const unsigned long long fs_start = (unsigned long long)clock64();
#endif
#if CRH_FRAME_STATS || CRH_FRAME_TIMELINE
  // the frame's time line on the 100 MHz wall clock (one frame per read of the statistics): [26] earliest wavefront start, [27] / [28] the first / last workgroup to
  // find the slot cursor exhausted, [29] / [30] the last / first wavefront to leave (minima are kept as maxima of the complement)
  if (lane == 0) atomicMax(&g_frame_stats[26], ~(unsigned long long)wall_clock64());
#endif
  uint32_t idle_spins = 0;
  for (;;) {
    const uint32_t nr = ring_count(rq_head, rq_tail);
#if CRH_FRAME_MISS_RING
    const uint32_t nh = ring_count(sq_head, sq_tail), nm = ring_count(mq_head, mq_tail), ns = nh + nm;
    // which ring a shading batch comes from: a full batch of surface hits first (they refill the ray ring), then a full batch of misses, else the fuller one
    const bool from_misses = nh >= 64u ? false : (nm >= 64u ? true : nm > nh);
    const bool full_batch = nh >= 64u || nm >= 64u;
#else
    const uint32_t ns = ring_count(sq_head, sq_tail);
    const bool from_misses = false, full_batch = ns >= 64u;
#endif
    // one call site per stage (each is a few thousand instructions, inlined): decide first, then act
    int act = 0;                                                     // 0 idle, 1 shade, 2 generate, 3 trace
    if (feeder) {
      if (full_batch) act = 1;
      else if (nr < A.low_water && may_generate()) act = 2;
      else if (ns != 0u && (nr < A.starve || nr == 0u)) act = 1;     // the tracers are about to starve: whatever waits is shaded now
    } else {
      if (nr < A.help_low && full_batch) act = 1;                    // a handful of rays would run at a few lanes of 64: 64 waiting hits make 64 new rays first
      else if (nr != 0u && nr >= claim_min) act = 3;
      else if (ns >= A.help || (A.n_feed == 0u && ns != 0u && !(nr == 0u && may_generate()))) act = 1;      // the feeders have fallen behind (or there are none)
      else if (nr == 0u && may_generate()) act = 2;
    }
    if (act == 1) { shade_some(from_misses); idle_spins = 0; continue; }
    if (act == 2) { if (generate()) { idle_spins = 0; continue; } }
    if (act == 3) { trace_some(); idle_spins = 0; continue; }
    // idle: other wavefronts of the workgroup still hold paths, or the frame is done
    if (load_u(cursor_out) != 0u && load_u(live) == 0u) break;
    if (++idle_spins >= kFrameSpinLimit / 4u) atomicOr(wg_err, kFrameErrIdle);
    if (load_u(wg_err) != 0u) break;
#if CRH_FRAME_STATS
    if (lane == 0) atomicAdd(&g_frame_stats[feeder ? 9 : 10], 1ull);
#endif
    __builtin_amdgcn_s_sleep(4);
  }
#if CRH_FRAME_STATS
  if (lane == 0) atomicAdd(&g_frame_stats[feeder ? 25 : 24], (unsigned long long)clock64() - fs_start);      // wave cycles from start to leaving the loop
#endif
#if CRH_FRAME_STATS || CRH_FRAME_TIMELINE
  if (lane == 0) { const unsigned long long t = (unsigned long long)wall_clock64(); atomicMax(&g_frame_stats[29], t); atomicMax(&g_frame_stats[30], ~t); }
#endif
  if (lane == 0) {
    if (n_near) atomicAdd(&C->rays_nearest, (unsigned long long)n_near);
    if (n_any) atomicAdd(&C->rays_any, (unsigned long long)n_any);
  }
  n_shaded = wave_sum(n_shaded);
  if (lane == 0 && n_shaded) atomicAdd(&C->shaded_hits, (unsigned long long)n_shaded);
  __syncthreads();
  if (threadIdx.x == 0 && s_ctl[6] != 0u && A.err) atomicOr(A.err, s_ctl[6]);
  if (threadIdx.x == 0 && atomicAdd(A.ctl + 1, 1u) == gridDim.x - 1u) { A.ctl[0] = 0u; A.ctl[1] = 0u; }      // the last workgroup leaves the control words as it found them
}

// k_traversal.h -- part of kernels.hip (ONE translation unit: included there inside namespace crh::(anonymous), in this order: k_common, k_traversal, k_packets, k_bsdf,
// k_lights_env, k_raygen, k_shade, k_accumulate).  The per-ray walk: trace_engine (persistent wavefronts, lane refill, LDS stacks), k_trace_nearest, k_trace_any.
// ================================================================== traversal
#ifndef CRH_TRACE_MINWAVES
#define CRH_TRACE_MINWAVES 0
#endif
#if CRH_TRACE_MINWAVES > 0
#define CRH_TRACE_BOUNDS __launch_bounds__(kBlock, CRH_TRACE_MINWAVES)
#else
#define CRH_TRACE_BOUNDS __launch_bounds__(kBlock)
#endif
#ifndef CRH_INNER_STEPS
#define CRH_INNER_STEPS 2      // 0: descend until every lane holds a leaf; k > 0: at most k inner steps per round
#endif
#ifndef CRH_POOL_DIV
#define CRH_POOL_DIV 2
#endif
#ifndef CRH_POSTPONE_LEAF
#define CRH_POSTPONE_LEAF 0    // round 6 experiment: 1 = the frame kernel's engine keeps descending with ONE triangle leaf postponed per lane (speculative while-while:
#endif                         // a lane that reaches a leaf does not drop out of the inner steps of its wavefront), 2 = every non-counting single-level engine
#ifndef CRH_HUNT_NOOP_CALLS
#define CRH_HUNT_NOOP_CALLS 0  // 1 re-creates the build of round 6 in which k_trace_rays<ANY, COUNT, TWO> answered wrong: calls of a lambda whose body is `if (false && ..)`
#endif                         // around the inner steps -- nothing at source level -- and hipcc 7.2's si-form-memory-clauses pass miscompiles that instantiation (DESIGN.md section 7,
                               // profiles/r6/hunt_anyhit_found.md, tests/hunts/anyhit_split_min.py).  The default source has no such call.
#ifndef CRH_FRAME_PARK
#define CRH_FRAME_PARK 0       // round 6 experiment: the frame engine runs its triangle step only when at least this many lanes hold a leaf, or none can descend (0: every turn)
#endif
#ifndef CRH_REFILL_IDLE
#define CRH_REFILL_IDLE 12     // refill a wavefront once this many of its 64 lanes have no ray
#endif
#ifndef CRH_FRAME_STATS
#define CRH_FRAME_STATS 0      // 1: an instrumented build (tools/ab_build.sh): the frame kernel's engine counts its turns, active rays, dry turns and step executions
#endif
#ifndef CRH_FRAME_TIMELINE
#define CRH_FRAME_TIMELINE 0   // 1: only the frame's time line (three wall-clock marks per workgroup: no measurable cost, unlike CRH_FRAME_STATS whose phase clocks slow a frame 3 x)
#endif
#if CRH_FRAME_STATS || CRH_FRAME_TIMELINE
__device__ unsigned long long g_frame_stats[32];      // tools/frame_stats.py names the entries
#endif
#ifndef CRH_EXP_EARLY_TRI
#define CRH_EXP_EARLY_TRI 0     // hunt scaffold only (tests/hunts/two_level_anyhit_counting.py)
#endif
#ifndef CRH_POOL_CHUNK
#define CRH_POOL_CHUNK 256     // measured: 64 -> 2257, 128 -> 2305, 256 -> 2308, 512 -> 2266, 1024 -> 2136 Mrays/s (big pools starve late bounces)
#endif
constexpr uint32_t kPoolChunk = CRH_POOL_CHUNK;   // rays a wavefront takes from the global cursor per atomic
constexpr uint32_t kDone = 0xFFFFFFFFu;

typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float inv_dir(float d)
{ return 1.0f / (crh_abs(d) < kDirEps ? (d < 0.f ? -kDirEps : kDirEps) : d); }

// Child order key (crh_spec.h #4).  Default: the entry distance's bits with the slot index in the two low mantissa bits -- unique
// 32-bit keys, unsigned order = near to far, ties by slot.  CRH_SPEC_ORDER_EXACT: the full bits with the slot appended (64-bit keys).
#if CRH_SPEC_ORDER_EXACT
typedef unsigned long long okey_t;
#define CRH_KEY_MISS 0xFFFFFFFFFFFFFFFFull
#define CRH_MAKE_KEY(BITS, K) ((((okey_t)((uint32_t)(BITS) & 0x7FFFFFFFu)) << 2) | (okey_t)(K))
#else
typedef uint32_t okey_t;
#define CRH_KEY_MISS 0xFFFFFFFFu
#define CRH_MAKE_KEY(BITS, K) (((uint32_t)(BITS) & 0x7FFFFFFCu) | (uint32_t)(K))
#endif
#define CRH_CE(a, b) { const okey_t lo_ = min(a, b); const okey_t hi_ = max(a, b); a = lo_; b = hi_; }
__device__ __forceinline__ uint32_t umed3(uint32_t a, uint32_t b, uint32_t c)
{ uint32_t r; asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }

// Persistent-wave traversal engine shared by every tracing kernel.
//
// A wavefront owns 64 ray slots.  It takes rays from a wave-local pool (kPoolChunk queue entries claimed with
// one atomic on the global cursor) and REFILLS idle lanes as soon as CRH_REFILL_IDLE of them have finished,
// instead of waiting for the slowest ray of a 64-ray packet.  Inside, the classic "while-while" shape keeps
// lanes convergent: (A) every lane descends inner nodes until it holds a leaf (or runs dry), (B) all lanes
// holding a leaf test its triangles together.  The per-ray sequence of node visits and triangle tests -- and
// therefore every result bit and counter -- is exactly the ordered stack traversal of DESIGN.md section 3.
//
// load(idx, o, d, tmax, tag) fetches queue entry idx; store(tag, hit, found) commits a finished ray.
// lds: this lane's column of the workgroup's stack (stride kBlock dwords), 16 entries; deeper entries
// spill to scratch (never touched on ordinary scenes).
// TWO: two-level scene -- traversal starts at the top-level root; an instance leaf re-expresses the ray in the object's
// space (direction not renormalised, so t keeps its meaning), a sentinel on the stack restores the world ray.
// position of the k-th (0-based) set bit of a wave mask
__device__ __forceinline__ uint32_t kth_bit(unsigned long long m, uint32_t k)
{
  uint32_t pos = 0, w32 = (uint32_t)m;
  const uint32_t c = (uint32_t)__popc(w32);
  if (k >= c) { k -= c; pos = 32u; w32 = (uint32_t)(m >> 32); }
#pragma unroll
  for (uint32_t w = 16u; w >= 1u; w >>= 1) {
    const uint32_t part = w32 & ((1u << w) - 1u), c2 = (uint32_t)__popc(part);
    if (k >= c2) { k -= c2; w32 >>= w; pos += w; } else w32 = part;
  }
  return pos;
}

constexpr uint32_t kNoLane = 64u;

// DON (work donation, small batches only).  A launch cannot end before its longest ray does -- ~400 node visits at ~1 us each on
// the benchmark scene, whatever the launch's size -- and a 1-spp frame is twenty such launches.  Once a
// wavefront's queue is exhausted, every lane that still walks hands the BOTTOM entry of its stack (the subtree it would visit
// last) to an idle lane of the wavefront, which walks it with a copy of the ray; helpers donate in turn, and a lane donates again
// as soon as another lane is idle, so a long ray fans out over the wavefront.
//   The lanes working on one ray form a list in traversal order: a helper is inserted right after its donor (everything the
// donor still has, and will push, comes before the donated subtree; everything donated earlier comes after it).  The sequential
// result is the earliest hit with the smallest t, i.e. a left-biased minimum over that list -- an associative fold.  A lane
// whose own part is walked and which has no successor left is finished; its predecessor absorbs its total in FRONT of what it
// has absorbed before (`chit`), and the head of the list stores fold(own, chit).  Hits are bit-identical to the sequential walk;
// only pruning differs (the parts do not see each other's `best`), i.e. the number of visits -- which is why the counting kernels
// never donate.
// Static / moved split of a two-level scene: the walk starts in the static world-space tree (`root`) and the top-level
// tree over the moved objects (`root2`) waits at the bottom of the stack -- pushed only when the ray touches the instances' bounds.
struct Top2 { uint32_t root2; float4 usph; const float4* isph; uint32_t n_isph; bool ask; };
__device__ __forceinline__ Top2 top2_of(const DScene& S, bool ask = true)
{ Top2 t; t.root2 = S.root2; t.usph = S.usph; t.isph = S.ibox; t.n_isph = S.n_ibox; t.ask = ask; return t; }

// Does the ray come near a moved object at all?  (spec: include/crh_math.h, crh_ray_near_sphere; the oracle's traverse() asks the same function.)
// The sphere around the bounds of ALL instances first, then -- when there are at most kMaxIBox of them -- the sphere of at least one.  ONE moved
// object (the gizmo drags one, ImRaytraceControls.cxx:64,88): the two spheres are the same numbers, one test.  Rays handed in through the API
// (any direction length) are not asked: they always walk the top level.
__device__ __forceinline__ bool touches_instances(const Top2& t2, v3 o, v3 d, float tmax)
{
  if (!t2.ask) return true;
  if (t2.n_isph != 1u && !crh_ray_near_sphere(o, d, tmax, t2.usph.x, t2.usph.y, t2.usph.z, t2.usph.w)) return false;
  if (t2.n_isph == 0u) return true;
  for (uint32_t i = 0; i < t2.n_isph; ++i) {
    const float4 sp = t2.isph[i];
    if (crh_ray_near_sphere(o, d, tmax, sp.x, sp.y, sp.z, sp.w)) return true;
  }
  return false;
}
__device__ __forceinline__ bool ray_touches_instances(const DScene& S, v3 o, v3 d, float tmax) { return touches_instances(top2_of(S), o, d, tmax); }

// FRM (the frame kernel, k_frame.h): the rays come from the workgroup's ray ring in LDS instead of a global queue -- `claim(want, base)` hands out up to `want`
// ring entries (0: the ring is empty RIGHT NOW; shading may refill it, so the walk goes on with donation and the engine returns only when no lane holds a ray and
// the ring has nothing); nearest-hit and any-hit rays travel mixed (load() says which, per lane: `any_l` takes the place of the template constant ANY); store() is
// called by the whole wavefront with a predicate (it appends to the shade ring with one LDS atomic per wavefront) and may hand the lane its path's NEXT ray -- the
// camera-path continuation that waited behind a shadow ray -- which the lane starts at once.
template <bool ANY, bool COUNT, bool TWO, bool DON, bool FRM = false, int BS = crh::kBlock, class Load, class Store, class Claim = int>
__device__ __forceinline__ void trace_engine(const float4* __restrict__ nodes, const float4* __restrict__ tris,
                                             const float4* __restrict__ inst, uint32_t root, float4 gbox, const Top2 t2,
                                             uint32_t* __restrict__ cursor, uint32_t n, uint32_t* lds,
                                             Load load, Store store, uint32_t& n_nodes, uint32_t& n_tris, uint32_t* bound = nullptr, Claim claim = Claim())
{
  constexpr int kBlock = BS;      // threads of the calling workgroup = row stride of its LDS stack and world-ray columns (the frame kernel runs wider workgroups)
  static_assert(!FRM || !COUNT, "the frame kernel's engine does not count visits");
  // bound (DON): one word per lane of this wavefront in LDS -- the smallest hit distance any part of the ray that STARTED in that
  // lane has found so far (float bits; distances are >= 0, so unsigned order = float order).  Every part prunes BOXES with it
  // (a box entered later than the bound holds nothing that can win the fold; equality is kept, ties are decided by order);
  // triangles are still accepted against the part's own `best`, which only knows what came earlier in traversal order.
  uint32_t ovf[kOvfStack];
  const uint32_t lane = lane_id();
  const unsigned long long lt_mask = (1ull << lane) - 1ull;
  // per-lane ray state
  bool have = false;
  bool any_l = ANY;                     // FRM: this lane's ray is an occlusion query (per lane); otherwise the template constant
#define CRH_ISANY (FRM ? any_l : ANY)
  uint32_t cur = kDone, tag = 0;
  // PL (CRH_POSTPONE_LEAF): a triangle leaf the lane has reached but not tested yet.  The lane goes on with the next stack entry and takes part in the inner steps
  // of its wavefront instead of waiting for the triangle phase; the leaf is tested there, BEFORE any leaf reached later (one at a time), so triangles are tested in
  // the walk's order.  Boxes visited in between are pruned with the `best` of before that test: more visits (never counted: not in COUNT instantiations), the
  // same hit -- a triangle in a box the up-to-date `best` would have culled lies strictly behind it (boxes are conservative), so it can neither win nor tie.
  // Single-level walks only: a postponed leaf of an object must not be tested with the world ray the lane holds again after popping the object's sentinel.
  constexpr bool PL = CRH_POSTPONE_LEAF != 0 && !COUNT && !TWO && (FRM || CRH_POSTPONE_LEAF > 1);
  uint32_t pleaf = kDone;               // kDone: none
#define CRH_LANE_DONE (cur == kDone && (!PL || pleaf == kDone))
  int sp = 0;
  v3 o = crh_mk3(0.f, 0.f, 0.f), d = o;
  // TWO: the world-space ray {origin, direction, reciprocal direction} of every lane waits in LDS while the lane walks inside an object
  // (restored, not recomputed, on leaving; nine registers fewer = one more wavefront per SIMD); column = lane, row stride kBlock
  __shared__ float s_world[TWO ? 9 * kBlock : 1];
  float* const wray = &s_world[TWO ? threadIdx.x : 0u];
  float ix = 0.f, iy = 0.f, iz = 0.f, gx = 0.f, gy = 0.f, gz = 0.f, best = 0.f;   // g: the slab test's guard band along each axis, in t
  auto save_world = [&]() {
    wray[0 * kBlock] = o.x; wray[1 * kBlock] = o.y; wray[2 * kBlock] = o.z; wray[3 * kBlock] = d.x; wray[4 * kBlock] = d.y; wray[5 * kBlock] = d.z;
    wray[6 * kBlock] = ix; wray[7 * kBlock] = iy; wray[8 * kBlock] = iz;
  };
  // guard band: entry / exit planes move apart by g = 2^-21 * |1/d| * R, R = |o - c|_1 + 3 h >= |origin - o| +
  // 256 * step of every node of the tree whose box has centre c and L1 half-extent h -- twice the worst rounding error of the
  // plane evaluation below, so a child box the exact ray touches is never culled
  auto set_guard = [&](float4 gb) {
    const float R = CRH_FMA(gb.w, 3.0f, (crh_abs(o.x - gb.x) + crh_abs(o.y - gb.y)) + crh_abs(o.z - gb.z)) * kSlabGuard;
    gx = crh_abs(ix) * R; gy = crh_abs(iy) * R; gz = crh_abs(iz) * R;
  };
  float4 hit = make_float4(0.f, 0.f, 0.f, __int_as_float(-1));
  bool found = false;
#if CRH_EXP_EARLY_TRI
  // hunt scaffold (tests/hunts/two_level_anyhit_counting.py): round 4's early triangle fetch, compiled in behind a run-time switch that is never on
  const bool early = gbox.w == -12345.0f;
  float4 pre_a = make_float4(0.f, 0.f, 0.f, 0.f);
#endif
  // donation state (DON): stack entries live in [sbase, sp); is_child: this lane walks a donated subtree, its total is absorbed by
  // its predecessor instead of stored; next: the lane that holds what comes right after this lane's part in traversal order;
  // chit / cfound: the folded totals of the successors absorbed so far (they come after everything this lane still walks)
  int sbase = 0; bool is_child = false, cfound = false; uint32_t next = kNoLane, head = 0;
  float4 chit = make_float4(0.f, 0.f, 0.f, __int_as_float(-1));
  // wave-uniform pool state.  Chunk per atomic: kPoolChunk for long queues (one cursor word sustains ~88 atomics/us); short
  // queues are cut finer so that every wavefront gets work -- a 75 K-ray launch in 256-ray chunks would keep 292 of the 5120
  // wavefronts busy with four 64-ray generations each (0.5 ms) instead of 1170 with one (CRH_POOL_DIV chunks per wavefront).
  const uint32_t per_wave = n / (gridDim.x * (uint32_t)(kBlock / 64) * (uint32_t)CRH_POOL_DIV);
  // DON, thin mode: a queue too short to give every wavefront 32 rays is dealt out in chunks of 8 ... 32 rays (about one per wavefront); a wavefront
  // takes ONE chunk at a time and all of its 64 lanes work on it (donation from the start), so the launch ends after ~the
  // average ray instead of after the longest one
  const bool thin = !FRM && DON && per_wave < 17u;                           // at most half of the lanes get a ray of their own
  const uint32_t chunk = thin ? max(8u, (2u * per_wave + 7u) & ~7u) : min(kPoolChunk, max(64u, (per_wave + 63u) & ~63u));
  uint32_t pool_next = 0, pool_end = 0;
  bool exhausted = false;

#if CRH_FRAME_STATS
  uint32_t fs_turns = 0, fs_have = 0, fs_dry = 0, fs_inner_w = 0, fs_tri_w = 0, fs_don = 0, fs_store_w = 0, fs_fin = 0;
  unsigned long long fc_refill = 0, fc_don = 0, fc_inner = 0, fc_leaf = 0, fc_retire = 0, fc_mark = 0;      // wave cycles by phase (s_memtime at the phase borders)
#define CRH_FS_MARK(ACC) if (FRM) { const unsigned long long now_ = (unsigned long long)clock64(); ACC += now_ - fc_mark; fc_mark = now_; }
  if (FRM) fc_mark = (unsigned long long)clock64();
#else
#define CRH_FS_MARK(ACC)
#endif
  for (;;) {
    // ------------------------------------------------------------------ refill idle lanes
    unsigned long long idle = __ballot(!have);
    if constexpr (FRM) {
      if ((uint32_t)__popcll(idle) >= (uint32_t)CRH_REFILL_IDLE) {
        uint32_t base = 0;
        const uint32_t take = claim((uint32_t)__popcll(idle), base);
        exhausted = take == 0u;                                  // "dry" for now: asked again on the next turn
        const bool mine = !have && (uint32_t)__popcll(idle & lt_mask) < take;
        if (mine) {
          float tmax;
          load(base + (uint32_t)__popcll(idle & lt_mask), o, d, tmax, tag, any_l);
          ix = inv_dir(d.x); iy = inv_dir(d.y); iz = inv_dir(d.z);
          set_guard(gbox);
          if (TWO) save_world();
          best = tmax; found = false; sp = 0; cur = root; have = true; pleaf = kDone;
          if (TWO && t2.root2 != kQEmpty && touches_instances(t2, o, d, tmax)) { lds[0] = t2.root2; sp = 1; }
          hit = make_float4(tmax, 0.f, 0.f, __int_as_float(-1));
          if (DON) { sbase = 0; is_child = false; cfound = false; next = kNoLane; head = lane; bound[lane] = __float_as_uint(tmax); }
        }
      }
    } else
    if (!exhausted && (thin ? idle == ~0ull : (uint32_t)__popcll(idle) >= (uint32_t)CRH_REFILL_IDLE)) {
      for (int round = 0; round < (thin ? 1 : 2) && idle != 0ull; ++round) {
        if (pool_next == pool_end) {
          uint32_t base = 0;
          if (lane == 0) base = atomicAdd(cursor, chunk);
          base = __shfl(base, 0);
          if (base >= n) { exhausted = true; break; }
          pool_next = base; pool_end = min(base + chunk, n);
        }
        const uint32_t avail = pool_end - pool_next;
        const uint32_t want = (uint32_t)__popcll(idle);
        const uint32_t take = min(avail, want);
        const uint32_t rank = (uint32_t)__popcll(idle & lt_mask);
        const bool mine = !have && ((idle >> lane) & 1ull) && rank < take;
        if (mine) {
          float tmax;
          load(pool_next + rank, o, d, tmax, tag);
          ix = inv_dir(d.x); iy = inv_dir(d.y); iz = inv_dir(d.z);
          set_guard(gbox);
          if (TWO) save_world();
          best = tmax; found = false; sp = 0; cur = root; have = true; pleaf = kDone;
          if (ANY && tmax < 0.f) cur = kDone;                      // second any-hit pass of a split scene: already occluded in the first (no visit, no test)
          if (TWO && t2.root2 != kQEmpty && touches_instances(t2, o, d, tmax)) { lds[0] = t2.root2; sp = 1; }
          hit = make_float4(tmax, 0.f, 0.f, __int_as_float(-1));
          if (DON) { sbase = 0; is_child = false; cfound = false; next = kNoLane; head = lane; bound[lane] = __float_as_uint(tmax); }
        }
        pool_next += take;
        idle &= ~__ballot(mine);
      }
    }
    if (__ballot(have) == 0ull) { if (exhausted) break; else continue; }
#if CRH_FRAME_STATS
    if (FRM) { ++fs_turns; fs_have += (uint32_t)__popcll(__ballot(have)); fs_dry += exhausted ? 1u : 0u; }
#endif
    CRH_FS_MARK(fc_refill)

    if (DON && (exhausted || thin)) {
      // ---------------------------------------------------------------- donation: bottom stack entries -> idle lanes
      const unsigned long long idle_m = __ballot(!have);
      if (idle_m != 0ull) {
#if CRH_FRAME_STATS
        if (FRM) ++fs_don;
#endif
        // a donor gives the FAR half of its stack (the entries below the middle, all of them in the LDS part); the helper
        // copies them into its own column and starts with the nearest of them
        bool can = have && cur != kDone && sp > sbase && sp <= kLdsStack && !(CRH_ISANY && found);
        if (TWO && can && lds[sbase * kBlock] == CRH_REF_SENTINEL) can = false;
        const int give_n = (sp - sbase + 1) >> 1;
        if (TWO && can)      // only world-level entries travel (the helper starts with the world ray): stop below an object sentinel
          for (int e = 0; e < give_n; ++e) if (lds[(sbase + e) * kBlock] == CRH_REF_SENTINEL) { can = false; break; }
        // lanes with a deep stack (much left to walk) are served first; within a class, by lane order
        const bool deep = can && sp - sbase >= 3;
        const unsigned long long deep_m = __ballot(deep), shal_m = __ballot(can && !deep);
        const uint32_t n_deep = (uint32_t)__popcll(deep_m);
        const uint32_t npair = min((uint32_t)__popcll(idle_m), n_deep + (uint32_t)__popcll(shal_m));
        if (npair != 0u) {
          const uint32_t rank_d = deep ? (uint32_t)__popcll(deep_m & lt_mask) : n_deep + (uint32_t)__popcll(shal_m & lt_mask);
          const uint32_t rank_i = (uint32_t)__popcll(idle_m & lt_mask);
          const bool gives = can && rank_d < npair, takes = !have && rank_i < npair;
          const uint32_t src = !takes ? lane : (rank_i < n_deep ? kth_bit(deep_m, rank_i) : kth_bit(shal_m, rank_i - n_deep));
          const int rcnt = __shfl(give_n, src), rsb = __shfl(sbase, src);
          if (takes) {
            const uint32_t* from = lds + ((int)src - (int)lane);                    // the donor's column of the same wavefront's stack
            for (int e = 0; e < rcnt; ++e) lds[e * kBlock] = from[(rsb + e) * kBlock];
          }
          // the helper walks in WORLD space (a donated entry sits below any object sentinel), with the donor's current bound
          float rox, roy, roz, rdx, rdy, rdz;
          if (TWO) {                                                                 // the donor's world ray: its column of s_world
            const float* from = wray + ((int)src - (int)lane);
            rox = from[0 * kBlock]; roy = from[1 * kBlock]; roz = from[2 * kBlock]; rdx = from[3 * kBlock]; rdy = from[4 * kBlock]; rdz = from[5 * kBlock];
          } else {
            rox = __shfl(o.x, src); roy = __shfl(o.y, src); roz = __shfl(o.z, src); rdx = __shfl(d.x, src); rdy = __shfl(d.y, src); rdz = __shfl(d.z, src);
          }
          const float rbest = __shfl(best, src);
          const int rany = FRM ? __shfl((int)any_l, src) : 0;
          const uint32_t rnext = __shfl(next, src), rhead = __shfl(head, src);
          if (gives) { next = kth_bit(idle_m, rank_d); sbase += give_n; }          // the helper comes right after the donor ...
          if (takes) {
            o = crh_mk3(rox, roy, roz); d = crh_mk3(rdx, rdy, rdz);
            ix = inv_dir(d.x); iy = inv_dir(d.y); iz = inv_dir(d.z);
            set_guard(gbox);
            if (TWO) save_world();
            best = rbest; found = false; sbase = 0; sp = rcnt - 1; cur = lds[sp * kBlock]; have = true; pleaf = kDone;      // the nearest of the entries received
            hit = make_float4(rbest, 0.f, 0.f, __int_as_float(-1));
            is_child = true; cfound = false; next = rnext; head = rhead;            // ... and before what the donor gave away earlier
            if (FRM) any_l = rany != 0;
          }
        }
      }
    }

    CRH_FS_MARK(fc_don)
    auto read_top = [&]() {
      --sp;
      if (__builtin_expect(sp < kLdsStack, 1)) cur = lds[sp * kBlock];
      else { cur = ovf[sp - kLdsStack]; asm volatile("" : "+v"(cur)); }
    };
    auto pop = [&]() {
      if ((CRH_ISANY && found) || sp == (DON ? sbase : 0)) { cur = kDone; return; }
      read_top();
      if (TWO && cur == CRH_REF_SENTINEL) {          // leaving an object: back to the world-space ray
        o = crh_mk3(wray[0 * kBlock], wray[1 * kBlock], wray[2 * kBlock]); d = crh_mk3(wray[3 * kBlock], wray[4 * kBlock], wray[5 * kBlock]);
        ix = wray[6 * kBlock]; iy = wray[7 * kBlock]; iz = wray[8 * kBlock];                         // the saved reciprocals are the bits inv_dir(d) would recompute
        set_guard(gbox);
        if (sp == (DON ? sbase : 0)) cur = kDone; else read_top();
      }
    };
    // one inner-node step of this lane: fetch the 48-B node (3 x dwordx4), slab-test and order its children, push / descend / pop
    auto inner_step = [&]() {
      const float4* np = nodes + (uint32_t)(CRH_NODE_DWORDS / 4) * cur;
      const float4 n0 = np[0], n1 = np[1], n2 = np[2];
      if (COUNT || (FRM && CRH_FRAME_STATS)) ++n_nodes;
#if CRH_FRAME_STATS
      if (FRM && lane == (uint32_t)__ffsll((long long)__ballot(true)) - 1u) ++fs_inner_w;      // once per execution by the wavefront (the lanes in it: n_nodes)
#endif
      // per-node grid: face t = fma(q, step * inv_d, fma(origin - o, inv_d, -+ guard)).  The difference is taken BEFORE the
      // multiplication: fma(origin, inv_d, -o * inv_d) cancels catastrophically when |o * inv_d| >> t (a ray grazing a box
      // corner was culled by 2e-5 of t); the guard is the per-ray constant above.
      const uint32_t ew = __float_as_uint(n0.w);
      // step * inv_d: the step is 2^k with k a signed byte of the node -- v_bfe_i32 + v_ldexp_f32, the same value as the product (a scaling by a
      // power of two is exact, and both round the same way where the result is subnormal)
      const float ax = __builtin_amdgcn_ldexpf(ix, (int)(ew << 24) >> 24), ay = __builtin_amdgcn_ldexpf(iy, (int)(ew << 16) >> 24),
                  az = __builtin_amdgcn_ldexpf(iz, (int)(ew << 8) >> 24);
      const float ddx = n0.x - o.x, ddy = n0.y - o.y, ddz = n0.z - o.z;
      // child references are implicit: slots < ni are the consecutive inner nodes from child_base, the others the leaves
      // with consecutive references from leaf_base (crh_bvh_format.h): ref(slot) = (slot < ni ? child_base : leaf_base - ni) + slot
      const uint32_t ni = (ew >> 24) & 7u, nch = ew >> 28;
      const uint32_t base_inner = __float_as_uint(n2.z), base_leaf = __float_as_uint(n2.w) - ni;
      // Along a negative direction the far plane is the one the ray enters through: swap the lo / hi byte words of that axis
      // once per node instead of a min + max per child and axis (fma is monotonic in q, so the values are the same bits).
      const bool sx = ix < 0.f, sy = iy < 0.f, sz = iz < 0.f;
      const uint32_t lx = __float_as_uint(sx ? n1.w : n1.x), ly = __float_as_uint(sy ? n2.x : n1.y), lz = __float_as_uint(sz ? n2.y : n1.z);
      const uint32_t hx = __float_as_uint(sx ? n1.x : n1.w), hy = __float_as_uint(sy ? n1.y : n2.x), hz = __float_as_uint(sz ? n1.z : n2.y);
      const float prune = DON ? fminf(best, __uint_as_float(bound[head])) : best;
      const f32x2 ax2 = {ax, ax}, ay2 = {ay, ay}, az2 = {az, az};
      const f32x2 bx2 = __builtin_elementwise_fma((f32x2){ddx, ddx}, (f32x2){ix, ix}, (f32x2){-gx, gx});      // {entry, exit} offsets
      const f32x2 by2 = __builtin_elementwise_fma((f32x2){ddy, ddy}, (f32x2){iy, iy}, (f32x2){-gy, gy});
      const f32x2 bz2 = __builtin_elementwise_fma((f32x2){ddz, ddz}, (f32x2){iz, iz}, (f32x2){-gz, gz});
      okey_t key[4];
      bool hitk[4];
#define CRH_QB(W, K) ((float)(((W) >> (8 * (K))) & 0xffu))      /* v_cvt_f32_ubyteK */
#define CRH_CHILD(K)                                                                                         \
      {                                                                                                     \
        const f32x2 tx = __builtin_elementwise_fma((f32x2){CRH_QB(lx, K), CRH_QB(hx, K)}, ax2, bx2);      /* v_pk_fma_f32: entry, exit */ \
        const f32x2 ty = __builtin_elementwise_fma((f32x2){CRH_QB(ly, K), CRH_QB(hy, K)}, ay2, by2);      \
        const f32x2 tz = __builtin_elementwise_fma((f32x2){CRH_QB(lz, K), CRH_QB(hz, K)}, az2, bz2);      \
        const float tmin = fmaxf(fmaxf(fmaxf(tx.x, ty.x), tz.x), 0.f);                                     \
        const float tmx  = fminf(fminf(fminf(tx.y, ty.y), tz.y), prune);                                   \
        /* tmin = max(.., 0) is >= 0 or -0 (never a negative number, and a NaN never passes the test below): clearing the sign bit IS max(bits, 0) */ \
        const uint32_t bits = __float_as_uint(tmin);                                                       \
        hitk[K] = (uint32_t)K < nch && tmin <= tmx;                                                        \
        key[K] = hitk[K] ? CRH_MAKE_KEY(bits, K) : CRH_KEY_MISS;                                           \
      }
      CRH_CHILD(0)
      CRH_CHILD(1)
      CRH_CHILD(2)
      CRH_CHILD(3)
#undef CRH_CHILD
#undef CRH_QB
      if (CRH_SPEC_ANYHIT_SLOT_ORDER && CRH_ISANY) {
        // crh_spec.h #8: an occlusion query needs no near-to-far order -- the hit children are taken in SLOT order (no sort, no keys): the
        // lowest hit slot continues, the others go onto the stack so that they pop in slot order; three unconditional stores, the ones of
        // children that were not hit (and of the one that continues) land in dead slots at / above the new top
        const uint32_t rs0 = (0u < ni ? base_inner : base_leaf) + 0u, rs1 = (1u < ni ? base_inner : base_leaf) + 1u,
                       rs2 = (2u < ni ? base_inner : base_leaf) + 2u, rs3 = (3u < ni ? base_inner : base_leaf) + 3u;
        const int f0 = hitk[0] ? 1 : 0, f1 = hitk[1] ? 1 : 0, f2 = hitk[2] ? 1 : 0, f3 = hitk[3] ? 1 : 0;
        const int a2 = f3, a1 = f3 + f2, nh = (a1 + f1) + f0;                   // hits in higher slots = position above the old top
        if (__builtin_expect(sp <= kLdsStack - 4, 1)) {
          uint32_t* top = lds + sp * kBlock;
          top[(f3 ? 0 : nh) * kBlock] = rs3; top[(f2 ? a2 : nh) * kBlock] = rs2; top[(f1 ? a1 : nh) * kBlock] = rs1;
          // the lowest hit slot sits at the top (position nh - 1) if it was stored at all: it continues in registers, its slot is dead
          sp += max(nh, 1) - 1;
        } else {
#define CRH_PUSH(V)                                                          \
          { const uint32_t v_ = (V);                                           \
            if (sp < kLdsStack) lds[sp * kBlock] = v_; else ovf[sp - kLdsStack] = v_; \
            ++sp; }
          const int first = f0 ? 0 : (f1 ? 1 : (f2 ? 2 : 3));
          if (f3 && first != 3) CRH_PUSH(rs3)
          if (f2 && first != 2) CRH_PUSH(rs2)
          if (f1 && first != 1) CRH_PUSH(rs1)
#undef CRH_PUSH
        }
        if (nh >= 1) cur = f0 ? rs0 : (f1 ? rs1 : (f2 ? rs2 : rs3)); else pop();
        return;
      }
#if CRH_SPEC_ORDER_EXACT
      CRH_CE(key[0], key[1]) CRH_CE(key[2], key[3]) CRH_CE(key[0], key[2]) CRH_CE(key[1], key[3]) CRH_CE(key[1], key[2])
#else
      {
        // four unique 32-bit keys in eight three-input operations (a five-comparator network is ten): sort three (v_min3 / v_med3 / v_max3), then the
        // fourth goes in -- the smallest and the largest of all are one min / max, the middle pair is {mid, med3(lo, hi, d)} in order
        const uint32_t lo = min(min(key[0], key[1]), key[2]), hi = max(max(key[0], key[1]), key[2]);
        const uint32_t mid = umed3(key[0], key[1], key[2]), d = key[3];
        const uint32_t m = umed3(lo, hi, d);
        key[0] = min(lo, d); key[3] = max(hi, d); key[1] = min(mid, m); key[2] = max(mid, m);
      }
#endif
      // The sorted keys put the nh hit children first (miss keys have bit 31 set).  Far .. near go onto the
      // stack, the nearest continues in registers.  Common case (room for three entries in the LDS part of
      // the stack): three UNCONDITIONAL stores -- hit children land at sp + (nh-1-j), the others in the dead
      // slots above the new top -- so the step has no per-child branches.
#define CRH_REF(KEY) ((((uint32_t)((KEY) & 3u) < ni) ? base_inner : base_leaf) + (uint32_t)((KEY) & 3u))
      const uint32_t r0 = CRH_REF(key[0]), r1 = CRH_REF(key[1]), r2 = CRH_REF(key[2]), r3 = CRH_REF(key[3]);
#undef CRH_REF
#if CRH_SPEC_ORDER_EXACT
      const int nh = 4 - (((key[0] == CRH_KEY_MISS) + (key[1] == CRH_KEY_MISS)) + ((key[2] == CRH_KEY_MISS) + (key[3] == CRH_KEY_MISS)));
#else
      const int nh = 4 + ((((int)key[0] >> 31) + ((int)key[1] >> 31)) + (((int)key[2] >> 31) + ((int)key[3] >> 31)));
#endif
      if (__builtin_expect(sp <= kLdsStack - 3, 1)) {
        uint32_t* top = lds + sp * kBlock;
        const int p1 = max(nh, 2) - 2, p2 = (nh == 3) ? 0 : 1, p3 = (nh == 4) ? 0 : 2;
        top[p3 * kBlock] = r3; top[p2 * kBlock] = r2; top[p1 * kBlock] = r1;
        sp += max(nh, 1) - 1;
      } else {
#define CRH_PUSH(V)                                                          \
        { const uint32_t v_ = (V);                                           \
          if (sp < kLdsStack) lds[sp * kBlock] = v_; else ovf[sp - kLdsStack] = v_; \
          ++sp; }
        if (nh == 4) CRH_PUSH(r3)
        if (nh >= 3) CRH_PUSH(r2)
        if (nh >= 2) CRH_PUSH(r1)
#undef CRH_PUSH
      }
      if (nh >= 1) cur = r0; else pop();
    };
    // one ray/triangle test of this lane against leaf-order triangle `ti`
    auto tri_step = [&](uint32_t ti) {
      const float4* tp = tris + kTriStride * ti;
#if CRH_EXP_EARLY_TRI
      const float4 a = early ? pre_a : tp[0], b = tp[1], c = tp[2];
#else
      const float4 a = tp[0], b = tp[1], c = tp[2];
#endif
      if (COUNT || (FRM && CRH_FRAME_STATS)) ++n_tris;
#if CRH_FRAME_STATS
      if (FRM && lane == (uint32_t)__ffsll((long long)__ballot(true)) - 1u) ++fs_tri_w;
#endif
      // record = {v0 | n.x}, {e0 = v1 - v0 | n.y}, {e1 = v0 - v2 | n.z}: the two edges and n = e1 x e0 are evaluated ONCE per triangle on the host
      // with the inline arithmetic this function used to apply per test (crh_sub3 / crh_cross3, same bits) -- 15 VALU instructions per test
      // fewer in a kernel that runs at the VALU issue limit
      const v3 v0 = xyz(a), e0 = xyz(b), e1 = xyz(c);
      const v3 nrm = crh_mk3(a.w, b.w, c.w);
      const v3 to = crh_sub3(v0, o);
      const float inv = 1.0f / crh_dot3(nrm, d);
      const v3 vc = crh_cross3(d, to);
      const float tt = crh_dot3(nrm, to) * inv;
      const float uu = crh_dot3(vc, e1) * inv;
      const float vv = crh_dot3(vc, e0) * inv;
      if (tt >= 0.f && uu >= 0.f && vv >= 0.f && (uu + vv) <= 1.0f && tt < best) {
        best = tt; found = true;
        if (DON) atomicMin(&bound[head], CRH_ISANY ? 0u : __float_as_uint(tt));      // any-hit: one occluder ends every part's walk
        hit = make_float4(tt, uu, vv, __int_as_float((int)ti));
      }
    };
    // ------------------------------------------------------------------ (A) inner nodes until a leaf is in hand
#if CRH_INNER_STEPS > 0
    // (the frame kernel's rays -- every bounce mixed in one wavefront -- hold leaves less often: three inner steps per turn there, lone frame 3.15 -> 3.05 ms on C3)
    constexpr int kInnerSteps = FRM ? CRH_INNER_STEPS + 1 : CRH_INNER_STEPS;
#if CRH_POSTPONE_LEAF || CRH_HUNT_NOOP_CALLS
    // PL: a triangle leaf in hand moves aside (if none waits yet) and the lane descends on
    auto postpone = [&]() { if (PL && have && (cur & kQLeafBit) && cur != kDone && pleaf == kDone) { pleaf = cur; pop(); } };
    postpone();
#pragma unroll 1
    for (int step_ = 0; step_ < kInnerSteps && have && !(cur & kQLeafBit); ++step_) { inner_step(); postpone(); }
#else
#pragma unroll 1
    for (int step_ = 0; step_ < kInnerSteps && have && !(cur & kQLeafBit); ++step_) inner_step();
#endif
#else
    while (have && !(cur & kQLeafBit)) inner_step();
#endif
#if CRH_EXP_EARLY_TRI
    if (early && have && (cur & kQLeafBit) && cur != kDone && (cur & 0xF0000000u) != CRH_REF_INSTANCE_TAG) pre_a = tris[kTriStride * (cur & 0x0FFFFFFFu)];
#endif
    CRH_FS_MARK(fc_inner)
    // ------------------------------------------------------------------ (B) the leaf in hand
    if (TWO && have && (cur & 0xF0000000u) == CRH_REF_INSTANCE_TAG && cur < CRH_REF_SENTINEL) {
      // top-level leaf: enter the object (ray := M^-1 ray), mark the stack, continue at the object's root
      const float4* ip = inst + 8u * (cur & 0x0FFFFFFFu);
      const float4 i0 = ip[0], i1 = ip[1], i2 = ip[2], meta = ip[6];
      const float m[12] = {i0.x, i0.y, i0.z, i0.w, i1.x, i1.y, i1.z, i1.w, i2.x, i2.y, i2.z, i2.w};
      // the lane holds the world ray here (instances do not nest), and s_world keeps it for the way out
      o = crh_xform_point(m, o); d = crh_xform_vector(m, d);
      // an instance that is only translated (inverse 3x3 == identity exactly, flagged by the host) leaves |d| and its signs
      // unchanged, so the reciprocals are the world ray's: three IEEE divisions saved on the common "placed, not rotated" part
      if (__float_as_uint(meta.z) == 0u) { ix = inv_dir(d.x); iy = inv_dir(d.y); iz = inv_dir(d.z); }
      set_guard(ip[7]);                                          // the object's own box: {centre, L1 half-extent}
      const uint32_t mark = CRH_REF_SENTINEL;
      if (sp < kLdsStack) lds[sp * kBlock] = mark; else ovf[sp - kLdsStack] = mark;
      ++sp;
      cur = __float_as_uint(meta.x);
    }
#if CRH_FRAME_PARK > 0
    else if (FRM && !TWO && (uint32_t)__popcll(__ballot(have && (cur & kQLeafBit) && cur != kDone)) < (uint32_t)CRH_FRAME_PARK && __ballot(have && !(cur & kQLeafBit)) != 0ull) {
      // parked: too few leaves in hand and somebody can still descend -- the leaves wait for the next turn
    }
#endif
#if CRH_POSTPONE_LEAF
    else if (PL) {
      if (have && pleaf != kDone) {
        tri_step(pleaf & 0x0FFFFFFFu);                           // the postponed leaf first: triangles in the walk's order
        pleaf = kDone;
        if (CRH_ISANY && found) cur = kDone;                     // an occluder ends the walk (what pop() does in the plain form)
      }
    }
#endif
    else if (have && (cur & kQLeafBit) && cur != kDone) {
      tri_step(cur & 0x0FFFFFFFu);                               // one triangle per leaf (crh_bvh_format.h)
      pop();
    }

    CRH_FS_MARK(fc_leaf)
    // ------------------------------------------------------------------ (C) retire finished rays
#if CRH_FRAME_STATS
    if (FRM) { const unsigned long long fm_ = __ballot(have && CRH_LANE_DONE); if (fm_ != 0ull) { ++fs_store_w; fs_fin += (uint32_t)__popcll(fm_); } }
#endif
    if (DON) {
      // own part walked and no successor left: fold what was absorbed behind the own hit (left-biased minimum: a later part
      // wins only with a strictly smaller t); helpers then wait to be absorbed by their predecessor, the head stores
      const bool finished = have && CRH_LANE_DONE && next == kNoLane;
      if (finished && cfound) { if (CRH_ISANY || !found || chit.x < hit.x) { hit = chit; found = true; } cfound = false; }
      const unsigned long long fin_children = __ballot(finished && is_child);
      if (fin_children != 0ull) {
        const bool takes = have && next != kNoLane && ((fin_children >> next) & 1ull);
        const uint32_t from = takes ? next : lane;
        const float hx = __shfl(hit.x, from), hy = __shfl(hit.y, from), hz = __shfl(hit.z, from), hw = __shfl(hit.w, from);
        const int hf = __shfl((int)found, from);
        if (takes) {
          // the successor's total goes IN FRONT of what this lane absorbed before (it was donated later = it comes earlier)
          if (hf && (CRH_ISANY || !cfound || !(chit.x < hx))) { chit = make_float4(hx, hy, hz, hw); cfound = true; }
          next = kNoLane;
        }
        if ((fin_children >> lane) & 1ull) { have = false; is_child = false; }           // absorbed: the lane is free again
      }
      const bool fin = have && !is_child && CRH_LANE_DONE && next == kNoLane;
      if (fin && cfound) { if (CRH_ISANY || !found || chit.x < hit.x) { hit = chit; found = true; } cfound = false; }
      if constexpr (FRM) {
        float tmax = CRH_MAXFLOAT;
        const bool go_on = store(fin, tag, hit, found, any_l, o, d, tmax);      // the whole wavefront calls; true: the lane's path continues with the ray in o, d
        if (fin) {
          have = go_on;
          if (go_on) {
            any_l = false;
            ix = inv_dir(d.x); iy = inv_dir(d.y); iz = inv_dir(d.z);
            set_guard(gbox);
            if (TWO) save_world();
            best = tmax; found = false; sp = 0; cur = root; pleaf = kDone;
            if (TWO && t2.root2 != kQEmpty && touches_instances(t2, o, d, tmax)) { lds[0] = t2.root2; sp = 1; }
            hit = make_float4(tmax, 0.f, 0.f, __int_as_float(-1));
            sbase = 0; is_child = false; cfound = false; next = kNoLane; head = lane; bound[lane] = __float_as_uint(tmax);
          }
        }
      } else if (fin) { store(tag, hit, found); have = false; }
    } else if constexpr (FRM) {
      const bool fin = have && CRH_LANE_DONE;
      float tmax = CRH_MAXFLOAT;
      const bool go_on = store(fin, tag, hit, found, any_l, o, d, tmax);        // the whole wavefront calls; true: the lane's path continues with the ray in o, d
      if (fin) {
        have = go_on;
        if (go_on) {
          any_l = false;
          ix = inv_dir(d.x); iy = inv_dir(d.y); iz = inv_dir(d.z);
          set_guard(gbox);
          if (TWO) save_world();
          best = tmax; found = false; sp = 0; cur = root; pleaf = kDone;
          if (TWO && t2.root2 != kQEmpty && touches_instances(t2, o, d, tmax)) { lds[0] = t2.root2; sp = 1; }
          hit = make_float4(tmax, 0.f, 0.f, __int_as_float(-1));
        }
      }
    } else { if (have && CRH_LANE_DONE) { store(tag, hit, found); have = false; } }
    CRH_FS_MARK(fc_retire)
  }
#undef CRH_ISANY
#undef CRH_LANE_DONE
#if CRH_FRAME_STATS
  if (FRM) {
    const uint32_t li = wave_sum(n_nodes), lt = wave_sum(n_tris);
    if (lane == 0) {
      atomicAdd(&g_frame_stats[0], (unsigned long long)fs_turns); atomicAdd(&g_frame_stats[1], (unsigned long long)fs_have); atomicAdd(&g_frame_stats[2], (unsigned long long)fs_dry);
      atomicAdd(&g_frame_stats[3], (unsigned long long)li); atomicAdd(&g_frame_stats[4], (unsigned long long)lt); atomicAdd(&g_frame_stats[5], (unsigned long long)fs_don);
      atomicAdd(&g_frame_stats[6], 1ull);
      atomicAdd(&g_frame_stats[13], (unsigned long long)fs_store_w); atomicAdd(&g_frame_stats[14], (unsigned long long)fs_fin);
      atomicAdd(&g_frame_stats[16], fc_refill); atomicAdd(&g_frame_stats[17], fc_don); atomicAdd(&g_frame_stats[18], fc_inner); atomicAdd(&g_frame_stats[19], fc_leaf);
      atomicAdd(&g_frame_stats[20], fc_retire);
    }
    const uint32_t wi = wave_sum(fs_inner_w), wt = wave_sum(fs_tri_w);
    if (lane == 0) { atomicAdd(&g_frame_stats[11], (unsigned long long)wi); atomicAdd(&g_frame_stats[12], (unsigned long long)wt); }
  }
#endif
#undef CRH_FS_MARK
}

// P2: the SECOND pass of a split scene (static tree + moved objects).  The first pass is the single-level instantiation over the
// whole queue, walking the static tree only; the kernels that produced the rays listed the ones that touch a moved object in `q` of this launch
// (DQueues::q2); this pass walks the top-level tree for them, from the distance the first pass found, and overwrites the hit when it finds a nearer
// one.  Same visits, same hits as one walk "static tree, then top level" -- the rays that never come near a moved object run the plain kernel.
// FB: the fall-back pass behind k_trace_packets (below) -- the few camera rays whose packet walk met two triangles at EXACTLY the same distance are walked
// again, one by one, in the order the spec prescribes; its own cursor word, no launch prologue (the packet kernel has done that).
template <bool COUNT, bool TWO, bool DON, bool P2 = false, bool FB = false>
__global__ CRH_TRACE_BOUNDS void k_trace_nearest(DScene S, DPaths P, int cur, const uint32_t* __restrict__ q,
                                                  const uint32_t* __restrict__ count, uint32_t* __restrict__ cursors,
                                                  uint32_t* zero_a, uint32_t* zero_b, uint32_t* zero_c, uint32_t* zero_d, DCounters* C)
{
  __shared__ uint32_t stk[kLdsStack * kBlock];
  __shared__ uint32_t s_bound[DON ? kBlock : 1];
  const uint32_t n = *count;
  if (!P2 && !FB && blockIdx.x == 0 && threadIdx.x == 0) {
    *zero_a = 0u; *zero_b = 0u; *zero_c = 0u; *zero_d = 0u;      // the other queue's count, the shadow count, the second-pass counts shading will fill
    cursors[1] = 0u; cursors[2] = 0u; cursors[5] = 0u;            // shade / any-hit / second-pass any-hit cursors for the launches that follow
    atomicAdd(&C->rays_nearest, (unsigned long long)n);
  }
  uint32_t nn = 0, nt = 0;
  const float4* __restrict__ ray_o = P.ray_o[cur]; const float4* __restrict__ ray_d = P.ray_d[cur];
  Top2 t2 = top2_of(S); if (P2) t2.root2 = kQEmpty;             // the second pass starts AT the top level
  trace_engine<false, COUNT, TWO, DON>(S.nodes, S.tris, S.inst_leaf, P2 ? S.root2 : S.root, S.guard_box, t2, cursors + (FB ? 8 : (P2 ? 4 : 0)), n, &stk[threadIdx.x],
    [&](uint32_t idx, v3& o, v3& d, float& tmax, uint32_t& tag) {
      tag = q[idx];
      const float4 o4 = ld_stream(&ray_o[tag]), d4 = ld_stream(&ray_d[tag]);      // .w lanes carry the path's rng state / slot + flags, not ray data
      o = xyz(o4); d = xyz(d4); tmax = P2 ? P.hit[tag].x : CRH_MAXFLOAT;           // first-pass distance (its miss record holds the ray's tmax)
    },
    [&](uint32_t tag, float4 h, bool f) { if (!P2 || f) st_stream(&P.hit[tag], h); }, nn, nt, DON ? &s_bound[threadIdx.x & ~63u] : nullptr);
  if (COUNT) {
    nn = wave_sum(nn); nt = wave_sum(nt);
    if (lane_id() == 0) { atomicAdd(&C->nodes_nearest, (unsigned long long)nn); atomicAdd(&C->tris_nearest, (unsigned long long)nt); }
  }
}

// P2 / S.split: shadow rays of a split scene.  First pass (single-level instantiation, static tree): a ray the producer flagged (sh_d.w != 0: it touches a
// moved object) does not add its contribution yet -- if the static tree occludes it, its pending contribution is zeroed instead; the second pass walks
// the top level for the flagged rays and adds what is left when that does not occlude either.
template <bool COUNT, bool TWO, bool DON, bool P2 = false>
__global__ CRH_TRACE_BOUNDS void k_trace_any(DScene S, DPaths P, const uint32_t* __restrict__ q,
                                              const uint32_t* __restrict__ count, uint32_t* __restrict__ cursors, DCounters* C)
{
  __shared__ uint32_t stk[kLdsStack * kBlock];
  __shared__ uint32_t s_bound[DON ? kBlock : 1];
  const uint32_t n = *count;
  if (!P2 && blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&C->rays_any, (unsigned long long)n);
  uint32_t nn = 0, nt = 0;
  Top2 t2 = top2_of(S); if (P2) t2.root2 = kQEmpty;
  const bool split = !P2 && S.split != 0;
  trace_engine<true, COUNT, TWO, DON>(S.nodes, S.tris, S.inst_leaf, P2 ? S.root2 : S.root, S.guard_box, t2, cursors + (P2 ? 5 : 2), n, &stk[threadIdx.x],
    [&](uint32_t idx, v3& o, v3& d, float& tmax, uint32_t& tag) {
      tag = q[idx];
      const float4 o4 = P.sh_o[tag], d4 = P.sh_d[tag];
      o = xyz(o4); d = xyz(d4); tmax = o4.w;
    },
    [&](uint32_t tag, float4, bool occluded) {
      if (split && P.sh_d[tag].w != 0.f) {                    // flagged: the second pass decides; an occluder found here cancels the contribution
        if (occluded) {                                       // nothing left to add, and nothing left to walk: the second pass retires it on sight
          P.sh_c[tag] = make_float4(0.f, 0.f, 0.f, P.sh_c[tag].w);
          float4 so = P.sh_o[tag]; so.w = -1.0f; P.sh_o[tag] = so;
        }
        return;
      }
      if (!occluded) {
        const float4 c = P.sh_c[tag];
        const uint32_t slot = __float_as_uint(c.w);
        float4 r = P.rad[slot];
        if (__float_as_uint(r.w) != P.stamp) r = make_float4(0.f, 0.f, 0.f, 0.f);      // not written by this batch yet: zero (DPaths::stamp)
        r.x += c.x; r.y += c.y; r.z += c.z; r.w = __uint_as_float(P.stamp);
        P.rad[slot] = r;
      }
    }, nn, nt, DON ? &s_bound[threadIdx.x & ~63u] : nullptr);
  if (COUNT) {
    nn = wave_sum(nn); nt = wave_sum(nt);
    if (lane_id() == 0) { atomicAdd(&C->nodes_any, (unsigned long long)nn); atomicAdd(&C->tris_any, (unsigned long long)nt); }
  }
}

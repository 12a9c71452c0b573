// kernels.hip -- hand-written gfx950 kernels of the progressive path tracer (wavefront formulation).
//
// One launch per stage and bounce, all on one HIP stream, no host round trip inside an iteration:
//   k_raygen -> [ k_trace_nearest -> k_shade -> k_trace_any ] x depth -> k_accumulate
// Stage boundaries exchange path-state POSITIONS through compacted queues built with wave64 ballot + prefix popcount,
// collected in LDS and appended with one atomic per 1024-8192 paths (a single counter word sustains only ~88 atomics/us on
// MI355X); the state itself is kept packed in runs (every shade chunk writes its survivors over its own first input
// positions in the other ray buffer), so all stages move whole cache lines at every bounce.  Traversal waves are persistent
// and refill idle lanes; they keep a per-lane stack in LDS (lane-interleaved, conflict free); BVH nodes are 48-B records on a
// 64-B stride with 8-bit child bounds and implicit child references (three dwordx4 fetches per visit), triangles 48-B records
// on a 64-B stride in leaf order.  No MFMA: there is no dense contraction on this path.
//
// Arithmetic follows DESIGN.md "Algorithm spec" operation by operation (fma only where written;
// built with -ffp-contract=off) so that results are bit-identical to the CPU oracle.
#include "kernels.h"

#include "../../include/crh_bvh_format.h"
#include "../../include/crh_xform.h"

namespace crh {
namespace {

typedef crh_v3 v3;

__device__ __forceinline__ v3 xyz(float4 a) { return crh_mk3(a.x, a.y, a.z); }
__device__ __forceinline__ float4 mk4(v3 a, float w) { return make_float4(a.x, a.y, a.z, w); }
__device__ __forceinline__ uint32_t wave_sum(uint32_t v)
{
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
  return v;   // total in lane 0
}
__device__ __forceinline__ uint32_t lane_id() { return threadIdx.x & 63u; }

// Path state is touched once per stage and never reused: streaming (non-temporal) accesses keep the L2 / Infinity Cache for the
// BVH, triangle and shading records that ARE reused (traversal kernel: +0.9 % C3, +0.7 % C5; the same treatment of the shading
// kernel's state accesses: -0.5 % C3, +0.1 % C2, not kept).
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ld_stream(const float4* p)
{ const f32x4 v = __builtin_nontemporal_load((const f32x4*)p); return make_float4(v.x, v.y, v.z, v.w); }
__device__ __forceinline__ void st_stream(float4* p, float4 v)
{ const f32x4 w = {v.x, v.y, v.z, v.w}; __builtin_nontemporal_store(w, (f32x4*)p); }


// Persistent-wave work distribution: each wavefront pulls the next 64 queue entries from a global cursor
// (one returning atomic per wave per chunk), so the grid only needs to fill the machine once and no
// workgroup is left running a statically assigned share after the others have drained.
__device__ __forceinline__ uint32_t wave_next_chunk(uint32_t* __restrict__ cursor)
{
  uint32_t base = 0;
  if (lane_id() == 0) base = atomicAdd(cursor, 64u);
  return __shfl(base, 0);
}

// Append to a workgroup-local LDS list: ballot + prefix popcount, one LDS atomic per wavefront.
__device__ __forceinline__ void lds_append(bool pred, uint32_t value, uint32_t* list, uint32_t* n)
{
  const unsigned long long mask = __ballot(pred);
  if (mask == 0ull) return;
  const uint32_t lane = lane_id();
  uint32_t base = 0;
  if (lane == 0) base = atomicAdd(n, (uint32_t)__popcll(mask));
  base = __shfl(base, 0);
  if (pred) list[base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull))] = value;
}

// Rank of this lane among the `pred` lanes of the workgroup's running list (ballot + prefix popcount, one LDS atomic per
// wavefront); only meaningful where pred holds.
__device__ __forceinline__ uint32_t lds_rank(bool pred, uint32_t* n)
{
  const unsigned long long mask = __ballot(pred);
  if (mask == 0ull) return 0u;
  const uint32_t lane = lane_id();
  uint32_t base = 0;
  if (lane == 0) base = atomicAdd(n, (uint32_t)__popcll(mask));
  base = __shfl(base, 0);
  return base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
}

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
#ifndef CRH_REFILL_IDLE
#define CRH_REFILL_IDLE 12     // refill a wavefront once this many of its 64 lanes have no ray
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

template <bool ANY, bool COUNT, bool TWO, bool DON, class Load, class Store>
__device__ __forceinline__ void trace_engine(const float4* __restrict__ nodes, const float4* __restrict__ tris,
                                             const float4* __restrict__ inst, uint32_t root, float4 gbox, const Top2 t2,
                                             uint32_t* __restrict__ cursor, uint32_t n, uint32_t* lds,
                                             Load load, Store store, uint32_t& n_nodes, uint32_t& n_tris, uint32_t* bound = nullptr)
{
  // bound (DON): one word per lane of this wavefront in LDS -- the smallest hit distance any part of the ray that STARTED in that
  // lane has found so far (float bits; distances are >= 0, so unsigned order = float order).  Every part prunes BOXES with it
  // (a box entered later than the bound holds nothing that can win the fold; equality is kept, ties are decided by order);
  // triangles are still accepted against the part's own `best`, which only knows what came earlier in traversal order.
  uint32_t ovf[kOvfStack];
  const uint32_t lane = lane_id();
  const unsigned long long lt_mask = (1ull << lane) - 1ull;
  // per-lane ray state
  bool have = false;
  uint32_t cur = kDone, tag = 0;
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
  const bool thin = DON && per_wave < 17u;                                   // at most half of the lanes get a ray of their own
  const uint32_t chunk = thin ? max(8u, (2u * per_wave + 7u) & ~7u) : min(kPoolChunk, max(64u, (per_wave + 63u) & ~63u));
  uint32_t pool_next = 0, pool_end = 0;
  bool exhausted = false;

  for (;;) {
    // ------------------------------------------------------------------ refill idle lanes
    unsigned long long idle = __ballot(!have);
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
          best = tmax; found = false; sp = 0; cur = root; have = true;
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

    if (DON && (exhausted || thin)) {
      // ---------------------------------------------------------------- donation: bottom stack entries -> idle lanes
      const unsigned long long idle_m = __ballot(!have);
      if (idle_m != 0ull) {
        // a donor gives the FAR half of its stack (the entries below the middle, all of them in the LDS part); the helper
        // copies them into its own column and starts with the nearest of them
        bool can = have && cur != kDone && sp > sbase && sp <= kLdsStack && !(ANY && found);
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
          const uint32_t rnext = __shfl(next, src), rhead = __shfl(head, src);
          if (gives) { next = kth_bit(idle_m, rank_d); sbase += give_n; }          // the helper comes right after the donor ...
          if (takes) {
            o = crh_mk3(rox, roy, roz); d = crh_mk3(rdx, rdy, rdz);
            ix = inv_dir(d.x); iy = inv_dir(d.y); iz = inv_dir(d.z);
            set_guard(gbox);
            if (TWO) save_world();
            best = rbest; found = false; sbase = 0; sp = rcnt - 1; cur = lds[sp * kBlock]; have = true;      // the nearest of the entries received
            hit = make_float4(rbest, 0.f, 0.f, __int_as_float(-1));
            is_child = true; cfound = false; next = rnext; head = rhead;            // ... and before what the donor gave away earlier
          }
        }
      }
    }

    auto read_top = [&]() {
      --sp;
      if (__builtin_expect(sp < kLdsStack, 1)) cur = lds[sp * kBlock];
      else { cur = ovf[sp - kLdsStack]; asm volatile("" : "+v"(cur)); }
    };
    auto pop = [&]() {
      if ((ANY && found) || sp == (DON ? sbase : 0)) { cur = kDone; return; }
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
      if (COUNT) ++n_nodes;
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
      if (ANY && CRH_SPEC_ANYHIT_SLOT_ORDER) {
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
      const float4 a = tp[0], b = tp[1], c = tp[2];
      if (COUNT) ++n_tris;
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
        if (DON) atomicMin(&bound[head], ANY ? 0u : __float_as_uint(tt));      // any-hit: one occluder ends every part's walk
        hit = make_float4(tt, uu, vv, __int_as_float((int)ti));
      }
    };
    // ------------------------------------------------------------------ (A) inner nodes until a leaf is in hand
#if CRH_INNER_STEPS > 0
#pragma unroll 1
    for (int step_ = 0; step_ < CRH_INNER_STEPS && have && !(cur & kQLeafBit); ++step_) inner_step();
#else
    while (have && !(cur & kQLeafBit)) inner_step();
#endif
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
    } else if (have && (cur & kQLeafBit) && cur != kDone) {
      tri_step(cur & 0x0FFFFFFFu);                               // one triangle per leaf (crh_bvh_format.h)
      pop();
    }

    // ------------------------------------------------------------------ (C) retire finished rays
    if (DON) {
      // own part walked and no successor left: fold what was absorbed behind the own hit (left-biased minimum: a later part
      // wins only with a strictly smaller t); helpers then wait to be absorbed by their predecessor, the head stores
      const bool finished = have && cur == kDone && next == kNoLane;
      if (finished && cfound) { if (ANY || !found || chit.x < hit.x) { hit = chit; found = true; } cfound = false; }
      const unsigned long long fin_children = __ballot(finished && is_child);
      if (fin_children != 0ull) {
        const bool takes = have && next != kNoLane && ((fin_children >> next) & 1ull);
        const uint32_t from = takes ? next : lane;
        const float hx = __shfl(hit.x, from), hy = __shfl(hit.y, from), hz = __shfl(hit.z, from), hw = __shfl(hit.w, from);
        const int hf = __shfl((int)found, from);
        if (takes) {
          // the successor's total goes IN FRONT of what this lane absorbed before (it was donated later = it comes earlier)
          if (hf && (ANY || !cfound || !(chit.x < hx))) { chit = make_float4(hx, hy, hz, hw); cfound = true; }
          next = kNoLane;
        }
        if ((fin_children >> lane) & 1ull) { have = false; is_child = false; }           // absorbed: the lane is free again
      }
      if (have && !is_child && cur == kDone && next == kNoLane) {
        if (cfound) { if (ANY || !found || chit.x < hit.x) { hit = chit; found = true; } cfound = false; }
        store(tag, hit, found); have = false;
      }
    } else if (have && cur == kDone) { store(tag, hit, found); have = false; }
  }
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

// ================================================================== camera-ray packets
// Bounce 0 of a wide batch: the 64 consecutive queue entries a wavefront takes are 64 samples of ONE pixel (or of 2 - 4 neighbouring ones: slot_to_pixel_sample)
// -- rays that differ by a sub-pixel jitter.  Walked one by one (trace_engine) they fetch the same nodes 64 times through the address path and still run at
// 0.71 of the lanes, because they reach their leaves in different rounds; it is the launch that costs most (20 % of the traversal time, VALU-issue bound:
// profiles/r4/per_bounce_counters.txt).  Here the WAVEFRONT walks the tree once for all of them:
//   * one stack per wavefront, held in three VGPRs (lane i = entry i: a select on push, v_readlane on pop), each entry a node reference + the 64-bit mask of the lanes
//     whose ray entered that child's box; node and triangle records are fetched with SCALAR loads (one request per wavefront instead of 64 lane requests);
//   * a lane takes part in a node / triangle test iff its bit is set -- exactly the rays that would get there in a walk of their own (the box test, its guard
//     band and the pruning against the lane's own `best` are the per-ray ones, bit for bit) -- so every lane computes something useful in every instruction;
//   * children are taken near to far as the FIRST participating lane sees them (its keys, sorted on the scalar unit); children only other lanes hit follow in
//     slot order.
// The hit a ray ends with does not depend on the order triangles are tested in -- the nearest one wins -- EXCEPT among triangles at exactly the same distance,
// where the spec says "the first in the ray's own near-to-far walk" (strict t < best).  A lane that meets such a tie (a valid hit at t == best) is flagged and
// written to the fall-back queue instead of the hit buffer; k_trace_nearest<.., FB> walks those rays (a handful per million in scenes with shared edges, none
// in a triangle soup) in the prescribed order afterwards.  Hits are therefore the spec's, bit for bit; node visits are not counted here -- the counting
// instantiations never use packets.
constexpr uint32_t kPacketChunk = 1024;      // queue entries (16 packets) per cursor fetch: one counter word sustains ~88 atomics / us
__device__ __forceinline__ uint32_t sgpr(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }

// lane `lane` (uniform) of three registers <- three uniform values: v_writelane with the lane in M0 (two different scalar registers in one VALU instruction
// exceed the constant bus; the compiler's builtin for it is not declared by this toolchain)
__device__ __forceinline__ void stack_put(uint32_t& a, uint32_t& b, uint32_t& c, uint32_t va, uint32_t vb, uint32_t vc, uint32_t lane)
{
  asm("s_mov_b32 m0, %6\n\tv_writelane_b32 %0, %3, m0\n\tv_writelane_b32 %1, %4, m0\n\tv_writelane_b32 %2, %5, m0"
      : "+v"(a), "+v"(b), "+v"(c) : "s"(va), "s"(vb), "s"(vc), "s"(lane) : "m0");
}
// one of four uniform masks by a uniform index: three scalar selects
__device__ __forceinline__ unsigned long long pick_mask(unsigned long long m0, unsigned long long m1, unsigned long long m2, unsigned long long m3, uint32_t i)
{
  unsigned long long m = m0;
  m = i == 1u ? m1 : m; m = i == 2u ? m2 : m; m = i == 3u ? m3 : m;
  return m;
}

// (t & ~3) | K as ONE vector instruction that the compiler may not move behind the v_readlane that follows it (it would: two scalar instructions on values that
// are uniform by then -- but the packet walk is bound by the scalar unit, k_trace_packets)
template <int K> __device__ __forceinline__ uint32_t key_bits(uint32_t t)
{
  uint32_t r;
  asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(r) : "v"(t), "s"(0x7FFFFFFCu), "n"(K));
  return r;
}

// Packet nodes (k_trace_packets<true>): node i of the tree as 8 x float4 -- {origin.xyz | exponents, counts}, x planes {lo0, hi0, lo1, hi1} {lo2, hi2, lo3, hi3},
// y planes, z planes, {first inner child, first leaf reference, -, -} -- the quantised bytes of the 64-B node converted once per scene instead of once per visit
__global__ __launch_bounds__(kBlock) void k_expand_packet_nodes(const float4* __restrict__ nodes, float4* __restrict__ pn, uint32_t n)
{
  const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  const float4* np = nodes + (uint32_t)(CRH_NODE_DWORDS / 4) * i;
  const float4 n0 = np[0], n1 = np[1], n2 = np[2];
  const uint32_t lo[3] = {__float_as_uint(n1.x), __float_as_uint(n1.y), __float_as_uint(n1.z)}, hi[3] = {__float_as_uint(n1.w), __float_as_uint(n2.x), __float_as_uint(n2.y)};
  const uint32_t nch = __float_as_uint(n0.w) >> 28;
  float4* out = pn + 8u * i;
  out[0] = n0;
  for (int a = 0; a < 3; ++a) {
    float f[8];
    for (int k = 0; k < 4; ++k) {
      f[2 * k] = (float)((lo[a] >> (8 * k)) & 0xffu); f[2 * k + 1] = (float)((hi[a] >> (8 * k)) & 0xffu);
      // a slot without a child: lower plane +inf, upper plane -inf -- the entry distance comes out +inf and the exit distance -inf on every axis whatever the
      // direction (inf x finite scale; a 0 x inf = NaN on ONE axis is dropped by max / min, and a unit direction cannot scale all three axes to zero), so no
      // ray enters it and the walk needs no child-count test
      if ((uint32_t)k >= nch) { f[2 * k] = __builtin_inff(); f[2 * k + 1] = -__builtin_inff(); }
    }
    out[1 + 2 * a] = make_float4(f[0], f[1], f[2], f[3]); out[2 + 2 * a] = make_float4(f[4], f[5], f[6], f[7]);
  }
  out[7] = make_float4(n2.z, n2.w, 0.f, 0.f);
}

// PN: the node is read from the PACKET-NODE array (k_expand_packet_nodes: the eight quantised planes of the four children as FLOATS, 128 B per node): a packed
// multiply-add takes a child's {lower, upper} plane pair straight from the scalar registers the node was loaded into, and the 24 byte-to-float conversions of a
// visit are gone.  The values are the same floats (0 .. 255), so is every result.  Only with uniform direction signs (OCT < 8).
template <int OCT, bool PN>
__device__ __forceinline__ void packet_walk(const float4* __restrict__ nodes, const float4* __restrict__ tris, uint32_t root, uint32_t lane, bool act,
                                            v3 o, v3 d, float ix, float iy, float iz, float gx, float gy, float gz, float4& hit, bool& amb)
{
  float best = CRH_MAXFLOAT; bool found = false;
  // OCT < 8: the direction signs of the whole packet (bit 0 / 1 / 2 = x / y / z negative), known at compile time -- the lower / upper byte words of a node are
  // then picked by REGISTER CHOICE (scalar operands of v_cvt_f32_ubyte) instead of six selects per visit; OCT = 8: mixed signs, per-lane selects
  const bool sx = OCT < 8 ? (OCT & 1) != 0 : ix < 0.f, sy = OCT < 8 ? (OCT & 2) != 0 : iy < 0.f, sz = OCT < 8 ? (OCT & 4) != 0 : iz < 0.f;
  // the wavefront's stack: lane i of these three registers is entry i
  uint32_t st_ref = 0, st_mlo = 0, st_mhi = 0; uint32_t sp = 0; bool ovf = false;
  unsigned long long cm = __ballot(act);
  uint32_t cur = root;
  while (cm != 0ull) {
    const bool in = (cm >> lane) & 1ull;
    if (!(cur & kQLeafBit)) {
      // uniform address: scalar loads
      const float4* np = PN ? nodes + 8u * cur : nodes + (uint32_t)(CRH_NODE_DWORDS / 4) * cur;
      const float4 n0 = np[0], n1 = np[1], n2 = np[2];
      float4 n3 = n0, n4 = n0, n5 = n0, n6 = n0, n7 = n0;
      if (PN) { n3 = np[3]; n4 = np[4]; n5 = np[5]; n6 = np[6]; n7 = np[7]; }
      const uint32_t ew = __float_as_uint(n0.w);
      const uint32_t ni = (ew >> 24) & 7u, nch = ew >> 28;
      const uint32_t base_inner = __float_as_uint(PN ? n7.x : n2.z), base_leaf = __float_as_uint(PN ? n7.y : n2.w) - ni;
      const float ax = __builtin_amdgcn_ldexpf(ix, (int)(ew << 24) >> 24), ay = __builtin_amdgcn_ldexpf(iy, (int)(ew << 16) >> 24), az = __builtin_amdgcn_ldexpf(iz, (int)(ew << 8) >> 24);
      const float ddx = n0.x - o.x, ddy = n0.y - o.y, ddz = n0.z - o.z;
      const uint32_t lx = __float_as_uint(sx ? n1.w : n1.x), ly = __float_as_uint(sy ? n2.x : n1.y), lz = __float_as_uint(sz ? n2.y : n1.z);
      const uint32_t hx = __float_as_uint(sx ? n1.x : n1.w), hy = __float_as_uint(sy ? n1.y : n2.x), hz = __float_as_uint(sz ? n1.z : n2.y);
      const f32x2 ax2 = {ax, ax}, ay2 = {ay, ay}, az2 = {az, az};
      // PN: component 0 of a pair belongs to the LOWER plane whatever the direction, so the guard band changes sides with the sign instead of the planes
      const f32x2 bx2 = __builtin_elementwise_fma((f32x2){ddx, ddx}, (f32x2){ix, ix}, (PN && sx) ? (f32x2){gx, -gx} : (f32x2){-gx, gx});
      const f32x2 by2 = __builtin_elementwise_fma((f32x2){ddy, ddy}, (f32x2){iy, iy}, (PN && sy) ? (f32x2){gy, -gy} : (f32x2){-gy, gy});
      const f32x2 bz2 = __builtin_elementwise_fma((f32x2){ddz, ddz}, (f32x2){iz, iz}, (PN && sz) ? (f32x2){gz, -gz} : (f32x2){-gz, gz});
      const uint32_t L = (uint32_t)__builtin_ctzll(cm);                      // the lane whose keys order the children
      const float best_in = in ? best : -1.0f;                               // PN: entry distances are >= 0, so a lane that is not in this node's mask enters no child
      unsigned long long mk[4] = {0ull, 0ull, 0ull, 0ull};
      uint32_t key[4] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
#define CRH_QB(W, K) ((float)(((W) >> (8 * (K))) & 0xffu))
#define CRH_PP(A, B, K) ((K) == 0 ? (f32x2){(A).x, (A).y} : (K) == 1 ? (f32x2){(A).z, (A).w} : (K) == 2 ? (f32x2){(B).x, (B).y} : (f32x2){(B).z, (B).w})
#define CRH_PCHILD(K)                                                                                          \
      if (PN || (uint32_t)K < nch) {      /* PN: all four slots, an absent child is masked out below (no branch: the node's loads stay one batch) */ \
        f32x2 tx, ty, tz;                                                                                  \
        if (PN) {                                                                                          \
          const f32x2 px = __builtin_elementwise_fma(CRH_PP(n1, n2, K), ax2, bx2), py = __builtin_elementwise_fma(CRH_PP(n3, n4, K), ay2, by2); \
          const f32x2 pz = __builtin_elementwise_fma(CRH_PP(n5, n6, K), az2, bz2);                         \
          tx = sx ? (f32x2){px.y, px.x} : px; ty = sy ? (f32x2){py.y, py.x} : py; tz = sz ? (f32x2){pz.y, pz.x} : pz;      /* {near, far}: a choice of registers */ \
        } else {                                                                                           \
          tx = __builtin_elementwise_fma((f32x2){CRH_QB(lx, K), CRH_QB(hx, K)}, ax2, bx2);                \
          ty = __builtin_elementwise_fma((f32x2){CRH_QB(ly, K), CRH_QB(hy, K)}, ay2, by2);                \
          tz = __builtin_elementwise_fma((f32x2){CRH_QB(lz, K), CRH_QB(hz, K)}, az2, bz2);                \
        }                                                                                                  \
        const float tmin = fmaxf(fmaxf(fmaxf(tx.x, ty.x), tz.x), 0.f);                                     \
        const float tmx  = fminf(fminf(fminf(tx.y, ty.y), tz.y), PN ? best_in : best);                     \
        const bool hitk = tmin <= tmx;                                                                     \
        /* every lane of the wavefront runs this loop: the vote is a plain compare into a scalar pair.  PN: an empty slot admits nobody (k_expand_packet_nodes), a lane \
           outside the node's mask prunes against -1 (best_in), and the ordering key is lane L's entry distance whether or not L itself enters the child */ \
        mk[K] = PN ? __builtin_amdgcn_ballot_w64(hitk) : (__builtin_amdgcn_ballot_w64(hitk) & cm);         \
        const uint32_t tv = __float_as_uint((PN || hitk) ? tmin : __builtin_inff());      /* !PN: lane L is in cm: +inf when it misses this child */ \
        const uint32_t tb = PN ? (uint32_t)__builtin_amdgcn_readlane((int)key_bits<K>(tv), (int)L)      /* PN: the key is finished on the vector unit (one v_and_or), the scalar one is the busier */ \
                               : (((uint32_t)__builtin_amdgcn_readlane((int)tv, (int)L) & 0x7FFFFFFCu) | (uint32_t)K);         \
        key[K] = mk[K] != 0ull ? tb : 0xFFFFFFFFu;                                                         \
      }
      CRH_PCHILD(0) CRH_PCHILD(1) CRH_PCHILD(2) CRH_PCHILD(3)
#undef CRH_PCHILD
#undef CRH_PP
#undef CRH_QB
      { // four unique scalar keys, ascending: the children somebody hit come first (0xFFFFFFFF = nobody)
        uint32_t a0 = min(key[0], key[1]), a1 = max(key[0], key[1]), b0 = min(key[2], key[3]), b1 = max(key[2], key[3]);
        const uint32_t lo = min(a0, b0), hi = max(a1, b1), m0 = max(a0, b0), m1 = min(a1, b1);
        key[0] = lo; key[1] = min(m0, m1); key[2] = max(m0, m1); key[3] = hi;
      }
#define CRH_PREF(I) (((I) < ni ? base_inner : base_leaf) + (I))
      if (key[0] != 0xFFFFFFFFu) {
        if (sp > 61u) { ovf = true; break; }      // deeper than any tree of the builder (<= 60 pending entries): the whole packet takes the fall-back pass
#define CRH_PPUSH(KEY)                                                                                         \
        if ((KEY) != 0xFFFFFFFFu) {                                                                        \
          const uint32_t ci = (KEY) & 3u;                                                                  \
          const unsigned long long pm = pick_mask(mk[0], mk[1], mk[2], mk[3], ci);                                                 \
          stack_put(st_ref, st_mlo, st_mhi, CRH_PREF(ci), (uint32_t)pm, (uint32_t)(pm >> 32), sp);      /* lane `sp` of the three registers takes the entry */ \
          ++sp;                                                                                            \
        }
        CRH_PPUSH(key[3]) CRH_PPUSH(key[2]) CRH_PPUSH(key[1])                 // far .. near
#undef CRH_PPUSH
        const uint32_t c0 = key[0] & 3u;
        cur = CRH_PREF(c0); cm = pick_mask(mk[0], mk[1], mk[2], mk[3], c0);
        continue;
      }
#undef CRH_PREF
    } else {
      const uint32_t ti = cur & 0x0FFFFFFFu;                                  // one triangle per leaf; uniform address: scalar loads
      const float4* tp = tris + kTriStride * ti;
      const float4 a = tp[0], b = tp[1], c = tp[2];
      const v3 v0 = xyz(a), e0 = xyz(b), e1 = xyz(c), nrm = crh_mk3(a.w, b.w, c.w);
      const v3 to = crh_sub3(v0, o);                                          // trace_engine::tri_step, operation by operation
      const float inv = 1.0f / crh_dot3(nrm, d);
      const v3 vc = crh_cross3(d, to);
      const float tt = crh_dot3(nrm, to) * inv, uu = crh_dot3(vc, e1) * inv, vv = crh_dot3(vc, e0) * inv;
      const bool ok = in && tt >= 0.f && uu >= 0.f && vv >= 0.f && (uu + vv) <= 1.0f;
      amb = amb || (ok && found && tt == best);                               // two triangles at exactly this distance: the ray's own walk decides (fall-back pass)
      const bool acc = ok && tt < best;                                       // selects, not branches: every lane of the wavefront is here anyway
      best = acc ? tt : best; found = found || acc;
      hit.x = acc ? tt : hit.x; hit.y = acc ? uu : hit.y; hit.z = acc ? vv : hit.z; hit.w = acc ? __int_as_float((int)ti) : hit.w;
    }
    if (sp == 0u) break;
    --sp;
    cur = (uint32_t)__builtin_amdgcn_readlane((int)st_ref, (int)sp);
    cm = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)st_mhi, (int)sp) << 32) | (unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)st_mlo, (int)sp);
  }
  if (ovf) amb = true;
}

template <bool PN>
__global__ __launch_bounds__(kBlock, 8) void k_trace_packets(DScene S, DPaths P, const float4* __restrict__ nodes, const float4* __restrict__ pnodes, const float4* __restrict__ tris,      // = S.nodes, S.pnodes, S.tris: as restrict-qualified PARAMETERS the compiler may read them with scalar loads
                                                          const uint32_t* __restrict__ q, const uint32_t* __restrict__ count,
                                                          uint32_t* __restrict__ cursors, uint32_t* zero_a, uint32_t* zero_b, uint32_t* zero_c, uint32_t* zero_d,
                                                          uint32_t* __restrict__ fb_q, uint32_t* __restrict__ fb_count, DCounters* C)
{
  const uint32_t n = *count;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    *zero_a = 0u; *zero_b = 0u; *zero_c = 0u; *zero_d = 0u;      // as k_trace_nearest: the other queue's count, the shadow count, the second-pass counts
    cursors[1] = 0u; cursors[2] = 0u; cursors[5] = 0u;
    atomicAdd(&C->rays_nearest, (unsigned long long)n);
  }
  const uint32_t lane = lane_id();
  const float4* __restrict__ ray_o = P.ray_o[0]; const float4* __restrict__ ray_d = P.ray_d[0];
  const float4 gb = S.guard_box;
  for (;;) {
    uint32_t cbase = 0;
    if (lane == 0) cbase = atomicAdd(cursors, kPacketChunk);
    cbase = sgpr(__shfl(cbase, 0));
    if (cbase >= n) break;
    const uint32_t cend = min(cbase + kPacketChunk, n);
    for (uint32_t pb = cbase; pb < cend; pb += 64u) {
      const uint32_t idx = pb + lane;
      const bool act = idx < cend;
      uint32_t tag = 0; v3 o = crh_mk3(0.f, 0.f, 0.f), d = crh_mk3(1.f, 0.f, 0.f);
      if (act) { tag = q[idx]; const float4 o4 = ld_stream(&ray_o[tag]), d4 = ld_stream(&ray_d[tag]); o = xyz(o4); d = xyz(d4); }
      const float ix = inv_dir(d.x), iy = inv_dir(d.y), iz = inv_dir(d.z);
      const float R = CRH_FMA(gb.w, 3.0f, (crh_abs(o.x - gb.x) + crh_abs(o.y - gb.y)) + crh_abs(o.z - gb.z)) * kSlabGuard;      // trace_engine::set_guard
      const float gx = crh_abs(ix) * R, gy = crh_abs(iy) * R, gz = crh_abs(iz) * R;
      float4 hit = make_float4(CRH_MAXFLOAT, 0.f, 0.f, __int_as_float(-1)); bool amb = false;
      {
        // the packet's direction signs: uniform over the wavefront for all but the packets that straddle an axis of the view -- one specialised walk per octant
        const unsigned long long am_ = __ballot(act), bx_ = __ballot(act && ix < 0.f), by_ = __ballot(act && iy < 0.f), bz_ = __ballot(act && iz < 0.f);
        const bool uni = (bx_ == 0ull || bx_ == am_) && (by_ == 0ull || by_ == am_) && (bz_ == 0ull || bz_ == am_);
        const uint32_t oct = uni ? ((bx_ ? 1u : 0u) | (by_ ? 2u : 0u) | (bz_ ? 4u : 0u)) : 8u;
#define CRH_WALK(O) case O: packet_walk<O, PN>(PN ? pnodes : nodes, tris, S.root, lane, act, o, d, ix, iy, iz, gx, gy, gz, hit, amb); break;
        switch (oct) { CRH_WALK(0) CRH_WALK(1) CRH_WALK(2) CRH_WALK(3) CRH_WALK(4) CRH_WALK(5) CRH_WALK(6) CRH_WALK(7) default: packet_walk<8, false>(nodes, tris, S.root, lane, act, o, d, ix, iy, iz, gx, gy, gz, hit, amb); }
#undef CRH_WALK
      }
      if (act && !amb) st_stream(&P.hit[tag], hit);
      const unsigned long long am = __ballot(act && amb);
      if (am != 0ull) {
        uint32_t fb = 0;
        if (lane == 0) fb = atomicAdd(fb_count, (uint32_t)__popcll(am));
        fb = __shfl(fb, 0);
        if (act && amb) fb_q[fb + (uint32_t)__popcll(am & ((1ull << lane) - 1ull))] = tag;
      }
    }
  }
}

// API-level tracing of a caller ray buffer {o.xyz, tmax, d.xyz, -}; `cursor` must be zero at launch
template <bool ANY, bool COUNT, bool TWO>
__global__ CRH_TRACE_BOUNDS void k_trace_rays(DScene S, const float4* __restrict__ rays, uint32_t n, uint32_t* __restrict__ cursor,
                                               float4* __restrict__ out_hit, uint32_t* __restrict__ out_vis, DCounters* C)
{
  __shared__ uint32_t stk[kLdsStack * kBlock];
  uint32_t nn = 0, nt = 0;
  trace_engine<ANY, COUNT, TWO, false>(S.nodes, S.tris, S.inst_leaf, S.root, S.guard_box, top2_of(S, false), cursor, n, &stk[threadIdx.x],
    [&](uint32_t idx, v3& o, v3& d, float& tmax, uint32_t& tag) {
      tag = idx;
      const float4 o4 = rays[2u * idx], d4 = rays[2u * idx + 1u];
      o = xyz(o4); d = xyz(d4); tmax = o4.w;
    },
    [&](uint32_t tag, float4 h, bool f) {
      if (ANY) out_vis[tag] = f ? 0u : 1u;
      else {
        const int k = __float_as_int(h.w);
        if (k >= 0) h.w = S.tris[kTriStride * (uint32_t)k + 3u].x;   // leaf order -> caller's triangle index (fourth quarter of the 64-B record)
        out_hit[tag] = h;
      }
    }, nn, nt);
  if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(ANY ? &C->rays_any : &C->rays_nearest, (unsigned long long)n);
  if (COUNT) {
    nn = wave_sum(nn); nt = wave_sum(nt);
    if (lane_id() == 0) {
      atomicAdd(ANY ? &C->nodes_any : &C->nodes_nearest, (unsigned long long)nn);
      atomicAdd(ANY ? &C->tris_any : &C->tris_nearest, (unsigned long long)nt);
    }
  }
}


// ================================================================== BSDF
struct Bsdf {
  v3 Kc, Kd, Ks, Kt, Le, Fc;
  float Rc, Rs;
  float4 fc, fb, ab;
};

__device__ v3 fresnel_media(float cosI, float4 f)
{
  if (f.x > -0.5f) {
    const float m = 1.0f - crh_abs(cosI); const float m2 = m * m; const float m5 = (m2 * m2) * m;
    return crh_mk3(CRH_FMA(1.0f - f.x, m5, f.x), CRH_FMA(1.0f - f.y, m5, f.y), CRH_FMA(1.0f - f.z, m5, f.z));
  }
  if (f.x > -1.5f) return crh_mk3(f.z, f.z, f.z);
  if (f.x > -2.5f) {
    const float ci = crh_abs(cosI), n = f.y, k = f.z;
    const float tmp = (2.0f * n) * ci;
    const float t1 = CRH_FMA(n, n, k * k);
    const float ci2 = ci * ci;
    const float sperp = ((t1 - tmp) + ci2) / ((t1 + tmp) + ci2);
    const float t2 = t1 * ci2;
    const float sparl = ((t2 - tmp) + 1.0f) / ((t2 + tmp) + 1.0f);
    const float r = (sperp + sparl) * 0.5f;
    return crh_mk3(r, r, r);
  }
  const float n = f.y;
  const float etaI = cosI > 0.f ? 1.0f : n, etaT = cosI > 0.f ? n : 1.0f;
  float r = 1.0f;
  const float ratio = etaI / etaT;
  const float sinT2 = (ratio * ratio) * CRH_FMA(-cosI, cosI, 1.0f);
  if (sinT2 < 1.0f) {
    const float ci = crh_abs(cosI), ct = crh_sqrt(1.0f - sinT2);
    const float p0 = etaT * ci, p1 = etaI * ct, q0 = etaI * ci, q1 = etaT * ct;
    const float parl = (p0 - p1) / (p0 + p1);
    const float perp = (q0 - q1) / (q0 + q1);
    const float pp = parl * parl, qq = perp * perp;
    r = (pp + qq) * 0.5f;
  }
  return crh_mk3(r, r, r);
}

__device__ float smith_g1(v3 w, v3 m, float rough)
{
  float r = 0.f;
  if (crh_dot3(w, m) * w.z > 0.f) {
    const float tanT = crh_sqrt(crh_max(CRH_FMA(-w.z, w.z, 1.0f), 0.f)) / w.z;
    if (tanT == 0.f) r = 1.0f;
    else {
      const float a = 1.0f / (rough * tanT);
      r = CRH_FMA(2.181f, a, 3.535f) / CRH_FMA(2.577f, a, 1.0f / a + 2.276f);
    }
  }
  return crh_min(r, 1.0f);
}

__device__ __forceinline__ float blinn_power(float rough) { return crh_max(2.0f / (rough * rough) - 2.0f, 0.f); }

__device__ v3 eval_blinn(v3 wi, v3 wo, float4 fr, float rough)
{
  if (wi.z <= 0.f || wo.z <= 0.f) return crh_mk3(0.f, 0.f, 0.f);
  const v3 h = crh_norm3(crh_add3(wi, wo));
  const float e = blinn_power(rough);
  const float D = ((e + 2.0f) * CRH_INV_TWOPI) * crh_pow(h.z, e);
  const float G = smith_g1(wo, h, rough) * smith_g1(wi, h, rough);
  const v3 F = fresnel_media(crh_dot3(wo, h), fr);
  const float s = (D * G) / (4.0f * wo.z);
  return crh_scale3(F, s);
}

__device__ v3 eval_layered(const Bsdf& b, v3 wi, v3 wo, int two_sided)
{
  if (two_sided) { const float sg = crh_sign(wo.z); wi.z *= sg; wo.z *= sg; }
  const float lam = (wi.z <= 0.f || wo.z <= 0.f) ? 0.f : wi.z * CRH_INV_PI;
  v3 r = crh_scale3(b.Kd, lam);
  if (b.Rs > kBsdfEps) r = crh_add3(r, crh_mul3(b.Ks, eval_blinn(wi, wo, b.fb, b.Rs)));
  r = crh_mul3(r, crh_mk3(1.0f - b.Fc.x, 1.0f - b.Fc.y, 1.0f - b.Fc.z));
  if (b.Rc > kBsdfEps) r = crh_add3(r, crh_mul3(b.Kc, eval_blinn(wi, wo, b.fc, b.Rc)));
  return r;
}

struct Lobes { float pc, pd, ps, pt, total; v3 Tc; };
__device__ __forceinline__ Lobes lobe_probs(const Bsdf& b, v3 W)
{
  Lobes L;
  L.Tc = crh_mk3(1.0f - b.Fc.x, 1.0f - b.Fc.y, 1.0f - b.Fc.z);
  L.pc = crh_dot3(crh_mul3(b.Kc, b.Fc), W);
  L.pd = crh_dot3(crh_mul3(b.Kd, L.Tc), W);
  L.ps = crh_dot3(crh_mul3(b.Ks, L.Tc), W);
  L.pt = crh_dot3(crh_mul3(b.Kt, L.Tc), W);
  L.total = ((L.pc + L.pd) + L.ps) + L.pt;
  return L;
}

__device__ float blinn_pdf(float hz, float dih, float rough)
{
  const float e = blinn_power(rough);
  return (((e + 2.0f) * CRH_INV_TWOPI) * crh_pow(crh_abs(hz), e + 1.0f)) / (4.0f * crh_abs(dih));
}

// lobe < 0: the mixture pdf over all non-delta lobes (the spec's MIS pdf); lobe = 0 coat / 1 diffuse / 2 glossy: that lobe's pdf times its
// selection probability only (crh_spec.h #3, mis_single_lobe)
__device__ float pdf_layered(const Bsdf& b, v3 wo, v3 wi, v3 W, int two_sided, int lobe = -1)
{
  const Lobes L = lobe_probs(b, W);
  if (!(L.total > kBsdfEps)) return 0.f;
  if (two_sided) { const float sg = crh_sign(wo.z); wi.z *= sg; wo.z *= sg; }
  float pdf = 0.f;
  if (wi.z > 0.f && wo.z > 0.f) {
    const v3 h = crh_norm3(crh_add3(wi, wo));
    const float dih = crh_dot3(wi, h);
    if (lobe < 0 || lobe == 1) pdf = L.pd * (wi.z * CRH_INV_PI);
    if (b.Rc > kBsdfEps && (lobe < 0 || lobe == 0)) pdf = CRH_FMA(L.pc, blinn_pdf(h.z, dih, b.Rc), pdf);
    if (b.Rs > kBsdfEps && (lobe < 0 || lobe == 2)) pdf = CRH_FMA(L.ps, blinn_pdf(h.z, dih, b.Rs), pdf);
  }
  return pdf / L.total;
}

__device__ v3 sample_blinn(v3 wo, v3& wi, float4 fr, float rough, uint32_t& rng, int two_sided, bool& ok, int u32)
{
  const float k1 = crh_rng_next_mode(&rng, u32), k2 = crh_rng_next_mode(&rng, u32);
  const float e = blinn_power(rough);
  const float cm = crh_pow(k1, 1.0f / (e + 2.0f));
  float s, c; crh_sincos2pi(k2, &s, &c);
  const float sm = crh_sqrt(crh_max(CRH_FMA(-cm, cm, 1.0f), 0.f));
  const v3 m = crh_mk3(c * sm, s * sm, cm);
  bool flip = false;
  if (two_sided && wo.z < 0.f) { flip = true; wo.z = -wo.z; }
  const float cd = crh_dot3(wo, m);
  const float cd2 = 2.0f * cd;
  wi = crh_mk3(CRH_FMA(cd2, m.x, -wo.x), CRH_FMA(cd2, m.y, -wo.y), CRH_FMA(cd2, m.z, -wo.z));
  if (wi.z <= 0.f || wo.z <= 0.f || !(cd > 0.f)) { ok = false; return crh_mk3(0.f, 0.f, 0.f); }
  const float G = smith_g1(wo, m, rough) * smith_g1(wi, m, rough);
  const v3 F = fresnel_media(cd, fr);
  const float w = (G * cd) / (wo.z * m.z);
  if (flip) wi.z = -wi.z;
  ok = true;
  return crh_scale3(F, w);
}

// the crh_spec.h switches the BSDF code sees (wave-uniform)
struct SpecB { int u32; float eta_nd; };

// returns alive; W multiplied by the lobe weight; inside toggled on transmission; lobe = 0 coat / 1 diffuse / 2 glossy / 3 transmission
__device__ bool sample_layered(const Bsdf& b, v3 wo, v3& wi, v3& W, bool& inside, bool& delta, uint32_t& rng, int two_sided, SpecB sp, int& lobe)
{
  const Lobes L = lobe_probs(b, W);
  const float ksi = L.total * crh_rng_next_mode(&rng, sp.u32);
  delta = false; lobe = 0;
  if (!(L.total > kBsdfEps)) { W = crh_mk3(0.f, 0.f, 0.f); return false; }
  const v3 mirror = crh_mk3(-wo.x, -wo.y, wo.z);
  bool ok = true; v3 k;
  if (ksi < L.pc) {
    k = crh_scale3(b.Kc, L.total / L.pc);
    if (b.Rc > kBsdfEps) k = crh_mul3(k, sample_blinn(wo, wi, b.fc, b.Rc, rng, two_sided, ok, sp.u32));
    else { k = crh_mul3(k, b.Fc); wi = mirror; delta = true; }
  } else if (ksi < L.pc + L.pd) {
    k = crh_scale3(crh_mul3(b.Kd, L.Tc), L.total / L.pd); lobe = 1;
    const float k1 = crh_rng_next_mode(&rng, sp.u32), k2 = crh_rng_next_mode(&rng, sp.u32);
    float s, c; crh_sincos2pi(k1, &s, &c);
    const float r = crh_sqrt(k2);
    wi = crh_mk3(c * r, s * r, crh_sqrt(1.0f - k2));
    if (two_sided) { if (wo.z < 0.f) wi.z = -wi.z; }
    else if (!(wo.z > 0.f)) ok = false;
  } else if (ksi < (L.pc + L.pd) + L.ps) {
    k = crh_scale3(crh_mul3(b.Ks, L.Tc), L.total / L.ps); lobe = 2;
    if (b.Rs > kBsdfEps) k = crh_mul3(k, sample_blinn(wo, wi, b.fb, b.Rs, rng, two_sided, ok, sp.u32));
    else { k = crh_mul3(k, fresnel_media(wo.z, b.fb)); wi = mirror; delta = true; }
  } else {
    k = crh_scale3(crh_mul3(b.Kt, L.Tc), L.total / L.pt); lobe = 3;
    const float ior = b.fc.x > -2.5f ? sp.eta_nd : b.fc.y;   // no dielectric coat: crh_spec.eta_no_dielectric (default 1 = index-matched, straight through)
    const float eta = wo.z > 0.f ? 1.0f / ior : ior;
    const float sinT2 = (eta * eta) * CRH_FMA(-wo.z, wo.z, 1.0f);
    if (!(sinT2 < 1.0f) || !(L.pt > 0.f)) ok = false;
    else {
      float ct = crh_sqrt(1.0f - sinT2); if (wo.z > 0.f) ct = -ct;
      wi = crh_norm3(crh_mk3(-(eta * wo.x), -(eta * wo.y), ct));
      inside = !inside; delta = true;
    }
  }
  if (!ok) { W = crh_mk3(0.f, 0.f, 0.f); return false; }
  W = crh_mul3(W, k);
  return true;
}

// ================================================================== frames, lights, environment
struct Frame { v3 t, b, n; };
__device__ __forceinline__ Frame make_frame(v3 n)
{
  Frame f; f.n = n;
  const v3 t = (crh_abs(n.x) > crh_abs(n.z)) ? crh_mk3(-n.y, n.x, 0.f) : crh_mk3(0.f, -n.z, n.y);
  f.t = crh_norm3(t); f.b = crh_cross3(n, f.t);
  return f;
}
__device__ __forceinline__ v3 to_local(const Frame& f, v3 v) { return crh_mk3(crh_dot3(v, f.t), crh_dot3(v, f.b), crh_dot3(v, f.n)); }
__device__ __forceinline__ v3 from_local(const Frame& f, v3 l)
{
  return crh_mk3(CRH_FMA(f.n.x, l.z, CRH_FMA(f.b.x, l.y, f.t.x * l.x)),
                 CRH_FMA(f.n.y, l.z, CRH_FMA(f.b.y, l.y, f.t.y * l.x)),
                 CRH_FMA(f.n.z, l.z, CRH_FMA(f.b.z, l.y, f.t.z * l.x)));
}
__device__ __forceinline__ float lerpf(float a, float b, float t) { return CRH_FMA(t, b - a, a); }

__device__ v3 env_lookup(const DScene& S, v3 d)
{
  if (!S.env) return crh_mk3(S.bg[0], S.bg[1], S.bg[2]);
  float u = (crh_atan2(d.y, d.x) + CRH_PI) * CRH_INV_TWOPI;
  float v = crh_acos(d.z) * CRH_INV_PI;
  if (S.spec_env_orient) { u = crh_atan2(d.y, d.x) * CRH_INV_TWOPI; v = crh_acos(-d.z) * CRH_INV_PI; }      // crh_spec.h #14
  const float x = CRH_FMA(u, (float)S.env_w, -0.5f), y = CRH_FMA(v, (float)S.env_h, -0.5f);
  float xf = (float)(int)x; if (xf > x) xf -= 1.0f;
  float yf = (float)(int)y; if (yf > y) yf -= 1.0f;
  const float fx = x - xf, fy = y - yf;
  const int W = (int)S.env_w, H = (int)S.env_h;
  int x0 = (int)xf % W; if (x0 < 0) x0 += W;
  int x1 = x0 + 1; if (x1 >= W) x1 = 0;
  int y0 = (int)yf; int y1 = y0 + 1;
  if (y0 < 0) y0 = 0; if (y0 > H - 1) y0 = H - 1; if (y1 < 0) y1 = 0; if (y1 > H - 1) y1 = H - 1;
  const float4 p00 = S.env[y0 * W + x0], p10 = S.env[y0 * W + x1], p01 = S.env[y1 * W + x0], p11 = S.env[y1 * W + x1];
  v3 r = crh_mk3(lerpf(lerpf(p00.x, p10.x, fx), lerpf(p01.x, p11.x, fx), fy),
                 lerpf(lerpf(p00.y, p10.y, fx), lerpf(p01.y, p11.y, fx), fy),
                 lerpf(lerpf(p00.z, p10.z, fx), lerpf(p01.z, p11.z, fx), fy));
  if (S.spec_gamma2) r = crh_mul3(r, r);            // crh_spec.h #2: the filtered texel squared
  return r;
}

__device__ __forceinline__ float cone_pdf(float cosmax) { return 1.0f / (CRH_TWO_PI * (1.0f - cosmax)); }
__device__ __forceinline__ float sphere_cosmax(float radius, float dist)
{ const float q = radius / dist; return 1.0f / crh_sqrt(CRH_FMA(q, q, 1.0f)); }

__device__ v3 intersect_light(const DScene& S, v3 o, v3 d, uint32_t bounce, float hit_t, float& exp_pdf)
{
  v3 rad = crh_mk3(0.f, 0.f, 0.f); float pdf = 0.f; float hd = hit_t;
  const float sel = S.n_lights ? 1.0f / (float)S.n_lights : 0.f;
  for (uint32_t i = 0; i < S.n_lights; ++i) {
    const float4 l0 = S.lights[2u * i], l1 = S.lights[2u * i + 1u];
    if (l0.w != 0.f) {
      const v3 tl = crh_sub3(xyz(l0), o);
      const float dist = crh_len3(tl);
      if (dist < hd) {
        const float cm = sphere_cosmax(l1.w, dist);
        if (cm < 1.0f && crh_dot3(d, tl) * (1.0f / dist) >= cm) { hd = dist; rad = xyz(l1); pdf = sel * cone_pdf(cm); }
      }
    } else if (hd == CRH_MAXFLOAT) {
      const float cm = l1.w;
      if (cm < 1.0f && crh_dot3(d, xyz(l0)) >= cm) { rad = crh_add3(rad, xyz(l1)); pdf += sel * cone_pdf(cm); }
    }
  }
  if (pdf == 0.f && hd == CRH_MAXFLOAT) {
    if (bounce == 0u && !S.env_as_bg) rad = crh_mk3(S.bg[0], S.bg[1], S.bg[2]);
    else rad = env_lookup(S, d);
  }
  exp_pdf = pdf;
  return rad;
}

__device__ __forceinline__ v3 offset_origin(v3 p, v3 dir, v3 ng, float eps)
{
  const v3 o = crh_madd3(p, dir, eps);
  const float s = crh_dot3(ng, dir) >= 0.f ? eps : -eps;
  return crh_madd3(o, ng, s);
}

// Diffuse texture lookup: bilinear, repeat wrap, row 0 of the image = v 1.  The call site is behind
// a wave-uniform "any texture bound" test.
__device__ __forceinline__ float4 sample_texture(const DScene& S, uint32_t slot, uint32_t tri, float bu, float bv, float w0, float sc_s, float sc_t)
{
  if (slot >= S.n_tex || !S.uvs) return make_float4(1.f, 1.f, 1.f, 1.f);
  const uint4 td = S.tex_desc[slot];
  if (td.y == 0u) return make_float4(1.f, 1.f, 1.f, 1.f);
  const float4 ua = S.uvs[2u * tri], ub = S.uvs[2u * tri + 1u];
  const float ss = sc_s != 0.f ? sc_s : 1.0f, st_ = sc_t != 0.f ? sc_t : 1.0f;
  float us = CRH_FMA(ub.x, bv, CRH_FMA(ua.z, bu, ua.x * w0)) * ss;
  float vs = CRH_FMA(ub.y, bv, CRH_FMA(ua.w, bu, ua.y * w0)) * st_;
  // beyond 2^22 a float has no fraction left worth sampling and (int) would saturate (texel index out of range): wrap to 0
  if (!(crh_abs(us) < 4194304.0f)) us = 0.f;
  if (!(crh_abs(vs) < 4194304.0f)) vs = 0.f;
  float uf = (float)(int)us; if (uf > us) uf -= 1.0f;
  float vf = (float)(int)vs; if (vf > vs) vf -= 1.0f;
  const float x = CRH_FMA(us - uf, (float)td.y, -0.5f), y = CRH_FMA(1.0f - (vs - vf), (float)td.z, -0.5f);
  float xf = (float)(int)x; if (xf > x) xf -= 1.0f;
  float yf = (float)(int)y; if (yf > y) yf -= 1.0f;
  const float fx = x - xf, fy = y - yf;
  const int W = (int)td.y, H = (int)td.z;
  int x0 = (int)xf; if (x0 < 0) x0 += W; if (x0 >= W) x0 -= W;
  int x1 = x0 + 1; if (x1 >= W) x1 = 0;
  int y0 = (int)yf; if (y0 < 0) y0 += H; if (y0 >= H) y0 -= H;
  int y1 = y0 + 1; if (y1 >= H) y1 = 0;
  const float4* tb = S.texels + td.x;
  const float4 p00 = tb[y0 * W + x0], p10 = tb[y0 * W + x1], p01 = tb[y1 * W + x0], p11 = tb[y1 * W + x1];
  float4 r = make_float4(lerpf(lerpf(p00.x, p10.x, fx), lerpf(p01.x, p11.x, fx), fy),
                         lerpf(lerpf(p00.y, p10.y, fx), lerpf(p01.y, p11.y, fx), fy),
                         lerpf(lerpf(p00.z, p10.z, fx), lerpf(p01.z, p11.z, fx), fy),
                         lerpf(lerpf(p00.w, p10.w, fx), lerpf(p01.w, p11.w, fx), fy));   // RGB images are stored with alpha 1
  if (S.spec_gamma2) { r.x *= r.x; r.y *= r.y; r.z *= r.z; }      // crh_spec.h #2 (the alpha is a coverage, never squared)
  return r;
}

constexpr uint32_t kGenIters = 32;     // k_raygen: 32 x 256 = 8192 path slots per queue reservation
static_assert(kGenIters * 4 == 128, "k_raygen scans its 128 (iteration, wave) counters with one wavefront, two per lane");
#ifndef CRH_SHADE_ITERS
#define CRH_SHADE_ITERS 4
#endif
constexpr uint32_t kShadeIters = CRH_SHADE_ITERS;    // k_shade : 4 x 256 = 1024 paths per cursor fetch / queue reservation

// ================================================================== path slot <-> pixel
// Slot layout inside one sample: tile-major, and inside a tile 8x8-pixel blocks so that one wavefront
// owns one 8x8 block (coherent primary rays, coalesced accumulator rows of 8 float4 = 128 B).
__device__ __forceinline__ bool slot_pixel(const DScene& S, const uint32_t* __restrict__ tile_ids, uint32_t local,
                                           uint32_t& px, uint32_t& py)
{
  const uint32_t ts = S.tile_size, tpp = ts * ts;
  const uint32_t ti = local / tpp, off = local - ti * tpp;
  const uint32_t tile = tile_ids ? tile_ids[ti] : ti;
  const uint32_t tx = (S.width + ts - 1u) / ts, ty = (S.height + ts - 1u) / ts;
  const uint32_t blk = off >> 6, l = off & 63u, bpr = ts >> 3;
  px = (tile % tx) * ts + (blk % bpr) * 8u + (l & 7u);
  py = (tile / tx) * ts + (blk / bpr) * 8u + (l >> 3);
  return tile < tx * ty && px < S.width && py < S.height;
}

// Path slot <-> (pixel slot, sample) inside one batch of `ns` samples.  An 8x8 pixel block owns 64 * ns consecutive slots = ns
// wavefronts.  With G = the largest power of two that divides ns (at most 64), a wavefront holds 64 / G pixels x G consecutive samples:
// ns = 1 (one Redraw): the 8x8 block, as ever; ns = 128 (the batch bench.py times): ONE pixel x 64 samples -- camera rays that differ
// only by their sub-pixel jitter walk the tree in lockstep and shade the same triangle (lane utilisation of the first launches), and
// the wavefronts in flight cover a few thousand pixels instead of a fifth of the image.  Only the ORDER of the slots changes: every
// path still owns (pixel, sample), seeds and per-pixel accumulation order are untouched, results are bit-identical.
#ifndef CRH_SAMPLE_GROUP_MAX
#define CRH_SAMPLE_GROUP_MAX 64
#endif
#ifndef CRH_SLOT_SAMPLE_MAJOR
#define CRH_SLOT_SAMPLE_MAJOR 0      // lane = pixel * G + sample (0) or sample * P + pixel (1) inside a wavefront's 64 slots
#endif
__device__ __forceinline__ uint32_t sample_group(uint32_t ns) { return min(ns & (0u - ns), (uint32_t)CRH_SAMPLE_GROUP_MAX); }
__device__ __forceinline__ void slot_to_pixel_sample(uint32_t pid, uint32_t ns, uint32_t& local, uint32_t& s)
{
  const uint32_t G = sample_group(ns), lg = 31u - (uint32_t)__clz((int)G);      // G is a power of two
  const uint32_t B = pid / (64u * ns), r = pid - B * 64u * ns, w = r >> 6, l = r & 63u;
  const uint32_t P = 64u >> lg;
  const uint32_t c = w >> lg, b = w & (G - 1u), pi_ = CRH_SLOT_SAMPLE_MAJOR ? l & (P - 1u) : l >> lg, si = CRH_SLOT_SAMPLE_MAJOR ? l / P : l & (G - 1u);
  local = B * 64u + b * P + pi_;
  s = (c << lg) + si;
}
__device__ __forceinline__ uint32_t pixel_sample_to_slot(uint32_t local, uint32_t s, uint32_t ns)
{
  const uint32_t G = sample_group(ns), lg = 31u - (uint32_t)__clz((int)G), P = 64u >> lg;
  const uint32_t B = local >> 6, p = local & 63u, b = p / P, pi_ = p - b * P, c = s >> lg, si = s & (G - 1u);
  return B * 64u * ns + (((c << lg) + b) << 6) + (CRH_SLOT_SAMPLE_MAJOR ? si * P + pi_ : (pi_ << lg) + si);
}

// SPLIT: the instantiation for split scenes (static tree + moved objects) also lists the rays that touch a moved object; the plain one carries none of it
template <bool SPLIT>
__global__ __launch_bounds__(kBlock) void k_raygen(DScene S, DPaths P, uint32_t* __restrict__ q, uint32_t* __restrict__ count,
                                                    uint32_t* __restrict__ q2, uint32_t* __restrict__ count2,
                                                    uint32_t* __restrict__ cursors,
                                                    const uint32_t* __restrict__ tile_ids, uint32_t n_tiles,
                                                    const uint32_t* __restrict__ seeds, uint32_t n_samples, int seed_per_tile,
                                                    const uint32_t* __restrict__ n_tiles_dev)
{
  if (n_tiles_dev) n_tiles = *n_tiles_dev;          // the tile list was drawn on the device (adaptive sampling): its length lives there too
  // Queue space is reserved ONCE per chunk of kGenIters x 256 slots: pass 1 counts the slots that map to a pixel
  // inside the image (edge tiles are partial), one atomic reserves the range, pass 2 generates the rays and writes
  // their ids at exclusive-scan offsets.  (Per-workgroup appends were atomic-rate bound: 261 K atomics per 67 M paths.)
  __shared__ uint32_t s_cnt[kGenIters * 4];
  __shared__ uint32_t s_base;
  // split scenes: the camera rays that touch a moved object are listed for the second traversal pass (collected per chunk, one atomic per chunk)
  __shared__ uint32_t s_q2[SPLIT ? kGenIters * kBlock : 1];
  __shared__ uint32_t s_n2, s_b2;
  if (blockIdx.x == 0 && threadIdx.x == 0) { cursors[0] = 0u; cursors[1] = 0u; cursors[2] = 0u; cursors[4] = 0u; cursors[5] = 0u; cursors[7] = 0u; cursors[8] = 0u; }      // [7], [8]: count and cursor of the packet kernel's fall-back queue
  if (threadIdx.x == 0) s_n2 = 0u;
  const uint32_t per_sample = n_tiles * S.tile_size * S.tile_size;
  const uint32_t total = per_sample * n_samples;
  const uint32_t chunk = kGenIters * kBlock;
  const uint32_t lane = lane_id(), wv = threadIdx.x >> 6;
  for (uint32_t cbase = blockIdx.x * chunk; cbase < total; cbase += gridDim.x * chunk) {
    for (uint32_t it = 0; it < kGenIters; ++it) {
      const uint32_t pid = cbase + it * kBlock + threadIdx.x;
      bool valid = pid < total;
      uint32_t px, py;
      if (valid) { uint32_t local, s; slot_to_pixel_sample(pid, n_samples, local, s); valid = slot_pixel(S, tile_ids, local, px, py); }
      const unsigned long long m = __ballot(valid);
      if (lane == 0) s_cnt[it * 4u + wv] = (uint32_t)__popcll(m);
    }
    __syncthreads();
    if (wv == 0) {                                  // exclusive scan of the kGenIters*4 (= 128) counters by one wavefront
      const uint32_t a = s_cnt[2u * lane], b = s_cnt[2u * lane + 1u];
      uint32_t incl = a + b;
      for (int o = 1; o < 64; o <<= 1) { const uint32_t t = __shfl_up(incl, o); if ((int)lane >= o) incl += t; }
      const uint32_t excl = incl - (a + b);
      s_cnt[2u * lane] = excl; s_cnt[2u * lane + 1u] = excl + a;
      if (lane == 63) s_base = incl ? atomicAdd(count, incl) : 0u;
    }
    __syncthreads();
    for (uint32_t it = 0; it < kGenIters; ++it) {
    const uint32_t pid = cbase + it * kBlock + threadIdx.x;
    bool valid = pid < total, flagged = false;
    uint32_t px = 0, py = 0, s = 0, local = 0;
    if (valid) { slot_to_pixel_sample(pid, n_samples, local, s); valid = slot_pixel(S, tile_ids, local, px, py); }
    if (valid) {
      const uint32_t pix = S.coherent ? ((py / 16u) * ((S.width + 15u) / 16u) + (px / 16u)) : (py * S.width + px);
      // whole-frame passes share one frame seed per sample; adaptive passes give every tile its own sample index
      const uint32_t fseed = seed_per_tile ? seeds[local / (S.tile_size * S.tile_size)] : seeds[s];
      uint32_t rng = crh_rng_seed(pix, fseed);
      const float jx = crh_rng_next_mode(&rng, S.spec_u32), jy = crh_rng_next_mode(&rng, S.spec_u32);
      const float nx = CRH_FMA(((float)px + jx) / (float)S.width, 2.0f, -1.0f);
      const float ny = CRH_FMA(((float)py + jy) / (float)S.height, -2.0f, 1.0f);
      v3 o, d;
      if (S.is_ortho) {
        const float sx = (nx * S.ortho_scale) * S.aspect, sy = ny * S.ortho_scale;
        o = crh_madd3(crh_madd3(S.eye, S.right, sx), S.up, sy);
        d = S.fwd;
      } else if (S.spec_raygen) {
        // crh_spec.h #13 (SURVEY a2, Appendix A GenerateRay): blend of the four frustum-corner directions by the pixel's position in [0,1]^2
        const float u = ((float)px + jx) / (float)S.width, v = 1.0f - ((float)py + jy) / (float)S.height;
        o = S.eye;
        d = crh_norm3(crh_lerp3(crh_lerp3(S.corner[0], S.corner[1], u), crh_lerp3(S.corner[2], S.corner[3], u), v));
      } else {
        const float sx = (nx * S.tan_half) * S.aspect, sy = ny * S.tan_half;
        o = S.eye;
        d = crh_norm3(crh_madd3(crh_madd3(S.fwd, S.right, sx), S.up, sy));
      }
      if (S.aperture > 0.f) {
        const float k1 = crh_rng_next_mode(&rng, S.spec_u32), k2 = crh_rng_next_mode(&rng, S.spec_u32);
        const float ft = S.focal / crh_dot3(d, S.fwd);
        const v3 focus = crh_madd3(o, d, ft);
        const float r = S.aperture * crh_sqrt(k1); float sn, cs; crh_sincos2pi(k2, &sn, &cs);
        o = crh_madd3(crh_madd3(o, S.right, r * cs), S.up, r * sn);
        d = crh_norm3(crh_sub3(focus, o));
      }
      P.ray_o[0][pid] = mk4(o, __uint_as_float(rng));           // .w = rng state; position = path slot at bounce 0
      P.ray_d[0][pid] = mk4(d, __uint_as_float(pid << 1));      // .w = (path slot << 1) | inside-a-medium flag
      if (SPLIT) flagged = ray_touches_instances(S, o, d, CRH_MAXFLOAT);
      // throughput (1,1,1 | no pending pdf) and radiance (0) are NOT written here: every generated path goes through
      // the bounce-0 k_shade, which takes them as constants and writes the radiance record unconditionally
    }
    const unsigned long long m = __ballot(valid);
    if (valid) q[s_base + s_cnt[it * 4u + wv] + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = pid;
    if (SPLIT) lds_append(flagged, pid, s_q2, &s_n2);
    }
    __syncthreads();
    if (SPLIT) {
      if (threadIdx.x == 0) { s_b2 = s_n2 ? atomicAdd(count2, s_n2) : 0u; }
      __syncthreads();
      for (uint32_t j = threadIdx.x; j < s_n2; j += kBlock) q2[s_b2 + j] = s_q2[j];
      __syncthreads();
      if (threadIdx.x == 0) s_n2 = 0u;
    }
  }
}

// ================================================================== shade
constexpr int kLdsMats = 64;   // materials staged in LDS (8 KB); larger tables are read from HBM/L2

// workgroups per CU the register allocation must allow: 4 => <= 128 VGPRs (143 unconstrained, 3 waves/SIMD).  Measured, Mrays/s at
// 0 / 4 / 5: C3 3457 / 3474 / 3200, C2 4803 / 4856 / 4569, C5 2416 / 2423, C1 13390 / 13765 (5 = 96 VGPRs spills the BSDF code)
#ifndef CRH_SHADE_MINWAVES
#define CRH_SHADE_MINWAVES 4
#endif
#if CRH_SHADE_MINWAVES > 0
#define CRH_SHADE_BOUNDS __launch_bounds__(kBlock, CRH_SHADE_MINWAVES)
#else
#define CRH_SHADE_BOUNDS __launch_bounds__(kBlock)
#endif
template <bool SPLIT>
__global__ CRH_SHADE_BOUNDS void k_shade(DScene S, DPaths P, int cur, uint32_t bounce,
                                                   const uint32_t* __restrict__ q_in, const uint32_t* __restrict__ count_in,
                                                   uint32_t* __restrict__ q_out, uint32_t* __restrict__ count_out,
                                                   uint32_t* __restrict__ q_sh, uint32_t* __restrict__ count_sh,
                                                   uint32_t* __restrict__ q2, uint32_t* __restrict__ count2,
                                                   uint32_t* __restrict__ q2_sh, uint32_t* __restrict__ count2_sh,
                                                   uint32_t* __restrict__ cursors, DCounters* C)
{
  __shared__ float4 s_mats[kLdsMats * 8];
  // survivors / shadow rays of kShadeIters x 256 paths are collected in LDS and appended with ONE global atomic each
  __shared__ uint32_t s_qc[kShadeIters * kBlock], s_qs[kShadeIters * kBlock];
  // split scenes: those of them that touch a moved object, for the second traversal pass
  __shared__ uint32_t s_q2c[SPLIT ? kShadeIters * kBlock : 1], s_q2s[SPLIT ? kShadeIters * kBlock : 1];
  __shared__ uint32_t s_base, s_nc, s_ns, s_gc, s_gs, s_n2c, s_n2s, s_g2c, s_g2s;
  if (blockIdx.x == 0 && threadIdx.x == 0) { cursors[0] = 0u; cursors[4] = 0u; }     // nearest-hit cursors (both passes) of the next bounce
  const bool mats_in_lds = S.n_mats <= (uint32_t)kLdsMats;
  if (mats_in_lds) {
    for (uint32_t i = threadIdx.x; i < S.n_mats * 8u; i += kBlock) s_mats[i] = S.mats[i];
    __syncthreads();
  }
  const uint32_t n = *count_in;
  const bool last = bounce + 1u >= S.max_depth;
  const bool first = bounce == 0u;
  const float4* __restrict__ in_o = P.ray_o[cur]; const float4* __restrict__ in_d = P.ray_d[cur]; const float4* __restrict__ in_t = P.thr[cur];
  float4* __restrict__ out_o = P.ray_o[1 - cur]; float4* __restrict__ out_d = P.ray_d[1 - cur]; float4* __restrict__ out_t = P.thr[1 - cur];
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
  uint32_t n_shaded = 0;
  for (;;) {
    __syncthreads();
    if (threadIdx.x == 0) { s_base = atomicAdd(cursors + 1, kShadeIters * (uint32_t)kBlock); s_nc = 0u; s_ns = 0u; s_n2c = 0u; s_n2s = 0u; }
    __syncthreads();
    const uint32_t base = s_base;
    if (base >= n) break;
#pragma unroll 1
    for (uint32_t it = 0; it < kShadeIters; ++it) {
    const uint32_t i = base + it * kBlock + threadIdx.x;
    bool cont = false, shadow = false;
    float4 n_o = zero4, n_d = zero4, n_t = zero4, s_o = zero4, s_d = zero4, s_c = zero4;     // successor ray / shadow ray, stored after the ranks are known
    if (i < n) {
      const uint32_t pos = q_in[i];
      const float4 o4 = in_o[pos], d4 = in_d[pos], h = P.hit[pos];
      const float4 t4 = first ? make_float4(1.0f, 1.0f, 1.0f, CRH_MAXFLOAT) : in_t[pos];     // k_raygen leaves thr / rad unwritten
      const uint32_t pid = __float_as_uint(d4.w) >> 1;                              // the path's slot (radiance record, pixel)
      const uint2 st = make_uint2(__float_as_uint(o4.w), __float_as_uint(d4.w) & 1u);   // rng state, flags
      const v3 o = xyz(o4), d = xyz(d4);
      v3 W = xyz(t4); float imp_pdf = t4.w;
      const int hk = __float_as_int(h.w);
      const bool found = hk >= 0;
      float exp_pdf;
      const v3 le = intersect_light(S, o, d, bounce, found ? h.x : CRH_MAXFLOAT, exp_pdf);
      if (le.x > 0.f || le.y > 0.f || le.z > 0.f || !found) {
        const float mis = (bounce == 0u || imp_pdf == CRH_MAXFLOAT) ? 1.0f : (imp_pdf * imp_pdf) / CRH_FMA(exp_pdf, exp_pdf, imp_pdf * imp_pdf);
        float4 r4 = first ? zero4 : P.rad[pid];              // the radiance record is touched only when something is added
        if (__float_as_uint(r4.w) != P.stamp) r4 = zero4;    // ... and holds this batch's stamp once it has been (DPaths::stamp): anything else reads as zero
        P.rad[pid] = mk4(crh_add3(xyz(r4), crh_scale3(crh_mul3(W, le), mis)), __uint_as_float(P.stamp));
      } else {
        ++n_shaded;
        // shading record: one 64-B sector {n0 | material, n1 | instance, n2, geometric normal}.  The geometric normal of a
        // single-level scene is precomputed on the host with the same inline arithmetic (crh_math.h) the kernel used to apply to
        // the three vertices -- same bits -- so shading no longer gathers the 48-B triangle record (a second 128-B line per hit);
        // a two-level scene still needs the vertices: the normal is taken from the TRANSFORMED corners
        const float4* sp = S.shade + 4u * (uint32_t)hk;
        const float4 s0 = sp[0], s1 = sp[1], s2 = sp[2];
        float M[12];
        v3 ng;
        // two-level scenes: s1.w = the object of a triangle that lives in an object tree (-1: a triangle of the static world-space tree)
        const bool in_object = S.two_level && __float_as_int(s1.w) >= 0;
        if (in_object) {                                     // object -> world through the instance's forward transform
          const float4* tp = S.verts + 3u * (uint32_t)hk;       // the three object-space vertices (the traversal record holds edges, not vertices)
          const float4 a = tp[0], b4 = tp[1], c4 = tp[2];
          const float4* ip = S.inst + 8u * (uint32_t)__float_as_int(s1.w);
          const float4 f0 = ip[3], f1 = ip[4], f2 = ip[5];
          M[0] = f0.x; M[1] = f0.y; M[2] = f0.z; M[3] = f0.w; M[4] = f1.x; M[5] = f1.y; M[6] = f1.z; M[7] = f1.w;
          M[8] = f2.x; M[9] = f2.y; M[10] = f2.z; M[11] = f2.w;
          const v3 p0 = crh_xform_point(M, xyz(a)), p1 = crh_xform_point(M, xyz(b4)), p2 = crh_xform_point(M, xyz(c4));
          ng = crh_norm3(crh_cross3(crh_sub3(p0, p2), crh_sub3(p1, p0)));
        } else ng = xyz(sp[3]);
        const float w0 = (1.0f - h.y) - h.z;
        v3 ns = crh_norm3(crh_mk3(CRH_FMA(s2.x, h.z, CRH_FMA(s1.x, h.y, s0.x * w0)),
                                  CRH_FMA(s2.y, h.z, CRH_FMA(s1.y, h.y, s0.y * w0)),
                                  CRH_FMA(s2.z, h.z, CRH_FMA(s1.z, h.y, s0.z * w0))));
        if (in_object) ns = crh_norm3(crh_xform_vector(M, ns));
        if (!(crh_dot3(ns, ns) > 0.f)) ns = ng;
        const v3 p = crh_madd3(o, d, h.x);
        int mat = __float_as_int(s0.w); if (mat < 0 || (uint32_t)mat >= S.n_mats) mat = 0;
        const float4* mp = mats_in_lds ? (s_mats + 8 * mat) : (S.mats + 8 * mat);
        Bsdf bs;
        { const float4 m0 = mp[0], m1 = mp[1], m2 = mp[2], m3 = mp[3], m4 = mp[4];
          bs.Kc = xyz(m0); bs.Rc = m0.w; bs.Kd = xyz(m1); bs.Ks = xyz(m2); bs.Rs = m2.w; bs.Kt = xyz(m3); bs.Le = xyz(m4);
          bs.ab = mp[5]; bs.fc = mp[6]; bs.fb = mp[7]; }
        if (S.n_tex != 0u) {                                   // wave-uniform: scenes without textures skip the call
          const int slot = (int)mp[1].w - 1;
          if (slot >= 0) {
            const float4 tx = sample_texture(S, (uint32_t)slot, (uint32_t)hk, h.y, h.z, w0, mp[3].w, mp[4].w);
            bs.Kd = crh_mul3(bs.Kd, xyz(tx));
            if (tx.w != 1.0f) {                                // alpha cut-out: the uncovered part transmits
              bs.Kd = crh_scale3(bs.Kd, tx.w);
              const float ia = 1.0f - tx.w;
              bs.Kt = crh_mk3(CRH_FMA(tx.w, bs.Kt.x, ia), CRH_FMA(tx.w, bs.Kt.y, ia), CRH_FMA(tx.w, bs.Kt.z, ia));
            }
          }
        }
        const Frame fr = make_frame(ns);
        const v3 wo = to_local(fr, crh_mk3(-d.x, -d.y, -d.z));
        bs.Fc = fresnel_media(wo.z, bs.fc);
        bool inside = (st.y & 1u) != 0u;
        if (inside) {
          const float k = -(h.x * bs.ab.w);
          W = crh_mul3(W, crh_mk3(crh_exp(k * (1.0f - bs.ab.x)), crh_exp(k * (1.0f - bs.ab.y)), crh_exp(k * (1.0f - bs.ab.z))));
        }
        if (bs.Le.x != 0.f || bs.Le.y != 0.f || bs.Le.z != 0.f) {      // emissive surfaces are rare: skip the read-modify-write otherwise
          float4 r4 = first ? zero4 : P.rad[pid];
          if (__float_as_uint(r4.w) != P.stamp) r4 = zero4;
          P.rad[pid] = mk4(crh_add3(xyz(r4), crh_mul3(W, bs.Le)), __uint_as_float(P.stamp));
        }                                                              // nobody initialises the record: an unstamped one reads as zero
        uint32_t rng = st.x;
        // ---- next event estimation
        {
          const v3 z3 = crh_mk3(0.f, 0.f, 0.f);
          const v3 nd = crh_add3(bs.Kd, crh_add3(bs.Rc > kBsdfEps ? bs.Kc : z3, bs.Rs > kBsdfEps ? bs.Ks : z3));
          if (S.n_lights > 0u && crh_dot3(nd, W) > kBsdfEps) {
            const float fl = crh_rng_next_mode(&rng, S.spec_u32) * (float)S.n_lights;
            uint32_t li = (uint32_t)fl; if (li > S.n_lights - 1u) li = S.n_lights - 1u;
            const float k1 = crh_rng_next_mode(&rng, S.spec_u32), k2 = crh_rng_next_mode(&rng, S.spec_u32);
            const float4 l0 = S.lights[2u * li], l1 = S.lights[2u * li + 1u];
            v3 axis; float dist, cm;
            if (l0.w != 0.f) { const v3 tl = crh_sub3(xyz(l0), p); dist = crh_len3(tl); axis = crh_scale3(tl, 1.0f / dist); cm = sphere_cosmax(l1.w, dist); }
            else { axis = xyz(l0); dist = CRH_MAXFLOAT; cm = l1.w; }
            const Frame lf = make_frame(axis);
            const float ct = CRH_FMA(-k2, 1.0f - cm, 1.0f);
            float sn, cs; crh_sincos2pi(k1, &sn, &cs);
            const float sq = crh_sqrt(crh_max(CRH_FMA(-ct, ct, 1.0f), 0.f));
            const v3 ld = crh_norm3(from_local(lf, crh_mk3(cs * sq, sn * sq, ct)));
            const float e_pdf = (cm < 1.0f) ? (1.0f / (float)S.n_lights) * cone_pdf(cm) : CRH_MAXFLOAT;
            const v3 wi = to_local(fr, ld);
            const float i_pdf = pdf_layered(bs, wo, wi, W, S.two_sided);
            const float mis = (e_pdf == CRH_MAXFLOAT) ? 1.0f : e_pdf / CRH_FMA(e_pdf, e_pdf, i_pdf * i_pdf);
            const v3 contrib = crh_scale3(crh_mul3(xyz(l1), eval_layered(bs, wi, wo, S.two_sided)), mis);
            const v3 wc = crh_mul3(W, contrib);
            if (contrib.x > S.spec_min_contrib || contrib.y > S.spec_min_contrib || contrib.z > S.spec_min_contrib) {      // crh_spec.h #11
              shadow = true;
              s_o = mk4(offset_origin(p, ld, ng, S.eps), dist);
              // split scenes: .w != 0 marks a shadow ray that touches a moved object (the first any-hit pass leaves its contribution to the second)
              s_d = mk4(ld, (SPLIT && ray_touches_instances(S, xyz(s_o), ld, dist)) ? 1.0f : 0.f);
              s_c = mk4(wc, __uint_as_float(pid));
            }
          }
        }
        // ---- BSDF sampling + Russian roulette (the last bounce has no successor ray)
        if (!last) {
          v3 wi; bool delta; const v3 Wsel = W; int lobe;
          const bool alive = sample_layered(bs, wo, wi, W, inside, delta, rng, S.two_sided, SpecB{S.spec_u32, S.spec_eta_nd}, lobe);
          if (alive) imp_pdf = delta ? CRH_MAXFLOAT : pdf_layered(bs, wo, wi, Wsel, S.two_sided, S.spec_mis1 ? lobe : -1);
          const bool roulette = S.rr && bounce >= S.spec_rr_start;      // crh_spec.h #9, #10, #12
          float survive = (W.x > S.spec_min_thr || W.y > S.spec_min_thr || W.z > S.spec_min_thr) ? 1.0f : 0.f;
          if (roulette)
            survive = crh_min(CRH_FMA(0.0722f, W.z, CRH_FMA(0.7152f, W.y, 0.2126f * W.x)), S.spec_rr_cap) * survive;
          const float kr = crh_rng_next_mode(&rng, S.spec_u32);
          if (alive && kr < survive) {
            if (roulette) W = crh_mk3(W.x / survive, W.y / survive, W.z / survive);
            const v3 nd2 = crh_norm3(from_local(fr, wi));
            n_o = mk4(offset_origin(p, nd2, ng, S.eps), __uint_as_float(rng));
            n_d = mk4(nd2, __uint_as_float((pid << 1) | (inside ? 1u : 0u)));
            n_t = mk4(W, imp_pdf);
            cont = true;
          }
        }
      }
    }
    // The r-th shadow ray / survivor of this chunk takes the position of the chunk's r-th input entry (in the other ray buffer for
    // survivors): positions stay packed in runs, no two chunks ever share one, and no global atomic is needed to find them.
    if (S.n_lights > 0u) {
      const uint32_t r = lds_rank(shadow, &s_ns);
      uint32_t ps = 0u;
      if (shadow) { ps = q_in[base + r]; s_qs[r] = ps; P.sh_o[ps] = s_o; P.sh_d[ps] = s_d; P.sh_c[ps] = s_c; }
      if (SPLIT) lds_append(shadow && s_d.w != 0.f, ps, s_q2s, &s_n2s);
    }
    {
      const uint32_t r = lds_rank(cont, &s_nc);
      uint32_t pn = 0u;
      if (cont) { pn = q_in[base + r]; s_qc[r] = pn; out_o[pn] = n_o; out_d[pn] = n_d; out_t[pn] = n_t; }
      if (SPLIT) lds_append(cont && ray_touches_instances(S, xyz(n_o), xyz(n_d), CRH_MAXFLOAT), pn, s_q2c, &s_n2c);
    }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      s_gc = s_nc ? atomicAdd(count_out, s_nc) : 0u; s_gs = s_ns ? atomicAdd(count_sh, s_ns) : 0u;
      if (SPLIT) { s_g2c = s_n2c ? atomicAdd(count2, s_n2c) : 0u; s_g2s = s_n2s ? atomicAdd(count2_sh, s_n2s) : 0u; }
    }
    __syncthreads();
    for (uint32_t j = threadIdx.x; j < s_nc; j += kBlock) q_out[s_gc + j] = s_qc[j];
    for (uint32_t j = threadIdx.x; j < s_ns; j += kBlock) q_sh[s_gs + j] = s_qs[j];
    if (SPLIT) {
      for (uint32_t j = threadIdx.x; j < s_n2c; j += kBlock) q2[s_g2c + j] = s_q2c[j];
      for (uint32_t j = threadIdx.x; j < s_n2s; j += kBlock) q2_sh[s_g2s + j] = s_q2s[j];
    }
  }
  n_shaded = wave_sum(n_shaded);
  if (lane_id() == 0 && n_shaded) atomicAdd(&C->shaded_hits, (unsigned long long)n_shaded);
}

// ================================================================== accumulate / display
__global__ __launch_bounds__(kBlock) void k_accumulate(DScene S, DPaths P, float4* __restrict__ accum, float* __restrict__ m2,
                                                        const uint32_t* __restrict__ tile_ids, uint32_t n_tiles,
                                                        uint32_t first_sample, uint32_t n_samples, uint32_t batch_samples, DCounters* C,
                                                        const uint32_t* __restrict__ n_tiles_dev)
{
  if (n_tiles_dev) n_tiles = *n_tiles_dev;
  // samples [first_sample, first_sample + n_samples) of the batch of batch_samples in the path buffer are folded in, in order
  const uint32_t per_sample = n_tiles * S.tile_size * S.tile_size;
  const uint32_t last = first_sample + n_samples;
  uint32_t done = 0;
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f); float q = 0.f;
  auto fold = [&](const float4 r) {
    const float w = 1.0f / (a.w + 1.0f);
    const bool written = __float_as_uint(r.w) == P.stamp;      // a path that never added anything left its record alone: zero radiance (DPaths::stamp)
    float v[3] = {written ? r.x : 0.f, written ? r.y : 0.f, written ? r.z : 0.f};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      if (!(v[k] == v[k])) v[k] = 0.f;
      if (S.clampv > 0.f && v[k] > S.clampv) v[k] = S.clampv;
    }
    a.x = CRH_FMA(v[0] - a.x, w, a.x);
    a.y = CRH_FMA(v[1] - a.y, w, a.y);
    a.z = CRH_FMA(v[2] - a.z, w, a.z);
    a.w = a.w + 1.0f;
    if (m2) {      // running mean of the squared luminance (adaptive sampling's variance estimate)
      const float l = CRH_FMA(0.0722f, v[2], CRH_FMA(0.7152f, v[1], 0.2126f * v[0]));
      q = CRH_FMA(l * l - q, w, q);
    }
    ++done;
  };
  if (sample_group(batch_samples) >= 8u) {
    // Wide batches: a pixel's samples sit in runs of >= 8 consecutive slots (one 128-B line), the pixels of a block far apart -- a
    // lane walking its own pixel would touch 64 lines per load.  A wavefront therefore takes one 8x8 block, fetches 64 pixels x 8
    // samples with eight coalesced loads (eight whole lines each) into LDS, and every lane folds ITS pixel's eight samples from
    // there, in sample order: the same arithmetic in the same order (4.05 -> ~1 ms per 128-spp step at 1080p).
    __shared__ float4 s_tile[4][64 * 9];
    const uint32_t wave = threadIdx.x >> 6, lane = lane_id();
    float4* tile = s_tile[wave];
    const uint32_t n_blocks = per_sample >> 6, stride_b = gridDim.x * 4u, rounds = (n_blocks + stride_b - 1u) / stride_b;
    for (uint32_t k = 0; k < rounds; ++k) {
      const uint32_t B = blockIdx.x * 4u + wave + k * stride_b;
      const bool live = B < n_blocks;                                  // uniform per wavefront; every wavefront keeps the barriers
      uint32_t px = 0, py = 0;
      const bool mine = live && slot_pixel(S, tile_ids, B * 64u + lane, px, py);
      const size_t pi = (size_t)py * S.width + px;
      if (mine) { a = accum[pi]; q = m2 ? m2[pi] : 0.f; }
      for (uint32_t s0 = first_sample & ~7u; s0 < last; s0 += 8u) {
        if (live) {
#pragma unroll
          for (uint32_t j = 0; j < 8u; ++j) {
            const uint32_t qp = 8u * j + (lane >> 3), ks = lane & 7u;
            tile[qp * 9u + ks] = P.rad[pixel_sample_to_slot(B * 64u + qp, s0 + ks, batch_samples)];
          }
        }
        __syncthreads();
        if (mine) {
#pragma unroll
          for (uint32_t ks = 0; ks < 8u; ++ks) { const uint32_t s = s0 + ks; if (s >= first_sample && s < last) fold(tile[lane * 9u + ks]); }
        }
        __syncthreads();
      }
      if (mine) { accum[pi] = a; if (m2) m2[pi] = q; }
    }
  } else {
    for (uint32_t local = blockIdx.x * kBlock + threadIdx.x; local < per_sample; local += gridDim.x * kBlock) {
      uint32_t px, py;
      if (!slot_pixel(S, tile_ids, local, px, py)) continue;
      const size_t pi = (size_t)py * S.width + px;
      a = accum[pi]; q = m2 ? m2[pi] : 0.f;
      for (uint32_t s = first_sample; s < last; ++s) fold(P.rad[pixel_sample_to_slot(local, s, batch_samples)]);
      accum[pi] = a;
      if (m2) m2[pi] = q;
    }
  }
  done = wave_sum(done);
  if (lane_id() == 0 && done) atomicAdd(&C->samples, (unsigned long long)done);
}

// Per-tile error estimate for the adaptive tile sampler (one workgroup per tile, fixed summation order so the
// CPU oracle reproduces every bit): pixel error = sqrt(max(E[l^2] - E[l]^2, 0) / n), unsampled or once-sampled
// pixels count as 1e3; lane j sums pixels j, j+256, ... of the row-major tile, then a stride-128..1 tree.
__global__ __launch_bounds__(kBlock) void k_tile_error(DScene S, const float4* __restrict__ accum, const float* __restrict__ m2,
                                                        float* __restrict__ tile_err, uint32_t* __restrict__ tile_min_count)
{
  __shared__ float s_e[kBlock];
  __shared__ float s_n[kBlock];
  __shared__ float s_c[kBlock];
  const uint32_t ts = S.tile_size, tx = (S.width + ts - 1u) / ts;
  const uint32_t tile = blockIdx.x, x0 = (tile % tx) * ts, y0 = (tile / tx) * ts;
  float e = 0.f, npx = 0.f, cmin = 3.0e38f;
  for (uint32_t i = threadIdx.x; i < ts * ts; i += kBlock) {
    const uint32_t px = x0 + i % ts, py = y0 + i / ts;
    if (px < S.width && py < S.height) {
      const size_t pi = (size_t)py * S.width + px;
      const float4 a = accum[pi];
      float pe = 1.0e3f;
      if (a.w >= 2.0f) {
        const float l = CRH_FMA(0.0722f, a.z, CRH_FMA(0.7152f, a.y, 0.2126f * a.x));
        pe = crh_sqrt(crh_max(m2[pi] - l * l, 0.f) / a.w);
      }
      e += pe; npx += 1.0f; cmin = crh_min(cmin, a.w);
    }
  }
  s_e[threadIdx.x] = e; s_n[threadIdx.x] = npx; s_c[threadIdx.x] = cmin;
  __syncthreads();
  for (uint32_t st = kBlock / 2; st > 0; st >>= 1) {
    if (threadIdx.x < st) {
      s_e[threadIdx.x] += s_e[threadIdx.x + st]; s_n[threadIdx.x] += s_n[threadIdx.x + st];
      s_c[threadIdx.x] = crh_min(s_c[threadIdx.x], s_c[threadIdx.x + st]);
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    tile_err[tile] = s_n[0] > 0.f ? s_e[0] / s_n[0] : 0.f;
    tile_min_count[tile] = s_n[0] > 0.f ? (uint32_t)s_c[0] : 0u;
  }
}

// Adaptive tile selection on the device (reference: AdaptiveScreenSampling / NbRayTracingTiles, SettingsWidget.cxx:427-477; the
// rule itself is DESIGN.md section 7 and the CPU oracle's adaptive_iteration): inverse-CDF draws driven by the base-2 radical
// inverse of a running pick counter, +1 sample on every distinct tile drawn, each at its own sample index.  ONE workgroup; the
// running sum of the errors is taken by one lane in tile order (the oracle's float summation order decides ties), through LDS
// in chunks; draws and the ordered compaction are parallel.  Nothing goes through the host: the tile list, its length and the
// per-tile frame seeds stay in HBM for k_raygen / k_accumulate.
constexpr uint32_t kPickChunk = 4096;
__global__ __launch_bounds__(kBlock) void k_adaptive_pick(const float* __restrict__ tile_err, const uint32_t* __restrict__ tile_cnt, uint32_t nt,
                                                           uint32_t pick0, uint32_t n_picks, uint32_t seed, float* __restrict__ cdf,
                                                           uint8_t* __restrict__ picked, uint32_t* __restrict__ tiles_out,
                                                           uint32_t* __restrict__ seeds_out, uint32_t* __restrict__ n_out)
{
  __shared__ float s_v[kPickChunk];
  __shared__ uint32_t s_part[kBlock];
  __shared__ float s_acc;
  if (threadIdx.x == 0) s_acc = 0.f;
  for (uint32_t i = threadIdx.x; i < nt; i += kBlock) picked[i] = 0;
  for (uint32_t c0 = 0; c0 < nt; c0 += kPickChunk) {
    const uint32_t m = min(kPickChunk, nt - c0);
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < m; i += kBlock) { const float e = tile_err[c0 + i]; s_v[i] = e > 0.f ? e : 0.f; }
    __syncthreads();
    if (threadIdx.x == 0) { float a = s_acc; for (uint32_t i = 0; i < m; ++i) { a += s_v[i]; s_v[i] = a; } s_acc = a; }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < m; i += kBlock) cdf[c0 + i] = s_v[i];
  }
  __threadfence_block();
  __syncthreads();
  const float acc = s_acc;
  for (uint32_t k = threadIdx.x; k < n_picks; k += kBlock) {
    const uint32_t v = __brev(pick0 + k);
    const float u = (float)(v >> 8) * 5.9604644775390625e-8f;
    uint32_t t;
    if (!(acc > 0.f)) t = (uint32_t)(u * (float)nt);                    // no estimate yet: uniform
    else {                                                              // first tile whose running sum exceeds x
      const float x = u * acc;
      uint32_t lo = 0, hi = nt;
      while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (cdf[mid] > x) hi = mid; else lo = mid + 1; }
      t = lo;
    }
    if (t >= nt) t = nt - 1u;
    picked[t] = 1;
  }
  __threadfence_block();
  __syncthreads();
  // ordered compaction: thread j owns tiles [j * per, (j + 1) * per)
  const uint32_t per = (nt + kBlock - 1) / kBlock, b0 = min(nt, threadIdx.x * per), b1 = min(nt, b0 + per);
  uint32_t mine = 0;
  for (uint32_t i = b0; i < b1; ++i) mine += picked[i];
  s_part[threadIdx.x] = mine;
  __syncthreads();
  if (threadIdx.x == 0) { uint32_t a = 0; for (int j = 0; j < kBlock; ++j) { const uint32_t v = s_part[j]; s_part[j] = a; a += v; } *n_out = a; }
  __syncthreads();
  uint32_t w = s_part[threadIdx.x];
  for (uint32_t i = b0; i < b1; ++i)
    if (picked[i]) {
      tiles_out[w] = i;
      // the tile's own sample index selects its frame seed: Bullard generator restarted at `seed`, frame n uses next() >> 2
      uint32_t hi = seed, lo = seed ^ 0x49616E42u, r = 0;
      const uint32_t n = tile_cnt[i];
      for (uint32_t j = 0; j <= n; ++j) { hi = (hi >> 2) + (hi << 2); hi += lo; lo += hi; r = hi; }
      seeds_out[w] = r >> 2;
      ++w;
    }
}

__device__ __forceinline__ float hable(float x)
{
  const float A = 0.22f, B = 0.30f, Cc = 0.10f, D = 0.20f, E = 0.01f, F = 0.30f;
  return (CRH_FMA(x, CRH_FMA(A, x, Cc * B), D * E) / CRH_FMA(x, CRH_FMA(A, x, B), D * F)) - E / F;
}
__global__ __launch_bounds__(kBlock) void k_tonemap(const float4* __restrict__ accum, uint8_t* __restrict__ out, uint32_t n,
                                                     int mode, float exposure, float white_point,
                                                     const uint8_t* __restrict__ tile_mask, uint32_t width, uint32_t tile_size)
{
  const uint32_t tiles_x = tile_mask ? (width + tile_size - 1u) / tile_size : 0u;
  const float gain = crh_exp(exposure * 0.69314718056f);
  const float wp = hable(white_point > 0.f ? white_point : 1.0f);
  for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
    const float4 a = accum[i];
    float v[3] = {a.x, a.y, a.z};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      float x = v[k];
      if (!(x == x) || x < 0.f) x = 0.f;
      x = x * gain;
      if (mode == 1) x = hable(x) / wp;
      x = crh_pow(crh_clamp(x, 0.f, 1.0f), 1.0f / 2.2f);
      out[3u * i + k] = (uint8_t)(int)CRH_FMA(x, 255.0f, 0.5f);
    }
    if (tile_mask) {                                     // ShowSamplingTiles: red outline around the tiles just sampled
      const uint32_t px = i % width, py = i / width, lx = px % tile_size, ly = py % tile_size;
      if (tile_mask[(py / tile_size) * tiles_x + px / tile_size] && (lx == 0u || ly == 0u || lx == tile_size - 1u || ly == tile_size - 1u)) {
        out[3u * i] = 255; out[3u * i + 1u] = 0; out[3u * i + 2u] = 0;
      }
    }
  }
}
__global__ __launch_bounds__(kBlock) void k_hdr(const float4* __restrict__ accum, float* __restrict__ out, uint32_t n)
{
  for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
    const float4 a = accum[i];
    out[3u * i] = a.x; out[3u * i + 1u] = a.y; out[3u * i + 2u] = a.z;
  }
}

// dst += src over n float4 (the same-device leg of crh_reduce: disjoint tile support, so every pixel adds zeros to one value)
__global__ __launch_bounds__(kBlock) void k_add4(float4* __restrict__ dst, const float4* __restrict__ src, uint32_t n)
{
  for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
    float4 a = dst[i]; const float4 b = src[i];
    a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    dst[i] = a;
  }
}

// crh_set_transforms, static / moved split: overwrite the 48-B triangle records at the listed leaf positions (an object leaving the static
// tree: all-zero vertices, whose test yields NaN and rejects; coming back: the original vertices)
__global__ __launch_bounds__(kBlock) void k_scatter_tris(float4* __restrict__ tris, const uint32_t* __restrict__ pos, const float4* __restrict__ recs, uint32_t n)
{
  for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
    float4* d = tris + kTriStride * pos[i];
    d[0] = recs[3u * i]; d[1] = recs[3u * i + 1u]; d[2] = recs[3u * i + 2u];
  }
}

__global__ void k_debug_math(int fn, const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out,
                             float* __restrict__ out2, uint32_t n)
{
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    switch (fn) {
      case 0: crh_sincos2pi(a[i], &out[i], &out2[i]); break;
      case 1: out[i] = crh_exp(a[i]); break;
      case 2: out[i] = crh_log(a[i]); break;
      case 3: out[i] = crh_pow(a[i], b[i]); break;
      case 4: out[i] = crh_acos(a[i]); break;
      case 5: out[i] = crh_atan2(a[i], b[i]); break;
      case 6: crh_sincos(a[i], &out[i], &out2[i]); break;
      case 7: out[i] = crh_sqrt(a[i]); break;
      case 8: out[i] = a[i] / b[i]; break;
      case 9: { uint32_t s = crh_rng_seed(__float_as_uint(a[i]), __float_as_uint(b[i])); out[i] = crh_rng_next(&s); out2[i] = crh_rng_next(&s); } break;
      case 10: { const v3 x = crh_norm3(crh_mk3(a[i], b[i], a[i] * b[i])); out[i] = x.x; out2[i] = crh_dot3(x, crh_mk3(b[i], a[i], 1.0f)); } break;
      default: out[i] = 0.f;
    }
  }
}

// Test hook behind crh_debug_bsdf: the layered BSDF functions k_shade uses, evaluated on caller-supplied directions (local
// frame, z = shading normal) so that the analytic known-answer tests (pdf integrates to 1, sample weight = f cos / pdf,
// Fresnel limits, Snell) run on the gfx950 code itself and not only on the CPU oracle.
//   fn 0: out[3i..]   = eval_layered(wi, wo)            (f * cos)
//   fn 1: out[i]      = pdf_layered(wo, wi, W = 1)
//   fn 2: out[8i..]   = sample_layered with rng state bits(b[3i]), inside flag b[3i+1] != 0: wi.xyz, weight.xyz,
//                       flags (1 alive | 2 delta | 4 inside after), rng state after (uint bits)
//   fn 3: out[3i..]   = fresnel_media(a[3i], m.FresnelCoat)
__global__ void k_debug_bsdf(int fn, const float4* __restrict__ m, const float* __restrict__ a, const float* __restrict__ b,
                             float* __restrict__ out, uint32_t n, int two_sided)
{
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const v3 wo = crh_mk3(a[3u * i], a[3u * i + 1u], a[3u * i + 2u]);
    Bsdf bs;
    bs.Kc = xyz(m[0]); bs.Rc = m[0].w; bs.Kd = xyz(m[1]); bs.Ks = xyz(m[2]); bs.Rs = m[2].w; bs.Kt = xyz(m[3]); bs.Le = xyz(m[4]);
    bs.ab = m[5]; bs.fc = m[6]; bs.fb = m[7];
    bs.Fc = fresnel_media(wo.z, bs.fc);
    const v3 one = crh_mk3(1.0f, 1.0f, 1.0f);
    if (fn == 0) {
      const v3 r = eval_layered(bs, crh_mk3(b[3u * i], b[3u * i + 1u], b[3u * i + 2u]), wo, two_sided);
      out[3u * i] = r.x; out[3u * i + 1u] = r.y; out[3u * i + 2u] = r.z;
    } else if (fn == 1) {
      out[i] = pdf_layered(bs, wo, crh_mk3(b[3u * i], b[3u * i + 1u], b[3u * i + 2u]), one, two_sided);
    } else if (fn == 2) {
      uint32_t rng = __float_as_uint(b[3u * i]);
      bool inside = b[3u * i + 1u] != 0.f, delta = false;
      v3 W = one, wi = crh_mk3(0.f, 0.f, 0.f); int lobe;
      const bool alive = sample_layered(bs, wo, wi, W, inside, delta, rng, two_sided, SpecB{0, 1.0f}, lobe);
      float* o = out + 8u * i;
      o[0] = wi.x; o[1] = wi.y; o[2] = wi.z; o[3] = W.x; o[4] = W.y; o[5] = W.z;
      o[6] = (float)((alive ? 1 : 0) | (delta ? 2 : 0) | (inside ? 4 : 0)); o[7] = __uint_as_float(rng);
    } else {
      const v3 r = fresnel_media(wo.x, bs.fc);
      out[3u * i] = r.x; out[3u * i + 1u] = r.y; out[3u * i + 2u] = r.z;
    }
  }
}

}  // namespace

// ================================================================== launch wrappers
void launch_raygen(const Launch& L, const DScene& S, const DPaths& P, const DQueues& Q, int qsel,
                   const uint32_t* d_tile_ids, uint32_t n_tiles, const uint32_t* d_seeds, uint32_t n_samples, int seed_per_tile,
                   const uint32_t* d_n_tiles)
{
  hipMemsetAsync(Q.counts + qsel, 0, sizeof(uint32_t), L.stream);
  if (S.split) hipMemsetAsync(Q.counts + 3, 0, sizeof(uint32_t), L.stream);                 // second-pass count of bounce 0 (even parity)
  if (S.split) hipLaunchKernelGGL(k_raygen<true>, dim3(L.grid), dim3(kBlock), 0, L.stream, S, P, Q.q[qsel], Q.counts + qsel, Q.q2, Q.counts + 3, Q.counts + 4, d_tile_ids, n_tiles, d_seeds, n_samples, seed_per_tile, d_n_tiles);
  else         hipLaunchKernelGGL(k_raygen<false>, dim3(L.grid), dim3(kBlock), 0, L.stream, S, P, Q.q[qsel], Q.counts + qsel, Q.q2, Q.counts + 3, Q.counts + 4, d_tile_ids, n_tiles, d_seeds, n_samples, seed_per_tile, d_n_tiles);
}
// A persistent traversal grid larger than what the register budget keeps resident leaves workgroups queued behind the first
// wave of them, i.e. a second, nearly empty round at the end of every launch: clamp the grid to occupancy x compute units.
template <auto Kernel> static int resident_grid(const Launch& L)
{
  if (L.cus <= 0) return L.grid;
  static int per_cu = 0;                      // one instance per kernel (the kernel is a template argument, not just its type)
  if (per_cu == 0) { int n = 0; per_cu = (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, Kernel, kBlock, 0) == hipSuccess && n > 0) ? n : -1; }
  return per_cu > 0 ? min(L.grid, per_cu * L.cus) : L.grid;
}

// second-pass counts ping-pong by bounce parity: bounce b's list is counted in counts[3] (b even) / counts[10] (b odd)
static inline int count2_slot(uint32_t bounce) { return (bounce & 1u) ? 10 : 3; }

void launch_trace_nearest(const Launch& L, const DScene& S, const DPaths& P, const DQueues& Q, int qin, uint32_t bounce, DCounters* C)
{
  // first pass (or the only one): a split scene walks its static tree with the SINGLE-LEVEL instantiation
  const bool two = S.two_level && !S.split;
  if (L.packets && bounce == 0u && !two && !S.split && !L.counters && !L.donate) {
    // camera rays of a wide batch: one walk per wavefront of 64 samples (k_trace_packets), then the rays that met a tie one by one (usually none)
    static int per_cu = 0;
    if (per_cu == 0) { int nb = 0; per_cu = (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_trace_packets<true>, kBlock, 0) == hipSuccess && nb > 0) ? nb : -1; }
    const int grid = (per_cu > 0 && L.cus > 0) ? min(max(L.grid, 8 * L.cus), per_cu * L.cus) : L.grid;
    if (S.pnodes) hipLaunchKernelGGL(k_trace_packets<true>, dim3(grid), dim3(kBlock), 0, L.stream, S, P, S.nodes, S.pnodes, S.tris, Q.q[qin], Q.counts + qin, Q.counts + 4, Q.counts + (1 - qin), Q.counts + 2,
                       Q.counts + count2_slot(bounce + 1u), Q.counts + 7, Q.q2, Q.counts + 11, C);
    else          hipLaunchKernelGGL(k_trace_packets<false>, dim3(grid), dim3(kBlock), 0, L.stream, S, P, S.nodes, S.nodes, S.tris, Q.q[qin], Q.counts + qin, Q.counts + 4, Q.counts + (1 - qin), Q.counts + 2,
                       Q.counts + count2_slot(bounce + 1u), Q.counts + 7, Q.q2, Q.counts + 11, C);
    // (the donating instantiation: a handful of rays, each walked by a whole wavefront)
    hipLaunchKernelGGL((k_trace_nearest<false, false, true, false, true>), dim3(64), dim3(kBlock), 0, L.stream, S, P, qin, Q.q2, Q.counts + 11, Q.counts + 4,
                       Q.counts + (1 - qin), Q.counts + 2, Q.counts + count2_slot(bounce + 1u), Q.counts + 7, C);
    return;
  }
#define CRH_LAUNCH_TN(CNT, TWO, DON, P2, QQ, CC) hipLaunchKernelGGL((k_trace_nearest<CNT, TWO, DON, P2>), dim3(resident_grid<k_trace_nearest<CNT, TWO, DON, P2>>(L)), dim3(kBlock), 0, L.stream, S, P, qin, QQ, \
                                                   CC, Q.counts + 4, Q.counts + (1 - qin), Q.counts + 2, Q.counts + count2_slot(bounce + 1u), Q.counts + 7, C)
  if (two) { if (L.counters) CRH_LAUNCH_TN(true, true, false, false, Q.q[qin], Q.counts + qin); else if (L.donate) CRH_LAUNCH_TN(false, true, true, false, Q.q[qin], Q.counts + qin); else CRH_LAUNCH_TN(false, true, false, false, Q.q[qin], Q.counts + qin); }
  else     { if (L.counters) CRH_LAUNCH_TN(true, false, false, false, Q.q[qin], Q.counts + qin); else if (L.donate) CRH_LAUNCH_TN(false, false, true, false, Q.q[qin], Q.counts + qin); else CRH_LAUNCH_TN(false, false, false, false, Q.q[qin], Q.counts + qin); }
  if (S.split) {     // second pass: the top level, for the rays listed by their producer
    if (L.counters) CRH_LAUNCH_TN(true, true, false, true, Q.q2, Q.counts + count2_slot(bounce)); else if (L.donate) CRH_LAUNCH_TN(false, true, true, true, Q.q2, Q.counts + count2_slot(bounce)); else CRH_LAUNCH_TN(false, true, false, true, Q.q2, Q.counts + count2_slot(bounce));
  }
#undef CRH_LAUNCH_TN
}
void launch_expand_packet_nodes(const Launch& L, const float4* nodes, float4* pnodes, uint32_t n)
{
  if (n) hipLaunchKernelGGL(k_expand_packet_nodes, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, L.stream, nodes, pnodes, n);
}
void launch_shade(const Launch& L, const DScene& S, const DPaths& P, const DQueues& Q, int qin, uint32_t bounce, DCounters* C)
{
  if (S.split) hipLaunchKernelGGL(k_shade<true>, dim3(L.grid), dim3(kBlock), 0, L.stream, S, P, qin, bounce, Q.q[qin], Q.counts + qin,
                     Q.q[1 - qin], Q.counts + (1 - qin), Q.q_sh, Q.counts + 2, Q.q2, Q.counts + count2_slot(bounce + 1u), Q.q2_sh, Q.counts + 7, Q.counts + 4, C);
  else         hipLaunchKernelGGL(k_shade<false>, dim3(L.grid), dim3(kBlock), 0, L.stream, S, P, qin, bounce, Q.q[qin], Q.counts + qin,
                     Q.q[1 - qin], Q.counts + (1 - qin), Q.q_sh, Q.counts + 2, Q.q2, Q.counts + count2_slot(bounce + 1u), Q.q2_sh, Q.counts + 7, Q.counts + 4, C);
}
void launch_trace_any(const Launch& L, const DScene& S, const DPaths& P, const DQueues& Q, DCounters* C)
{
  const bool two = S.two_level && !S.split;
#define CRH_LAUNCH_TA(CNT, TWO, DON, P2, QQ, CC) hipLaunchKernelGGL((k_trace_any<CNT, TWO, DON, P2>), dim3(resident_grid<k_trace_any<CNT, TWO, DON, P2>>(L)), dim3(kBlock), 0, L.stream, S, P, QQ, CC, Q.counts + 4, C)
  if (two) { if (L.counters) CRH_LAUNCH_TA(true, true, false, false, Q.q_sh, Q.counts + 2); else if (L.donate) CRH_LAUNCH_TA(false, true, true, false, Q.q_sh, Q.counts + 2); else CRH_LAUNCH_TA(false, true, false, false, Q.q_sh, Q.counts + 2); }
  else     { if (L.counters) CRH_LAUNCH_TA(true, false, false, false, Q.q_sh, Q.counts + 2); else if (L.donate) CRH_LAUNCH_TA(false, false, true, false, Q.q_sh, Q.counts + 2); else CRH_LAUNCH_TA(false, false, false, false, Q.q_sh, Q.counts + 2); }
  if (S.split) {
    if (L.counters) CRH_LAUNCH_TA(true, true, false, true, Q.q2_sh, Q.counts + 7); else if (L.donate) CRH_LAUNCH_TA(false, true, true, true, Q.q2_sh, Q.counts + 7); else CRH_LAUNCH_TA(false, true, false, true, Q.q2_sh, Q.counts + 7);
  }
#undef CRH_LAUNCH_TA
}
void launch_accumulate(const Launch& L, const DScene& S, const DPaths& P, float4* accum, float* m2, const uint32_t* d_tile_ids,
                       uint32_t n_tiles, uint32_t first_sample, uint32_t n_samples, uint32_t batch_samples, DCounters* C, const uint32_t* d_n_tiles)
{
  hipLaunchKernelGGL(k_accumulate, dim3(L.grid), dim3(kBlock), 0, L.stream, S, P, accum, m2, d_tile_ids, n_tiles, first_sample, n_samples, batch_samples, C, d_n_tiles);
}
void launch_tile_error(const Launch& L, const DScene& S, const float4* accum, const float* m2, float* tile_err, uint32_t* tile_min_count,
                       uint32_t n_tiles_total)
{
  hipLaunchKernelGGL(k_tile_error, dim3(n_tiles_total), dim3(kBlock), 0, L.stream, S, accum, m2, tile_err, tile_min_count);
}
void launch_adaptive_pick(const Launch& L, const float* tile_err, const uint32_t* tile_cnt, uint32_t n_tiles_total, uint32_t pick0, uint32_t n_picks,
                          uint32_t seed, float* cdf, uint8_t* picked, uint32_t* tiles_out, uint32_t* seeds_out, uint32_t* n_out)
{
  hipLaunchKernelGGL(k_adaptive_pick, dim3(1), dim3(kBlock), 0, L.stream, tile_err, tile_cnt, n_tiles_total, pick0, n_picks, seed, cdf, picked, tiles_out, seeds_out, n_out);
}
void launch_tonemap(const Launch& L, const float4* accum, uint8_t* out, uint32_t n, int mode, float exposure, float wp,
                    const uint8_t* tile_mask, uint32_t width, uint32_t tile_size)
{
  hipLaunchKernelGGL(k_tonemap, dim3(L.grid), dim3(kBlock), 0, L.stream, accum, out, n, mode, exposure, wp, tile_mask, width, tile_size);
}
void launch_hdr(const Launch& L, const float4* accum, float* out, uint32_t n)
{
  hipLaunchKernelGGL(k_hdr, dim3(L.grid), dim3(kBlock), 0, L.stream, accum, out, n);
}
void launch_scatter_tris(const Launch& L, float4* tris, const uint32_t* pos, const float4* recs, uint32_t n)
{
  hipLaunchKernelGGL(k_scatter_tris, dim3(n ? (n + kBlock - 1) / kBlock : 1u), dim3(kBlock), 0, L.stream, tris, pos, recs, n);
}
void launch_add4(const Launch& L, float4* dst, const float4* src, uint32_t n)
{
  hipLaunchKernelGGL(k_add4, dim3(L.grid), dim3(kBlock), 0, L.stream, dst, src, n);
}
void launch_trace_rays(const Launch& L, const DScene& S, const float4* rays, uint32_t n, int any_hit, float4* out_hit,
                       uint32_t* out_vis, uint32_t* cursor, DCounters* C)
{
  hipMemsetAsync(cursor, 0, sizeof(uint32_t), L.stream);
#define CRH_LAUNCH_TR(ANY, CNT, TWO) hipLaunchKernelGGL((k_trace_rays<ANY, CNT, TWO>), dim3(resident_grid<k_trace_rays<ANY, CNT, TWO>>(L)), dim3(kBlock), 0, L.stream, S, rays, n, cursor, out_hit, out_vis, C)
#define CRH_LAUNCH_TR2(ANY, CNT) { if (S.two_level) CRH_LAUNCH_TR(ANY, CNT, true); else CRH_LAUNCH_TR(ANY, CNT, false); }
  if (any_hit) { if (L.counters) CRH_LAUNCH_TR2(true, true) else CRH_LAUNCH_TR2(true, false) }
  else         { if (L.counters) CRH_LAUNCH_TR2(false, true) else CRH_LAUNCH_TR2(false, false) }
#undef CRH_LAUNCH_TR2
#undef CRH_LAUNCH_TR
}
void launch_debug_math(const Launch& L, int fn, const float* a, const float* b, float* out, float* out2, uint32_t n)
{
  hipLaunchKernelGGL(k_debug_math, dim3(L.grid), dim3(kBlock), 0, L.stream, fn, a, b, out, out2, n);
}

void launch_debug_bsdf(const Launch& L, int fn, const float4* m, const float* a, const float* b, float* out, uint32_t n, int two_sided)
{
  hipLaunchKernelGGL(k_debug_bsdf, dim3(L.grid), dim3(kBlock), 0, L.stream, fn, m, a, b, out, n, two_sided);
}

}  // namespace crh

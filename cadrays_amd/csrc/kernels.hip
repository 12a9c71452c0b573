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

#include "k_common.h"
#include "k_traversal.h"
#include "k_packets.h"
#include "k_bsdf.h"
#include "k_lights_env.h"
#include "k_raygen.h"
#include "k_shade.h"
#include "k_frame.h"
#include "k_accumulate.h"

}  // namespace

// ================================================================== launch wrappers
void launch_raygen(const Launch& L, const DScene& S, const DPaths& P, const DQueues& Q, int qsel,
                   const uint32_t* d_tile_ids, uint32_t n_tiles, const uint32_t* d_seeds, uint32_t n_samples, int seed_per_tile,
                   const uint32_t* d_n_tiles, const uint32_t* h_seeds)
{
  SeedVals sv;
  for (int i = 0; i < 16; ++i) sv.v[i] = (h_seeds && (uint32_t)i < n_samples) ? h_seeds[i] : 0u;
  hipMemsetAsync(Q.counts + qsel, 0, sizeof(uint32_t), L.stream);
  if (S.split) hipMemsetAsync(Q.counts + 3, 0, sizeof(uint32_t), L.stream);                 // second-pass count of bounce 0 (even parity)
  if (S.split) hipLaunchKernelGGL(k_raygen<true>, dim3(L.grid), dim3(kBlock), 0, L.stream, S, P, Q.q[qsel], Q.counts + qsel, Q.q2, Q.counts + 3, Q.counts + 4, d_tile_ids, n_tiles, d_seeds, n_samples, seed_per_tile, d_n_tiles, sv);
  else         hipLaunchKernelGGL(k_raygen<false>, dim3(L.grid), dim3(kBlock), 0, L.stream, S, P, Q.q[qsel], Q.counts + qsel, Q.q2, Q.counts + 3, Q.counts + 4, d_tile_ids, n_tiles, d_seeds, n_samples, seed_per_tile, d_n_tiles, sv);
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
    // (the donating instantiation: usually a handful of rays, each walked by a whole wavefront -- but coincident or duplicated faces, common in CAD assemblies,
    // send every camera ray that meets them here: the grid is what is resident, and a workgroup that finds the list empty leaves at once)
    hipLaunchKernelGGL((k_trace_nearest<false, false, true, false, true>), dim3(resident_grid<k_trace_nearest<false, false, true, false, true>>(L)), dim3(kBlock), 0, L.stream, S, P, qin, Q.q2, Q.counts + 11, Q.counts + 4,
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
// The frame kernel (k_frame.h): ray generation, every bounce's traversal and shading of a small batch in ONE launch.  `ctl`: two words, zero at launch (the kernel
// leaves them zero).  The grid is what the register budget keeps resident (every workgroup is a persistent streaming path tracer), or less for a frame that
// shares the chip with others in flight.
// LDS of the frame kernel: stack 16 x 1024 x 4 B = 64 KB + two rings of kFrameRing x 4 B = 32 KB + bounds 4 KB + materials 4 KB = ~104 KB (two-level scenes: + 36 KB of
// world rays = ~140 KB).  Only a part with gfx950's 160 KB of LDS per compute unit can launch it.
static_assert((size_t)kLdsStack * kFrameBlock * 4 + (2 + CRH_FRAME_MISS_RING) * (size_t)kFrameRing * 4 + (size_t)kFrameBlock * 4 + (size_t)kFrameMats * 128 + 9 * (size_t)kFrameBlock * 4 <= 160 * 1024, "k_frame<true> does not fit the LDS of a gfx950 compute unit");
static int frame_occupancy(bool two_level)
{
  static int per_cu[2] = {0, 0};
  int& pc = per_cu[two_level ? 1 : 0];
  if (pc == 0) { int n = 0; pc = (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, two_level ? k_frame<true> : k_frame<false>, kFrameBlock, 0) == hipSuccess && n > 0) ? n : -1; }
  return pc;
}
// on a device (or after a build-flag change) where not even one workgroup of the frame kernel is resident, small batches take the staged schedule
// (crh_schedule.cpp frame_ok) -- bit-identical -- instead of a launch that fails
bool frame_launchable(bool two_level) { return frame_occupancy(two_level) > 0; }
int frame_resident_grid(int cus, bool two_level)
{
  const int pc = frame_occupancy(two_level);
  return (pc > 0 ? pc : 1024 / kFrameBlock) * (cus > 0 ? cus : 256);
}
void launch_frame(const Launch& L, const DScene& S, const DPaths& P, uint32_t* ctl, const uint32_t* d_tile_ids, uint32_t n_tiles, const uint32_t* d_seeds,
                  uint32_t n_samples, int seed_per_tile, const uint32_t* d_n_tiles, uint32_t max_live, uint32_t gen_chunk, uint32_t low_water, uint32_t n_feed, uint32_t claim_step, uint32_t starve, DCounters* C, const uint32_t* h_seeds, uint32_t* d_err, uint32_t help)
{
  FrameArgs A;
  A.err = d_err; A.help = (help & 0xFFFFu) ? (help & 0xFFFFu) : 256u; A.help_low = help >> 16;      // (the two tracer-helps-shading thresholds travel in one argument)
  A.tile_ids = d_tile_ids; A.n_tiles = n_tiles; A.n_tiles_dev = d_n_tiles; A.seeds = d_seeds; A.n_samples = n_samples; A.seed_per_tile = seed_per_tile;
  for (int i = 0; i < 16; ++i) A.seed_vals[i] = (h_seeds && (uint32_t)i < n_samples) ? h_seeds[i] : 0u;
  A.ctl = ctl; A.gen_chunk = min(max(gen_chunk & ~63u, 64u), kFrameRing); A.max_live = min(max(max_live, A.gen_chunk), kFrameRing); A.low_water = low_water;
  A.n_feed = min(n_feed, (uint32_t)kFrameBlock / 64u - 1u); A.claim_step = claim_step; A.starve = starve;
  DScene S1 = S; S1.split = 0;                  // a split scene is walked in one go: static tree, then the top level (the two-pass form is a wavefront-schedule device)
  const int grid = min(L.grid, frame_resident_grid(L.cus, S.two_level != 0));
  if (S.two_level) hipLaunchKernelGGL(k_frame<true>, dim3(grid), dim3(kFrameBlock), 0, L.stream, S1, P, A, C);
  else             hipLaunchKernelGGL(k_frame<false>, dim3(grid), dim3(kFrameBlock), 0, L.stream, S1, P, A, C);
}
#if CRH_COHERENCE_STATS
}  // namespace crh
// instrumented builds only (tools/ab_build.sh coh "-DCRH_COHERENCE_STATS=1"): k_shade's vote counters per bounce since the last call (k_shade.h)
extern "C" __attribute__((visibility("default"))) int crh_exp_coherence(unsigned long long* out256)
{
  static unsigned long long zero[256] = {0};
  if (hipMemcpyFromSymbol(out256, HIP_SYMBOL(crh::g_coherence), sizeof zero) != hipSuccess) return -1;
  return hipMemcpyToSymbol(HIP_SYMBOL(crh::g_coherence), zero, sizeof zero) == hipSuccess ? 0 : -1;
}
namespace crh {
#endif
#if CRH_FRAME_STATS || CRH_FRAME_TIMELINE
}  // namespace crh
// instrumented builds only (tools/ab_build.sh NAME "-DCRH_FRAME_STATS=1"): what the frame kernel's engines counted since the last call
extern "C" __attribute__((visibility("default"))) int crh_exp_frame_stats(unsigned long long* out32)
{
  unsigned long long zero[32] = {0};
  if (hipMemcpyFromSymbol(out32, HIP_SYMBOL(crh::g_frame_stats), sizeof zero) != hipSuccess) return -1;
  return hipMemcpyToSymbol(HIP_SYMBOL(crh::g_frame_stats), zero, sizeof zero) == hipSuccess ? 0 : -1;
}
namespace crh {
#endif
void launch_accumulate(const Launch& L, const DScene& S, const DPaths& P, float4* accum, float* m2, const uint32_t* d_tile_ids,
                       uint32_t n_tiles, uint32_t first_sample, uint32_t n_samples, uint32_t batch_samples, DCounters* C, const uint32_t* d_n_tiles, uint32_t* d_tile_cost)
{
  hipLaunchKernelGGL(k_accumulate, dim3(L.grid), dim3(kBlock), 0, L.stream, S, P, accum, m2, d_tile_ids, n_tiles, first_sample, n_samples, batch_samples, C, d_n_tiles, d_tile_cost);
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
void launch_tonemap(const Launch& L, const float4* accum, uint8_t* out, uint32_t n, int mode, float exposure, float wp, int gamma22,
                    const uint8_t* tile_mask, uint32_t width, uint32_t tile_size)
{
  hipLaunchKernelGGL(k_tonemap, dim3(L.grid), dim3(kBlock), 0, L.stream, accum, out, n, mode, exposure, wp, gamma22, tile_mask, width, tile_size);
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

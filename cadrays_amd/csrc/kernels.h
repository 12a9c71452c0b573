// kernels.h -- launch wrappers of the gfx950 kernels (implemented in kernels.hip).
#pragma once
#include "device_types.h"

namespace crh {

struct Launch {
  hipStream_t stream;
  int grid;          // workgroups for the grid-stride kernels
  bool counters;     // collect node / triangle counters (slower)
  int cus = 0;       // compute units of the device: persistent traversal grids are clamped to what is resident at once (0: as given)
  bool donate = false;   // small batches: traversal kernels whose idle lanes take over parts of the long rays' stacks (kernels.hip, DON)
  bool packets = false;  // wide batches: the camera rays (bounce 0) are walked as wavefront packets (kernels.hip, k_trace_packets)
};

// camera rays for n_samples x n_tiles x tile^2 path slots; fills queue `qsel` and its count
void launch_raygen(const Launch&, const DScene&, const DPaths&, const DQueues&, int qsel,
                   const uint32_t* d_tile_ids, uint32_t n_tiles, const uint32_t* d_frame_seeds, uint32_t n_samples,
                   int seed_per_tile = 0, const uint32_t* d_n_tiles = nullptr /* tile count in HBM (device-drawn tile list) */,
                   const uint32_t* h_seeds = nullptr /* with d_frame_seeds == nullptr: the <= 16 frame seeds on the host, passed by value */);
// nearest-hit traversal of queue `qin`; also zeroes the other queue's count and the shadow count
void launch_trace_nearest(const Launch&, const DScene&, const DPaths&, const DQueues&, int qin, uint32_t bounce, DCounters*);
// emission, NEE, BSDF sampling, Russian roulette; survivors -> queue 1-qin, shadow rays -> q_sh
void launch_expand_packet_nodes(const Launch&, const float4* nodes, float4* pnodes, uint32_t n_nodes);      // k_trace_packets<true>'s node array
void launch_shade(const Launch&, const DScene&, const DPaths&, const DQueues&, int qin, uint32_t bounce, DCounters*);
// any-hit traversal of the shadow queue; unoccluded contributions are added to the path radiance
void launch_trace_any(const Launch&, const DScene&, const DPaths&, const DQueues&, DCounters*);
// the frame kernel (k_frame.h): a whole small batch -- camera rays, every bounce's traversal, shading and shadow rays -- in one launch; `ctl` = two control words, zero
// at launch and left zero; L.grid = workgroups wanted (clamped to what is resident); slots must stay below kFrameMaxPaths
constexpr uint32_t kFrameMaxPaths = 1u << 25;
int frame_resident_grid(int cus, bool two_level);
bool frame_launchable(bool two_level);      // at least one workgroup of the frame kernel is resident on this device
void launch_frame(const Launch&, const DScene&, const DPaths&, uint32_t* ctl, const uint32_t* d_tile_ids, uint32_t n_tiles, const uint32_t* d_frame_seeds,
                  uint32_t n_samples, int seed_per_tile, const uint32_t* d_n_tiles, uint32_t max_live, uint32_t gen_chunk, uint32_t low_water, uint32_t n_feed, uint32_t claim_step, uint32_t starve, DCounters*,
                  const uint32_t* h_seeds = nullptr /* with d_frame_seeds == nullptr: the <= 16 frame seeds on the host, passed by value */, uint32_t* d_err = nullptr /* the context's device error word */, uint32_t help = 256u /* FrameArgs::help */);
// clamp + running mean of the finished paths of batch samples [first_sample, first_sample + n_samples) into the float4
// accumulator, sample by sample; batch_samples = the sample count the batch was generated with (it fixes the slot layout)
void launch_accumulate(const Launch&, const DScene&, const DPaths&, float4* accum, float* m2 /* or nullptr */,
                       const uint32_t* d_tile_ids, uint32_t n_tiles, uint32_t first_sample, uint32_t n_samples, uint32_t batch_samples, DCounters*,
                       const uint32_t* d_n_tiles = nullptr, uint32_t* d_tile_cost = nullptr /* small batches of the frame kernel: + rays traced per tile */);
// adaptive tile sampler, device side: running sum of the tile errors, n_picks inverse-CDF draws (radical inverse of pick0 + k),
// the distinct tiles in ascending order with their per-tile frame seeds and their number -- all left in HBM
void launch_adaptive_pick(const Launch&, const float* tile_err, const uint32_t* tile_cnt, uint32_t n_tiles_total, uint32_t pick0, uint32_t n_picks,
                          uint32_t seed, float* cdf, uint8_t* picked, uint32_t* tiles_out, uint32_t* seeds_out, uint32_t* n_out);
// adaptive tile sampler: per-tile mean standard error and minimum per-pixel sample count (one workgroup per tile)
void launch_tile_error(const Launch&, const DScene&, const float4* accum, const float* m2, float* tile_err,
                       uint32_t* tile_min_count, uint32_t n_tiles_total);
void launch_tonemap(const Launch&, const float4* accum, uint8_t* out_rgb, uint32_t n_pixels, int mode, float exposure, float white_point, int gamma22,
                    const uint8_t* tile_mask /* nullptr: no overlay */, uint32_t width, uint32_t tile_size);
void launch_hdr(const Launch&, const float4* accum, float* out_rgb, uint32_t n_pixels);
void launch_add4(const Launch&, float4* dst, const float4* src, uint32_t n_float4);
// overwrite the triangle records at leaf positions pos[0..n) with recs (3 x float4 each): static / moved split of crh_set_transforms
void launch_scatter_tris(const Launch&, float4* tris, const uint32_t* pos, const float4* recs, uint32_t n);
// API-level ray tracing on a plain ray buffer (8 floats per ray)
void launch_trace_rays(const Launch&, const DScene&, const float4* rays, uint32_t n, int any_hit,
                       float4* out_hit, uint32_t* out_vis, uint32_t* d_cursor, DCounters*);
void launch_debug_math(const Launch&, int fn, const float* a, const float* b, float* out, float* out2, uint32_t n);
// test hook: eval / pdf / sample / Fresnel of the layered BSDF on caller-supplied local directions (m: 8 x float4 = crh_bsdf)
void launch_debug_bsdf(const Launch&, int fn, const float4* m, const float* a, const float* b, float* out, uint32_t n, int two_sided);

}  // namespace crh

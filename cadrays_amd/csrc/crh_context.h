// crh_context.h -- the context behind the C ABI of libcadrays_hip.so (include/cadrays_hip.h) and the helpers its translation units share.
//
//   crh_context.cpp    create / destroy, device-memory and event helpers, accumulator, crh_reset / crh_sync, the CRH_* environment table
//   crh_scene.cpp      geometry, transforms, materials, lights, environment, textures, camera, params, spec; crh_build; the kernels' DScene
//   crh_schedule.cpp   the wavefront schedule: lanes, batches, frame pipelining, look-ahead, adaptive iterations; crh_render / crh_render_tiles
//   crh_readback.cpp   HDR / LDR read-back (synchronous and asynchronous), accumulator checkpoints, statistics and kernel timing
//   crh_reduce.cpp     crh_reduce (RCCL over xGMI, or peer copies on one device)
//   crh_debug.cpp      API-level ray tracing, micro-benchmarks and the math / BSDF test hooks
//
// This is the code that sits behind CADRays' `myInternal->View->Redraw()` (reference src/Launcher/AppViewer.cxx:1047).  There is NO CPU fallback:
// every entry point that renders or traces launches the gfx950 kernels and reports CRH_E_DEVICE when the HIP runtime refuses.
#pragma once
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>     // types only: the library is loaded with dlopen on the first multi-device crh_reduce

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <string>
#include <thread>
#include <vector>

#include "../../include/cadrays_hip.h"
#include "../../include/crh_xform.h"
#include "bvh_builder.h"
#include "kernels.h"

namespace crh {

// ---- host copies of the inputs (what the setters received; crh_build and the per-frame setters work from these)
struct HostInputs {
  std::vector<float> pos, nrm, uv;
  std::vector<int32_t> tri;
  std::vector<crh_bsdf> mats;
  std::vector<crh_light> lights;
  std::vector<float> env; uint32_t envW = 0, envH = 0;
  struct HostTex { std::vector<float> rgba; uint32_t w = 0, h = 0; };
  std::vector<HostTex> textures; bool textures_dirty = false;
  crh_camera cam{};
  crh_params par{};
  crh_spec spec = CRH_SPEC_DEFAULTS;      // include/crh_spec.h
};

// ---- two-level mode (per-object transforms) and the static / moved split (reference: the gizmo moves ONE object per drag,
// ImRaytraceControls.cxx:64,88): every object is baked into one world-space tree with the transform it has when the scene is built; an object that is
// moved away from that placement has its triangles there disabled and gets an object tree of its own (built on first need, kept); nothing else is ever
// rebuilt by crh_set_transforms but the top-level tree
struct TwoLevelState {
  bool two_level = false; uint32_t nO = 0;
  std::vector<float> xf; std::vector<int32_t> tri_obj;
  std::vector<uint8_t> hidden;            // per object: 1 = erased from the view (crh_set_visibility); empty = all displayed
  std::vector<float> xf0, pos_w, nrm_w;   // the transforms the scene was built with; per vertex: position / unit normal under its object's build-time transform (what the static tree holds)
  struct Inst { float fwd[12], inv[12], bmin[3], bmax[3]; uint32_t root, obj; };
  std::vector<Inst> inst;                 // the objects rendered as instances RIGHT NOW, ascending object index (empty: the scene is one world-space tree)
  uint32_t n_blas_nodes = 0, root = 0;    // nodes of the static tree + the object trees built so far (the top-level tree follows them); entry point of the walk
  struct Obj { bool static0 = false, built = false, is_inst = false, in_static = false; uint32_t root = 0, first = 0, ntri = 0; float bmin[3] = {0, 0, 0}, bmax[3] = {0, 0, 0}; };
  std::vector<Obj> objs; std::vector<uint32_t> obj_tris, static_pos, pos_obj;   // pos_obj: object of the triangle at a leaf position >= n_static
  uint32_t n_static = 0, n_static_live = 0, n_pos = 0; float sbmin[3] = {0, 0, 0}, sbmax[3] = {0, 0, 0};
  uint32_t root2 = 0xFFFFFFFFu; float tlas_lo[3] = {0, 0, 0}, tlas_hi[3] = {0, 0, 0};
  size_t cap_pos = 0;                     // leaf positions the triangle / shading / uv arrays have room for
  void* d_patch = nullptr; size_t cap_patch = 0;
  int split_passes = -1;                  // CRH_SPLIT_PASSES: -1 auto, 0 one walk, 1 two passes
  float4* d_ibox = nullptr;               // spheres around the instances' world boxes when there are at most kMaxIBox (the "does the ray come near a moved object" test)
  float usph[4] = {0.f, 0.f, 0.f, 0.f};  // ... and around the bounds of all of them
  float4* d_inst = nullptr;
};

// ---- the built scene: host tree + leaf-ordered records, and their residency in HBM
struct BuiltScene {
  QBvh bvh;
  std::vector<float> h_tris;      // 12 floats per triangle, leaf order
  bool built = false;
  float4* d_pnodes = nullptr; size_t cap_pnodes = 0;      // packet nodes of a single-level scene (128 B per node; kernels.hip k_expand_packet_nodes)
  float4 *d_nodes = nullptr, *d_tris = nullptr, *d_shade = nullptr, *d_mats = nullptr, *d_lights = nullptr, *d_env = nullptr;
  float4 *d_uvs = nullptr, *d_texels = nullptr; uint4* d_tex_desc = nullptr;
  float4* d_verts = nullptr;      // two-level scenes: object-space vertices per leaf position (shading of instance hits)
  // setters called every GUI frame (material-editor drags MaterialEditor.cxx:331-337, manipulator moves ImRaytraceControls.cxx:58-89)
  // reuse their device allocations (capacities below) and copy through a small ring of pinned staging buffers on the context's
  // stream: no hipFree / hipMalloc, no device-wide synchronisation, kernels still in flight keep reading the old bytes
  size_t cap_nodes = 0, cap_inst = 0, cap_mats = 0, cap_lights = 0, cap_env = 0;
  struct Stage { void* p = nullptr; size_t cap = 0; hipEvent_t ev = nullptr; bool used = false; } stage[4];
  uint32_t stage_next = 0;
};

// ---- accumulator, the frame assembled by crh_reduce, adaptive sampler state
struct FrameState {
  float4* d_accum = nullptr; uint32_t accumW = 0, accumH = 0;
  float* d_m2 = nullptr;            // running mean of squared luminance (adaptive sampling only)
  // crh_reduce: the frame assembled from all shards lives beside the root's own accumulator (rendering continues into that)
  float4* d_assembled = nullptr; float4* d_peer_stage = nullptr; uint32_t assembledW = 0, assembledH = 0; bool assembled_valid = false;
  std::vector<ncclComm_t> comms; std::vector<crh_ctx*> comm_ctxs;      // RCCL communicators of the last multi-device group (kept on the root)
  float* d_tile_err = nullptr; uint32_t* d_tile_cnt = nullptr; uint32_t tile_stat_cap = 0;
  bool show_tiles = false; bool picked_valid = false;                                   // ShowSamplingTiles: d_picked marks the tiles of the last adaptive iteration
  float* d_tile_cdf = nullptr; uint8_t* d_picked = nullptr; uint32_t* d_adapt_n = nullptr;   // adaptive sampler state in HBM (running sum, drawn-tile mask, tile count)
  bool adaptive = false; uint32_t adaptive_tiles = 128; uint32_t adaptive_picks = 0;   // NbRayTracingTiles, Halton index
  uint32_t frames_done = 0;       // whole-frame iterations since reset (crh_render continues from here)
};

// ---- path state, queues and the launch schedule
struct ScheduleState {
  // speculative look-ahead for the +1-spp-per-Redraw boundary: frames [pending_first, pending_first + pending_n) are traced
  // and wait in the path buffer (batch sample index pending_off ...) to be folded in by the next crh_render calls
  uint32_t lookahead = 1, pending_first = 0, pending_n = 0, pending_off = 0, pending_tiles = 0;
  uint32_t lookahead_auto = 0, ramp_k = 1;               // crh_set_lookahead_auto: the batch grows 1, 4, 16, ... after every restart of the accumulation
  DPaths paths{}; DQueues queues{}; uint32_t path_cap = 0;
  uint32_t stamp_counter = 0;        // DPaths::stamp of the last batch traced (never 0: the radiance buffer is zeroed when it is allocated)
  uint32_t* d_tile_ids = nullptr; uint32_t tile_cap = 0;
  uint32_t* d_tile_ids2 = nullptr; uint32_t tile_cap2 = 0; int tile_list_last = 0;      // render_impl keeps two lists resident (round 6: row-major and sorted)
  uint32_t* d_seeds = nullptr; uint32_t seed_cap = 0;
  // Counters of the CURRENT accumulation (crh_stats counts since the last restart).  Four blocks, used in turn: crh_reset moves on to the next one -- zeroed one
  // restart earlier, stream-ordered -- instead of zeroing the one in use, so the first frame after a restart need not wait for the frames of the old accumulation
  // that are still in flight to have added their last counts (render_impl, frame pipeline).
  DCounters* d_counters = nullptr; DCounters* d_counters_ring = nullptr; uint32_t counter_epoch = 0; hipEvent_t counters_zeroed[4] = {nullptr, nullptr, nullptr, nullptr};
  hipEvent_t reset_ev = nullptr; bool reset_pending = false;       // crh_reset's memsets of the accumulator: the next frame's ACCUMULATE waits for them, its tracing does not
  uint64_t stream_uses = 0;                                        // cstream() calls (anything enqueued on the context's stream) ...
  uint64_t lane_stream_uses[8] = {~0ull, ~0ull, ~0ull, ~0ull, ~0ull, ~0ull, ~0ull, ~0ull};      // ... and the count at which each pipeline stream last forked from it
  uint32_t lane_counter_epoch[8] = {~0u, ~0u, ~0u, ~0u, ~0u, ~0u, ~0u, ~0u};                    // the accumulation whose counter block each pipeline stream has waited for
  bool pipe_running[8] = {false, false, false, false, false, false, false, false};      // a frame was submitted on that pipeline stream and has not been seen finished (the frames-in-flight estimate)
  uint32_t* d_api_cursor = nullptr;   // work cursor of the API-level trace kernels
  void* d_scratch = nullptr; size_t scratch_bytes = 0;
  bool clamp_grid = true;   // persistent traversal grids are clamped to what the register budget keeps resident (kernels.hip, resident_grid)
  int packets = 64;         // smallest run of consecutive samples per pixel (sample_group) from which wide batches walk their camera rays as wavefront packets (kernels.hip, k_trace_packets): 64 = a wavefront is ONE pixel (C3 +3.3 % at 128 spp per batch, +1.6 % at 64); with two / four pixels per wavefront (32 / 16 samples each) the shared walk loses (C3 -2 % / -5.6 %); CRH_PACKETS=<n> sets the threshold, 0 switches packets off
  bool donate = true;       // small batches use the work-donating traversal kernels (kernels.hip, DON); CRH_DONATE=0 switches them off
                            // (measured with plain kernels + wider grids for the first 1-4 bounces: 232 -> 232 / 226 / 222 / 218 Redraw/s: donate from bounce 0)
  // Small batches (one Redraw() = +1 spp of one frame, AppViewer.cxx:1045-1047) are launch- and drain-bound: every traversal
  // launch ends with the longest rays of a few wavefronts while the rest of the chip idles.  Such a batch is cut into `n_lanes`
  // tile ranges that run the same wavefront schedule on their own streams and their own slice of the path state, so one
  // range's drain phases overlap the others' busy phases.  Pixels, seeds and the per-pixel accumulation order do not change.
  // Measured on C3 at 1080p, 1 spp per call (tools/bench_interactive.py): 1 / 2 / 4 / 8 ranges -> 162 / 175 / 114 / 84 Redraw/s: two
  // concurrent schedules overlap, more of them only add launches that each end in their own ~0.4 ms drain.
  uint32_t n_lanes = 2, lane_max_paths = 12u << 20;
  hipStream_t lane_stream[8] = {}; hipEvent_t lane_fork = nullptr, lane_join[8] = {}; uint32_t* d_lane_counts = nullptr;
  std::vector<uint32_t> h_tile_ids;      // what d_tile_ids holds (an unchanged tile list is not uploaded again)
  std::vector<uint32_t> h_tile_ids2;     // ... and d_tile_ids2
  // frame pipelining: consecutive small whole batches (one Redraw() each) run on alternating streams and path-state halves, so
  // the drain-bound late bounces of frame n overlap the throughput-bound first bounces of frame n + 1; accumulation stays in
  // frame order (an event between the two accumulate launches)
  bool pipeline = true; bool pipe_pending[8] = {false, false, false, false, false, false, false, false}; uint32_t pipe_seq = 0; int pipe_div = 4096;
  uint64_t pipe_total = 0;         // batch size of the frames in flight (their path-state slices are laid out by it)
  hipEvent_t pipe_resized = nullptr; bool pipe_resized_pending = false;      // complete when the last frame of the PREVIOUS batch size is done (render_impl)
  uint32_t pipe_last_depth = 0;    // frames in flight the last pipelined frame was submitted with
  std::chrono::steady_clock::time_point pipe_last_submit{};      // when the previous pipelined frame was submitted
  int pipe_grid_min = 192, pipe_grid_min_shade = 512;      // floors of a pipelined frame's traversal / streaming grids
  uint32_t pipe_depth = 3;         // frames in flight: 2 / 3 / 4 -> 323 / 391 / 312 Redraw/s on C3, 448 / 558 / 453 on C2
  bool read_since_render = true;   // a host that looks at every frame (read-back / sync between Redraws) gets the two-range schedule instead
  // path slots per batch (196 B each = 105 GB of the 288 GB; allocated on demand, so small renders stay small).  Every launch of
  // the wavefront schedule ends in a drain phase whose length does not depend on the launch's size (~0.24 ms per launch on C3), so
  // the batch is made as wide as the memory comfortably allows: 32 M / 64 M / 128 M / 256 M / 512 M slots -> 2745 / 2960 / 3114 /
  // 3205 / 3243 Mrays/s on C3 in round 2; round 4, 512 samples per call: 128 M / 256 M / 512 M / 1 G -> 3981 / 4132 / 4204 / 4180
  uint32_t max_paths = 512u << 20;
  // The frame kernel (k_frame.h): a small batch in ONE launch -- every workgroup streams its own paths through ray generation, traversal and shading, no
  // launch boundary between bounces.  Takes the place of the staged small-batch schedule (lanes / pipelined launches per bounce) wherever a batch is small,
  // not counted and not timed per kernel; CRH_FRAME_KERNEL=0 or crh_set_schedule(CRH_SCHEDULE_STAGED) keeps the staged form (the reference of the sequence tests).
  bool frame_kernel = true, auto_frame_kernel = true;
  uint32_t frame_live = 4096, frame_chunk = 128;    // paths a workgroup (16 wavefronts, one per compute unit) keeps alive at most; path slots a wavefront claims at a time
                                                    // (round 6, with the miss ring: chunk 128 / 192 / 256 / 512 -> lone frame on C3 2.85 / 2.97 / 2.95 / 3.46 ms, profiles/r6/lone_frame.md)
  uint32_t frame_low_water = 512;                   // a feeder wavefront claims the next chunk once fewer rays than this wait in the workgroup's ring
  uint32_t frame_starve = 1u << 20;                 // a feeder shades fewer than 64 hits only while fewer rays than this wait in the ring
  uint32_t frame_feeders = 3, frame_claim_step = 0;    // wavefronts that only shade and generate (round 6: 2 / 3 / 4 / 5 -> 3.07 / 2.85 / 2.92 / 3.39 ms lone frame on C3, 3.0 / 2.60 / 2.30 / 2.31 on the CAD-like scene,
                                                    // whose short walks leave more shading and generating per traced ray: CRH_FRAME_FEED=4 there); tracer w takes rays only
                                                       // while >= w * claim_step wait: 0 / 16 / 32 / 64 -> 2.90 / 2.95 / 3.13 / 3.46 ms -- the shared rings gather the late bounces by themselves
  // The feeder count is chosen by measurement (round 6) unless CRH_FRAME_FEED fixes it: scenes with short walks (tessellated parts, 13.6 node visits per ray) want 4
  // feeders -- lone frame 2.62 -> 2.36 ms, drag 425 -> 507 frames/s on CAD1M -- triangle soups 3 (4 costs them 1 - 2 %).  After every crh_build (and a change of the
  // target size or the depth) the first pipelined frame-kernel frames run in blocks of 6 -- a warm-up block, then 3 / 4 / 3 / 4 feeders; the device time of each
  // frame KERNEL (its own event pair, first frame of a block left out) decides: 4 if that is >= 5 % faster, else 3 (CAD1M: 16 %; the soups: within +- 3 %).  No image depends on the count (tests/test_frame_kernel.py).
  struct FeedTune {
    bool on = true; uint32_t chosen = 0, frames = 0, n[2] = {0, 0}; double ms[2] = {0.0, 0.0};
    struct Pend { hipEvent_t e0, e1; int which; };
    std::deque<Pend> pend;
    void restart() { chosen = 0; frames = 0; n[0] = n[1] = 0; ms[0] = ms[1] = 0.0; for (Pend& q : pend) q.which = -1; }
  } feed_tune;
  // The ORDER in which a lone frame's tiles are claimed (round 6; CRH_TILE_ORDER=0: row-major as up to round 5).  Whatever the frame kernel claims last runs out its
  // bounces on an emptying chip (the drain is 0.43 of a lone frame); pixels do not depend on the order (the RNG is seeded per pixel).  k_accumulate sums the rays
  // each tile's paths traced (frame-kernel frames of a host that waits for its frames only); a restart copies the sums to the host and zeroes them; crh_render
  // then lists the tiles most-rays-first -- for a host whose frames start on an idle chip (six calls in a row with nothing in flight); otherwise row-major.
  // Both lists stay resident on the device (render_impl).  Measured (profiles/r6/lone_frame.md 2c, tile_order_product_ab.txt): lone frame -6 % on CAD1M, -2 % on
  // C2, +-1 % on C3; the drag, display and free-running loops unchanged.
  struct TileOrder {
    bool on = true;
    uint32_t* d_cost = nullptr; uint32_t* h_cost = nullptr; uint32_t n = 0;     // per tile id: rays since the last restart (device), the last restart's copy (pinned host)
    hipEvent_t copied = nullptr; bool pending = false, dirty = false;           // a copy is under way; frames have added to d_cost since the last copy
    std::vector<uint8_t> cls; std::vector<uint32_t> order;                      // the classes the current list was made from; the list (empty: row-major)
    uint64_t reorders = 0, calls_sorted = 0, calls_row_major = 0, frames_collected = 0; uint32_t streak = 0;
    // ... and whether the sorted list pays on THIS scene is measured, once per crh_build (a soup that does not fit the caches loses more by the scattered tiles than
    // the shorter drain gives back: C5's lone 4K frame 10.9 -> 12.1 ms): once a sorted list exists and the feeder count is settled, lone frames take the two lists
    // in turn, each frame kernel between two events; sorted stays if its mean is >= 2 % shorter.  verdict: 0 measuring, 1 sorted, 2 row-major.
    uint32_t verdict = 0, trials = 0, tn[2] = {0, 0}; double tms[2] = {0.0, 0.0}; int tag_next = -1;
    struct Pend { hipEvent_t e0, e1; int which; };
    std::deque<Pend> pend;
    void remeasure() { verdict = 0; trials = 0; tn[0] = tn[1] = 0; tms[0] = tms[1] = 0.0; tag_next = -1; for (Pend& q : pend) q.which = -1; }                                 // crh_render calls in a row that found no frame in flight
  } tile_order;
  uint32_t last_running = 0;                        // frames in flight when the last pipelined frame was submitted (render_impl)
  uint32_t frame_help = 256;                        // a tracer wavefront shades a batch itself once this many hit records wait
  uint32_t frame_help_low = 0;                      // ... and prefers a full shading batch to tracing while fewer rays than this wait (CRH_FRAME_HELP_LOW)
  int frame_grid = 0;                               // workgroups of a lone frame (0: what is resident, 4 per CU)
  uint32_t frame_pipe_depth = 2;                    // a pipelined frame takes the frame kernel while fewer than this many frames are running, and at most this many frame
                                                    // kernels run at a time (they share the chip by compute units); beyond it the staged form carries the deeper pipeline
  int schedule = CRH_SCHEDULE_AUTO; uint32_t auto_lane_max_paths = 12u << 20; bool auto_donate = true, auto_pipeline = true;   // crh_set_schedule
};

// ---- asynchronous read-back (crh_read_ldr_begin / _end, crh_read_hdr_begin / _end): tone map + device-to-host copy of the frame as submitted so far run on
// their own stream into one of two device / pinned-host buffer pairs while the next Redraw()s are already rendering; only the NEXT
// accumulate waits (for the tone map, which reads the accumulator), nothing else does
struct ReadbackState {
  hipStream_t rb_stream = nullptr; hipEvent_t rb_fork = nullptr, rb_tm[2] = {nullptr, nullptr}, rb_done[2] = {nullptr, nullptr};
  uint8_t* d_rb[2] = {nullptr, nullptr}; uint8_t* h_rb[2] = {nullptr, nullptr}; size_t rb_cap = 0, rb_bytes[2] = {0, 0}; bool rb_hdr[2] = {false, false};
  uint32_t rb_head = 0, rb_outstanding = 0; bool rb_guard_pending = false; hipEvent_t rb_guard = nullptr;
};

// ---- counters and timing
struct TimingState {
  bool counters_on = false, timing_on = false;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> render_ev, trace_ev;
  std::vector<hipEvent_t> ev_pool;
  double seconds_acc = 0.0, trace_ms_acc = 0.0, all_ms_acc = 0.0; uint64_t trace_launches = 0;
};

}  // namespace crh

struct crh_ctx : crh::HostInputs, crh::TwoLevelState, crh::BuiltScene, crh::FrameState, crh::ScheduleState, crh::ReadbackState, crh::TimingState {
  int device = 0;
  hipStream_t stream_ = nullptr;   // use cstream(c): it first joins frames still in flight on the pipeline streams
  int cus = 0;            // compute units (0: unknown)
  int grid = 1024;        // streaming / shading kernels: 4 workgroups per CU = what k_shade's 128 VGPRs keep resident; every workgroup of the
                          // persistent loops then starts at once and the next bounce's queue keeps the block-major order.  Measured, workgroups
                          // 512 / 768 / 1024 / 1280 / 2048: C5 2842 / 2874 / 2925 / 2839 / 2813, C3 3507 / 3688 / 3807 / 3787 / 3777, C2 4649 / 4849 / 4967 / 4975 / 4941 Mrays/s
  int grid_trace = 1536;  // traversal kernels: 6 workgroups (= 6 waves/SIMD) per CU -- measured: 4 / 5 / 6 / 7 / 8 per CU -> 3300 / 3424 /
                          // 3448 / 3448 / 3443 Mrays/s on C3 (more rays in flight enlarge the working set the 4 MB-per-XCD L2s hold)
  std::string err;
};

// The context's stream.  Small whole-frame batches alternate between two pipeline streams (render_impl) and are joined lazily:
// whoever wants to enqueue on, or wait for, the context's stream first makes it wait for the frames still in flight.
static inline hipStream_t cstream(crh_ctx* c)
{
  for (int k = 0; k < 8; ++k)
    if (c->pipe_pending[k]) { hipStreamWaitEvent(c->stream_, c->lane_join[k], 0); c->pipe_pending[k] = false; }
  if (c->rb_guard_pending) { hipStreamWaitEvent(c->stream_, c->rb_guard, 0); c->rb_guard_pending = false; }      // an asynchronous read-back still tone-maps the accumulator
  c->read_since_render = true;      // something other than the next frame used the stream (render_impl clears this when it is done)
  ++c->stream_uses;
  return c->stream_;
}

#define CRH_HIP(call)                                                                             \
  do { hipError_t e_ = (call); if (e_ != hipSuccess) {                                            \
      char b_[512]; snprintf(b_, sizeof b_, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
      c->err = b_; return CRH_E_DEVICE; } } while (0)

namespace crh {
namespace api {

// ---- crh_context.cpp
int fail(crh_ctx* c, int code, const char* msg);
bool all_finite(const float* v, size_t n, float limit = 3.0e38f);
int stage_copy(crh_ctx* c, void* dst, const void* src, size_t bytes, hipStream_t on = nullptr);
hipEvent_t get_event(crh_ctx* c);
void drain_events(crh_ctx* c);
int check_device_error(crh_ctx* c);      // CRH_E_DEVICE + message if a frame-kernel workgroup gave up (crh_context.cpp)
int trim_events(crh_ctx* c);
void discard_events(crh_ctx* c);
int ensure_paths(crh_ctx* c, uint32_t need);
int ensure_scratch(crh_ctx* c, size_t bytes);
uint32_t next_stamp(crh_ctx* c);     // DPaths::stamp for the batch about to be traced
int alloc_accum(crh_ctx* c);
int do_reset(crh_ctx* c);
int build_threads_env();      // CRH_BUILD_THREADS (0: every usable CPU)
int hw_queues();              // GPU_MAX_HW_QUEUES as this library found it when it was loaded (default 4)
uint32_t pipeline_capacity(); // frames crh_set_pipeline_depth accepts in this process: min(8, max(3, hw_queues - 2))

// Every copy and memset goes through the context's own stream: it is created non-blocking, so work on the null stream (plain
// hipMemset / hipMemcpy) is NOT ordered with it -- a hipMemset of the queue counters on the null stream used to land in the middle
// of the first batch of a fresh context when other contexts kept the device busy (lost and doubled paths).
template <class T> int dev_upload(crh_ctx* c, T*& dptr, const void* src, size_t bytes)
{
  CRH_HIP(hipStreamSynchronize(cstream(c)));
  if (dptr) { CRH_HIP(hipFree(dptr)); dptr = nullptr; }
  if (!bytes) return CRH_OK;
  CRH_HIP(hipMalloc((void**)&dptr, bytes));
  CRH_HIP(hipMemcpyAsync(dptr, src, bytes, hipMemcpyHostToDevice, cstream(c)));
  CRH_HIP(hipStreamSynchronize(cstream(c)));            // `src` may be a temporary of the caller
  return CRH_OK;
}

// Refresh a device array in place: the allocation is kept (and grown with head-room only when it is too small), the bytes travel
// stream-ordered.  Large blocks (environment maps, whole node arrays) are copied straight from the caller's memory and waited for.
template <class T> int dev_put(crh_ctx* c, T*& dptr, size_t& cap, const void* src, size_t bytes, size_t headroom = 0)
{
  if (bytes > cap || !dptr) {
    CRH_HIP(hipStreamSynchronize(cstream(c)));          // kernels in flight may still read the old allocation
    if (dptr) { CRH_HIP(hipFree(dptr)); dptr = nullptr; cap = 0; }
    const size_t want = std::max<size_t>(bytes + headroom, 256);
    CRH_HIP(hipMalloc((void**)&dptr, want));
    cap = want;
  }
  if (bytes <= (4u << 20)) return stage_copy(c, dptr, src, bytes);
  CRH_HIP(hipMemcpyAsync(dptr, src, bytes, hipMemcpyHostToDevice, cstream(c)));
  CRH_HIP(hipStreamSynchronize(cstream(c)));
  return CRH_OK;
}

// ---- crh_scene.cpp
void fill_scene(const crh_ctx* c, DScene& S);
int upload_textures(crh_ctx* c);

// ---- crh_schedule.cpp
uint32_t frame_seed(uint32_t seed, uint32_t n);

// ---- crh_reduce.cpp
void release_comms(crh_ctx* c);
int reduce_fake_devices(crh_ctx* const* ctxs, uint32_t n, uint32_t root);      // crh_reduce.cpp: the RCCL branch of crh_reduce on contexts that share a device (test hook)

}  // namespace api
}  // namespace crh

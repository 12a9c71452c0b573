// crh_api.cpp -- the C ABI of libcadrays_hip.so (include/cadrays_hip.h): context, HBM residency of the
// scene, and the per-iteration wavefront schedule.  This is the code that sits behind CADRays'
// `myInternal->View->Redraw()` (reference src/Launcher/AppViewer.cxx:1047).
//
// There is NO CPU fallback: every entry point that renders or traces launches the gfx950 kernels and
// reports CRH_E_DEVICE when the HIP runtime refuses.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <rccl/rccl.h>     // types only: the library is loaded with dlopen on the first multi-device crh_reduce

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../include/cadrays_hip.h"
#include "../../include/crh_xform.h"
#include "bvh_builder.h"
#include "kernels.h"

using namespace crh;

struct crh_ctx {
  int device = 0;
  hipStream_t stream_ = nullptr;   // use cstream(c): it first joins frames still in flight on the pipeline streams
  int cus = 0;            // compute units (0: unknown)
  int grid = 1024;        // streaming / shading kernels: 4 workgroups per CU = what k_shade's 128 VGPRs keep resident; every workgroup of the
                          // persistent loops then starts at once and the next bounce's queue keeps the block-major order.  Measured, workgroups
                          // 512 / 768 / 1024 / 1280 / 2048: C5 2842 / 2874 / 2925 / 2839 / 2813, C3 3507 / 3688 / 3807 / 3787 / 3777, C2 4649 / 4849 / 4967 / 4975 / 4941 Mrays/s
  int grid_trace = 1536;  // traversal kernels: 6 workgroups (= 6 waves/SIMD) per CU -- measured: 4 / 5 / 6 / 7 / 8 per CU -> 3300 / 3424 /
                          // 3448 / 3448 / 3443 Mrays/s on C3 (more rays in flight enlarge the working set the 4 MB-per-XCD L2s hold)
  std::string err;
  // ---- host copies of the inputs
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
  // ---- two-level mode (per-object transforms)
  bool two_level = false; uint32_t nO = 0;
  std::vector<float> xf; std::vector<int32_t> tri_obj;
  std::vector<float> xf0, pos_w, nrm_w;   // the transforms the scene was built with; per vertex: position / unit normal under its object's build-time transform (what the static tree holds)
  struct Inst { float fwd[12], inv[12], bmin[3], bmax[3]; uint32_t root, obj; };
  std::vector<Inst> inst;                 // the objects rendered as instances RIGHT NOW, ascending object index (empty: the scene is one world-space tree)
  uint32_t n_blas_nodes = 0, root = 0;    // nodes of the static tree + the object trees built so far (the top-level tree follows them); entry point of the walk
  // static / moved split (reference: the gizmo moves ONE object per drag, ImRaytraceControls.cxx:64,88): every object is baked into
  // one world-space tree with the transform it has when the scene is built; an object that is moved away from that placement has its triangles there
  // disabled and gets an object tree of its own (built on first need, kept); nothing else is ever rebuilt by crh_set_transforms but the top-level tree
  struct Obj { bool static0 = false, built = false, is_inst = false; uint32_t root = 0, first = 0, ntri = 0; float bmin[3] = {0, 0, 0}, bmax[3] = {0, 0, 0}; };
  std::vector<Obj> objs; std::vector<uint32_t> obj_tris, static_pos, pos_obj;   // pos_obj: object of the triangle at a leaf position >= n_static
  uint32_t n_static = 0, n_static_live = 0, n_pos = 0; float sbmin[3] = {0, 0, 0}, sbmax[3] = {0, 0, 0};
  uint32_t root2 = 0xFFFFFFFFu; float tlas_lo[3] = {0, 0, 0}, tlas_hi[3] = {0, 0, 0};
  size_t cap_pos = 0;                     // leaf positions the triangle / shading / uv arrays have room for
  void* d_patch = nullptr; size_t cap_patch = 0;
  int split_passes = -1;                  // CRH_SPLIT_PASSES: -1 auto, 0 one walk, 1 two passes
  float4* d_ibox = nullptr;               // spheres around the instances' world boxes when there are at most kMaxIBox (the "does the ray come near a moved object" test)
  float usph[4] = {0.f, 0.f, 0.f, 0.f};  // ... and around the bounds of all of them
  float4* d_inst = nullptr;
  // ---- built scene
  QBvh bvh;
  std::vector<float> h_tris;      // 12 floats per triangle, leaf order
  bool built = false;
  // ---- device
  float4 *d_nodes = nullptr, *d_tris = nullptr, *d_shade = nullptr, *d_mats = nullptr, *d_lights = nullptr, *d_env = nullptr;
  float4 *d_uvs = nullptr, *d_texels = nullptr; uint4* d_tex_desc = nullptr;
  float4* d_verts = nullptr;      // two-level scenes: object-space vertices per leaf position (shading of instance hits)
  float4* d_accum = nullptr; uint32_t accumW = 0, accumH = 0;
  float* d_m2 = nullptr;            // running mean of squared luminance (adaptive sampling only)
  // crh_reduce: the frame assembled from all shards lives beside the root's own accumulator (rendering continues into that)
  float4* d_assembled = nullptr; float4* d_peer_stage = nullptr; uint32_t assembledW = 0, assembledH = 0; bool assembled_valid = false;
  std::vector<ncclComm_t> comms; std::vector<crh_ctx*> comm_ctxs;      // RCCL communicators of the last multi-device group (kept on the root)
  float* d_tile_err = nullptr; uint32_t* d_tile_cnt = nullptr; uint32_t tile_stat_cap = 0;
  bool show_tiles = false; bool picked_valid = false;                                   // ShowSamplingTiles: d_picked marks the tiles of the last adaptive iteration
  float* d_tile_cdf = nullptr; uint8_t* d_picked = nullptr; uint32_t* d_adapt_n = nullptr;   // adaptive sampler state in HBM (running sum, drawn-tile mask, tile count)
  bool adaptive = false; uint32_t adaptive_tiles = 128; uint32_t adaptive_picks = 0;   // NbRayTracingTiles, Halton index
  // speculative look-ahead for the +1-spp-per-Redraw boundary: frames [pending_first, pending_first + pending_n) are traced
  // and wait in the path buffer (batch sample index pending_off ...) to be folded in by the next crh_render calls
  uint32_t lookahead = 1, pending_first = 0, pending_n = 0, pending_off = 0, pending_tiles = 0;
  uint32_t lookahead_auto = 0, ramp_k = 1;               // crh_set_lookahead_auto: the batch grows 1, 4, 16, ... after every restart of the accumulation
  DPaths paths{}; DQueues queues{}; uint32_t path_cap = 0;
  uint32_t* d_tile_ids = nullptr; uint32_t tile_cap = 0;
  uint32_t* d_seeds = nullptr; uint32_t seed_cap = 0;
  DCounters* d_counters = nullptr;
  uint32_t* d_api_cursor = nullptr;   // work cursor of the API-level trace kernels
  void* d_scratch = nullptr; size_t scratch_bytes = 0;
  // setters called every GUI frame (material-editor drags MaterialEditor.cxx:331-337, manipulator moves ImRaytraceControls.cxx:58-89)
  // reuse their device allocations (capacities below) and copy through a small ring of pinned staging buffers on the context's
  // stream: no hipFree / hipMalloc, no device-wide synchronisation, kernels still in flight keep reading the old bytes
  size_t cap_nodes = 0, cap_inst = 0, cap_mats = 0, cap_lights = 0, cap_env = 0;
  struct Stage { void* p = nullptr; size_t cap = 0; hipEvent_t ev = nullptr; bool used = false; } stage[4];
  uint32_t stage_next = 0;
  bool clamp_grid = true;   // CRH_CLAMP_GRID=0: A/B switch for the occupancy clamp of the traversal grids
  bool donate = true;       // small batches use the work-donating traversal kernels (kernels.hip, DON); CRH_DONATE=0 switches them off
                            // (measured with plain kernels + wider grids for the first 1-4 bounces: 232 -> 232 / 226 / 222 / 218 Redraw/s: donate from bounce 0)
  // Small batches (one Redraw() = +1 spp of one frame, AppViewer.cxx:1045-1047) are launch- and drain-bound: every traversal
  // launch ends with the longest rays of a few wavefronts while the rest of the chip idles.  Such a batch is cut into `n_lanes`
  // tile ranges that run the same wavefront schedule on their own streams and their own slice of the path state, so one
  // range's drain phases overlap the others' busy phases.  Pixels, seeds and the per-pixel accumulation order do not change.
  // Measured on C3 at 1080p, 1 spp per call (tools/bench_interactive.py): 1 / 2 / 4 / 8 ranges -> 162 / 175 / 114 / 84 Redraw/s: two
  // concurrent schedules overlap, more of them only add launches that each end in their own ~0.4 ms drain.
  uint32_t n_lanes = 2, lane_max_paths = 12u << 20; int lane_grid = 0, lane_grid_trace = 0;
  hipStream_t lane_stream[8] = {}; hipEvent_t lane_fork = nullptr, lane_join[8] = {}; uint32_t* d_lane_counts = nullptr;
  std::vector<uint32_t> h_tile_ids;      // what d_tile_ids holds (an unchanged tile list is not uploaded again)
  // frame pipelining: consecutive small whole batches (one Redraw() each) run on alternating streams and path-state halves, so
  // the drain-bound late bounces of frame n overlap the throughput-bound first bounces of frame n + 1; accumulation stays in
  // frame order (an event between the two accumulate launches)
  bool pipeline = true; bool pipe_pending[8] = {false, false, false, false, false, false, false, false}; uint32_t pipe_seq = 0; uint32_t* d_pipe_seeds = nullptr; int pipe_div = 4096;
  uint64_t pipe_total = 0;         // batch size of the frames in flight (their path-state slices are laid out by it)
  std::chrono::steady_clock::time_point pipe_last_submit{};      // when the previous pipelined frame was submitted
  int pipe_grid_min = 192, pipe_grid_min_shade = 512;      // floors of a pipelined frame's traversal / streaming grids
  uint32_t pipe_depth = 3;         // frames in flight: 2 / 3 / 4 -> 323 / 391 / 312 Redraw/s on C3, 448 / 558 / 453 on C2
  // asynchronous LDR read-back (crh_read_ldr_begin / _end): tone map + device-to-host copy of the frame as submitted so far run on their
  // own stream into one of two device / pinned-host buffer pairs while the next Redraw()s are already rendering; only the NEXT
  // accumulate waits (for the tone map, which reads the accumulator), nothing else does
  hipStream_t rb_stream = nullptr; hipEvent_t rb_fork = nullptr, rb_tm[2] = {nullptr, nullptr}, rb_done[2] = {nullptr, nullptr};
  uint8_t* d_rb[2] = {nullptr, nullptr}; uint8_t* h_rb[2] = {nullptr, nullptr}; size_t rb_cap = 0, rb_bytes[2] = {0, 0}; bool rb_hdr[2] = {false, false};
  uint32_t rb_head = 0, rb_outstanding = 0; bool rb_guard_pending = false; hipEvent_t rb_guard = nullptr;
  bool read_since_render = true;   // a host that looks at every frame (read-back / sync between Redraws) gets the two-range schedule instead
  bool counters_on = false, timing_on = false;
  uint32_t frames_done = 0;       // whole-frame iterations since reset (crh_render continues from here)
  std::vector<std::pair<hipEvent_t, hipEvent_t>> render_ev, trace_ev;
  std::vector<hipEvent_t> ev_pool;
  double seconds_acc = 0.0, trace_ms_acc = 0.0, all_ms_acc = 0.0; uint64_t trace_launches = 0;
  // path slots per batch (196 B each = 53 GB of the 288 GB; allocated on demand, so small renders stay small).  Every launch of
  // the wavefront schedule ends in a drain phase whose length does not depend on the launch's size (~0.24 ms per launch on C3), so
  // the batch is made as wide as the memory comfortably allows: 32 M / 64 M / 128 M / 256 M / 512 M slots -> 2745 / 2960 / 3114 /
  // 3205 / 3243 Mrays/s on C3
  uint32_t max_paths = 256u << 20;
  int schedule = CRH_SCHEDULE_AUTO; uint32_t auto_lane_max_paths = 12u << 20; bool auto_donate = true, auto_pipeline = true;   // crh_set_schedule
};

// The context's stream.  Small whole-frame batches alternate between two pipeline streams (render_impl) and are joined lazily:
// whoever wants to enqueue on, or wait for, the context's stream first makes it wait for the frames still in flight.
static inline hipStream_t cstream(crh_ctx* c)
{
  for (int k = 0; k < 8; ++k)
    if (c->pipe_pending[k]) { hipStreamWaitEvent(c->stream_, c->lane_join[k], 0); c->pipe_pending[k] = false; }
  if (c->rb_guard_pending) { hipStreamWaitEvent(c->stream_, c->rb_guard, 0); c->rb_guard_pending = false; }      // an asynchronous read-back still tone-maps the accumulator
  c->read_since_render = true;      // something other than the next frame used the stream (render_impl clears this when it is done)
  return c->stream_;
}

namespace {

#define CRH_HIP(call)                                                                             \
  do { hipError_t e_ = (call); if (e_ != hipSuccess) {                                            \
      char b_[512]; snprintf(b_, sizeof b_, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
      c->err = b_; return CRH_E_DEVICE; } } while (0)

int fail(crh_ctx* c, int code, const char* msg) { if (c) c->err = msg; return code; }

// The boundary takes finite numbers only (coordinates additionally |x| <= 1e30, so that box centres and extents stay finite):
// NaN / Inf would otherwise reach the BVH builder's binning and the kernels' float -> int conversions.
bool all_finite(const float* v, size_t n, float limit = 3.0e38f)
{
  for (size_t i = 0; i < n; ++i) if (!(v[i] >= -limit && v[i] <= limit)) return false;
  return true;
}

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

// Stream-ordered copy of a small host block: staged through one of four pinned buffers, so the call returns at once and `src`
// can be reused; a slot is waited for only when the copy issued four uploads earlier has not finished yet.
int stage_copy(crh_ctx* c, void* dst, const void* src, size_t bytes, hipStream_t on = nullptr)
{
  if (!bytes) return CRH_OK;
  const hipStream_t stream = on ? on : cstream(c);
  crh_ctx::Stage& st = c->stage[c->stage_next++ & 3u];
  if (st.used) CRH_HIP(hipEventSynchronize(st.ev));
  if (st.cap < bytes) {
    if (st.p) { CRH_HIP(hipHostFree(st.p)); st.p = nullptr; st.cap = 0; }
    const size_t want = bytes + bytes / 2 + 4096;
    CRH_HIP(hipHostMalloc(&st.p, want, hipHostMallocDefault));
    st.cap = want;
  }
  if (!st.ev) CRH_HIP(hipEventCreateWithFlags(&st.ev, hipEventDisableTiming));
  std::memcpy(st.p, src, bytes);
  CRH_HIP(hipMemcpyAsync(dst, st.p, bytes, hipMemcpyHostToDevice, stream));
  CRH_HIP(hipEventRecord(st.ev, stream));
  st.used = true;
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

hipEvent_t get_event(crh_ctx* c)
{
  if (!c->ev_pool.empty()) { hipEvent_t e = c->ev_pool.back(); c->ev_pool.pop_back(); return e; }
  hipEvent_t e = nullptr; hipEventCreate(&e); return e;
}

// Fold finished event pairs into the accumulated times (stream must be idle).
void drain_events(crh_ctx* c)
{
  for (auto& p : c->render_ev) { float ms = 0.f; if (hipEventElapsedTime(&ms, p.first, p.second) == hipSuccess) { c->seconds_acc += 1e-3 * ms; c->all_ms_acc += ms; } c->ev_pool.push_back(p.first); c->ev_pool.push_back(p.second); }
  c->render_ev.clear();
  for (auto& p : c->trace_ev) { float ms = 0.f; if (hipEventElapsedTime(&ms, p.first, p.second) == hipSuccess) { c->trace_ms_acc += ms; c->trace_launches++; } c->ev_pool.push_back(p.first); c->ev_pool.push_back(p.second); }
  c->trace_ev.clear();
}

// Event pairs wait in the context until someone asks for the times; an interactive loop that never does (one crh_render(1)
// per GUI frame, adaptive or look-ahead) must not grow the lists without bound: fold them in every 4096 pairs.
int trim_events(crh_ctx* c)
{
  if (c->render_ev.size() + c->trace_ev.size() > 4096) { CRH_HIP(hipStreamSynchronize(cstream(c))); drain_events(c); }
  return CRH_OK;
}

// A restart does not need the old epoch's times: hand the events back without waiting for them (no stream synchronisation).
void discard_events(crh_ctx* c)
{
  for (auto& p : c->render_ev) { c->ev_pool.push_back(p.first); c->ev_pool.push_back(p.second); }
  for (auto& p : c->trace_ev) { c->ev_pool.push_back(p.first); c->ev_pool.push_back(p.second); }
  c->render_ev.clear(); c->trace_ev.clear();
}

uint32_t frame_seed(uint32_t seed, uint32_t n)   // Bullard generator, SURVEY.md a14
{
  uint32_t hi = seed, lo = seed ^ 0x49616E42u, r = 0;
  for (uint32_t i = 0; i <= n; ++i) { hi = (hi >> 2) + (hi << 2); hi += lo; lo += hi; r = hi; }
  return r >> 2;
}

int ensure_paths(crh_ctx* c, uint32_t need)
{
  if (need <= c->path_cap) return CRH_OK;
  CRH_HIP(hipStreamSynchronize(cstream(c)));
  void** ptrs[] = {(void**)&c->paths.ray_o[0], (void**)&c->paths.ray_d[0], (void**)&c->paths.hit, (void**)&c->paths.thr[0], (void**)&c->paths.rad,
                   (void**)&c->paths.sh_o, (void**)&c->paths.sh_d, (void**)&c->paths.sh_c,
                   (void**)&c->queues.q[0], (void**)&c->queues.q[1], (void**)&c->queues.q_sh,
                   (void**)&c->paths.ray_o[1], (void**)&c->paths.ray_d[1], (void**)&c->paths.thr[1], (void**)&c->queues.q2, (void**)&c->queues.q2_sh};
  const size_t sz[] = {16, 16, 16, 16, 16, 16, 16, 16, 4, 4, 4, 16, 16, 16, 4, 4};
  c->path_cap = 0;                                          // stays 0 if an allocation below fails
  for (int i = 0; i < 16; ++i) {
    if (*ptrs[i]) { CRH_HIP(hipFree(*ptrs[i])); *ptrs[i] = nullptr; }
    CRH_HIP(hipMalloc(ptrs[i], sz[i] * (size_t)need));
  }
  if (!c->queues.counts) { CRH_HIP(hipMalloc((void**)&c->queues.counts, kCounts * sizeof(uint32_t))); CRH_HIP(hipMemsetAsync(c->queues.counts, 0, kCounts * sizeof(uint32_t), cstream(c))); }
  c->path_cap = need;
  return CRH_OK;
}

int ensure_scratch(crh_ctx* c, size_t bytes)
{
  if (bytes <= c->scratch_bytes) return CRH_OK;
  if (c->d_scratch) { CRH_HIP(hipFree(c->d_scratch)); c->d_scratch = nullptr; }
  CRH_HIP(hipMalloc(&c->d_scratch, bytes));
  c->scratch_bytes = bytes;
  return CRH_OK;
}

void fill_scene(const crh_ctx* c, DScene& S)
{
  std::memset(&S, 0, sizeof S);
  S.nodes = c->d_nodes; S.tris = c->d_tris; S.verts = c->d_verts; S.shade = c->d_shade; S.mats = c->d_mats; S.lights = c->d_lights; S.env = (c->envW && c->envH) ? c->d_env : nullptr;
  S.inst = c->d_inst; S.inst_leaf = c->d_inst ? c->d_inst + 8 * (size_t)c->nO : nullptr; S.root = c->root; S.two_level = c->inst.empty() ? 0 : 1;
  S.root2 = c->inst.empty() ? kQEmpty : c->root2;
  // render path of a split scene: with at most kMaxIBox moved objects the producers can tell precisely which rays come near one -- two traversal passes
  // (the plain single-level kernels over everything, the two-level ones over the few flagged rays); with more, most rays would be flagged: one walk
  // "static tree, then top level" in the two-level kernels, like the API-level tracers.  Same hits, same counters either way (CRH_SPLIT_PASSES=0/1 forces one).
  S.split = (S.root2 != kQEmpty && (c->split_passes < 0 ? c->inst.size() <= kMaxIBox : c->split_passes != 0)) ? 1 : 0;
  for (int a = 0; a < 3; ++a) { S.tlas_lo[a] = c->tlas_lo[a]; S.tlas_hi[a] = c->tlas_hi[a]; }
  S.ibox = c->d_ibox; S.n_ibox = (c->d_ibox && c->inst.size() <= kMaxIBox) ? (uint32_t)c->inst.size() : 0u;
  S.usph = make_float4(c->usph[0], c->usph[1], c->usph[2], c->usph[3]);
  {
    const float* lo = c->bvh.bbmin; const float* hi = c->bvh.bbmax;      // bounds of the tree the walk starts in (the world box of a two-level scene)
    S.guard_box = make_float4((lo[0] + hi[0]) * 0.5f, (lo[1] + hi[1]) * 0.5f, (lo[2] + hi[2]) * 0.5f, (((hi[0] - lo[0]) + (hi[1] - lo[1])) + (hi[2] - lo[2])) * 0.5f);
  }
  S.uvs = c->d_uvs; S.texels = c->d_texels; S.tex_desc = c->d_tex_desc; S.n_tex = c->d_tex_desc ? (uint32_t)c->textures.size() : 0u;
  S.n_mats = (uint32_t)c->mats.size(); S.n_lights = (uint32_t)c->lights.size(); S.env_w = c->envW; S.env_h = c->envH;
  for (int k = 0; k < 3; ++k) S.bg[k] = c->par.background[k];
  S.env_as_bg = c->par.env_as_background;
  S.eye = crh_mk3(c->cam.eye[0], c->cam.eye[1], c->cam.eye[2]);
  S.fwd = crh_norm3(crh_mk3(c->cam.dir[0], c->cam.dir[1], c->cam.dir[2]));
  S.right = crh_norm3(crh_cross3(S.fwd, crh_mk3(c->cam.up[0], c->cam.up[1], c->cam.up[2])));
  S.up = crh_cross3(S.right, S.fwd);
  float s, cs; crh_sincos((c->cam.fovy_deg * 0.5f) * (CRH_PI / 180.0f), &s, &cs);
  S.tan_half = s / cs;
  S.aspect = c->cam.aspect > 0.f ? c->cam.aspect : (float)c->par.width / (float)c->par.height;
  S.ortho_scale = c->cam.ortho_scale; S.aperture = c->cam.aperture_radius; S.focal = c->cam.focal_dist; S.is_ortho = c->cam.is_ortho;
  S.width = c->par.width; S.height = c->par.height; S.max_depth = c->par.max_depth; S.tile_size = c->par.tile_size;
  S.clampv = c->par.radiance_clamp;
  const crh_v3 dg = crh_mk3(c->bvh.bbmax[0] - c->bvh.bbmin[0], c->bvh.bbmax[1] - c->bvh.bbmin[1], c->bvh.bbmax[2] - c->bvh.bbmin[2]);
  S.eps = c->par.scene_epsilon > 0.f ? c->par.scene_epsilon
        : (c->spec.eps_rule ? crh_max(1.0e-6f, 1.0e-4f * (crh_len3(dg) * 0.5f)) : crh_max(1.0e-6f, 1.0e-5f * crh_len3(dg)));      // crh_spec.h #6
  S.two_sided = c->par.two_sided; S.coherent = c->par.coherent_rng; S.rr = c->par.russian_roulette;
  S.spec_u32 = c->spec.uniform_32bit; S.spec_gamma2 = c->spec.texel_gamma2; S.spec_mis1 = c->spec.mis_single_lobe; S.spec_eta_nd = c->spec.eta_no_dielectric;
}

int upload_lights(crh_ctx* c)
{
  std::vector<float> l(8 * std::max<size_t>(c->lights.size(), 1), 0.f);
  for (size_t i = 0; i < c->lights.size(); ++i) {
    const crh_light& s = c->lights[i]; float* o = &l[8 * i];
    if (s.is_point != 0.f) { o[0] = s.vec[0]; o[1] = s.vec[1]; o[2] = s.vec[2]; o[3] = 1.f; o[7] = s.smoothness; }
    else {
      const crh_v3 d = crh_norm3(crh_mk3(-s.vec[0], -s.vec[1], -s.vec[2]));
      float sn, cn; crh_sincos(s.smoothness, &sn, &cn);
      o[0] = d.x; o[1] = d.y; o[2] = d.z; o[3] = 0.f; o[7] = s.smoothness > 0.f ? cn : 1.0f;
    }
    o[4] = s.emission[0]; o[5] = s.emission[1]; o[6] = s.emission[2];
  }
  return dev_put(c, c->d_lights, c->cap_lights, l.data(), l.size() * sizeof(float), 8 * 32);
}

static bool is_identity(const float* m)
{
  static const float I[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
  for (int k = 0; k < 12; ++k) if (m[k] != I[k]) return false;
  return true;
}

// (Re)build the top-level tree over the world boxes of the objects rendered as instances right now, behind the static tree and the object
// trees [0, n_blas_nodes), and refresh the instance table on the device.  Sets the walk's entry points (fill_scene): no instance -> the static
// tree alone; instances + live static triangles -> the static tree, then the top level (root2); no live static triangle -> the top level.
int build_tlas(crh_ctx* c)
{
  c->bvh.nodes.resize(c->n_blas_nodes);
  c->inst.clear();
  for (uint32_t ob = 0; ob < c->nO; ++ob) {
    const crh_ctx::Obj& o = c->objs[ob];
    if (!o.is_inst) continue;
    crh_ctx::Inst in{}; in.obj = ob; in.root = o.root;
    std::memcpy(in.bmin, o.bmin, sizeof in.bmin); std::memcpy(in.bmax, o.bmax, sizeof in.bmax);
    c->inst.push_back(in);
  }
  const uint32_t n = (uint32_t)c->inst.size();
  c->root2 = kQEmpty;
  if (n == 0) {                                 // one world-space tree
    c->root = 0;
    for (int a = 0; a < 3; ++a) { c->bvh.bbmin[a] = c->n_static ? c->sbmin[a] : 0.f; c->bvh.bbmax[a] = c->n_static ? c->sbmax[a] : 0.f; }
    return CRH_OK;
  }
  // instance table: one record per OBJECT (shading looks the hit triangle's object up; only the instances' records are filled) followed
  // by the instances' records in top-level LEAF order (a top-level leaf reference is a position in that order)
  std::vector<float> boxes(6 * (size_t)n, 0.f), table(32 * ((size_t)c->nO + n), 0.f);
  for (uint32_t i = 0; i < n; ++i) {
    crh_ctx::Inst& in = c->inst[i];
    float* rec = &table[32 * (size_t)in.obj];
    std::memcpy(in.fwd, &c->xf[12 * (size_t)in.obj], sizeof in.fwd);
    if (!crh_xform_inverse(in.fwd, in.inv)) std::memset(in.inv, 0, sizeof in.inv);
    crh_xform_box(in.fwd, in.bmin, in.bmax, &boxes[6 * (size_t)i], &boxes[6 * (size_t)i + 3]);
    std::memcpy(rec, in.inv, 48); std::memcpy(rec + 12, in.fwd, 48);
    std::memcpy(rec + 24, &in.root, 4); std::memcpy(rec + 25, &in.obj, 4);
    const float* iv = in.inv;                    // meta.z = 1: the inverse's 3x3 part is exactly the identity (translation only)
    const uint32_t pure_translation = (iv[0] == 1.f && iv[5] == 1.f && iv[10] == 1.f && iv[1] == 0.f && iv[2] == 0.f && iv[4] == 0.f &&
                                       iv[6] == 0.f && iv[8] == 0.f && iv[9] == 0.f) ? 1u : 0u;
    std::memcpy(rec + 26, &pure_translation, 4);
    // the object's own box {centre, L1 half-extent}: the guard band of the slab test inside the object
    for (int a = 0; a < 3; ++a) rec[28 + a] = (in.bmin[a] + in.bmax[a]) * 0.5f;
    rec[31] = (((in.bmax[0] - in.bmin[0]) + (in.bmax[1] - in.bmin[1])) + (in.bmax[2] - in.bmin[2])) * 0.5f;
  }
  std::vector<uint32_t> order;
  const uint32_t troot = build_tree(boxes.data(), n, true, 0, c->bvh.nodes, order, c->tlas_lo, c->tlas_hi, n >= 4096 ? 0 : 1);   // small trees: one thread beats the hand-off
  for (uint32_t p = 0; p < n; ++p) std::memcpy(&table[32 * ((size_t)c->nO + p)], &table[32 * (size_t)c->inst[order[p]].obj], 128);
  if (c->n_static_live) {
    c->root = 0; c->root2 = troot;
    for (int a = 0; a < 3; ++a) { c->bvh.bbmin[a] = crh_min(c->tlas_lo[a], c->sbmin[a]); c->bvh.bbmax[a] = crh_max(c->tlas_hi[a], c->sbmax[a]); }
  } else {
    c->root = troot;
    for (int a = 0; a < 3; ++a) { c->bvh.bbmin[a] = c->tlas_lo[a]; c->bvh.bbmax[a] = c->tlas_hi[a]; }
  }
  if (!c->d_ibox) CRH_HIP(hipMalloc((void**)&c->d_ibox, sizeof(float4) * kMaxIBox));
  crh_box_sphere(c->tlas_lo, c->tlas_hi, c->usph);
  if (n <= kMaxIBox) {
    float ib[4 * kMaxIBox] = {0};
    for (uint32_t i = 0; i < n; ++i) crh_box_sphere(&boxes[6 * (size_t)i], &boxes[6 * (size_t)i + 3], &ib[4 * i]);
    int rc_b = stage_copy(c, c->d_ibox, ib, sizeof(float) * 4 * n); if (rc_b) return rc_b;
  }
  return dev_put(c, c->d_inst, c->cap_inst, table.data(), table.size() * sizeof(float), 32 * sizeof(float) * ((size_t)c->nO + 64));
}

// the object-space tree of object ob, appended behind the trees built so far; its triangles take the next leaf positions
void build_object_tree(crh_ctx* c, uint32_t ob, int threads)
{
  crh_ctx::Obj& o = c->objs[ob];
  const uint32_t m = o.ntri; const uint32_t* mem = &c->obj_tris[o.first];
  std::vector<float> boxes(6 * (size_t)m); std::vector<uint32_t> order;
  for (uint32_t i = 0; i < m; ++i)
    for (int a = 0; a < 3; ++a) {
      const uint32_t t = mem[i];
      const float v0 = c->pos[3 * c->tri[4 * t + 0] + a], v1 = c->pos[3 * c->tri[4 * t + 1] + a], v2 = c->pos[3 * c->tri[4 * t + 2] + a];
      boxes[6 * (size_t)i + a] = std::min(v0, std::min(v1, v2)); boxes[6 * (size_t)i + 3 + a] = std::max(v0, std::max(v1, v2));
    }
  o.root = build_tree(boxes.data(), m, false, c->n_pos, c->bvh.nodes, order, o.bmin, o.bmax, threads);
  for (uint32_t i = 0; i < m; ++i) { c->bvh.prim_order.push_back(mem[order[i]]); c->pos_obj.push_back(ob); }
  c->n_pos += m; o.built = true;
}

// Leaf-ordered device records of positions [p0, p1): 16 floats of triangle (48 B used), 16 floats of shading record, 8 floats of uv.
// Also refreshes the host copy h_tris (12 floats per position, crh_get_bvh).
// device form of one triangle record from its host form q = {v0 | id, v1, v2}: {v0 | n.x}, {e0 | n.y}, {e1 | n.z}, {id} with e0 = v1 - v0, e1 = v0 - v2,
// n = e1 x e0 -- the expressions the traversal kernel used to evaluate per test, evaluated here with the same inline arithmetic (same bits).  A
// disabled triangle (all-zero vertices) stays all zero: n . d = 0, the test yields NaN and rejects.
static void device_tri_record(const float* q, float* d)
{
  const crh_v3 v0 = crh_mk3(q[0], q[1], q[2]), v1 = crh_mk3(q[4], q[5], q[6]), v2 = crh_mk3(q[8], q[9], q[10]);
  const crh_v3 e0 = crh_sub3(v1, v0), e1 = crh_sub3(v0, v2), n = crh_cross3(e1, e0);
  d[0] = v0.x; d[1] = v0.y; d[2] = v0.z; d[3] = n.x;
  d[4] = e0.x; d[5] = e0.y; d[6] = e0.z; d[7] = n.y;
  d[8] = e1.x; d[9] = e1.y; d[10] = e1.z; d[11] = n.z;
  std::memcpy(&d[12], &q[3], 4); d[13] = d[14] = d[15] = 0.f;
}

void fill_records(crh_ctx* c, uint32_t p0, uint32_t p1, std::vector<float>& tr, std::vector<float>& sh, std::vector<float>& uvr, std::vector<float>* verts = nullptr)
{
  const size_t n = p1 - p0;
  tr.assign(4 * (size_t)kTriStride * std::max<size_t>(n, 1), 0.f); sh.assign(16 * std::max<size_t>(n, 1), 0.f);
  if (verts) verts->assign(12 * std::max<size_t>(n, 1), 0.f);
  if (!c->uv.empty()) uvr.assign(8 * std::max<size_t>(n, 1), 0.f); else uvr.clear();
  c->h_tris.resize(12 * (size_t)std::max(p1, 1u), 0.f);
  // every position is independent (gathers from the vertex arrays, writes its own records): ranges of positions on the builder's threads -- at
  // 10 M triangles this loop was 2 s of the 9 s a scene hand-over takes
  auto fill = [&](uint32_t q0, uint32_t q1) {
  for (uint32_t p = q0; p < q1; ++p) {
    const uint32_t t = c->bvh.prim_order[p]; const size_t i = p - p0;
    float* q = &c->h_tris[12 * (size_t)p]; float* s_ = &sh[16 * i];
    // a position of the static tree of a two-level scene holds the BAKED vertex (its object's build-time transform applied); an object tree the object-space one
    const bool baked = c->two_level && p < c->n_static;
    const float* VP = baked ? c->pos_w.data() : c->pos.data(); const float* VN = baked ? c->nrm_w.data() : c->nrm.data();
    for (int k = 0; k < 3; ++k) {
      const int32_t vi = c->tri[4 * t + k];
      for (int a = 0; a < 3; ++a) { q[4 * k + a] = VP[3 * vi + a]; s_[4 * k + a] = VN[3 * vi + a]; }
      q[4 * k + 3] = 0.f;
      if (!uvr.empty()) { uvr[8 * i + 2 * k] = c->uv[2 * vi]; uvr[8 * i + 2 * k + 1] = c->uv[2 * vi + 1]; }
    }
    {
      // the kernel's former expression on the three vertices, evaluated once here with the same inline arithmetic (same bits)
      const crh_v3 a0 = crh_mk3(q[0], q[1], q[2]), a1 = crh_mk3(q[4], q[5], q[6]), a2 = crh_mk3(q[8], q[9], q[10]);
      const crh_v3 ng = crh_norm3(crh_cross3(crh_sub3(a0, a2), crh_sub3(a1, a0)));
      s_[12] = ng.x; s_[13] = ng.y; s_[14] = ng.z;
    }
    std::memcpy(&q[3], &t, 4);
    const int32_t mat = c->tri[4 * t + 3];
    std::memcpy(&s_[3], &mat, 4);
    const int32_t ob = (c->two_level && p >= c->n_static) ? (int32_t)c->pos_obj[p - c->n_static] : -1;      // n1.w: the object whose transform shading applies (-1: world space)
    std::memcpy(&s_[7], &ob, 4);
    device_tri_record(q, &tr[4 * (size_t)kTriStride * i]);
    if (verts) std::memcpy(&(*verts)[12 * i], q, 48);
  }
  };
  int threads = 0; if (const char* e = getenv("CRH_BUILD_THREADS")) threads = atoi(e);
  const uint32_t nth = n >= 65536u ? (uint32_t)std::min<size_t>((size_t)build_threads(threads), n / 32768u) : 1u;
  if (nth <= 1u) fill(p0, p1);
  else {
    std::vector<std::thread> pool;
    for (uint32_t k = 0; k < nth; ++k) pool.emplace_back(fill, p0 + (uint32_t)((uint64_t)n * k / nth), p0 + (uint32_t)((uint64_t)n * (k + 1) / nth));
    for (auto& th : pool) th.join();
  }
}

int upload_textures(crh_ctx* c)
{
  if (!c->textures_dirty) return CRH_OK;
  std::vector<float> all; std::vector<uint32_t> desc(4 * std::max<size_t>(c->textures.size(), 1), 0u);
  for (size_t i = 0; i < c->textures.size(); ++i) {
    desc[4 * i] = (uint32_t)(all.size() / 4); desc[4 * i + 1] = c->textures[i].w; desc[4 * i + 2] = c->textures[i].h;
    all.insert(all.end(), c->textures[i].rgba.begin(), c->textures[i].rgba.end());
  }
  if (all.empty()) all.assign(4, 0.f);
  int rc;
  if ((rc = dev_upload(c, c->d_texels, all.data(), all.size() * sizeof(float)))) return rc;
  if ((rc = dev_upload(c, c->d_tex_desc, desc.data(), desc.size() * sizeof(uint32_t)))) return rc;
  c->textures_dirty = false;
  return CRH_OK;
}

int alloc_accum(crh_ctx* c)
{
  if (c->d_accum && c->accumW == c->par.width && c->accumH == c->par.height) return CRH_OK;
  CRH_HIP(hipStreamSynchronize(cstream(c)));
  if (c->d_accum) { CRH_HIP(hipFree(c->d_accum)); c->d_accum = nullptr; }
  CRH_HIP(hipMalloc((void**)&c->d_accum, sizeof(float4) * (size_t)c->par.width * c->par.height));
  if (c->d_m2) { CRH_HIP(hipFree(c->d_m2)); c->d_m2 = nullptr; }
  CRH_HIP(hipMalloc((void**)&c->d_m2, sizeof(float) * (size_t)c->par.width * c->par.height));
  c->accumW = c->par.width; c->accumH = c->par.height;
  return CRH_OK;
}

int do_reset(crh_ctx* c)
{
  // stream-ordered: kernels still in flight finish into the old accumulator contents first, nothing is waited for
  CRH_HIP(hipSetDevice(c->device));
  int rc = alloc_accum(c); if (rc) return rc;
  CRH_HIP(hipMemsetAsync(c->d_accum, 0, sizeof(float4) * (size_t)c->par.width * c->par.height, cstream(c)));
  CRH_HIP(hipMemsetAsync(c->d_m2, 0, sizeof(float) * (size_t)c->par.width * c->par.height, cstream(c)));
  c->adaptive_picks = 0; c->pending_n = 0; c->ramp_k = 1; c->picked_valid = false; c->assembled_valid = false;
  CRH_HIP(hipMemsetAsync(c->d_counters, 0, sizeof(DCounters), cstream(c)));
  discard_events(c);
  c->seconds_acc = c->trace_ms_acc = c->all_ms_acc = 0.0; c->trace_launches = 0; c->frames_done = 0;
  return CRH_OK;
}

// One batch: `ns` samples of `nt` tiles whose ids sit at d_tiles; seeds at d_seeds.
// One wavefront schedule: `ns` samples of `nt` tiles (ids at d_tiles, seeds at d_seeds) on one stream and one slice of the path state.
struct Lane { hipStream_t stream; DPaths P; DQueues Q; int grid, grid_trace; bool timed; const uint32_t* n_tiles_dev = nullptr; bool donate = false; };

int run_lane(crh_ctx* c, const Lane& ln, const DScene& S, const uint32_t* d_tiles, uint32_t nt, const uint32_t* d_seeds, uint32_t ns, int seed_per_tile,
             bool accumulate, hipEvent_t before_accumulate = nullptr, hipEvent_t before_accumulate2 = nullptr)
{
  Launch L{ln.stream, ln.grid, c->counters_on};
  Launch LT{ln.stream, ln.grid_trace, c->counters_on, c->clamp_grid ? c->cus : 0, ln.donate && !c->counters_on};

  launch_raygen(L, S, ln.P, ln.Q, 0, d_tiles, nt, d_seeds, ns, seed_per_tile, ln.n_tiles_dev);
  int qin = 0;
  for (uint32_t b = 0; b < S.max_depth; ++b) {
    const Launch& T = LT;
    if (ln.timed && c->timing_on) {
      hipEvent_t e0 = get_event(c), e1 = get_event(c);
      hipEventRecord(e0, ln.stream);
      launch_trace_nearest(T, S, ln.P, ln.Q, qin, b, c->d_counters);
      hipEventRecord(e1, ln.stream);
      c->trace_ev.emplace_back(e0, e1);
    } else launch_trace_nearest(T, S, ln.P, ln.Q, qin, b, c->d_counters);
    launch_shade(L, S, ln.P, ln.Q, qin, b, c->d_counters);
    if (S.n_lights > 0) launch_trace_any(T, S, ln.P, ln.Q, c->d_counters);
    qin = 1 - qin;
  }
  if (accumulate && before_accumulate) CRH_HIP(hipStreamWaitEvent(ln.stream, before_accumulate, 0));      // samples are folded in in frame order
  if (accumulate && before_accumulate2) CRH_HIP(hipStreamWaitEvent(ln.stream, before_accumulate2, 0));    // ... and not while a read-back tone-maps the accumulator
  if (accumulate) launch_accumulate(L, S, ln.P, c->d_accum, c->adaptive ? c->d_m2 : nullptr, d_tiles, nt, 0, ns, ns, c->d_counters, ln.n_tiles_dev);
  CRH_HIP(hipGetLastError());
  return CRH_OK;
}

int ensure_lanes(crh_ctx* c)
{
  if (c->d_lane_counts) return CRH_OK;
  for (uint32_t k = 0; k < 8u; ++k) {      // tile ranges of one batch (n_lanes <= 8), or frames in flight (pipe_depth <= 8)
    CRH_HIP(hipStreamCreateWithFlags(&c->lane_stream[k], hipStreamNonBlocking));
    CRH_HIP(hipEventCreateWithFlags(&c->lane_join[k], hipEventDisableTiming));
  }
  CRH_HIP(hipEventCreateWithFlags(&c->lane_fork, hipEventDisableTiming));
  CRH_HIP(hipMalloc((void**)&c->d_pipe_seeds, 8 * 16 * sizeof(uint32_t)));
  CRH_HIP(hipMalloc((void**)&c->d_lane_counts, kCounts * sizeof(uint32_t) * 8));
  CRH_HIP(hipMemsetAsync(c->d_lane_counts, 0, kCounts * sizeof(uint32_t) * 8, cstream(c)));
  return CRH_OK;
}

// One batch: `ns` samples of `nt` tiles whose ids sit at d_tiles; seeds at d_seeds.
int run_batch(crh_ctx* c, const DScene& S, const uint32_t* d_tiles, uint32_t nt, const uint32_t* d_seeds, uint32_t ns, int seed_per_tile = 0,
              bool accumulate = true)
{
  c->pending_n = 0;                                  // the path buffer is about to be overwritten
  const uint32_t tpp = S.tile_size * S.tile_size;
  const uint64_t total = (uint64_t)nt * tpp * ns;
  // two ranges pay from about a frame's worth of paths (2 M: 173 -> 182 Redraw/s); below that one schedule with a small grid is
  // faster (128 tiles per call: 303 vs 284 calls/s)
  const uint32_t K = total >= (1u << 20) ? std::min<uint32_t>(c->n_lanes, nt / 4u) : 1u;
  // A persistent grid far larger than the batch only queues wavefronts for work fetches that return nothing (one cursor word
  // sustains ~88 atomics/us: 6144 wavefronts = 70 us per launch): small batches get grids that follow their size.  Measured on
  // C3 at 1080p, 1 spp per call: traversal grids of 1536 / 1024 / 768 / 512 workgroups -> 163 / 174 / 173 / 165 Redraw/s; 128
  // tiles per call: 257 / 275 / 284 / 293 calls/s.
  auto small_grid = [&](uint64_t paths, int full, int per) { return (int)std::min<uint64_t>((uint64_t)full, std::max<uint64_t>(512u, paths / (uint64_t)per)); };
  if (K < 2 || total > c->lane_max_paths || !accumulate || c->counters_on) {
    const bool small = total <= c->lane_max_paths && !c->counters_on;
    Lane one{cstream(c), c->paths, c->queues, small ? small_grid(total, c->grid, 2048) : c->grid, small ? small_grid(total, c->grid_trace, 2048) : c->grid_trace, true};
    one.donate = small && c->donate;
    return run_lane(c, one, S, d_tiles, nt, d_seeds, ns, seed_per_tile, accumulate);
  }
  // small batch: K tile ranges on K streams, each with its own slice [base, base + n_k * tpp * ns) of every path-state array
  // (queue entries are positions relative to the slice) and its own counter block; fork from / join into the context's stream
  int rc = ensure_lanes(c); if (rc) return rc;
  CRH_HIP(hipEventRecord(c->lane_fork, cstream(c)));
  size_t base = 0;
  for (uint32_t k = 0; k < K; ++k) {
    const uint32_t t0 = (uint32_t)((uint64_t)nt * k / K), t1 = (uint32_t)((uint64_t)nt * (k + 1) / K);
    Lane ln; ln.stream = c->lane_stream[k]; ln.timed = false; ln.donate = c->donate;
    ln.grid = c->lane_grid > 0 ? c->lane_grid : small_grid(total / K, c->grid, 2048);
    ln.grid_trace = c->lane_grid_trace > 0 ? c->lane_grid_trace : small_grid(total / K, c->grid_trace, 2048);
    const DPaths& P = c->paths; const DQueues& Q = c->queues;
    ln.P.ray_o[0] = P.ray_o[0] + base; ln.P.ray_o[1] = P.ray_o[1] + base; ln.P.ray_d[0] = P.ray_d[0] + base; ln.P.ray_d[1] = P.ray_d[1] + base;
    ln.P.thr[0] = P.thr[0] + base; ln.P.thr[1] = P.thr[1] + base; ln.P.hit = P.hit + base; ln.P.rad = P.rad + base;
    ln.P.sh_o = P.sh_o + base; ln.P.sh_d = P.sh_d + base; ln.P.sh_c = P.sh_c + base;
    ln.Q.q[0] = Q.q[0] + base; ln.Q.q[1] = Q.q[1] + base; ln.Q.q_sh = Q.q_sh + base; ln.Q.q2 = Q.q2 + base; ln.Q.q2_sh = Q.q2_sh + base; ln.Q.counts = c->d_lane_counts + kCounts * k;
    CRH_HIP(hipStreamWaitEvent(ln.stream, c->lane_fork, 0));
    rc = run_lane(c, ln, S, d_tiles + t0, t1 - t0, seed_per_tile ? d_seeds + t0 : d_seeds, ns, seed_per_tile, true); if (rc) return rc;
    CRH_HIP(hipEventRecord(c->lane_join[k], ln.stream));
    CRH_HIP(hipStreamWaitEvent(cstream(c), c->lane_join[k], 0));
    base += (size_t)(t1 - t0) * tpp * ns;
  }
  return CRH_OK;
}

int render_impl(crh_ctx* c, const uint32_t* tiles, uint32_t nt, uint32_t first, uint32_t ns)
{
  if (!c->built) return fail(c, CRH_E_NOTBUILT, "crh_build has not been called");
  if (ns == 0 || nt == 0) return CRH_OK;
  CRH_HIP(hipSetDevice(c->device));
  { int rc_t = upload_textures(c); if (rc_t) return rc_t; }
  const uint32_t ts = c->par.tile_size, tpp = ts * ts;
  const uint32_t tx = (c->par.width + ts - 1) / ts, ty = (c->par.height + ts - 1) / ts;
  {
    std::vector<uint8_t> seen((size_t)tx * ty, 0);          // a tile listed twice would be accumulated by two threads at once
    for (uint32_t i = 0; i < nt; ++i) {
      if (tiles[i] >= tx * ty) return fail(c, CRH_E_INVALID, "tile id out of range");
      if (seen[tiles[i]]) return fail(c, CRH_E_INVALID, "duplicate tile id");
      seen[tiles[i]] = 1;
    }
  }
  // tile ids + frame seeds to the device, stream-ordered behind any kernels still reading the old ones, through pinned staging:
  // a Redraw() does not wait for the previous one (an unchanged tile list is not sent again)
  if (nt > c->tile_cap) { CRH_HIP(hipStreamSynchronize(cstream(c))); if (c->d_tile_ids) CRH_HIP(hipFree(c->d_tile_ids)); CRH_HIP(hipMalloc((void**)&c->d_tile_ids, sizeof(uint32_t) * nt)); c->tile_cap = nt; c->h_tile_ids.clear(); }
  if (c->h_tile_ids.size() != nt || std::memcmp(c->h_tile_ids.data(), tiles, sizeof(uint32_t) * nt) != 0) {
    int rc_u = stage_copy(c, c->d_tile_ids, tiles, sizeof(uint32_t) * nt); if (rc_u) return rc_u;
    c->h_tile_ids.assign(tiles, tiles + nt);
  }
  const uint32_t cap_tiles = std::max<uint32_t>(1u, c->max_paths / tpp);
  const uint32_t group = std::min(nt, cap_tiles);
  const uint32_t spb = std::max<uint32_t>(1u, std::min(ns, c->max_paths / (group * tpp)));
  const uint64_t total = (uint64_t)nt * tpp * ns;
  // Measured on C3 at 1080p, 1 spp per call: free-running 232 -> 326 Redraw/s (C2: 323 -> 442); a host that reads every frame back
  // would get 187 instead of 225 (one schedule per frame is slower than two tile ranges when nothing overlaps it), so a
  // read-back / synchronisation since the last render selects the two-range schedule for this frame.
  const bool host_runs_ahead = !c->read_since_render;
  if (c->pipeline && host_runs_ahead && !c->counters_on && !c->timing_on && !c->adaptive && group == nt && spb == ns && ns <= 16u && total >= (1u << 20) &&
      total <= c->lane_max_paths && (uint64_t)c->pipe_depth * total <= c->max_paths) {
    // ---- frame pipelining: this batch (one Redraw() worth) goes to pipeline stream k with its own half of the path state; it
    // starts as soon as the previous frame ON THAT STREAM is done and overlaps the frame on the other stream; its samples are
    // folded in after that frame's.  Nothing is joined into the context's stream here -- cstream() does that on demand.
    int rc = ensure_paths(c, (uint32_t)(c->pipe_depth * total)); if (rc) return rc;
    rc = ensure_lanes(c); if (rc) return rc;
    const uint32_t k = c->pipe_seq % c->pipe_depth, prev = (c->pipe_seq + c->pipe_depth - 1u) % c->pipe_depth;      // this frame's stream, the previous frame's
    ++c->pipe_seq;
    const hipStream_t cs = c->stream_;                 // raw: no join
    if (c->pipe_pending[k]) CRH_HIP(hipStreamWaitEvent(cs, c->lane_join[k], 0));      // this stream's seed slot is free once its last frame is done
    uint32_t* d_seeds_k = c->d_pipe_seeds + 16u * k;
    {
      std::vector<uint32_t> seeds(ns);
      uint32_t hi = c->par.seed, lo = c->par.seed ^ 0x49616E42u;
      for (uint32_t i = 0; i < first + ns; ++i) { hi = (hi >> 2) + (hi << 2); hi += lo; lo += hi; if (i >= first) seeds[i - first] = hi >> 2; }
      rc = stage_copy(c, d_seeds_k, seeds.data(), sizeof(uint32_t) * ns, cs); if (rc) return rc;
    }
    DScene S; fill_scene(c, S);
    CRH_HIP(hipEventRecord(c->lane_fork, cs));
    Lane ln; ln.stream = c->lane_stream[k]; ln.timed = false; ln.donate = c->donate;
    ln.grid = (int)std::min<uint64_t>((uint64_t)c->grid, std::max<uint64_t>((uint64_t)c->pipe_grid_min_shade, total / 2048u));
    // traversal grid of this frame: the chip's resident workgroups (6 per CU) shared among the frames that are in flight RIGHT NOW -- a host that runs far
    // ahead has pipe_depth of them (eight: 192 workgroups each), one that waits for every other frame's read-back has two or three (512 each); measured
    // optima at 3 / 4 / 6 / 8 frames in flight: 512 / 384 / 256 / 192-256 (profiles/r3/interactive_counters.txt)
    uint32_t in_flight = 1u;
    for (uint32_t j = 0; j < 8u; ++j) if (j != k && c->pipe_pending[j] && hipEventQuery(c->lane_join[j]) == hipErrorNotReady) ++in_flight;
    {
      // a host that submitted the previous frame a moment ago is not waiting for anything: the pipeline is about to fill (counting what is in flight NOW
      // would give the first frames of a burst grids for a nearly empty chip: eight of them, 2900 workgroups)
      const auto now = std::chrono::steady_clock::now();
      if (c->pipe_last_submit.time_since_epoch().count() != 0 && now - c->pipe_last_submit < std::chrono::microseconds(300)) in_flight = std::max(in_flight, c->pipe_depth);
      c->pipe_last_submit = now;
    }
    const uint64_t share = std::min<uint64_t>(512u, std::max<uint64_t>((uint64_t)c->pipe_grid_min, (uint64_t)c->grid_trace / in_flight));
    ln.grid_trace = (int)std::min<uint64_t>((uint64_t)c->grid_trace, std::max<uint64_t>(share, total / (uint64_t)c->pipe_div));
    const size_t base = (size_t)k * total;
    const DPaths& P = c->paths; const DQueues& Q = c->queues;
    ln.P.ray_o[0] = P.ray_o[0] + base; ln.P.ray_o[1] = P.ray_o[1] + base; ln.P.ray_d[0] = P.ray_d[0] + base; ln.P.ray_d[1] = P.ray_d[1] + base;
    ln.P.thr[0] = P.thr[0] + base; ln.P.thr[1] = P.thr[1] + base; ln.P.hit = P.hit + base; ln.P.rad = P.rad + base;
    ln.P.sh_o = P.sh_o + base; ln.P.sh_d = P.sh_d + base; ln.P.sh_c = P.sh_c + base;
    ln.Q.q[0] = Q.q[0] + base; ln.Q.q[1] = Q.q[1] + base; ln.Q.q_sh = Q.q_sh + base; ln.Q.q2 = Q.q2 + base; ln.Q.q2_sh = Q.q2_sh + base; ln.Q.counts = c->d_lane_counts + kCounts * k;
    CRH_HIP(hipStreamWaitEvent(ln.stream, c->lane_fork, 0));
    // the frames in flight own path-state slices [k * total, (k + 1) * total): a batch of ANOTHER size (crh_render_tiles with another
    // sample count or tile list) would lay its slice across theirs -- it starts only when they are all done
    if (total != c->pipe_total) {
      for (int j = 0; j < 8; ++j) if (c->pipe_pending[j]) CRH_HIP(hipStreamWaitEvent(ln.stream, c->lane_join[j], 0));
      c->pipe_total = total;
    }
    c->pending_n = 0;
    hipEvent_t e0 = get_event(c), e1 = get_event(c);          // crh_stats.seconds: device time of this frame (frames in flight overlap)
    hipEventRecord(e0, ln.stream);
    const hipEvent_t guard = c->rb_guard_pending ? c->rb_guard : nullptr; c->rb_guard_pending = false;      // later frames are ordered behind this one's accumulate
    rc = run_lane(c, ln, S, c->d_tile_ids, nt, d_seeds_k, ns, 0, true, c->pipe_pending[prev] ? c->lane_join[prev] : nullptr, guard); if (rc) return rc;
    hipEventRecord(e1, ln.stream);
    c->render_ev.emplace_back(e0, e1);
    CRH_HIP(hipEventRecord(c->lane_join[k], ln.stream));
    c->pipe_pending[k] = true;
    rc = trim_events(c);                                        // every 4096 frames: waits for the device once
    c->read_since_render = false;
    return rc;
  }
  if (ns > c->seed_cap) { CRH_HIP(hipStreamSynchronize(cstream(c))); if (c->d_seeds) CRH_HIP(hipFree(c->d_seeds)); CRH_HIP(hipMalloc((void**)&c->d_seeds, sizeof(uint32_t) * ns)); c->seed_cap = ns; }
  {
    // frame seeds: Bullard generator restarted at par.seed, frame n uses next() >> 2
    std::vector<uint32_t> seeds(ns);
    uint32_t hi = c->par.seed, lo = c->par.seed ^ 0x49616E42u;
    for (uint32_t i = 0; i < first + ns; ++i) { hi = (hi >> 2) + (hi << 2); hi += lo; lo += hi; if (i >= first) seeds[i - first] = hi >> 2; }
    int rc_u = stage_copy(c, c->d_seeds, seeds.data(), sizeof(uint32_t) * ns); if (rc_u) return rc_u;
  }

  int rc = ensure_paths(c, group * tpp * spb); if (rc) return rc;
  DScene S; fill_scene(c, S);
  hipEvent_t e0 = get_event(c), e1 = get_event(c);
  hipEventRecord(e0, cstream(c));
  for (uint32_t t0 = 0; t0 < nt; t0 += group) {
    const uint32_t g = std::min(group, nt - t0);
    for (uint32_t s0 = 0; s0 < ns; s0 += spb) {
      rc = run_batch(c, S, c->d_tile_ids + t0, g, c->d_seeds + s0, std::min(spb, ns - s0));
      if (rc) return rc;
    }
  }
  hipEventRecord(e1, cstream(c));
  c->render_ev.emplace_back(e0, e1);
  rc = trim_events(c);
  c->read_since_render = false;      // from here on, only a call other than the next render sets it again
  return rc;
}

// ---- adaptive screen sampling (reference: AdaptiveScreenSampling / NbRayTracingTiles, SettingsWidget.cxx:427-477) ----------
int tile_stats(crh_ctx* c, std::vector<float>& err, std::vector<uint32_t>& cnt)
{
  const uint32_t ts = c->par.tile_size;
  const uint32_t nt = ((c->par.width + ts - 1) / ts) * ((c->par.height + ts - 1) / ts);
  if (nt > c->tile_stat_cap) {
    CRH_HIP(hipStreamSynchronize(cstream(c)));
    if (c->d_tile_err) CRH_HIP(hipFree(c->d_tile_err));
    if (c->d_tile_cnt) CRH_HIP(hipFree(c->d_tile_cnt));
    for (void* q : {(void*)c->d_tile_cdf, (void*)c->d_picked, (void*)c->d_adapt_n}) if (q) CRH_HIP(hipFree(q));
    c->d_tile_cdf = nullptr; c->d_picked = nullptr; c->d_adapt_n = nullptr; c->picked_valid = false;      // the sampler allocates them again at this size
    CRH_HIP(hipMalloc((void**)&c->d_tile_err, sizeof(float) * nt)); CRH_HIP(hipMalloc((void**)&c->d_tile_cnt, sizeof(uint32_t) * nt));
    c->tile_stat_cap = nt;
  }
  DScene S; fill_scene(c, S);
  Launch L{cstream(c), c->grid, false};
  launch_tile_error(L, S, c->d_accum, c->d_m2, c->d_tile_err, c->d_tile_cnt, nt);
  err.resize(nt); cnt.resize(nt);
  CRH_HIP(hipMemcpyAsync(err.data(), c->d_tile_err, sizeof(float) * nt, hipMemcpyDeviceToHost, cstream(c)));
  CRH_HIP(hipMemcpyAsync(cnt.data(), c->d_tile_cnt, sizeof(uint32_t) * nt, hipMemcpyDeviceToHost, cstream(c)));
  CRH_HIP(hipStreamSynchronize(cstream(c)));
  return CRH_OK;
}

// One adaptive iteration: draw `adaptive_tiles` tiles with probability proportional to their error estimate (inverse CDF
// driven by the base-2 radical inverse of a running pick counter), render +1 sample on the distinct tiles drawn.  Everything --
// error estimate, running sum, draws, tile list, per-tile seeds -- happens in HBM, stream-ordered: the host only advances the
// pick counter, so a GUI loop of crh_render(1) calls never waits for the device (the reference offers this mode as its
// responsiveness feature, SettingsWidget.cxx:427-477).
int adaptive_iteration(crh_ctx* c)
{
  int rc = upload_textures(c); if (rc) return rc;
  const uint32_t ts = c->par.tile_size, tpp = ts * ts;
  const uint32_t nt = ((c->par.width + ts - 1) / ts) * ((c->par.height + ts - 1) / ts);
  const uint32_t most = std::min(c->adaptive_tiles, nt);                // distinct tiles one iteration can draw
  if (nt > c->tile_stat_cap || !c->d_tile_cdf) {
    CRH_HIP(hipStreamSynchronize(cstream(c)));
    for (void* q : {(void*)c->d_tile_err, (void*)c->d_tile_cnt, (void*)c->d_tile_cdf, (void*)c->d_picked, (void*)c->d_adapt_n}) if (q) CRH_HIP(hipFree(q));
    c->d_tile_err = nullptr; c->d_tile_cnt = nullptr; c->d_tile_cdf = nullptr; c->d_picked = nullptr; c->d_adapt_n = nullptr; c->tile_stat_cap = 0;
    CRH_HIP(hipMalloc((void**)&c->d_tile_err, sizeof(float) * nt)); CRH_HIP(hipMalloc((void**)&c->d_tile_cnt, sizeof(uint32_t) * nt));
    CRH_HIP(hipMalloc((void**)&c->d_tile_cdf, sizeof(float) * nt)); CRH_HIP(hipMalloc((void**)&c->d_picked, nt)); CRH_HIP(hipMalloc((void**)&c->d_adapt_n, 64));
    c->tile_stat_cap = std::max(nt, c->tile_stat_cap);
  }
  if (most > c->tile_cap) { CRH_HIP(hipStreamSynchronize(cstream(c))); if (c->d_tile_ids) CRH_HIP(hipFree(c->d_tile_ids)); CRH_HIP(hipMalloc((void**)&c->d_tile_ids, sizeof(uint32_t) * most)); c->tile_cap = most; }
  if (most > c->seed_cap) { CRH_HIP(hipStreamSynchronize(cstream(c))); if (c->d_seeds) CRH_HIP(hipFree(c->d_seeds)); CRH_HIP(hipMalloc((void**)&c->d_seeds, sizeof(uint32_t) * most)); c->seed_cap = most; }
  c->h_tile_ids.clear();                                                // the device is about to write its own list there
  rc = ensure_paths(c, most * tpp); if (rc) return rc;
  DScene S; fill_scene(c, S);
  hipEvent_t e0 = get_event(c), e1 = get_event(c);
  hipEventRecord(e0, cstream(c));
  Launch L{cstream(c), c->grid, false};
  launch_tile_error(L, S, c->d_accum, c->d_m2, c->d_tile_err, c->d_tile_cnt, nt);
  launch_adaptive_pick(L, c->d_tile_err, c->d_tile_cnt, nt, c->adaptive_picks, c->adaptive_tiles, c->par.seed, c->d_tile_cdf, c->d_picked,
                       c->d_tile_ids, c->d_seeds, c->d_adapt_n);
  c->adaptive_picks += c->adaptive_tiles; c->picked_valid = true;
  c->pending_n = 0;
  Lane ln{cstream(c), c->paths, c->queues, (int)std::min<uint64_t>((uint64_t)c->grid, std::max<uint64_t>(512u, (uint64_t)most * tpp / 1024u)),
          (int)std::min<uint64_t>((uint64_t)c->grid_trace, std::max<uint64_t>(512u, (uint64_t)most * tpp / 2048u)), true};      // grids follow the batch (run_batch)
  ln.n_tiles_dev = c->d_adapt_n; ln.donate = c->donate;
  rc = run_lane(c, ln, S, c->d_tile_ids, most, c->d_seeds, 1, 1, true); if (rc) return rc;
  hipEventRecord(e1, cstream(c));
  c->render_ev.emplace_back(e0, e1);
  return trim_events(c);
}


// ---- RCCL, loaded on first use (a single-GPU host never pays for it; a Python host that already loaded torch's librccl
// gets that copy back from dlopen by SONAME)
struct RcclApi {
  void* lib = nullptr;
  ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*Reduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, int, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
RcclApi* rccl()
{
  static RcclApi api;
  static bool tried = false;
  if (!tried) {
    tried = true;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* nm : names) { api.lib = dlopen(nm, RTLD_NOW | RTLD_GLOBAL); if (api.lib) break; }
    if (api.lib) {
      api.CommInitAll = (decltype(api.CommInitAll))dlsym(api.lib, "ncclCommInitAll");
      api.CommDestroy = (decltype(api.CommDestroy))dlsym(api.lib, "ncclCommDestroy");
      api.GroupStart = (decltype(api.GroupStart))dlsym(api.lib, "ncclGroupStart");
      api.GroupEnd = (decltype(api.GroupEnd))dlsym(api.lib, "ncclGroupEnd");
      api.Reduce = (decltype(api.Reduce))dlsym(api.lib, "ncclReduce");
      api.GetErrorString = (decltype(api.GetErrorString))dlsym(api.lib, "ncclGetErrorString");
      if (!api.CommInitAll || !api.CommDestroy || !api.GroupStart || !api.GroupEnd || !api.Reduce) api.lib = nullptr;
    }
  }
  return api.lib ? &api : nullptr;
}

void release_comms(crh_ctx* c)
{
  if (c->comms.empty()) return;
  if (RcclApi* R = rccl()) for (ncclComm_t m : c->comms) if (m) R->CommDestroy(m);
  c->comms.clear(); c->comm_ctxs.clear();
}

#define CRH_NCCL(call)                                                                                   \
  do { ncclResult_t r_ = (call); if (r_ != ncclSuccess) {                                                \
      char b_[512]; snprintf(b_, sizeof b_, "%s failed: %s", #call, R->GetErrorString ? R->GetErrorString(r_) : "rccl error"); \
      c->err = b_; return CRH_E_DEVICE; } } while (0)

int reduce_impl(crh_ctx* const* ctxs, uint32_t n, uint32_t root)
{
  crh_ctx* c = ctxs[root];                                   // errors are reported on the root
  const uint32_t W = c->par.width, H = c->par.height;
  const size_t n4 = (size_t)W * H;
  std::vector<int> devs(n);
  bool distinct = true;
  for (uint32_t i = 0; i < n; ++i) {
    crh_ctx* x = ctxs[i];
    if (!x || !x->d_accum || x->par.width != W || x->par.height != H) return fail(c, CRH_E_INVALID, "crh_reduce: every context needs an accumulator of the root's size");
    for (uint32_t j = 0; j < i; ++j) { if (ctxs[j] == x) return fail(c, CRH_E_INVALID, "crh_reduce: context listed twice"); if (ctxs[j]->device == x->device) distinct = false; }
    devs[i] = x->device;
  }
  CRH_HIP(hipSetDevice(c->device));
  if (!c->d_assembled || c->assembledW != W || c->assembledH != H) {
    if (c->d_assembled) { CRH_HIP(hipFree(c->d_assembled)); c->d_assembled = nullptr; }
    if (c->d_peer_stage) { CRH_HIP(hipFree(c->d_peer_stage)); c->d_peer_stage = nullptr; }
    CRH_HIP(hipMalloc((void**)&c->d_assembled, sizeof(float4) * n4));
    c->assembledW = W; c->assembledH = H;
  }
  // every shard's queued rendering must have landed before its accumulator is read by another stream / device
  for (uint32_t i = 0; i < n; ++i) { CRH_HIP(hipSetDevice(ctxs[i]->device)); CRH_HIP(hipStreamSynchronize(cstream(ctxs[i]))); }

  // CRH_REDUCE_RCCL_SINGLE=1 sends even a one-context group through RCCL (exercises the library binding on a 1-GPU box)
  RcclApi* R = (distinct && (n > 1 || getenv("CRH_REDUCE_RCCL_SINGLE"))) ? rccl() : nullptr;
  if (R) {
    // one process, one communicator per context, a single grouped ncclReduce: on xGMI the peers' contributions arrive
    // over distinct links; message = W*H*16 B (33 MB at 1080p, 133 MB at 4K)
    if (c->comm_ctxs.size() != n || !std::equal(c->comm_ctxs.begin(), c->comm_ctxs.end(), ctxs)) {
      release_comms(c);
      c->comms.assign(n, nullptr);
      CRH_NCCL(R->CommInitAll(c->comms.data(), (int)n, devs.data()));
      c->comm_ctxs.assign(ctxs, ctxs + n);
    }
    CRH_NCCL(R->GroupStart());
    for (uint32_t i = 0; i < n; ++i) {
      hipSetDevice(ctxs[i]->device);
      ncclResult_t r = R->Reduce(ctxs[i]->d_accum, i == root ? (void*)c->d_assembled : (void*)ctxs[i]->d_accum, 4 * n4, ncclFloat, ncclSum, (int)root,
                                 c->comms[i], cstream(ctxs[i]));
      if (r != ncclSuccess) { R->GroupEnd(); c->err = "ncclReduce failed"; return CRH_E_DEVICE; }
    }
    CRH_NCCL(R->GroupEnd());
    for (uint32_t i = 0; i < n; ++i) { CRH_HIP(hipSetDevice(ctxs[i]->device)); CRH_HIP(hipStreamSynchronize(cstream(ctxs[i]))); }
    CRH_HIP(hipSetDevice(c->device));
  } else {
    // contexts that share a device (rehearsal of the sharded flow on one GPU) or no RCCL in the process: the root pulls
    // every shard (peer copy when it lives on another device) and adds it in context order -- same sums, since every pixel
    // is non-zero in exactly one shard
    CRH_HIP(hipMemcpyAsync(c->d_assembled, c->d_accum, sizeof(float4) * n4, hipMemcpyDeviceToDevice, cstream(c)));
    Launch L{cstream(c), c->grid, false};
    for (uint32_t i = 0; i < n; ++i) {
      if (i == root) continue;
      const float4* src = ctxs[i]->d_accum;
      if (ctxs[i]->device != c->device) {
        if (!c->d_peer_stage) CRH_HIP(hipMalloc((void**)&c->d_peer_stage, sizeof(float4) * n4));
        CRH_HIP(hipMemcpyPeerAsync(c->d_peer_stage, c->device, ctxs[i]->d_accum, ctxs[i]->device, sizeof(float4) * n4, cstream(c)));
        src = c->d_peer_stage;
      }
      launch_add4(L, c->d_assembled, src, (uint32_t)n4);
    }
    CRH_HIP(hipGetLastError());
    CRH_HIP(hipStreamSynchronize(cstream(c)));
  }
  c->assembled_valid = true;
  return CRH_OK;
}

}  // namespace

// =============================================================================================== C ABI
extern "C" {

crh_ctx* crh_create(int device_ordinal)
{
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || device_ordinal < 0 || device_ordinal >= n) {
    fprintf(stderr, "crh_create: no HIP device %d (device count %d) -- this backend has no CPU fallback\n", device_ordinal, n);
    return nullptr;
  }
  crh_ctx* c = new crh_ctx();
  c->device = device_ordinal;
  if (hipSetDevice(device_ordinal) != hipSuccess || hipStreamCreateWithFlags(&c->stream_, hipStreamNonBlocking) != hipSuccess ||
      hipMalloc((void**)&c->d_counters, sizeof(DCounters)) != hipSuccess || hipMalloc((void**)&c->d_api_cursor, 64) != hipSuccess || hipMemsetAsync(c->d_counters, 0, sizeof(DCounters), cstream(c)) != hipSuccess ||
      hipStreamSynchronize(cstream(c)) != hipSuccess) {
    fprintf(stderr, "crh_create: HIP initialisation failed: %s\n", hipGetErrorString(hipGetLastError()));
    delete c; return nullptr;
  }
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device_ordinal) == hipSuccess && prop.multiProcessorCount > 0) { c->cus = prop.multiProcessorCount; c->grid = prop.multiProcessorCount * 4; c->grid_trace = prop.multiProcessorCount * 6; }
  if (const char* e = getenv("CRH_MAX_PATHS")) { long v = atol(e); if (v >= 1024) c->max_paths = (uint32_t)std::min<long>(v, 1l << 30); }   // a path slot travels in 31 bits
  if (const char* e = getenv("CRH_GRID")) { int v = atoi(e); if (v > 0) c->grid = v; }
  if (const char* e = getenv("CRH_GRID_TRACE")) { int v = atoi(e); if (v > 0) c->grid_trace = v; }
  if (const char* e = getenv("CRH_CLAMP_GRID")) c->clamp_grid = atoi(e) != 0;
  if (const char* e = getenv("CRH_DONATE")) c->donate = atoi(e) != 0;
  if (const char* e = getenv("CRH_SPLIT_PASSES")) c->split_passes = atoi(e);
  if (const char* e = getenv("CRH_PIPELINE")) c->pipeline = atoi(e) != 0;
  if (const char* e = getenv("CRH_PIPE_DEPTH")) { int v = atoi(e); if (v >= 2 && v <= 8) c->pipe_depth = (uint32_t)v; }
  if (const char* e = getenv("CRH_PIPE_DIV")) { int v = atoi(e); if (v > 0) c->pipe_div = v; }
  if (const char* e = getenv("CRH_PIPE_GRID_MIN")) { int v = atoi(e); if (v > 0) c->pipe_grid_min = v; }
  if (const char* e = getenv("CRH_PIPE_GRID_MIN_SHADE")) { int v = atoi(e); if (v > 0) c->pipe_grid_min_shade = v; }
  if (const char* e = getenv("CRH_LANES")) { int v = atoi(e); if (v >= 1 && v <= 8) c->n_lanes = (uint32_t)v; }
  if (const char* e = getenv("CRH_LANE_MAX_PATHS")) { long v = atol(e); if (v >= 0) c->lane_max_paths = (uint32_t)std::min<long>(v, 1l << 30); }
  if (const char* e = getenv("CRH_LANE_GRID")) { int v = atoi(e); if (v > 0) c->lane_grid = v; }
  if (const char* e = getenv("CRH_LANE_GRID_TRACE")) { int v = atoi(e); if (v > 0) c->lane_grid_trace = v; }
  // reference defaults: GI on, depth as vrenderparams default, two-sided (SettingsWidget.cxx:65-90)
  c->par.width = 64; c->par.height = 64; c->par.max_depth = 5; c->par.two_sided = 1; c->par.seed = 1; c->par.tile_size = 32;
  c->par.white_point = 1.0f; c->par.russian_roulette = 1; c->par.env_as_background = 1;
  c->cam.dir[1] = 1.0f; c->cam.up[2] = 1.0f; c->cam.fovy_deg = 45.0f;
  return c;
}

void crh_destroy(crh_ctx* c)
{
  if (!c) return;
  hipSetDevice(c->device);
  hipStreamSynchronize(cstream(c));
  drain_events(c);
  for (hipEvent_t e : c->ev_pool) hipEventDestroy(e);
  void* ptrs[] = {c->d_nodes, c->d_tris, c->d_shade, c->d_mats, c->d_lights, c->d_env, c->d_accum, c->paths.ray_o[0], c->paths.ray_d[0], c->paths.ray_o[1], c->paths.ray_d[1], c->paths.thr[1],
                  c->paths.hit, c->paths.thr[0], c->paths.rad, c->paths.sh_o, c->paths.sh_d, c->paths.sh_c,
                  c->queues.q[0], c->queues.q[1], c->queues.q_sh, c->queues.counts, c->d_tile_ids, c->d_seeds, c->d_counters, c->d_api_cursor, c->d_scratch,
                  c->d_m2, c->d_tile_err, c->d_tile_cnt, c->d_uvs, c->d_texels, c->d_tex_desc, c->d_inst, c->d_patch, c->d_verts, c->d_ibox, c->queues.q2, c->queues.q2_sh};
  for (void* p : ptrs) if (p) hipFree(p);
  if (c->d_assembled) hipFree(c->d_assembled);
  if (c->d_peer_stage) hipFree(c->d_peer_stage);
  for (crh_ctx::Stage& st : c->stage) { if (st.p) hipHostFree(st.p); if (st.ev) hipEventDestroy(st.ev); }
  for (int k = 0; k < 8; ++k) { if (c->lane_stream[k]) { hipStreamSynchronize(c->lane_stream[k]); hipStreamDestroy(c->lane_stream[k]); } if (c->lane_join[k]) hipEventDestroy(c->lane_join[k]); }
  if (c->lane_fork) hipEventDestroy(c->lane_fork);
  if (c->d_lane_counts) hipFree(c->d_lane_counts);
  if (c->d_pipe_seeds) hipFree(c->d_pipe_seeds);
  if (c->rb_stream) { hipStreamSynchronize(c->rb_stream); hipStreamDestroy(c->rb_stream); }
  for (int k = 0; k < 2; ++k) { if (c->d_rb[k]) hipFree(c->d_rb[k]); if (c->h_rb[k]) hipHostFree(c->h_rb[k]); if (c->rb_tm[k]) hipEventDestroy(c->rb_tm[k]); if (c->rb_done[k]) hipEventDestroy(c->rb_done[k]); }
  if (c->rb_fork) hipEventDestroy(c->rb_fork);
  for (void* q : {(void*)c->d_tile_cdf, (void*)c->d_picked, (void*)c->d_adapt_n}) if (q) hipFree(q);
  release_comms(c);
  hipStreamDestroy(c->stream_);
  delete c;
}

const char* crh_last_error(crh_ctx* c) { return c ? c->err.c_str() : "null context"; }

int crh_set_geometry(crh_ctx* c, const float* pos, const float* nrm, const float* uv, uint32_t nV, const int32_t* tri, uint32_t nT,
                     const int32_t* tri_obj, const float* xf, uint32_t nO)
{
  if (!c) return CRH_E_INVALID;
  if ((nV && (!pos || !nrm)) || (nT && !tri)) return fail(c, CRH_E_INVALID, "null geometry array");
  if (nT >= (1u << 28)) return fail(c, CRH_E_INVALID, "too many triangles (limit 2^28)");
  if (!all_finite(pos, 3 * (size_t)nV, 1.0e30f) || !all_finite(nrm, 3 * (size_t)nV) || (uv && !all_finite(uv, 2 * (size_t)nV)) ||
      (xf && !all_finite(xf, 12 * (size_t)nO, 1.0e30f)))
    return fail(c, CRH_E_INVALID, "geometry holds a NaN / Inf (or a coordinate beyond 1e30)");
  for (uint32_t t = 0; t < nT; ++t)
    for (int k = 0; k < 3; ++k)
      if (tri[4 * t + k] < 0 || (uint32_t)tri[4 * t + k] >= nV) { char b[96]; snprintf(b, sizeof b, "triangle %u index out of range", t); return fail(c, CRH_E_INVALID, b); }
  if (tri_obj && xf && nO)
    for (uint32_t t = 0; t < nT; ++t)
      if (tri_obj[t] < 0 || (uint32_t)tri_obj[t] >= nO) return fail(c, CRH_E_INVALID, "triangle object id out of range");
  // every check passed: only now is the context's state replaced
  c->pos.assign(pos, pos + 3 * (size_t)nV); c->nrm.assign(nrm, nrm + 3 * (size_t)nV);
  if (uv) c->uv.assign(uv, uv + 2 * (size_t)nV); else c->uv.clear();
  c->tri.assign(tri, tri + 4 * (size_t)nT);
  c->two_level = false; c->nO = 0; c->xf.clear(); c->tri_obj.clear();
  if (tri_obj && xf && nO) {
    // two-level mode: vertices stay in object space; every object gets its own tree (crh_build), the top-level tree
    // over the instances carries the transforms (crh_set_transforms rebuilds only that)
    c->two_level = true; c->nO = nO;
    c->xf.assign(xf, xf + 12 * (size_t)nO); c->tri_obj.assign(tri_obj, tri_obj + nT);
  }
  c->built = false; c->pending_n = 0;
  return CRH_OK;
}

int crh_set_transforms(crh_ctx* c, const float* xf, uint32_t nO)
{
  if (!c || !xf) return fail(c, CRH_E_INVALID, "null transforms");
  if (!c->two_level || nO != c->nO) return fail(c, CRH_E_INVALID, "crh_set_transforms needs a two-level scene with the same object count");
  if (!all_finite(xf, 12 * (size_t)nO, 1.0e30f)) return fail(c, CRH_E_INVALID, "transform holds a NaN / Inf");
  CRH_HIP(hipSetDevice(c->device));
  c->xf.assign(xf, xf + 12 * (size_t)nO);
  if (!c->built) return do_reset(c);
  // The manipulator calls this every frame (ImRaytraceControls.cxx:58-89).  Nothing big is ever rebuilt here: an object of the static tree that
  // leaves the identity has its triangles THERE disabled (a scatter of all-zero records) and, the first time, gets an object tree of its own,
  // appended behind the trees built so far; back at the identity its triangles are restored and the instance dropped.  Then the top-level tree
  // over the instances of this moment is rebuilt on the host and only the new nodes and the instance table travel, stream-ordered.
  int threads = 0; if (const char* e = getenv("CRH_BUILD_THREADS")) threads = atoi(e);
  const uint32_t old_nodes = c->n_blas_nodes, old_pos = c->n_pos;
  c->bvh.nodes.resize(c->n_blas_nodes);
  std::vector<uint32_t> ppos; std::vector<float> prec;
  for (uint32_t ob = 0; ob < nO; ++ob) {
    crh_ctx::Obj& o = c->objs[ob];
    if (!o.ntri) continue;
    const bool want = std::memcmp(&xf[12 * (size_t)ob], &c->xf0[12 * (size_t)ob], 12 * sizeof(float)) != 0;      // off its build-time placement: an instance
    if (want == o.is_inst) continue;
    // static0 object changing sides: its records in the static tree die / come back
    for (uint32_t i = 0; i < o.ntri; ++i) {
      const uint32_t t = c->obj_tris[o.first + i], p = c->static_pos[t];
      float* q = &c->h_tris[12 * (size_t)p];
      if (want) { std::memset(q, 0, 48); std::memcpy(&q[3], &t, 4); }
      else {
        for (int k = 0; k < 3; ++k) { const int32_t vi = c->tri[4 * t + k]; for (int a = 0; a < 3; ++a) q[4 * k + a] = c->pos_w[3 * vi + a]; q[4 * k + 3] = 0.f; }
        std::memcpy(&q[3], &t, 4);
      }
      float d[16]; device_tri_record(q, d);
      ppos.push_back(p); prec.insert(prec.end(), d, d + 12);
    }
    if (want) { c->n_static_live -= o.ntri; if (!o.built) build_object_tree(c, ob, threads); }
    else c->n_static_live += o.ntri;
    o.is_inst = want;
  }
  c->n_blas_nodes = (uint32_t)c->bvh.nodes.size();
  int rc;
  if (c->n_pos > old_pos) {                               // records of the object trees just built
    if (c->n_pos > c->cap_pos || c->n_pos >= (1u << 28)) return fail(c, CRH_E_NOMEM, "leaf positions exhausted (object trees of moved objects)");
    std::vector<float> tr, sh, uvr, vt;
    fill_records(c, old_pos, c->n_pos, tr, sh, uvr, &vt);
    const size_t n = c->n_pos - old_pos;
    auto put = [&](void* dst, const std::vector<float>& v, size_t rec_floats) -> int {
      const size_t bytes = n * rec_floats * sizeof(float);
      if (bytes <= (4u << 20)) return stage_copy(c, dst, v.data(), bytes);
      CRH_HIP(hipMemcpyAsync(dst, v.data(), bytes, hipMemcpyHostToDevice, cstream(c))); CRH_HIP(hipStreamSynchronize(cstream(c))); return CRH_OK;
    };
    if ((rc = put(c->d_tris + (size_t)kTriStride * old_pos, tr, 4 * kTriStride))) return rc;
    if ((rc = put(c->d_shade + 4 * (size_t)old_pos, sh, 16))) return rc;
    if (c->d_uvs && !uvr.empty() && (rc = put(c->d_uvs + 2 * (size_t)old_pos, uvr, 8))) return rc;
    if ((rc = put(c->d_verts + 3 * (size_t)old_pos, vt, 12))) return rc;
  }
  if (!ppos.empty()) {
    const size_t nb = ppos.size() * 4, rb = prec.size() * 4, need = ((nb + 255) & ~(size_t)255) + rb;
    if (need > c->cap_patch) {
      CRH_HIP(hipStreamSynchronize(cstream(c)));
      if (c->d_patch) { CRH_HIP(hipFree(c->d_patch)); c->d_patch = nullptr; c->cap_patch = 0; }
      CRH_HIP(hipMalloc(&c->d_patch, need + need / 2)); c->cap_patch = need + need / 2;
    }
    char* base = (char*)c->d_patch; char* recs = base + ((nb + 255) & ~(size_t)255);
    auto put = [&](void* dst, const void* src, size_t bytes) -> int {
      if (bytes <= (4u << 20)) return stage_copy(c, dst, src, bytes);
      CRH_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, cstream(c))); CRH_HIP(hipStreamSynchronize(cstream(c))); return CRH_OK;
    };
    if ((rc = put(base, ppos.data(), nb))) return rc;
    if ((rc = put(recs, prec.data(), rb))) return rc;
    Launch L{cstream(c), c->grid, false};
    launch_scatter_tris(L, c->d_tris, (const uint32_t*)base, (const float4*)recs, (uint32_t)ppos.size());
    CRH_HIP(hipGetLastError());
  }
  if ((rc = build_tlas(c))) return rc;
  const size_t tail = c->bvh.nodes.size() - old_nodes;
  if (c->bvh.nodes.size() * sizeof(QNode) > c->cap_nodes) {          // more object trees than crh_build left room for: the whole array again, with head-room
    if ((rc = dev_put(c, c->d_nodes, c->cap_nodes, c->bvh.nodes.data(), c->bvh.nodes.size() * sizeof(QNode), c->bvh.nodes.size() * sizeof(QNode) / 2 + (size_t)(4 * c->nO + 64) * sizeof(QNode)))) return rc;
  } else if (tail) {
    const size_t bytes = tail * sizeof(QNode); void* dst = (char*)c->d_nodes + (size_t)old_nodes * sizeof(QNode);
    if (bytes <= (4u << 20)) { if ((rc = stage_copy(c, dst, c->bvh.nodes.data() + old_nodes, bytes))) return rc; }
    else { CRH_HIP(hipMemcpyAsync(dst, c->bvh.nodes.data() + old_nodes, bytes, hipMemcpyHostToDevice, cstream(c))); CRH_HIP(hipStreamSynchronize(cstream(c))); }
  }
  return do_reset(c);
}

int crh_get_tlas(crh_ctx* c, uint32_t* root, uint32_t* n_inst, uint32_t* n_blas)
{
  if (!c) return CRH_E_INVALID;
  if (!c->built) return fail(c, CRH_E_NOTBUILT, "crh_build has not been called");
  if (root) *root = c->inst.empty() ? 0u : (c->root2 != kQEmpty ? c->root2 : c->root);
  if (n_inst) *n_inst = (uint32_t)c->inst.size(); if (n_blas) *n_blas = c->n_blas_nodes;
  return CRH_OK;
}

int crh_set_materials(crh_ctx* c, const crh_bsdf* m, uint32_t n)
{
  if (!c || (n && !m)) return fail(c, CRH_E_INVALID, "null materials");
  if (!all_finite((const float*)m, 32 * (size_t)n)) return fail(c, CRH_E_INVALID, "material holds a NaN / Inf");
  CRH_HIP(hipSetDevice(c->device));
  c->mats.assign(m, m + n); c->pending_n = 0;
  return dev_put(c, c->d_mats, c->cap_mats, c->mats.data(), sizeof(crh_bsdf) * n, 16 * sizeof(crh_bsdf));
}

int crh_set_lights(crh_ctx* c, const crh_light* l, uint32_t n)
{
  if (!c || (n && !l)) return fail(c, CRH_E_INVALID, "null lights");
  if (!all_finite((const float*)l, 8 * (size_t)n, 1.0e30f)) return fail(c, CRH_E_INVALID, "light holds a NaN / Inf");
  CRH_HIP(hipSetDevice(c->device));
  c->lights.assign(l, l + n); c->pending_n = 0;
  return upload_lights(c);
}

int crh_set_envmap(crh_ctx* c, const float* rgb, uint32_t w, uint32_t h)
{
  if (!c) return CRH_E_INVALID;
  if (rgb && w && h && !all_finite(rgb, 3 * (size_t)w * h)) return fail(c, CRH_E_INVALID, "environment map holds a NaN / Inf");
  CRH_HIP(hipSetDevice(c->device));
  c->envW = c->envH = 0; c->pending_n = 0;
  if (rgb && w && h) {
    std::vector<float> t(4 * (size_t)w * h);
    for (size_t i = 0; i < (size_t)w * h; ++i) { t[4 * i] = rgb[3 * i]; t[4 * i + 1] = rgb[3 * i + 1]; t[4 * i + 2] = rgb[3 * i + 2]; t[4 * i + 3] = 0.f; }
    int rc = dev_put(c, c->d_env, c->cap_env, t.data(), t.size() * sizeof(float)); if (rc) return rc;
    c->envW = w; c->envH = h;
  }
  return CRH_OK;      // without a map the kernels take the background colour (fill_scene hands them a null pointer); the allocation is kept
}

int crh_set_texture(crh_ctx* c, uint32_t slot, const float* rgb, uint32_t w, uint32_t h, uint32_t channels)
{
  if (!c || slot >= 4096u) return fail(c, CRH_E_INVALID, "texture slot out of range");
  if (rgb && channels != 3u && channels != 4u) return fail(c, CRH_E_INVALID, "texture channels must be 3 or 4");
  if (rgb && w && h && !all_finite(rgb, (size_t)channels * w * h)) return fail(c, CRH_E_INVALID, "texture holds a NaN / Inf");
  CRH_HIP(hipSetDevice(c->device));
  CRH_HIP(hipStreamSynchronize(cstream(c)));
  if (c->textures.size() <= slot) c->textures.resize(slot + 1);
  crh_ctx::HostTex& t = c->textures[slot];
  t.rgba.clear(); t.w = t.h = 0;
  if (rgb && w && h) {
    t.rgba.resize(4 * (size_t)w * h);
    for (size_t i = 0; i < (size_t)w * h; ++i) {
      t.rgba[4 * i] = rgb[channels * i]; t.rgba[4 * i + 1] = rgb[channels * i + 1]; t.rgba[4 * i + 2] = rgb[channels * i + 2];
      t.rgba[4 * i + 3] = channels == 4u ? rgb[4 * i + 3] : 1.f;
    }
    t.w = w; t.h = h;
  }
  c->textures_dirty = true; c->pending_n = 0;
  return do_reset(c);
}

int crh_set_camera(crh_ctx* c, const crh_camera* cam)
{
  if (!c || !cam) return fail(c, CRH_E_INVALID, "null camera");
  const float f[] = {cam->eye[0], cam->eye[1], cam->eye[2], cam->dir[0], cam->dir[1], cam->dir[2], cam->up[0], cam->up[1], cam->up[2],
                     cam->fovy_deg, cam->aspect, cam->ortho_scale, cam->aperture_radius, cam->focal_dist};
  if (!all_finite(f, sizeof f / sizeof f[0], 1.0e30f)) return fail(c, CRH_E_INVALID, "camera holds a NaN / Inf");
  c->cam = *cam; c->pending_n = 0;                      // samples traced ahead with the old camera are dropped
  c->read_since_render = true;
  return CRH_OK;
}

int crh_set_params(crh_ctx* c, const crh_params* p)
{
  if (!c || !p) return fail(c, CRH_E_INVALID, "null params");
  if (!p->width || !p->height || p->max_depth < 1 || p->max_depth > 32) return fail(c, CRH_E_INVALID, "width/height must be > 0 and max_depth in 1..32");
  if (p->width > 32768u || p->height > 32768u || (uint64_t)p->width * p->height > (1ull << 28)) return fail(c, CRH_E_INVALID, "render target too large (limit 32768 per side, 2^28 pixels)");
  if (p->tile_size < 8 || (p->tile_size & 7u) || p->tile_size > 1024) return fail(c, CRH_E_INVALID, "tile_size must be a multiple of 8 in 8..1024");
  { const float f[] = {p->radiance_clamp, p->exposure, p->white_point, p->background[0], p->background[1], p->background[2], p->scene_epsilon};
    if (!all_finite(f, sizeof f / sizeof f[0])) return fail(c, CRH_E_INVALID, "params hold a NaN / Inf"); }
  c->par = *p;
  return do_reset(c);
}

int crh_set_spec(crh_ctx* c, const crh_spec* sp)
{
  if (!c || !sp) return fail(c, CRH_E_INVALID, "null spec");
  if (sp->size != sizeof(crh_spec)) return fail(c, CRH_E_INVALID, "crh_spec.size does not match this library's struct");
  if (!(sp->eta_no_dielectric >= 1.0e-2f && sp->eta_no_dielectric <= 1.0e3f)) return fail(c, CRH_E_INVALID, "eta_no_dielectric must be in 1e-2 .. 1e3");
  c->spec = *sp;
  c->spec.uniform_32bit = sp->uniform_32bit != 0; c->spec.texel_gamma2 = sp->texel_gamma2 != 0; c->spec.mis_single_lobe = sp->mis_single_lobe != 0;
  c->spec.eps_rule = sp->eps_rule != 0;
  return do_reset(c);                                     // like every rendering-parameter change (pending look-ahead samples are dropped there)
}

int crh_get_spec(crh_ctx* c, crh_spec* out)
{
  if (!c || !out) return fail(c, CRH_E_INVALID, "null spec");
  *out = c->spec; out->size = (uint32_t)sizeof(crh_spec);
  return CRH_OK;
}

int crh_spec_order_exact(void) { return CRH_SPEC_ORDER_EXACT; }
int crh_spec_anyhit_slot_order(void) { return CRH_SPEC_ANYHIT_SLOT_ORDER; }

int crh_build(crh_ctx* c)
{
  if (!c) return CRH_E_INVALID;
  const uint32_t nT = (uint32_t)(c->tri.size() / 4);
  if (nT && c->mats.empty()) return fail(c, CRH_E_INVALID, "no materials");
  CRH_HIP(hipSetDevice(c->device));
  CRH_HIP(hipStreamSynchronize(cstream(c)));
  int threads = 0; if (const char* e = getenv("CRH_BUILD_THREADS")) threads = atoi(e);
  const bool verbose = getenv("CRH_BUILD_VERBOSE") != nullptr; auto tp = std::chrono::steady_clock::now();
  auto phase = [&](const char* what) { if (verbose) { const auto now = std::chrono::steady_clock::now(); fprintf(stderr, "crh_build: %-28s %.3f s\n", what, std::chrono::duration<double>(now - tp).count()); tp = now; } };
  c->inst.clear(); c->root = 0; c->root2 = kQEmpty; c->objs.clear(); c->obj_tris.clear(); c->static_pos.clear(); c->pos_obj.clear();
  c->bvh.nodes.clear(); c->bvh.prim_order.clear();
  if (!c->two_level) {
    build_qbvh(c->pos.data(), c->tri.data(), nT, c->bvh, threads);
    c->n_static = c->n_static_live = c->n_pos = nT;
    for (int a = 0; a < 3; ++a) { c->sbmin[a] = c->bvh.bbmin[a]; c->sbmax[a] = c->bvh.bbmax[a]; }
  } else {
    // static / moved split: the objects at the identity share ONE world-space tree (first in the node array, its triangles first in leaf
    // order); every other non-empty object gets an object-space tree (triangles in input order); then the top-level tree over the instances
    c->objs.assign(c->nO, crh_ctx::Obj{}); c->obj_tris.resize(nT ? nT : 1); c->static_pos.assign(nT ? nT : 1, 0u);
    for (uint32_t t = 0; t < nT; ++t) c->objs[c->tri_obj[t]].ntri++;
    { uint32_t acc = 0; for (uint32_t ob = 0; ob < c->nO; ++ob) { crh_ctx::Obj& o = c->objs[ob]; o.first = acc; acc += o.ntri; o.ntri = 0; o.static0 = true; } }
    for (uint32_t t = 0; t < nT; ++t) { crh_ctx::Obj& o = c->objs[c->tri_obj[t]]; c->obj_tris[o.first + o.ntri++] = t; }
    // bake: every vertex under the transform its object has NOW (each vertex belongs to one object; an object at the identity keeps its bits) -- a
    // loaded scene whose objects all carry a location (vlocation lines of model.tcl) renders as ONE tree at the single-level rate until one is dragged
    c->xf0 = c->xf; c->pos_w = c->pos; c->nrm_w = c->nrm;
    {
      std::vector<uint8_t> done(c->pos.size() / 3 + 1, 0);
      for (uint32_t t = 0; t < nT; ++t) {
        const float* M = &c->xf0[12 * (size_t)c->tri_obj[t]];
        if (is_identity(M)) continue;
        for (int k = 0; k < 3; ++k) {
          const int32_t vi = c->tri[4 * t + k];
          if (done[vi]) continue;
          done[vi] = 1;
          const crh_v3 pw = crh_xform_point(M, crh_mk3(c->pos[3 * vi], c->pos[3 * vi + 1], c->pos[3 * vi + 2]));
          crh_v3 nn = crh_norm3(crh_xform_vector(M, crh_mk3(c->nrm[3 * vi], c->nrm[3 * vi + 1], c->nrm[3 * vi + 2])));
          if (!(crh_dot3(nn, nn) > 0.f)) nn = crh_mk3(c->nrm[3 * vi], c->nrm[3 * vi + 1], c->nrm[3 * vi + 2]);
          c->pos_w[3 * vi] = pw.x; c->pos_w[3 * vi + 1] = pw.y; c->pos_w[3 * vi + 2] = pw.z;
          c->nrm_w[3 * vi] = nn.x; c->nrm_w[3 * vi + 1] = nn.y; c->nrm_w[3 * vi + 2] = nn.z;
        }
      }
    }
    std::vector<uint32_t> list; list.reserve(nT);
    for (uint32_t t = 0; t < nT; ++t) if (c->objs[c->tri_obj[t]].static0) list.push_back(t);
    const uint32_t nS = (uint32_t)list.size();
    c->n_static = c->n_static_live = nS; c->n_pos = nS;
    if (nS) {
      std::vector<float> boxes(6 * (size_t)nS); std::vector<uint32_t> order;
      for (uint32_t i = 0; i < nS; ++i)
        for (int a = 0; a < 3; ++a) {
          const uint32_t t = list[i];
          const float v0 = c->pos_w[3 * c->tri[4 * t + 0] + a], v1 = c->pos_w[3 * c->tri[4 * t + 1] + a], v2 = c->pos_w[3 * c->tri[4 * t + 2] + a];
          boxes[6 * (size_t)i + a] = std::min(v0, std::min(v1, v2)); boxes[6 * (size_t)i + 3 + a] = std::max(v0, std::max(v1, v2));
        }
      c->bvh.nodes.reserve(nS / 2 + 16);
      build_tree(boxes.data(), nS, false, 0, c->bvh.nodes, order, c->sbmin, c->sbmax, threads);
      c->bvh.prim_order.resize(nS);
      for (uint32_t i = 0; i < nS; ++i) { c->bvh.prim_order[i] = list[order[i]]; c->static_pos[list[order[i]]] = i; }
    }
    for (uint32_t ob = 0; ob < c->nO; ++ob) {
      crh_ctx::Obj& o = c->objs[ob];
      if (o.static0 || !o.ntri) continue;
      build_object_tree(c, ob, threads);
      o.is_inst = true;
    }
  }
  phase("trees");
  c->n_blas_nodes = (uint32_t)c->bvh.nodes.size();
  if (c->n_pos >= (1u << 28)) return fail(c, CRH_E_INVALID, "too many leaf positions (limit 2^28)");
  if (c->two_level) {
    // room for the instance table of ANY later placement (one record per object + one per instance), so that the first crh_set_transforms --
    // the user has just grabbed the gizmo -- allocates nothing
    const size_t want_inst = 128 * (2 * (size_t)c->nO + 64);
    if (c->cap_inst < want_inst) {
      if (c->d_inst) { CRH_HIP(hipFree(c->d_inst)); c->d_inst = nullptr; c->cap_inst = 0; }
      CRH_HIP(hipMalloc((void**)&c->d_inst, want_inst)); c->cap_inst = want_inst;
      CRH_HIP(hipMemsetAsync(c->d_inst, 0, want_inst, cstream(c)));
    }
  }
  { int rc_t = build_tlas(c); if (rc_t) return rc_t; }
  // leaf-ordered triangle, shading and uv records.  A two-level scene keeps room for an object tree of every object of the static tree
  // (each may be dragged away once; the copies cost 2 x 64 B per triangle of HBM, nothing at run time)
  std::vector<float> tr, sh, uvr, vt;
  c->h_tris.clear();
  fill_records(c, 0, c->n_pos, tr, sh, uvr, c->two_level ? &vt : nullptr);
  phase("leaf-ordered records");
  c->cap_pos = (size_t)std::max(c->n_pos, 1u) + (c->two_level ? c->n_static : 0u);
  if (c->two_level) {
    // host arrays that grow when an object tree is built later: reserve now -- the first growth of a 33 MB node vector or a 48 MB record vector is a
    // reallocation + copy of 10-20 ms, which used to land in the first dragged frame
    c->bvh.nodes.reserve(c->bvh.nodes.size() + 2 * (size_t)c->n_static + 4 * (size_t)c->nO + 64);
    c->h_tris.reserve(12 * c->cap_pos); c->bvh.prim_order.reserve(c->cap_pos); c->pos_obj.reserve(c->cap_pos);
  }
  int rc;
  // head-room behind the node array: object trees built later by crh_set_transforms (<= ~1.5 nodes per triangle incl. alignment holes) and the top-level tree
  if ((rc = dev_put(c, c->d_nodes, c->cap_nodes, c->bvh.nodes.data(), c->bvh.nodes.size() * sizeof(QNode),
                    (size_t)((c->two_level ? 2 * (size_t)c->n_static : 0) + 4 * (size_t)c->nO + 64) * sizeof(QNode)))) return rc;
  auto alloc_put = [&](float4*& dptr, const std::vector<float>& v, size_t rec_floats) -> int {
    CRH_HIP(hipStreamSynchronize(cstream(c)));
    if (dptr) { CRH_HIP(hipFree(dptr)); dptr = nullptr; }
    CRH_HIP(hipMalloc((void**)&dptr, c->cap_pos * rec_floats * sizeof(float)));
    CRH_HIP(hipMemcpyAsync(dptr, v.data(), (size_t)std::max(c->n_pos, 1u) * rec_floats * sizeof(float), hipMemcpyHostToDevice, cstream(c)));
    CRH_HIP(hipStreamSynchronize(cstream(c)));
    return CRH_OK;
  };
  if ((rc = alloc_put(c->d_tris, tr, 4 * kTriStride))) return rc;
  if ((rc = alloc_put(c->d_shade, sh, 16))) return rc;
  if (!c->uv.empty()) { if ((rc = alloc_put(c->d_uvs, uvr, 8))) return rc; }
  else if (c->d_uvs) { CRH_HIP(hipFree(c->d_uvs)); c->d_uvs = nullptr; }
  if (c->two_level) { if ((rc = alloc_put(c->d_verts, vt, 12))) return rc; }
  else if (c->d_verts) { CRH_HIP(hipFree(c->d_verts)); c->d_verts = nullptr; }
  phase("upload");
  if (c->two_level) {
    // what the FIRST crh_set_transforms would otherwise allocate while the user is dragging: the staging of the triangle patches of the largest object
    uint32_t biggest = 0; for (const crh_ctx::Obj& o : c->objs) biggest = std::max(biggest, o.ntri);
    // ... and the pinned staging buffers the records of one object travel through (stage_copy grows them lazily: four hipHostMalloc of a few
    // milliseconds each would otherwise land in the first dragged frame)
    const size_t want_stage = std::min<size_t>(4u << 20, (size_t)biggest * 64 + (size_t)c->nO * 160 + 65536);
    for (crh_ctx::Stage& st : c->stage)
      if (st.cap < want_stage) {
        if (st.used) CRH_HIP(hipEventSynchronize(st.ev));
        if (st.p) { CRH_HIP(hipHostFree(st.p)); st.p = nullptr; st.cap = 0; }
        CRH_HIP(hipHostMalloc(&st.p, want_stage, hipHostMallocDefault)); st.cap = want_stage;
      }
    const size_t want_patch = 2 * ((size_t)biggest * 52 + 512);
    if (c->cap_patch < want_patch) {
      if (c->d_patch) { CRH_HIP(hipFree(c->d_patch)); c->d_patch = nullptr; c->cap_patch = 0; }
      CRH_HIP(hipMalloc(&c->d_patch, want_patch)); c->cap_patch = want_patch;
    }
  }
  c->built = true;
  if (c->two_level && c->inst.empty()) {
    // Every object sits at the identity: the single-level kernels render this scene.  The first crh_set_transforms (the user has just grabbed the
    // gizmo) switches to the two-level instantiations and the record scatter -- launch each of them once now, on empty queues, so that their
    // first-launch cost (function lookup, code upload: ~20 ms for the set) is paid while the scene loads and not in the first dragged frame.
    if ((rc = ensure_paths(c, 4096))) return rc;
    CRH_HIP(hipMemsetAsync(c->queues.counts, 0, kCounts * sizeof(uint32_t), cstream(c)));
    DScene S; fill_scene(c, S); S.two_level = 1; S.root2 = 0;
    for (int don = 0; don < 2; ++don) {
      Launch LT{cstream(c), 64, false, c->clamp_grid ? c->cus : 0, don != 0};      // with the occupancy query of resident_grid<>
      S.split = 0; launch_trace_nearest(LT, S, c->paths, c->queues, 0, 0, c->d_counters); launch_trace_any(LT, S, c->paths, c->queues, c->d_counters);      // the instantiations of an all-moved scene
      S.split = 1; launch_trace_nearest(LT, S, c->paths, c->queues, 0, 0, c->d_counters); launch_trace_any(LT, S, c->paths, c->queues, c->d_counters);      // and the second-pass ones
    }
    Launch L{cstream(c), 64, false};
    launch_scatter_tris(L, c->d_tris, (const uint32_t*)c->d_patch, (const float4*)c->d_patch, 0);
    CRH_HIP(hipGetLastError());
  }
  rc = do_reset(c); if (rc) return rc;
  CRH_HIP(hipStreamSynchronize(cstream(c)));
  return CRH_OK;
}

int crh_reset(crh_ctx* c) { if (!c) return CRH_E_INVALID; return do_reset(c); }

int crh_render(crh_ctx* c, uint32_t n)
{
  if (!c) return CRH_E_INVALID;
  if (!c->built) return fail(c, CRH_E_NOTBUILT, "crh_build has not been called");
  c->assembled_valid = false;                            // reads show this context's own accumulator again
  if (c->adaptive) {
    CRH_HIP(hipSetDevice(c->device));
    for (uint32_t i = 0; i < n; ++i) { int rc = adaptive_iteration(c); if (rc) return rc; }
    c->frames_done += n;
    return CRH_OK;
  }
  const uint32_t ts = c->par.tile_size;
  const uint32_t nt = ((c->par.width + ts - 1) / ts) * ((c->par.height + ts - 1) / ts);
  std::vector<uint32_t> all(nt);
  for (uint32_t i = 0; i < nt; ++i) all[i] = i;
  // crh_set_lookahead_auto: the first frame after a restart is ONE sample (what the user sees while dragging), then batches of 4, 16, ... max
  const bool ramp = c->lookahead_auto > 1;
  const uint32_t k_max = ramp ? c->lookahead_auto : c->lookahead;
  if (k_max > 1 && (uint64_t)nt * ts * ts * k_max <= c->max_paths) {
    // look-ahead: one wide batch of `lookahead` frames is traced at once; each call folds in only the samples it asked for
    CRH_HIP(hipSetDevice(c->device));
    while (n > 0) {
      if (c->pending_n == 0 || c->pending_first != c->frames_done || c->pending_tiles != nt) {
        uint32_t k = c->lookahead;
        if (ramp) {
          if (n >= k_max) {                                                   // the caller asks for a whole batch itself: nothing to speculate on,
            int rcn = render_impl(c, all.data(), nt, c->frames_done, n); if (rcn) return rcn;      // and render_impl cuts it into the widest batches that fit
            c->frames_done += n; n = 0; c->ramp_k = k_max;
            break;
          }
          k = std::max(c->ramp_k, n);                                         // a call that asks for n samples at once is not cut finer than that
          c->ramp_k = std::min(4u * k, k_max);
          if (k == 1) {                                                       // right after a restart: a plain frame
            int rc1 = render_impl(c, all.data(), nt, c->frames_done, 1); if (rc1) return rc1;
            c->frames_done += 1; n -= 1;
            continue;
          }
        }
        int rc_t = upload_textures(c); if (rc_t) return rc_t;
        if (nt > c->tile_cap) { CRH_HIP(hipStreamSynchronize(cstream(c))); if (c->d_tile_ids) CRH_HIP(hipFree(c->d_tile_ids)); CRH_HIP(hipMalloc((void**)&c->d_tile_ids, sizeof(uint32_t) * nt)); c->tile_cap = nt; }
        if (k > c->seed_cap) { CRH_HIP(hipStreamSynchronize(cstream(c))); if (c->d_seeds) CRH_HIP(hipFree(c->d_seeds)); CRH_HIP(hipMalloc((void**)&c->d_seeds, sizeof(uint32_t) * k)); c->seed_cap = k; }
        std::vector<uint32_t> seeds(k);
        { uint32_t hi = c->par.seed, lo = c->par.seed ^ 0x49616E42u;
          for (uint32_t i = 0; i < c->frames_done + k; ++i) { hi = (hi >> 2) + (hi << 2); hi += lo; lo += hi; if (i >= c->frames_done) seeds[i - c->frames_done] = hi >> 2; } }
        c->h_tile_ids.clear();
        CRH_HIP(hipMemcpyAsync(c->d_tile_ids, all.data(), sizeof(uint32_t) * nt, hipMemcpyHostToDevice, cstream(c)));
        CRH_HIP(hipMemcpyAsync(c->d_seeds, seeds.data(), sizeof(uint32_t) * k, hipMemcpyHostToDevice, cstream(c)));
        CRH_HIP(hipStreamSynchronize(cstream(c)));
        int rc_p = ensure_paths(c, nt * ts * ts * k); if (rc_p) return rc_p;
        DScene S; fill_scene(c, S);
        hipEvent_t e0 = get_event(c), e1 = get_event(c);
        hipEventRecord(e0, cstream(c));
        int rc_b = run_batch(c, S, c->d_tile_ids, nt, c->d_seeds, k, 0, false); if (rc_b) return rc_b;
        hipEventRecord(e1, cstream(c));
        c->render_ev.emplace_back(e0, e1);
        { int rc_e = trim_events(c); if (rc_e) return rc_e; }
        c->pending_first = c->frames_done; c->pending_n = k; c->pending_off = 0; c->pending_tiles = nt;
      }
      const uint32_t m = std::min(n, c->pending_n);
      DScene S; fill_scene(c, S);
      Launch L{cstream(c), c->grid, false};
      launch_accumulate(L, S, c->paths, c->d_accum, nullptr, c->d_tile_ids, nt, c->pending_off, m, c->pending_off + c->pending_n, c->d_counters);      // the batch held pending_off + pending_n samples
      CRH_HIP(hipGetLastError());
      c->pending_off += m; c->pending_n -= m; c->pending_first += m; c->frames_done += m; n -= m;
    }
    return CRH_OK;
  }
  int rc = render_impl(c, all.data(), nt, c->frames_done, n);
  if (rc == CRH_OK) c->frames_done += n;
  return rc;
}

int crh_render_tiles(crh_ctx* c, const uint32_t* tiles, uint32_t nt, uint32_t first, uint32_t ns)
{
  if (!c || (nt && !tiles)) return fail(c, CRH_E_INVALID, "null tile list");
  c->assembled_valid = false;
  return render_impl(c, tiles, nt, first, ns);
}

int crh_set_adaptive(crh_ctx* c, int on, uint32_t tiles_per_iteration)
{
  if (!c || (on && tiles_per_iteration == 0)) return fail(c, CRH_E_INVALID, "tiles_per_iteration must be > 0");
  c->adaptive = on != 0; if (on) c->adaptive_tiles = tiles_per_iteration;
  return do_reset(c);                                   // like every rendering-parameter change, restarts accumulation
}

int crh_set_show_tiles(crh_ctx* c, int on)
{
  if (!c) return CRH_E_INVALID;
  c->show_tiles = on != 0;                              // display-only: accumulation goes on
  return CRH_OK;
}

int crh_set_lookahead(crh_ctx* c, uint32_t frames)
{
  if (!c || frames == 0) return fail(c, CRH_E_INVALID, "lookahead must be >= 1");
  c->lookahead = frames; c->pending_n = 0;
  return CRH_OK;
}

int crh_set_lookahead_auto(crh_ctx* c, uint32_t max_frames)
{
  if (!c) return CRH_E_INVALID;
  c->lookahead_auto = max_frames; c->ramp_k = 1; c->pending_n = 0;
  return CRH_OK;
}

int crh_set_schedule(crh_ctx* c, int mode)
{
  if (!c || mode < CRH_SCHEDULE_AUTO || mode > CRH_SCHEDULE_SMALL) return fail(c, CRH_E_INVALID, "schedule must be CRH_SCHEDULE_AUTO / _WIDE / _SMALL");
  if (c->schedule == CRH_SCHEDULE_AUTO) { c->auto_lane_max_paths = c->lane_max_paths; c->auto_donate = c->donate; c->auto_pipeline = c->pipeline; }
  c->schedule = mode; c->pending_n = 0;
  c->read_since_render = true;                           // the next frame is not pipelined behind frames of the other schedule
  // WIDE: no batch counts as small (run_batch: one stream, full grids, plain kernels; render_impl: no frame pipelining; adaptive
  // iterations: plain kernels).  SMALL: every batch up to the path budget does.
  c->lane_max_paths = mode == CRH_SCHEDULE_WIDE ? 0u : (mode == CRH_SCHEDULE_SMALL ? (1u << 30) : c->auto_lane_max_paths);
  c->donate = mode == CRH_SCHEDULE_WIDE ? false : c->auto_donate;
  c->pipeline = mode == CRH_SCHEDULE_WIDE ? false : c->auto_pipeline;
  return CRH_OK;
}

int crh_set_pipeline_depth(crh_ctx* c, uint32_t frames)
{
  if (!c || frames < 2u || frames > 8u) return fail(c, CRH_E_INVALID, "pipeline depth must be in 2 .. 8 frames");
  CRH_HIP(hipSetDevice(c->device));
  CRH_HIP(hipStreamSynchronize(cstream(c)));            // the frames in flight own slices of the path state laid out for the old depth
  c->pipe_depth = frames; c->pipe_seq = 0; c->pipe_total = 0;
  return CRH_OK;
}

int crh_set_path_budget(crh_ctx* c, uint64_t max_paths)
{
  if (!c || max_paths < 1024u || max_paths > (1ull << 30)) return fail(c, CRH_E_INVALID, "path budget must be in 1024 .. 2^30 slots");
  CRH_HIP(hipSetDevice(c->device));
  c->max_paths = (uint32_t)max_paths; c->pending_n = 0;
  if (c->path_cap > c->max_paths) {                      // give the memory back now; the next render allocates what it needs
    CRH_HIP(hipStreamSynchronize(cstream(c)));
    void** ptrs[] = {(void**)&c->paths.ray_o[0], (void**)&c->paths.ray_d[0], (void**)&c->paths.hit, (void**)&c->paths.thr[0], (void**)&c->paths.rad,
                     (void**)&c->paths.sh_o, (void**)&c->paths.sh_d, (void**)&c->paths.sh_c, (void**)&c->queues.q[0], (void**)&c->queues.q[1], (void**)&c->queues.q_sh,
                     (void**)&c->paths.ray_o[1], (void**)&c->paths.ray_d[1], (void**)&c->paths.thr[1], (void**)&c->queues.q2, (void**)&c->queues.q2_sh};
    for (void** q : ptrs) if (*q) { CRH_HIP(hipFree(*q)); *q = nullptr; }
    c->path_cap = 0;
  }
  return CRH_OK;
}

int crh_get_tile_stats(crh_ctx* c, float* err, uint32_t* counts, uint32_t* n_tiles)
{
  if (!c || !c->d_accum) return fail(c, CRH_E_INVALID, "no accumulator");
  CRH_HIP(hipSetDevice(c->device));
  std::vector<float> e; std::vector<uint32_t> n;
  int rc = tile_stats(c, e, n); if (rc) return rc;
  if (n_tiles) *n_tiles = (uint32_t)e.size();
  if (err) std::memcpy(err, e.data(), sizeof(float) * e.size());
  if (counts) std::memcpy(counts, n.data(), sizeof(uint32_t) * n.size());
  return CRH_OK;
}

int crh_sync(crh_ctx* c) { if (!c) return CRH_E_INVALID; c->read_since_render = true; CRH_HIP(hipSetDevice(c->device)); CRH_HIP(hipStreamSynchronize(cstream(c))); return CRH_OK; }

int crh_read_hdr(crh_ctx* c, float* out)
{
  if (c) c->read_since_render = true;
  if (!c || !out || !c->d_accum) return fail(c, CRH_E_INVALID, "no accumulator / null output");
  CRH_HIP(hipSetDevice(c->device));
  const uint32_t n = c->par.width * c->par.height;
  int rc = ensure_scratch(c, sizeof(float) * 3 * (size_t)n); if (rc) return rc;
  Launch L{cstream(c), c->grid, false};
  launch_hdr(L, c->assembled_valid ? c->d_assembled : c->d_accum, (float*)c->d_scratch, n);
  CRH_HIP(hipMemcpyAsync(out, c->d_scratch, sizeof(float) * 3 * (size_t)n, hipMemcpyDeviceToHost, cstream(c)));
  CRH_HIP(hipStreamSynchronize(cstream(c)));
  return CRH_OK;
}

int crh_read_ldr(crh_ctx* c, uint8_t* out)
{
  if (c) c->read_since_render = true;
  if (!c || !out || !c->d_accum) return fail(c, CRH_E_INVALID, "no accumulator / null output");
  CRH_HIP(hipSetDevice(c->device));
  const uint32_t n = c->par.width * c->par.height;
  const uint32_t ts = c->par.tile_size, n_tiles = ((c->par.width + ts - 1) / ts) * ((c->par.height + ts - 1) / ts);
  const bool overlay = c->show_tiles && c->adaptive && c->picked_valid && c->d_picked && c->tile_stat_cap >= n_tiles;
  int rc = ensure_scratch(c, 3 * (size_t)n); if (rc) return rc;
  const uint8_t* d_mask = overlay ? c->d_picked : nullptr;            // written by the device-side tile draw of the last iteration
  Launch L{cstream(c), c->grid, false};
  launch_tonemap(L, c->assembled_valid ? c->d_assembled : c->d_accum, (uint8_t*)c->d_scratch, n, c->par.tonemap_mode, c->par.exposure, c->par.white_point, d_mask, c->par.width, ts);
  CRH_HIP(hipMemcpyAsync(out, c->d_scratch, 3 * (size_t)n, hipMemcpyDeviceToHost, cstream(c)));
  CRH_HIP(hipStreamSynchronize(cstream(c)));
  return CRH_OK;
}

// Asynchronous read-back of the frame as submitted so far, LDR (tone-mapped RGB8) or HDR (linear float RGB): tone map / unpack + device-to-host copy
// run on a stream of their own into one of two device / pinned-host buffer pairs while the next Redraw()s are already rendering; only the NEXT
// accumulate waits (for the kernel that reads the accumulator), nothing else does.
static int read_begin(crh_ctx* c, bool hdr)
{
  if (!c || !c->d_accum) return fail(c, CRH_E_INVALID, "no accumulator");
  if (c->rb_outstanding >= 2) return fail(c, CRH_E_INVALID, "two read-backs are already in flight (crh_read_ldr_end / crh_read_hdr_end first)");
  CRH_HIP(hipSetDevice(c->device));
  const uint32_t n = c->par.width * c->par.height;
  const size_t bytes = (hdr ? 12 : 3) * (size_t)n;
  if (!c->rb_stream) {
    CRH_HIP(hipStreamCreateWithFlags(&c->rb_stream, hipStreamNonBlocking));
    CRH_HIP(hipEventCreateWithFlags(&c->rb_fork, hipEventDisableTiming));
    for (int k = 0; k < 2; ++k) { CRH_HIP(hipEventCreateWithFlags(&c->rb_tm[k], hipEventDisableTiming)); CRH_HIP(hipEventCreateWithFlags(&c->rb_done[k], hipEventDisableTiming)); }
  }
  if (bytes > c->rb_cap) {
    CRH_HIP(hipStreamSynchronize(c->rb_stream));
    if (c->rb_outstanding) return fail(c, CRH_E_INVALID, "the read-back buffers must grow while a read-back is in flight (crh_read_*_end first)");
    for (int k = 0; k < 2; ++k) { if (c->d_rb[k]) CRH_HIP(hipFree(c->d_rb[k])); if (c->h_rb[k]) CRH_HIP(hipHostFree(c->h_rb[k])); c->d_rb[k] = nullptr; c->h_rb[k] = nullptr; }
    for (int k = 0; k < 2; ++k) { CRH_HIP(hipMalloc((void**)&c->d_rb[k], bytes)); CRH_HIP(hipHostMalloc((void**)&c->h_rb[k], bytes, hipHostMallocDefault)); }
    c->rb_cap = bytes;
  }
  const uint32_t slot = c->rb_head & 1u;
  // everything submitted so far comes first: the frames in flight on the pipeline streams (not joined, they stay in flight) and
  // whatever sits on the context's stream
  for (int k = 0; k < 8; ++k) if (c->pipe_pending[k]) CRH_HIP(hipStreamWaitEvent(c->rb_stream, c->lane_join[k], 0));
  CRH_HIP(hipEventRecord(c->rb_fork, c->stream_));
  CRH_HIP(hipStreamWaitEvent(c->rb_stream, c->rb_fork, 0));
  Launch L{c->rb_stream, c->grid, false};
  const float4* src = c->assembled_valid ? c->d_assembled : c->d_accum;
  if (hdr) launch_hdr(L, src, (float*)c->d_rb[slot], n);
  else {
    const uint32_t ts = c->par.tile_size, n_tiles = ((c->par.width + ts - 1) / ts) * ((c->par.height + ts - 1) / ts);
    const bool overlay = c->show_tiles && c->adaptive && c->picked_valid && c->d_picked && c->tile_stat_cap >= n_tiles;
    launch_tonemap(L, src, c->d_rb[slot], n, c->par.tonemap_mode, c->par.exposure, c->par.white_point, overlay ? c->d_picked : nullptr, c->par.width, ts);
  }
  CRH_HIP(hipGetLastError());
  CRH_HIP(hipEventRecord(c->rb_tm[slot], c->rb_stream));
  CRH_HIP(hipMemcpyAsync(c->h_rb[slot], c->d_rb[slot], bytes, hipMemcpyDeviceToHost, c->rb_stream));
  CRH_HIP(hipEventRecord(c->rb_done[slot], c->rb_stream));
  c->rb_bytes[slot] = bytes; c->rb_hdr[slot] = hdr;
  c->rb_guard = c->rb_tm[slot]; c->rb_guard_pending = true;
  ++c->rb_head; ++c->rb_outstanding;
  return CRH_OK;
}

static int read_end(crh_ctx* c, void* out, bool hdr)
{
  if (!c || !out) return fail(c, CRH_E_INVALID, "null output");
  if (!c->rb_outstanding) return fail(c, CRH_E_INVALID, "no read-back in flight (crh_read_*_begin first)");
  CRH_HIP(hipSetDevice(c->device));
  const uint32_t slot = (c->rb_head - c->rb_outstanding) & 1u;          // the oldest one
  if (c->rb_hdr[slot] != hdr) return fail(c, CRH_E_INVALID, "the oldest read-back in flight is of the other kind (LDR / HDR): end it with its own call");
  CRH_HIP(hipEventSynchronize(c->rb_done[slot]));
  std::memcpy(out, c->h_rb[slot], c->rb_bytes[slot]);
  --c->rb_outstanding;
  return CRH_OK;
}

int crh_read_ldr_begin(crh_ctx* c) { return read_begin(c, false); }
int crh_read_ldr_end(crh_ctx* c, uint8_t* out) { return read_end(c, out, false); }
int crh_read_hdr_begin(crh_ctx* c) { return read_begin(c, true); }
int crh_read_hdr_end(crh_ctx* c, float* out) { return read_end(c, out, true); }

int crh_save_accum(crh_ctx* c, float* out, uint32_t* frames_done)
{
  if (c) c->read_since_render = true;
  if (!c || !out || !c->d_accum) return fail(c, CRH_E_INVALID, "no accumulator / null output");
  CRH_HIP(hipSetDevice(c->device));
  CRH_HIP(hipStreamSynchronize(cstream(c)));
  CRH_HIP(hipMemcpyAsync(out, c->assembled_valid ? c->d_assembled : c->d_accum, sizeof(float4) * (size_t)c->par.width * c->par.height, hipMemcpyDeviceToHost, cstream(c)));
  CRH_HIP(hipStreamSynchronize(cstream(c)));
  if (frames_done) *frames_done = c->frames_done;
  return CRH_OK;
}

int crh_load_accum(crh_ctx* c, const float* in, uint32_t frames_done)
{
  if (!c || !in || !c->d_accum) return fail(c, CRH_E_INVALID, "no accumulator / null input");
  if (c->adaptive) return fail(c, CRH_E_INVALID, "checkpoints do not carry the adaptive sampler's second moments");
  CRH_HIP(hipSetDevice(c->device));
  CRH_HIP(hipStreamSynchronize(cstream(c)));
  CRH_HIP(hipMemcpyAsync(c->d_accum, in, sizeof(float4) * (size_t)c->par.width * c->par.height, hipMemcpyHostToDevice, cstream(c)));
  CRH_HIP(hipStreamSynchronize(cstream(c)));
  c->frames_done = frames_done; c->pending_n = 0; c->assembled_valid = false;
  return CRH_OK;
}

int crh_accum_device_ptr(crh_ctx* c, void** p, uint64_t* nbytes)
{
  if (!c || !p || !c->d_accum) return fail(c, CRH_E_INVALID, "no accumulator");
  *p = c->d_accum; if (nbytes) *nbytes = sizeof(float4) * (uint64_t)c->par.width * c->par.height;
  return CRH_OK;
}

int crh_reduce(crh_ctx* const* ctxs, uint32_t n, uint32_t root)
{
  if (!ctxs || n == 0 || root >= n || !ctxs[root]) return CRH_E_INVALID;
  return reduce_impl(ctxs, n, root);
}

int crh_enable_counters(crh_ctx* c, int on) { if (!c) return CRH_E_INVALID; c->counters_on = on != 0; return CRH_OK; }

int crh_get_stats(crh_ctx* c, crh_stats* out)
{
  if (c) c->read_since_render = true;
  if (!c || !out) return fail(c, CRH_E_INVALID, "null stats");
  CRH_HIP(hipSetDevice(c->device));
  CRH_HIP(hipStreamSynchronize(cstream(c)));
  drain_events(c);
  DCounters h;
  CRH_HIP(hipMemcpyAsync(&h, c->d_counters, sizeof h, hipMemcpyDeviceToHost, cstream(c)));
  CRH_HIP(hipStreamSynchronize(cstream(c)));
  out->rays_nearest = h.rays_nearest; out->rays_any = h.rays_any; out->nodes_nearest = h.nodes_nearest; out->tris_nearest = h.tris_nearest;
  out->nodes_any = h.nodes_any; out->tris_any = h.tris_any; out->shaded_hits = h.shaded_hits; out->samples = h.samples;
  out->seconds = c->seconds_acc;
  return CRH_OK;
}

static int trace_api(crh_ctx* c, const float* rays, uint32_t n, int any_hit, float* out_hit, uint32_t* out_vis)
{
  if (!c || (n && (!rays || (!out_hit && !out_vis)))) return fail(c, CRH_E_INVALID, "null ray buffers");
  if (!c->built) return fail(c, CRH_E_NOTBUILT, "crh_build has not been called");
  if (!n) return CRH_OK;
  CRH_HIP(hipSetDevice(c->device));
  const size_t in_b = 32 * (size_t)n, out_b = (any_hit ? 4 : 16) * (size_t)n;
  int rc = ensure_scratch(c, in_b + out_b); if (rc) return rc;
  char* base = (char*)c->d_scratch;
  CRH_HIP(hipMemcpyAsync(base, rays, in_b, hipMemcpyHostToDevice, cstream(c)));
  DScene S; fill_scene(c, S);
  Launch L{cstream(c), c->grid_trace, c->counters_on, c->clamp_grid ? c->cus : 0};
  launch_trace_rays(L, S, (const float4*)base, n, any_hit, (float4*)(base + in_b), (uint32_t*)(base + in_b), c->d_api_cursor, c->d_counters);
  CRH_HIP(hipGetLastError());
  CRH_HIP(hipMemcpyAsync(any_hit ? (void*)out_vis : (void*)out_hit, base + in_b, out_b, hipMemcpyDeviceToHost, cstream(c)));
  CRH_HIP(hipStreamSynchronize(cstream(c)));
  return CRH_OK;
}
int crh_trace_nearest(crh_ctx* c, const float* rays, uint32_t n, float* out_hit) { return trace_api(c, rays, n, 0, out_hit, nullptr); }
int crh_trace_any(crh_ctx* c, const float* rays, uint32_t n, uint32_t* out_vis) { return trace_api(c, rays, n, 1, nullptr, out_vis); }

int crh_get_bvh(crh_ctx* c, float* nodes, uint32_t* nn, float* tris, uint32_t* nt)
{
  if (!c) return CRH_E_INVALID;
  if (!c->built) return fail(c, CRH_E_NOTBUILT, "crh_build has not been called");
  const uint32_t nP = c->n_pos;                         // leaf positions in use: the triangles, + the object-tree copies of objects dragged out of the static tree
  if (nn) *nn = (uint32_t)c->bvh.nodes.size();
  if (nt) *nt = nP;
  if (nodes) std::memcpy(nodes, c->bvh.nodes.data(), c->bvh.nodes.size() * sizeof(QNode));
  if (tris && nP) std::memcpy(tris, c->h_tris.data(), 48 * (size_t)nP);
  return CRH_OK;
}

int crh_build_bvh_host(const float* pos, uint32_t nV, const int32_t* tri, uint32_t nT, int threads, float* nodes, uint32_t* n_nodes,
                       uint32_t* prim_order)
{
  if ((nT && (!pos || !tri)) || !n_nodes) return CRH_E_INVALID;
  for (uint32_t t = 0; t < nT; ++t) for (int k = 0; k < 3; ++k) if (tri[4 * t + k] < 0 || (uint32_t)tri[4 * t + k] >= nV) return CRH_E_INVALID;
  if (nV && !all_finite(pos, 3 * (size_t)nV, 1.0e30f)) return CRH_E_INVALID;
  QBvh b; build_qbvh(pos, tri, nT, b, threads);
  *n_nodes = (uint32_t)b.nodes.size();
  if (nodes) std::memcpy(nodes, b.nodes.data(), b.nodes.size() * sizeof(QNode));
  if (prim_order && nT) std::memcpy(prim_order, b.prim_order.data(), sizeof(uint32_t) * nT);
  return CRH_OK;
}

int crh_bench_trace(crh_ctx* c, const float* rays, uint32_t n, int any_hit, uint32_t repeat, float* avg_ms)
{
  if (!c || !rays || !avg_ms || !n || !repeat) return fail(c, CRH_E_INVALID, "bad bench arguments");
  if (!c->built) return fail(c, CRH_E_NOTBUILT, "crh_build has not been called");
  CRH_HIP(hipSetDevice(c->device));
  const size_t in_b = 32 * (size_t)n, out_b = 16 * (size_t)n;
  int rc = ensure_scratch(c, in_b + out_b); if (rc) return rc;
  char* base = (char*)c->d_scratch;
  CRH_HIP(hipMemcpyAsync(base, rays, in_b, hipMemcpyHostToDevice, cstream(c)));
  DScene S; fill_scene(c, S);
  Launch L{cstream(c), c->grid_trace, false, c->clamp_grid ? c->cus : 0};
  launch_trace_rays(L, S, (const float4*)base, n, any_hit, (float4*)(base + in_b), (uint32_t*)(base + in_b), c->d_api_cursor, c->d_counters);
  hipEvent_t e0 = get_event(c), e1 = get_event(c);
  CRH_HIP(hipEventRecord(e0, cstream(c)));
  for (uint32_t r = 0; r < repeat; ++r)
    launch_trace_rays(L, S, (const float4*)base, n, any_hit, (float4*)(base + in_b), (uint32_t*)(base + in_b), c->d_api_cursor, c->d_counters);
  CRH_HIP(hipEventRecord(e1, cstream(c)));
  CRH_HIP(hipStreamSynchronize(cstream(c)));
  float ms = 0.f; CRH_HIP(hipEventElapsedTime(&ms, e0, e1));
  c->ev_pool.push_back(e0); c->ev_pool.push_back(e1);
  *avg_ms = ms / (float)repeat;
  return CRH_OK;
}

int crh_debug_math(crh_ctx* c, int fn, const float* a, const float* b, float* out, float* out2, uint32_t n)
{
  if (!c || !a || !b || !out || !out2 || !n) return fail(c, CRH_E_INVALID, "bad debug_math arguments");
  CRH_HIP(hipSetDevice(c->device));
  const size_t bytes = sizeof(float) * (size_t)n;
  int rc = ensure_scratch(c, 4 * bytes); if (rc) return rc;
  float* d = (float*)c->d_scratch;
  CRH_HIP(hipMemcpyAsync(d, a, bytes, hipMemcpyHostToDevice, cstream(c)));
  CRH_HIP(hipMemcpyAsync(d + n, b, bytes, hipMemcpyHostToDevice, cstream(c)));
  CRH_HIP(hipMemsetAsync(d + 2 * (size_t)n, 0, 2 * bytes, cstream(c)));
  Launch L{cstream(c), c->grid, false};
  launch_debug_math(L, fn, d, d + n, d + 2 * (size_t)n, d + 3 * (size_t)n, n);
  CRH_HIP(hipGetLastError());
  CRH_HIP(hipMemcpyAsync(out, d + 2 * (size_t)n, bytes, hipMemcpyDeviceToHost, cstream(c)));
  CRH_HIP(hipMemcpyAsync(out2, d + 3 * (size_t)n, bytes, hipMemcpyDeviceToHost, cstream(c)));
  CRH_HIP(hipStreamSynchronize(cstream(c)));
  return CRH_OK;
}

int crh_debug_bsdf(crh_ctx* c, int fn, const crh_bsdf* m, const float* a, const float* b, float* out, uint32_t n, int two_sided)
{
  if (!c || !m || !a || !out || !n || fn < 0 || fn > 3 || (fn != 3 && !b)) return fail(c, CRH_E_INVALID, "bad debug_bsdf arguments");
  CRH_HIP(hipSetDevice(c->device));
  const size_t per_out = fn == 2 ? 8 : (fn == 1 ? 1 : 3);
  const size_t in_b = sizeof(float) * 3 * (size_t)n, out_b = sizeof(float) * per_out * (size_t)n;
  int rc = ensure_scratch(c, 256 + 2 * in_b + out_b); if (rc) return rc;
  char* base = (char*)c->d_scratch;
  float* d_a = (float*)(base + 256); float* d_b = (float*)(base + 256 + in_b); float* d_o = (float*)(base + 256 + 2 * in_b);
  CRH_HIP(hipMemcpyAsync(base, m, sizeof(crh_bsdf), hipMemcpyHostToDevice, cstream(c)));
  CRH_HIP(hipMemcpyAsync(d_a, a, in_b, hipMemcpyHostToDevice, cstream(c)));
  if (b) CRH_HIP(hipMemcpyAsync(d_b, b, in_b, hipMemcpyHostToDevice, cstream(c)));
  Launch L{cstream(c), c->grid, false};
  launch_debug_bsdf(L, fn, (const float4*)base, d_a, d_b, d_o, n, two_sided);
  CRH_HIP(hipGetLastError());
  CRH_HIP(hipMemcpyAsync(out, d_o, out_b, hipMemcpyDeviceToHost, cstream(c)));
  CRH_HIP(hipStreamSynchronize(cstream(c)));
  return CRH_OK;
}

int crh_enable_kernel_timing(crh_ctx* c, int on) { if (!c) return CRH_E_INVALID; c->timing_on = on != 0; return CRH_OK; }

int crh_get_kernel_timing(crh_ctx* c, double* trace_ms_total, uint64_t* trace_launches, double* all_ms_total)
{
  if (!c) return CRH_E_INVALID;
  CRH_HIP(hipSetDevice(c->device));
  CRH_HIP(hipStreamSynchronize(cstream(c)));
  drain_events(c);
  if (trace_ms_total) *trace_ms_total = c->trace_ms_acc;
  if (trace_launches) *trace_launches = c->trace_launches;
  if (all_ms_total) *all_ms_total = c->all_ms_acc;
  return CRH_OK;
}

}  // extern "C"

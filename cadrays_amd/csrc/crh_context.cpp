// crh_context.cpp -- context life cycle, device-memory / event helpers, accumulator, the CRH_* environment table
// (one of the translation units behind include/cadrays_hip.h; the context, the shared helpers and the map of the files: crh_context.h)
#include "crh_context.h"

using namespace crh;
using namespace crh::api;

namespace crh {
namespace api {

int fail(crh_ctx* c, int code, const char* msg) { if (c) c->err = msg; return code; }

// The device error word (d_api_cursor[8]): raised by a frame-kernel workgroup whose spin loop gave up (k_frame.h, kFrameSpinLimit) -- never in a healthy run.
// Read where the host has just synchronised anyway (crh_sync, crh_get_stats, the synchronous read-backs): one 4-byte copy.
int check_device_error(crh_ctx* c)
{
  uint32_t e = 0;
  if (!c->d_api_cursor || hipMemcpyAsync(&e, c->d_api_cursor + 8, sizeof e, hipMemcpyDeviceToHost, cstream(c)) != hipSuccess || hipStreamSynchronize(cstream(c)) != hipSuccess) return CRH_OK;
  if (e == 0u) return CRH_OK;
  hipMemsetAsync(c->d_api_cursor + 8, 0, sizeof e, cstream(c));
  char b[160]; snprintf(b, sizeof b, "frame kernel gave up: a workgroup's %s%s%sloop did not end (code %u); the frame is incomplete -- crh_reset and render again",
                        (e & 1u) ? "ring-take " : "", (e & 2u) ? "ring-push " : "", (e & 4u) ? "idle " : "", e);
  c->err = b;
  return CRH_E_DEVICE;
}

// The boundary takes finite numbers only (coordinates additionally |x| <= 1e30, so that box centres and extents stay finite):
// NaN / Inf would otherwise reach the BVH builder's binning and the kernels' float -> int conversions.
bool all_finite(const float* v, size_t n, float limit)
{
  for (size_t i = 0; i < n; ++i) if (!(v[i] >= -limit && v[i] <= limit)) return false;
  return true;
}

// Stream-ordered copy of a small host block: staged through one of four pinned buffers, so the call returns at once and `src`
// can be reused; a slot is waited for only when the copy issued four uploads earlier has not finished yet.
int stage_copy(crh_ctx* c, void* dst, const void* src, size_t bytes, hipStream_t on)
{
  if (!bytes) return CRH_OK;
  const hipStream_t stream = on ? on : cstream(c);
  BuiltScene::Stage& st = c->stage[c->stage_next++ & 3u];
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
  CRH_HIP(hipMemsetAsync(c->paths.rad, 0, 16 * (size_t)need, cstream(c)));      // no record carries a batch stamp yet (DPaths::stamp is never 0)
  if (!c->queues.counts) { CRH_HIP(hipMalloc((void**)&c->queues.counts, kCounts * sizeof(uint32_t))); CRH_HIP(hipMemsetAsync(c->queues.counts, 0, kCounts * sizeof(uint32_t), cstream(c))); }
  c->path_cap = need;
  return CRH_OK;
}

uint32_t next_stamp(crh_ctx* c) { if (++c->stamp_counter == 0u) ++c->stamp_counter; return c->stamp_counter; }

int ensure_scratch(crh_ctx* c, size_t bytes)
{
  if (bytes <= c->scratch_bytes) return CRH_OK;
  if (c->d_scratch) { CRH_HIP(hipFree(c->d_scratch)); c->d_scratch = nullptr; }
  CRH_HIP(hipMalloc(&c->d_scratch, bytes));
  c->scratch_bytes = bytes;
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
  // A restart touches the ACCUMULATOR, not what a frame's tracing reads: the frame pipeline (render_impl) lets the first frame of the new accumulation trace while
  // the last frames of the old one finish -- its accumulate waits for reset_ev, its counters go to the next block of the ring (zeroed one restart ago).
  const uint64_t uses = c->stream_uses;
  const hipStream_t cs = cstream(c);                     // joins the frames in flight: the memsets come after their accumulates and after a read-back's tone map
  CRH_HIP(hipMemsetAsync(c->d_accum, 0, sizeof(float4) * (size_t)c->par.width * c->par.height, cs));
  CRH_HIP(hipMemsetAsync(c->d_m2, 0, sizeof(float) * (size_t)c->par.width * c->par.height, cs));
  c->adaptive_picks = 0; c->pending_n = 0; c->ramp_k = 1; c->picked_valid = false; c->assembled_valid = false;
  {   // per-tile costs of the accumulation that ends here: to the host (pinned, asynchronous, behind the frames in flight), then zero for the next one
    crh_ctx::TileOrder& to = c->tile_order;
    // a host that restarts while frames are still running is dragging: it gets no new list (replacing the list waits for the frames that read it -- measured:
    // drag loop -5 ... -10 % when every restart may do that); `streak` = crh_render calls in a row that found nothing in flight
    if (to.on && to.d_cost && to.dirty && !to.pending && to.streak >= 2u) {
      if (!to.copied) CRH_HIP(hipEventCreateWithFlags(&to.copied, hipEventDisableTiming));
      CRH_HIP(hipMemcpyAsync(to.h_cost, to.d_cost, sizeof(uint32_t) * to.n, hipMemcpyDeviceToHost, cs));
      CRH_HIP(hipMemsetAsync(to.d_cost, 0, sizeof(uint32_t) * to.n, cs));
      CRH_HIP(hipEventRecord(to.copied, cs));
      to.pending = true; to.dirty = false;
    }
  }
  if (!c->reset_ev) CRH_HIP(hipEventCreateWithFlags(&c->reset_ev, hipEventDisableTiming));
  CRH_HIP(hipEventRecord(c->reset_ev, cs)); c->reset_pending = true;
  {
    const uint32_t now = ++c->counter_epoch & 3u, next = (c->counter_epoch + 1u) & 3u;
    c->d_counters = c->d_counters_ring + now;                                        // zeroed at the previous restart (counters_zeroed[now]) or at creation
    CRH_HIP(hipMemsetAsync(c->d_counters_ring + next, 0, sizeof(DCounters), cs));   // last used three restarts ago; its frames were joined above
    if (!c->counters_zeroed[next]) CRH_HIP(hipEventCreateWithFlags(&c->counters_zeroed[next], hipEventDisableTiming));
    CRH_HIP(hipEventRecord(c->counters_zeroed[next], cs));
  }
  c->stream_uses = uses;                                 // none of this is input of a frame's tracing
  discard_events(c);
  c->seconds_acc = c->trace_ms_acc = c->all_ms_acc = 0.0; c->trace_launches = 0; c->frames_done = 0;
  return CRH_OK;
}

// ---- every environment variable the library reads, in ONE table (crh_env_table() hands it to the host; tests/test_cpu_host.py checks that no other
// getenv("CRH_...") exists in the sources).  These are reference-schedule selectors for tests and diagnostics, never needed for correct output: images
// do not depend on any of them.  Tuning knobs whose A/B is settled (grid sizes, pipeline grid floors, lane sizes) are gone -- their values are constants
// with their measurements beside them in crh_context.h.
struct EnvKnob { const char* name; const char* what; };
static const EnvKnob kEnvKnobs[] = {
  {"CRH_BUILD_THREADS",      "threads of the host BVH builder and the record fill (default: the CPUs this process may use; bench.py gives each of N ranks its share)"},
  {"CRH_BUILD_VERBOSE",      "crh_build prints its phases and their times on stderr"},
  {"CRH_MAX_PATHS",          "path slots per batch, 1024 .. 2^30 (default 2^29); the same knob as crh_set_path_budget"},
  {"CRH_DONATE",             "0: small batches use the plain traversal kernels instead of the work-donating ones (reference schedule of the sequence tests)"},
  {"CRH_PACKETS",            "smallest run of consecutive samples per pixel from which the camera rays of a wide batch are walked as wavefront packets (k_trace_packets); default 64 = one pixel per wavefront, 0 = never"},
  {"CRH_PIPELINE",           "0: free-running Redraw()s are not pipelined across streams (reference schedule of the sequence tests)"},
  {"CRH_PIPE_DEPTH",         "frames in flight of free-running Redraw()s, 2 .. 8; the same knob as crh_set_pipeline_depth (crh_query_pipeline_capacity says what the process supports)"},
  {"CRH_FRAME_KERNEL",       "0: small batches take the staged small-batch schedule (one launch per stage and bounce) instead of the frame kernel (reference schedule of the sequence tests)"},
  {"CRH_FRAME_LIVE",         "frame kernel: paths a workgroup (16 wavefronts, one per compute unit) keeps alive at most, 64 .. 16384, cut to the ring size 4096 (default 4096)"},
  {"CRH_FRAME_CHUNK",        "frame kernel: path slots a wavefront claims at a time, multiples of 64 up to 1024 (default 128)"},
  {"CRH_FRAME_LOW",          "frame kernel: a feeder wavefront claims the next chunk once fewer rays than this wait in the workgroup's ring (default 512)"},
  {"CRH_FRAME_FEED",         "frame kernel: wavefronts of a workgroup that only shade and generate, 0 .. 15 (default: 3 or 4 of 16, chosen by measurement after every crh_build -- 4 pays on scenes with short walks: profiles/r6/lone_frame.md)"},
  {"CRH_FRAME_STARVE",       "frame kernel: a feeder shades fewer than 64 waiting hits only while fewer rays than this wait in the ring (default 2^20: always)"},
  {"CRH_FRAME_STEP",         "frame kernel: tracer wavefront w takes rays only while w x this many wait in the ring (default 0: every tracer takes what is there)"},
  {"CRH_FRAME_HELP",         "frame kernel: a tracer wavefront shades a batch itself once this many hit records wait in the workgroup's rings, 64 .. 4096 (default 256)"},
  {"CRH_FRAME_HELP_LOW",     "frame kernel: a tracer wavefront prefers a full shading batch (64 waiting hits) to tracing while fewer rays than this wait in the ring, 0 .. 4096 (default 0: it traces whatever is there)"},
  {"CRH_TILE_ORDER",         "0: crh_render lists the tiles row-major as up to round 5 (default 1: a host whose frames start on an idle chip gets the tiles that cost the most rays in the last accumulation first -- lone frame -6 % on a model in front of a background; no pixel depends on the order)"},
  {"CRH_FRAME_GRID",         "frame kernel: workgroups of a lone frame (default: what is resident = ONE 1024-thread workgroup per compute unit); never fewer than min(resident, 32), never more than resident"},
  {"CRH_FRAME_PIPE",         "frame pipeline: a frame takes the frame kernel while fewer than this many frames are running, and at most this many frame kernels run at a time, 1 .. 8 (default 2)"},
  {"CRH_LANES",              "tile ranges a small batch is cut into, 1 .. 8 (default 2); 1 = one stream (reference schedule of the sequence tests)"},
  {"CRH_SPLIT_PASSES",       "split scenes (static tree + moved objects): 0 one walk in the two-level kernels, 1 two traversal passes, unset: by the number of moved objects"},
  {"CRH_REDUCE_RCCL_SINGLE", "crh_reduce sends even a one-context group through RCCL (exercises the library binding on a 1-GPU box)"},
};

// Hardware queues the HIP runtime of this process maps streams onto: GPU_MAX_HW_QUEUES as it stands at this library's FIRST use (crh_create or
// crh_query_pipeline_capacity, whichever comes first), default 4.  The runtime reads the variable at its own initialisation -- the process's first HIP
// call, which a host without another HIP user makes through crh_create -- so a host exports it before that; a value exported later changes nothing in
// the runtime and is not seen here either.  A pipelined frame needs a queue of its own beside the context's stream and the read-back stream.
int hw_queues()
{
  static const int n = [] { const char* e = getenv("GPU_MAX_HW_QUEUES"); const int v = e ? atoi(e) : 4; return v > 0 ? v : 4; }();
  return n;
}
uint32_t pipeline_capacity() { return (uint32_t)std::min(8, std::max(3, hw_queues() - 2)); }

int build_threads_env() { if (const char* e = getenv("CRH_BUILD_THREADS")) return atoi(e); return 0; }

static void read_env(crh_ctx* c)
{
  if (const char* e = getenv("CRH_MAX_PATHS")) { long v = atol(e); if (v >= 1024) c->max_paths = (uint32_t)std::min<long>(v, 1l << 30); }   // a path slot travels in 31 bits
  if (const char* e = getenv("CRH_DONATE")) c->donate = atoi(e) != 0;
  if (const char* e = getenv("CRH_PACKETS")) c->packets = atoi(e);
  if (const char* e = getenv("CRH_SPLIT_PASSES")) c->split_passes = atoi(e);
  if (const char* e = getenv("CRH_PIPELINE")) c->pipeline = atoi(e) != 0;
  if (const char* e = getenv("CRH_PIPE_DEPTH")) { int v = atoi(e); if (v >= 2 && v <= (int)pipeline_capacity()) c->pipe_depth = (uint32_t)v; }
  if (const char* e = getenv("CRH_FRAME_KERNEL")) c->frame_kernel = c->auto_frame_kernel = atoi(e) != 0;
  if (const char* e = getenv("CRH_FRAME_LIVE")) { int v = atoi(e); if (v >= 64 && v <= 16384) c->frame_live = (uint32_t)v; }
  if (const char* e = getenv("CRH_FRAME_CHUNK")) { int v = atoi(e); if (v >= 64 && v <= 1024) c->frame_chunk = (uint32_t)v & ~63u; }
  if (const char* e = getenv("CRH_FRAME_LOW")) { int v = atoi(e); if (v >= 0 && v <= 16384) c->frame_low_water = (uint32_t)v; }
  if (const char* e = getenv("CRH_FRAME_FEED")) { int v = atoi(e); if (v >= 0 && v <= 15) { c->frame_feeders = (uint32_t)v; c->feed_tune.on = false; } }      // fixed: no tuning
  if (const char* e = getenv("CRH_FRAME_STARVE")) { int v = atoi(e); if (v >= 0) c->frame_starve = (uint32_t)v; }
  if (const char* e = getenv("CRH_FRAME_STEP")) { int v = atoi(e); if (v >= 0 && v <= 256) c->frame_claim_step = (uint32_t)v; }
  if (const char* e = getenv("CRH_FRAME_HELP")) { int v = atoi(e); if (v >= 64 && v <= 4096) c->frame_help = (uint32_t)v; }
  if (const char* e = getenv("CRH_TILE_ORDER")) c->tile_order.on = atoi(e) != 0;
  if (const char* e = getenv("CRH_FRAME_HELP_LOW")) { int v = atoi(e); if (v >= 0 && v <= 4096) c->frame_help_low = (uint32_t)v; }
  if (const char* e = getenv("CRH_FRAME_GRID")) { int v = atoi(e); if (v >= 1) c->frame_grid = v; }
  if (const char* e = getenv("CRH_FRAME_PIPE")) { int v = atoi(e); if (v >= 1 && v <= 8) c->frame_pipe_depth = (uint32_t)v; }
  if (const char* e = getenv("CRH_LANES")) { int v = atoi(e); if (v >= 1 && v <= 8) c->n_lanes = (uint32_t)v; }
}

}  // namespace api
}  // namespace crh

extern "C" {

crh_ctx* crh_create(int device_ordinal)
{
  (void)api::hw_queues();      // what the HIP runtime is about to read, if this is the process's first HIP call
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || device_ordinal < 0 || device_ordinal >= n) {
    fprintf(stderr, "crh_create: no HIP device %d (device count %d) -- this backend has no CPU fallback\n", device_ordinal, n);
    return nullptr;
  }
  crh_ctx* c = new crh_ctx();
  c->device = device_ordinal;
  if (hipSetDevice(device_ordinal) != hipSuccess || hipStreamCreateWithFlags(&c->stream_, hipStreamNonBlocking) != hipSuccess ||
      hipMalloc((void**)&c->d_counters_ring, 4 * sizeof(DCounters)) != hipSuccess || hipMalloc((void**)&c->d_api_cursor, 64) != hipSuccess || hipMemsetAsync(c->d_counters_ring, 0, 4 * sizeof(DCounters), cstream(c)) != hipSuccess || hipMemsetAsync(c->d_api_cursor, 0, 64, cstream(c)) != hipSuccess ||
      hipStreamSynchronize(cstream(c)) != hipSuccess) {
    fprintf(stderr, "crh_create: HIP initialisation failed: %s\n", hipGetErrorString(hipGetLastError()));
    delete c; return nullptr;
  }
  c->d_counters = c->d_counters_ring;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device_ordinal) == hipSuccess && prop.multiProcessorCount > 0) { c->cus = prop.multiProcessorCount; c->grid = prop.multiProcessorCount * 4; c->grid_trace = prop.multiProcessorCount * 6; }
  api::read_env(c);
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
  for (auto& q : c->feed_tune.pend) { c->ev_pool.push_back(q.e0); c->ev_pool.push_back(q.e1); }
  for (auto& q : c->tile_order.pend) { hipEventDestroy(q.e0); hipEventDestroy(q.e1); }
  c->tile_order.pend.clear();
  if (c->pipe_resized) hipEventDestroy(c->pipe_resized);
  if (c->d_tile_ids2) hipFree(c->d_tile_ids2);
  if (c->tile_order.d_cost) hipFree(c->tile_order.d_cost);
  if (c->tile_order.h_cost) hipHostFree(c->tile_order.h_cost);
  if (c->tile_order.copied) hipEventDestroy(c->tile_order.copied);
  c->feed_tune.pend.clear();
  for (hipEvent_t e : c->ev_pool) hipEventDestroy(e);
  void* ptrs[] = {c->d_nodes, c->d_pnodes, c->d_tris, c->d_shade, c->d_mats, c->d_lights, c->d_env, c->d_accum, c->paths.ray_o[0], c->paths.ray_d[0], c->paths.ray_o[1], c->paths.ray_d[1], c->paths.thr[1],
                  c->paths.hit, c->paths.thr[0], c->paths.rad, c->paths.sh_o, c->paths.sh_d, c->paths.sh_c,
                  c->queues.q[0], c->queues.q[1], c->queues.q_sh, c->queues.counts, c->d_tile_ids, c->d_seeds, c->d_counters_ring, c->d_api_cursor, c->d_scratch,
                  c->d_m2, c->d_tile_err, c->d_tile_cnt, c->d_uvs, c->d_texels, c->d_tex_desc, c->d_inst, c->d_patch, c->d_verts, c->d_ibox, c->queues.q2, c->queues.q2_sh};
  for (void* p : ptrs) if (p) hipFree(p);
  if (c->d_assembled) hipFree(c->d_assembled);
  if (c->d_peer_stage) hipFree(c->d_peer_stage);
  for (BuiltScene::Stage& st : c->stage) { if (st.p) hipHostFree(st.p); if (st.ev) hipEventDestroy(st.ev); }
  for (int k = 0; k < 8; ++k) { if (c->lane_stream[k]) { hipStreamSynchronize(c->lane_stream[k]); hipStreamDestroy(c->lane_stream[k]); } if (c->lane_join[k]) hipEventDestroy(c->lane_join[k]); }
  if (c->lane_fork) hipEventDestroy(c->lane_fork);
  if (c->reset_ev) hipEventDestroy(c->reset_ev);
  for (hipEvent_t e : c->counters_zeroed) if (e) hipEventDestroy(e);
  if (c->d_lane_counts) hipFree(c->d_lane_counts);
  if (c->rb_stream) { hipStreamSynchronize(c->rb_stream); hipStreamDestroy(c->rb_stream); }
  for (int k = 0; k < 2; ++k) { if (c->d_rb[k]) hipFree(c->d_rb[k]); if (c->h_rb[k]) hipHostFree(c->h_rb[k]); if (c->rb_tm[k]) hipEventDestroy(c->rb_tm[k]); if (c->rb_done[k]) hipEventDestroy(c->rb_done[k]); }
  if (c->rb_fork) hipEventDestroy(c->rb_fork);
  for (void* q : {(void*)c->d_tile_cdf, (void*)c->d_picked, (void*)c->d_adapt_n}) if (q) hipFree(q);
  release_comms(c);
  hipStreamDestroy(c->stream_);
  delete c;
}

const char* crh_last_error(crh_ctx* c) { return c ? c->err.c_str() : "null context"; }

int crh_reset(crh_ctx* c) { if (!c) return CRH_E_INVALID; return do_reset(c); }

int crh_sync(crh_ctx* c) { if (!c) return CRH_E_INVALID; c->read_since_render = true; CRH_HIP(hipSetDevice(c->device)); CRH_HIP(hipStreamSynchronize(cstream(c))); return check_device_error(c); }

int crh_get_frame_tuning(crh_ctx* c, uint32_t out[5])
{
  if (!c || !out) return fail(c, CRH_E_INVALID, "null argument");
  const crh_ctx::FeedTune& ft = c->feed_tune;
  out[0] = ft.on ? 1u : 0u; out[1] = ft.on ? ft.chosen : c->frame_feeders; out[2] = ft.n[0] + ft.n[1];
  out[3] = ft.n[0] ? (uint32_t)(1.0e3 * ft.ms[0] / ft.n[0]) : 0u; out[4] = ft.n[1] ? (uint32_t)(1.0e3 * ft.ms[1] / ft.n[1]) : 0u;
  return CRH_OK;
}

int crh_get_tile_order(crh_ctx* c, uint32_t* order, uint32_t* n_tiles, uint64_t counts[7])
{
  if (!c) return CRH_E_INVALID;
  const uint32_t ts = c->par.tile_size;
  const uint32_t nt = ts ? ((c->par.width + ts - 1) / ts) * ((c->par.height + ts - 1) / ts) : 0u;
  const bool have = c->tile_order.on && c->tile_order.order.size() == nt;
  if (order) for (uint32_t t = 0; t < nt; ++t) order[t] = have ? c->tile_order.order[t] : t;
  if (n_tiles) *n_tiles = nt;
  if (counts) { counts[0] = c->tile_order.reorders; counts[1] = c->tile_order.calls_sorted; counts[2] = c->tile_order.calls_row_major; counts[3] = c->tile_order.frames_collected; counts[4] = c->tile_order.on ? c->tile_order.verdict : 2u; counts[5] = c->tile_order.tn[0] ? (uint64_t)(1.0e3 * c->tile_order.tms[0] / c->tile_order.tn[0]) : 0; counts[6] = c->tile_order.tn[1] ? (uint64_t)(1.0e3 * c->tile_order.tms[1] / c->tile_order.tn[1]) : 0; }
  return CRH_OK;
}

int crh_get_path_budget(crh_ctx* c, uint64_t* max_paths)
{
  if (!c || !max_paths) return fail(c, CRH_E_INVALID, "null argument");
  *max_paths = c->max_paths;
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

int crh_query_pipeline_capacity(uint32_t* max_frames, int* hw_queues)
{
  if (max_frames) *max_frames = api::pipeline_capacity();
  if (hw_queues) *hw_queues = api::hw_queues();
  return CRH_OK;
}

const char* crh_env_table(void)
{
  static std::string t;
  if (t.empty()) for (const api::EnvKnob& k : api::kEnvKnobs) { t += k.name; t += "\t"; t += k.what; t += "\n"; }
  return t.c_str();
}

}  // extern "C"

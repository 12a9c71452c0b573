// crh_schedule.cpp -- the per-iteration wavefront schedule: lanes, batches, frame pipelining, look-ahead, adaptive iterations
// (one of the translation units behind include/cadrays_hip.h; the context, the shared helpers and the map of the files: crh_context.h)
#include "crh_context.h"

using namespace crh;
using namespace crh::api;

namespace crh {
namespace api {

uint32_t frame_seed(uint32_t seed, uint32_t n)   // Bullard generator, SURVEY.md a14
{
  uint32_t hi = seed, lo = seed ^ 0x49616E42u, r = 0;
  for (uint32_t i = 0; i <= n; ++i) { hi = (hi >> 2) + (hi << 2); hi += lo; lo += hi; r = hi; }
  return r >> 2;
}

}  // namespace api
}  // namespace crh

namespace {

// One batch: `ns` samples of `nt` tiles whose ids sit at d_tiles; seeds at d_seeds.
// One wavefront schedule: `ns` samples of `nt` tiles (ids at d_tiles, seeds at d_seeds) on one stream and one slice of the path state.
struct Lane { hipStream_t stream; DPaths P; DQueues Q; int grid, grid_trace; bool timed; const uint32_t* n_tiles_dev = nullptr; bool donate = false;
              bool frame = false; int grid_frame = 0;         // frame: the whole batch in one launch of the frame kernel (k_frame.h) with grid_frame workgroups
              const uint32_t* h_seeds = nullptr;              // ... whose <= 16 frame seeds travel by value (d_seeds == nullptr)
              hipEvent_t tune_e0 = nullptr, tune_e1 = nullptr;      // ... bracketed by these two events when the feeder count is being tuned (crh_context.h FeedTune)
              uint32_t* tile_cost = nullptr; };                     // ... and whose accumulate adds the rays its paths traced to the per-tile sums (crh_context.h TileOrder)

// May this batch take the frame kernel?  Small, not counted, not timed per kernel, slots that fit the bits the kernel keeps them in.
bool frame_ok(const crh_ctx* c, uint64_t total)
{ return c->frame_kernel && !c->counters_on && !c->timing_on && total <= c->lane_max_paths && total < (uint64_t)kFrameMaxPaths && frame_launchable(!c->inst.empty()); }
int frame_grid(const crh_ctx* c, const DScene& S, uint32_t share = 1u)
{
  const int res = frame_resident_grid(c->clamp_grid ? c->cus : 0, S.two_level != 0);
  const int want = c->frame_grid > 0 ? std::min(c->frame_grid, res) : res;
  return std::max(std::min(res, 32), want / (int)std::max(1u, share));
}

int run_lane(crh_ctx* c, const Lane& ln, const DScene& S, const uint32_t* d_tiles, uint32_t nt, const uint32_t* d_seeds, uint32_t ns, int seed_per_tile,
             bool accumulate, hipEvent_t before_accumulate = nullptr, hipEvent_t before_accumulate2 = nullptr, hipEvent_t before_accumulate3 = nullptr)
{
  Launch L{ln.stream, ln.grid, c->counters_on};
  Launch LT{ln.stream, ln.grid_trace, c->counters_on, c->clamp_grid ? c->cus : 0, ln.donate && !c->counters_on};
  // wide batches (a wavefront = >= 16 consecutive samples of one pixel): the camera rays are walked as packets (kernels.hip, k_trace_packets)
  // (small batches -- a wavefront = an 8 x 8 pixel block -- lose with packets: 468 -> 391 Redraw/s, profiles/r4/ab_camera_ray_packets.txt)
  LT.packets = c->packets > 0 && !c->counters_on && !ln.donate && std::min<uint32_t>(ns & (0u - ns), 64u) >= (uint32_t)c->packets;

  if (ln.frame) {
    // the frame kernel: camera rays, every bounce's traversal, shading and shadow rays of this batch in ONE launch (k_frame.h); its two control words live behind
    // the queue counters of this lane (zero when allocated, left zero by every launch)
    Launch LF{ln.stream, ln.grid_frame, false, c->clamp_grid ? c->cus : 0};
    if (ln.tune_e0) hipEventRecord(ln.tune_e0, ln.stream);
    launch_frame(LF, S, ln.P, ln.Q.counts + 12, d_tiles, nt, d_seeds, ns, seed_per_tile, ln.n_tiles_dev, c->frame_live, c->frame_chunk, c->frame_low_water, c->frame_feeders, c->frame_claim_step, c->frame_starve, c->d_counters, ln.h_seeds, c->d_api_cursor + 8, c->frame_help | (c->frame_help_low << 16));      // + 8: the context's device error word (crh_context.cpp check_device_error)
    if (ln.tune_e1) hipEventRecord(ln.tune_e1, ln.stream);
  } else {
  launch_raygen(L, S, ln.P, ln.Q, 0, d_tiles, nt, d_seeds, ns, seed_per_tile, ln.n_tiles_dev, ln.h_seeds);
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
  }
  if (accumulate && before_accumulate) CRH_HIP(hipStreamWaitEvent(ln.stream, before_accumulate, 0));      // samples are folded in in frame order
  if (accumulate && before_accumulate2) CRH_HIP(hipStreamWaitEvent(ln.stream, before_accumulate2, 0));    // ... and not while a read-back tone-maps the accumulator
  if (accumulate && before_accumulate3) CRH_HIP(hipStreamWaitEvent(ln.stream, before_accumulate3, 0));    // ... and after a restart has zeroed it
  if (accumulate) launch_accumulate(L, S, ln.P, c->d_accum, c->adaptive ? c->d_m2 : nullptr, d_tiles, nt, 0, ns, ns, c->d_counters, ln.n_tiles_dev, ln.frame && ns == 1u ? ln.tile_cost : nullptr);
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
  CRH_HIP(hipMalloc((void**)&c->d_lane_counts, kCounts * sizeof(uint32_t) * 8));
  CRH_HIP(hipMemsetAsync(c->d_lane_counts, 0, kCounts * sizeof(uint32_t) * 8, cstream(c)));
  return CRH_OK;
}

// One batch: `ns` samples of `nt` tiles whose ids sit at d_tiles; seeds at d_seeds.
int run_batch(crh_ctx* c, const DScene& S, const uint32_t* d_tiles, uint32_t nt, const uint32_t* d_seeds, uint32_t ns, int seed_per_tile = 0,
              bool accumulate = true)
{
  c->pending_n = 0;                                  // the path buffer is about to be overwritten
  c->paths.stamp = next_stamp(c);                    // lanes copy it; a look-ahead batch is folded in later by launch_accumulate(c->paths, ...)
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
  const bool frame = frame_ok(c, total);
  if (frame || K < 2 || total > c->lane_max_paths || !accumulate || c->counters_on) {
    const bool small = total <= c->lane_max_paths && !c->counters_on;
    Lane one{cstream(c), c->paths, c->queues, small ? small_grid(total, c->grid, 2048) : c->grid, small ? small_grid(total, c->grid_trace, 2048) : c->grid_trace, true};
    one.donate = small && c->donate;
    one.frame = frame; one.grid_frame = frame ? frame_grid(c, S) : 0;
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
    ln.grid = small_grid(total / K, c->grid, 2048);
    ln.grid_trace = small_grid(total / K, c->grid_trace, 2048);
    const DPaths& P = c->paths; const DQueues& Q = c->queues;
    ln.P.ray_o[0] = P.ray_o[0] + base; ln.P.ray_o[1] = P.ray_o[1] + base; ln.P.ray_d[0] = P.ray_d[0] + base; ln.P.ray_d[1] = P.ray_d[1] + base;
    ln.P.thr[0] = P.thr[0] + base; ln.P.thr[1] = P.thr[1] + base; ln.P.hit = P.hit + base; ln.P.rad = P.rad + base;
    ln.P.sh_o = P.sh_o + base; ln.P.sh_d = P.sh_d + base; ln.P.sh_c = P.sh_c + base; ln.P.stamp = P.stamp;
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
  // Round 6: TWO lists stay on the device -- crh_render hands over the image's tiles either row-major or most-expensive-first (crh_context.h TileOrder), by what
  // is in flight; switching between two resident lists costs nothing, and a frame in flight keeps reading the one it was launched with.  A third list replaces the
  // one the previous call did not use (through the context's stream: behind every frame that still reads it).
  uint32_t* d_list = nullptr;
  if (c->h_tile_ids.size() == nt && std::memcmp(c->h_tile_ids.data(), tiles, sizeof(uint32_t) * nt) == 0) { d_list = c->d_tile_ids; c->tile_list_last = 0; }
  else if (c->h_tile_ids2.size() == nt && std::memcmp(c->h_tile_ids2.data(), tiles, sizeof(uint32_t) * nt) == 0) { d_list = c->d_tile_ids2; c->tile_list_last = 1; }
  else if (c->tile_list_last == 0 && !c->h_tile_ids.empty()) {      // the first list is the one in use: the new one goes to the second
    if (nt > c->tile_cap2) { CRH_HIP(hipStreamSynchronize(cstream(c))); if (c->d_tile_ids2) CRH_HIP(hipFree(c->d_tile_ids2)); CRH_HIP(hipMalloc((void**)&c->d_tile_ids2, sizeof(uint32_t) * nt)); c->tile_cap2 = nt; }
    int rc_u = stage_copy(c, c->d_tile_ids2, tiles, sizeof(uint32_t) * nt); if (rc_u) return rc_u;
    c->h_tile_ids2.assign(tiles, tiles + nt); d_list = c->d_tile_ids2; c->tile_list_last = 1;
  } else {
    int rc_u = stage_copy(c, c->d_tile_ids, tiles, sizeof(uint32_t) * nt); if (rc_u) return rc_u;
    c->h_tile_ids.assign(tiles, tiles + nt); d_list = c->d_tile_ids; c->tile_list_last = 0;
  }
  // path budget of this call: crh_set_path_budget's, cut to what the device has free when the state would have to grow (196 B per slot; a shared or smaller
  // device gets the narrower batches the budget table of cadrays_hip.h prices at a few percent, not a hipMalloc error -- ADVICE r4)
  uint32_t budget = c->max_paths;
  const uint64_t slots_wanted = (uint64_t)nt * tpp * ns * ((c->pipeline && ns <= 16u && (uint64_t)nt * tpp * ns <= c->lane_max_paths) ? c->pipe_depth : 1u);      // a pipelined frame holds pipe_depth slices
  if (slots_wanted > c->path_cap) {
    size_t fr = 0, tot = 0;
    if (hipMemGetInfo(&fr, &tot) == hipSuccess) {
      const uint64_t fit = (uint64_t)((double)(fr + (size_t)c->path_cap * 196u) * 0.85) / 196u;
      if (fit < budget) budget = (uint32_t)std::max<uint64_t>(fit, std::min<uint64_t>(budget, 1u << 20));
    }
  }
  const uint32_t cap_tiles = std::max<uint32_t>(1u, budget / tpp);
  uint32_t group = std::min(nt, cap_tiles);
  uint32_t spb = std::max<uint32_t>(1u, std::min(ns, budget / (group * tpp)));
  // A wide request that does not fit one batch is cut into TILE groups first and sample batches second: as many of the call's samples per batch as fit
  // (multiples of 64, so that a wavefront is 64 samples of one pixel and the camera rays walk as packets; at most kBatchSamples), the tiles per batch follow
  // (at least kBatchMinTiles).  The more samples of a pixel travel together, the more of their later bounces start from the same few triangles and the
  // smaller the part of the tree a batch touches: on the 10 M-triangle tree at 4K 32 / 64 / 128 / 256 samples per batch give 2956 / 3301 / 3382 / 3410
  // Mrays/s, on the 1 M-triangle one at 1080p 128 / 256 / 512 / 1024 give 4089 / 4145 / 4204 / 4224 (profiles/r4/ab_batch_shape.txt).  Samples of a pixel
  // are folded in in the same order whatever the cut, so the frame does not change by a bit.
  constexpr uint32_t kBatchSamples = 1024u, kBatchMinTiles = 256u;
  if (c->packets > 0 && c->packets <= 64 && !c->counters_on && ns >= 64u && (uint64_t)nt * tpp * ns > c->lane_max_paths) {
    uint32_t want = std::min<uint32_t>(ns & ~63u, kBatchSamples);
    const uint32_t min_group = std::min<uint32_t>(nt, kBatchMinTiles);
    while (want > 64u && budget / (want * tpp) < min_group) want = (want / 2u) & ~63u;
    if (want >= 64u && budget / (want * tpp) >= 1u) {
      group = std::min(nt, budget / (want * tpp));
      spb = want;
      if (group == nt) spb = std::max<uint32_t>(want, std::min(ns, budget / (group * tpp)) & ~63u);
    }
  }
  const uint64_t total = (uint64_t)nt * tpp * ns;
  // Measured on C3 at 1080p, 1 spp per call: free-running 232 -> 326 Redraw/s (C2: 323 -> 442); a host that reads every frame back
  // would get 187 instead of 225 (one schedule per frame is slower than two tile ranges when nothing overlaps it), so a
  // read-back / synchronisation since the last render selects the two-range schedule for this frame.
  const bool host_runs_ahead = !c->read_since_render;
  // Small whole-frame batches -- one Redraw() each -- are pipelined: frame n + 1 starts tracing on another stream and another slice of the path state while frame n
  // finishes; accumulation stays in frame order.  Two kinds of frame travel through the same pipeline:
  //   * the FRAME KERNEL (k_frame.h, one launch): the fastest way through a chip that is empty or nearly so -- a lone frame (2.98 ms against 4.26 staged on C3),
  //     and the frames of a host that shows every frame or restarts with every frame (AppViewer.cxx:979-984: the camera drag);
  //   * the STAGED form (a launch per stage and bounce; the stages of eight frames in flight overlap one another): the higher throughput once the host runs far
  //     ahead -- 467-480 Redraw/s against 406-426 with frame kernels only.
  // A frame takes the frame kernel when fewer than frame_pipe_depth frames are still running at its submission.  Neither kind waits for the context's stream
  // unless something was enqueued there that its tracing reads (seeds travel by value, a restart's memsets gate the accumulate only).
  const bool frame_able = frame_ok(c, total);
  if (c->pipeline && (host_runs_ahead || frame_able) && !c->counters_on && !c->timing_on && !c->adaptive && group == nt && spb == ns && ns <= 16u && total >= (1u << 20) &&
      total <= c->lane_max_paths && (uint64_t)c->pipe_depth * total <= budget) {      // `budget`: the device's free memory counts here too
    const uint32_t depth = c->pipe_depth;               // streams / path-state slices the frames rotate through
    int rc = ensure_paths(c, (uint32_t)(c->pipe_depth * total)); if (rc) return rc;
    rc = ensure_lanes(c); if (rc) return rc;
    const uint32_t k = c->pipe_seq % depth, prev = (c->pipe_seq + depth - 1u) % depth;      // this frame's stream, the previous frame's
    ++c->pipe_seq;
    const hipStream_t cs = c->stream_;                 // raw: no join
    uint32_t seeds[16];
    {
      uint32_t hi = c->par.seed, lo = c->par.seed ^ 0x49616E42u;
      for (uint32_t i = 0; i < first + ns; ++i) { hi = (hi >> 2) + (hi << 2); hi += lo; lo += hi; if (i >= first) seeds[i - first] = hi >> 2; }
    }
    DScene S; fill_scene(c, S);
    // What this frame's TRACING must wait for: whatever was enqueued on the context's stream since this pipeline stream last forked from it -- scene uploads, a new
    // tile list, a read-back's kernels.  A restart's memsets are not among them (do_reset): the first frame after crh_reset starts at once, only its accumulate
    // waits (reset_ev below).  Per pipeline stream: the frame after the one that forked runs on ANOTHER stream and must see the same uploads.
    const bool fork = c->stream_uses != c->lane_stream_uses[k];
    if (fork) { CRH_HIP(hipEventRecord(c->lane_fork, cs)); c->lane_stream_uses[k] = c->stream_uses; }
    // frames still running right now (this stream's own last frame included: this one queues behind it)
    uint32_t running = 0u;
    for (uint32_t j = 0; j < 8u; ++j) if (c->pipe_running[j]) { if (hipEventQuery(c->lane_join[j]) != hipErrorNotReady) c->pipe_running[j] = false; else ++running; }
    c->last_running = running;
    const bool frame = frame_able && (running < c->frame_pipe_depth || !host_runs_ahead);
    Lane ln; ln.stream = c->lane_stream[k]; ln.timed = false; ln.donate = c->donate; ln.h_seeds = seeds;
    ln.grid = (int)std::min<uint64_t>((uint64_t)c->grid, std::max<uint64_t>((uint64_t)c->pipe_grid_min_shade, total / 2048u));
    // staged traversal grid of this frame: the chip's resident workgroups (6 per CU) shared among the frames that are in flight RIGHT NOW -- a host that runs far
    // ahead has pipe_depth of them (eight: 192 workgroups each), one that waits for every other frame's read-back has two or three (512 each); measured
    // optima at 3 / 4 / 6 / 8 frames in flight: 512 / 384 / 256 / 192-256 (profiles/r3/interactive_counters.txt)
    uint32_t in_flight = 1u + running - ((c->pipe_running[k]) ? 1u : 0u);
    {
      // a host that submitted the previous frame a moment ago is not waiting for anything: the pipeline is about to fill (counting what is in flight NOW
      // would give the first frames of a burst grids for a nearly empty chip: eight of them, 2900 workgroups)
      const auto now = std::chrono::steady_clock::now();
      if (!frame && c->pipe_last_submit.time_since_epoch().count() != 0 && now - c->pipe_last_submit < std::chrono::microseconds(300)) in_flight = std::max(in_flight, depth);
      c->pipe_last_submit = now;
    }
    const uint64_t share = std::min<uint64_t>(512u, std::max<uint64_t>((uint64_t)c->pipe_grid_min, (uint64_t)c->grid_trace / in_flight));
    ln.grid_trace = (int)std::min<uint64_t>((uint64_t)c->grid_trace, std::max<uint64_t>(share, total / (uint64_t)c->pipe_div));
    // frame kernel: one workgroup fills a compute unit, so the frames in flight share the chip by compute units (every frame asking for all of them was
    // measured: the second frame's workgroups then wait for whole workgroups of the first to leave -- 368 against 399 Redraw/s)
    ln.frame = frame; ln.grid_frame = frame ? frame_grid(c, S, std::min(in_flight, std::max(1u, c->frame_pipe_depth))) : 0;
    if (frame && c->tile_order.on && ns == 1u && true) {
      // per-tile ray sums of this accumulation (crh_context.h TileOrder), indexed by tile id: allocated for the image's tiles
      crh_ctx::TileOrder& to = c->tile_order;
      const uint32_t ts_ = c->par.tile_size, all_tiles = ((c->par.width + ts_ - 1u) / ts_) * ((c->par.height + ts_ - 1u) / ts_);
      if (to.n != all_tiles) {
        CRH_HIP(hipStreamSynchronize(cstream(c)));
        if (to.d_cost) CRH_HIP(hipFree(to.d_cost)); if (to.h_cost) CRH_HIP(hipHostFree(to.h_cost));
        to.d_cost = nullptr; to.h_cost = nullptr;
        CRH_HIP(hipMalloc((void**)&to.d_cost, sizeof(uint32_t) * all_tiles)); CRH_HIP(hipMemsetAsync(to.d_cost, 0, sizeof(uint32_t) * all_tiles, cstream(c)));      // (never the NULL stream: once it exists every launch on the context's streams pays for it -- the drag loop lost 2.3 % to one hipMemset here)
        CRH_HIP(hipHostMalloc((void**)&to.h_cost, sizeof(uint32_t) * all_tiles, hipHostMallocDefault));
        to.n = all_tiles; to.pending = false; to.dirty = false; to.cls.clear(); to.order.clear();
      }
      if (to.streak >= 2u && to.verdict != 2u) { ln.tile_cost = to.d_cost; to.dirty = true; ++to.frames_collected; }      // (a host that keeps frames in flight gets no new list anyway: nothing is collected for it)
    }
    if (frame && c->feed_tune.on) {
      // the feeder count by measurement (crh_context.h FeedTune): collect the frame kernels that have finished, then either take the next measurement or the verdict
      crh_ctx::FeedTune& ft = c->feed_tune;
      while (!ft.pend.empty() && hipEventQuery(ft.pend.front().e1) != hipErrorNotReady) {
        const crh_ctx::FeedTune::Pend q = ft.pend.front(); ft.pend.pop_front();
        float ms = 0.f;
        if (q.which >= 0 && hipEventElapsedTime(&ms, q.e0, q.e1) == hipSuccess) { ft.ms[q.which] += ms; ++ft.n[q.which]; }
        c->ev_pool.push_back(q.e0); c->ev_pool.push_back(q.e1);
      }
      if (!ft.chosen && ft.n[0] >= 8u && ft.n[1] >= 8u) {
        ft.chosen = ft.ms[1] / ft.n[1] < 0.95 * (ft.ms[0] / ft.n[0]) ? 4u : 3u;
        for (crh_ctx::FeedTune::Pend& q : ft.pend) q.which = -1;
      }
      if (ft.chosen) c->frame_feeders = ft.chosen;
      else {
        const uint32_t block = ft.frames / 6u, which = (block + 1u) & 1u;      // blocks: warm-up (4, not measured), 3, 4, 3, 4, ...
        c->frame_feeders = which ? 4u : 3u;
        if (block != 0u && ft.frames % 6u != 0u && ft.pend.size() < 64u) {     // (the first frame of a block runs beside a frame of the other setting; the first block after
                                                                               // a build touches the path state for the first time)
          ln.tune_e0 = get_event(c); ln.tune_e1 = get_event(c);
          ft.pend.push_back({ln.tune_e0, ln.tune_e1, (int)which});
        }
        ++ft.frames;
      }
    }
    if (frame && c->tile_order.on && c->tile_order.tag_next >= 0) {      // this lone frame is a sample of the sorted-against-row-major measurement (crh_render decided which list it got)
      crh_ctx::TileOrder& to = c->tile_order;
      if (!ln.tune_e0 && running == 0u && to.pend.size() < 32u) {
        ln.tune_e0 = get_event(c); ln.tune_e1 = get_event(c);
        to.pend.push_back({ln.tune_e0, ln.tune_e1, to.tag_next});
      }
      to.tag_next = -1;
    }
    const size_t base = (size_t)k * total;
    const DPaths& P = c->paths; const DQueues& Q = c->queues;
    ln.P.ray_o[0] = P.ray_o[0] + base; ln.P.ray_o[1] = P.ray_o[1] + base; ln.P.ray_d[0] = P.ray_d[0] + base; ln.P.ray_d[1] = P.ray_d[1] + base;
    ln.P.thr[0] = P.thr[0] + base; ln.P.thr[1] = P.thr[1] + base; ln.P.hit = P.hit + base; ln.P.rad = P.rad + base;
    ln.P.sh_o = P.sh_o + base; ln.P.sh_d = P.sh_d + base; ln.P.sh_c = P.sh_c + base; ln.P.stamp = next_stamp(c);      // this frame's own stamp
    ln.Q.q[0] = Q.q[0] + base; ln.Q.q[1] = Q.q[1] + base; ln.Q.q_sh = Q.q_sh + base; ln.Q.q2 = Q.q2 + base; ln.Q.q2_sh = Q.q2_sh + base; ln.Q.counts = c->d_lane_counts + kCounts * k;
    if (fork) CRH_HIP(hipStreamWaitEvent(ln.stream, c->lane_fork, 0));
    // a frame kernel wants compute units of its own: at most frame_pipe_depth of them run at a time (the frame submitted that many frames ago comes first;
    // measured with three in flight on halves of the chip: 272 against 379 frames/s in the drag loop)
    if (frame && c->frame_pipe_depth < depth) {
      const uint32_t j = (k + depth - std::max(1u, c->frame_pipe_depth)) % depth;
      if (c->pipe_running[j]) CRH_HIP(hipStreamWaitEvent(ln.stream, c->lane_join[j], 0));
    }
    if (c->lane_counter_epoch[k] != c->counter_epoch) {   // this stream's first frame of an accumulation: the counter block was zeroed one restart ago, stream-ordered
      const hipEvent_t z = c->counters_zeroed[c->counter_epoch & 3u];
      if (z) CRH_HIP(hipStreamWaitEvent(ln.stream, z, 0));
      c->lane_counter_epoch[k] = c->counter_epoch;
    }
    // the frames in flight own path-state slices [k * total, (k + 1) * total): a batch of ANOTHER size (crh_render_tiles with another
    // sample count or tile list) would lay its slice across theirs -- it starts only when they are all done (joined into the context's stream or not)
    if (total != c->pipe_total || depth != c->pipe_last_depth) {
      for (int j = 0; j < 8; ++j) if (c->pipe_running[j]) CRH_HIP(hipStreamWaitEvent(ln.stream, c->lane_join[j], 0));
      c->pipe_total = total; c->pipe_last_depth = depth;
      // ... and so must every LATER frame of the new size: its slice may lie across a slice of the old size too, and it runs on another stream (round 6,
      // tests/hunts/tile_order_sequences.py seed 166: crh_render(2) followed at once by three one-sample frames -- the third one's slice [6.5 M, 7.6 M) lay inside the
      // two-sample frame's [6.5 M, 8.7 M) and nothing made it wait: a GPU memory fault).  One event, recorded behind the waits above, says "the old size is gone".
      if (!c->pipe_resized) CRH_HIP(hipEventCreateWithFlags(&c->pipe_resized, hipEventDisableTiming));
      CRH_HIP(hipEventRecord(c->pipe_resized, ln.stream));
      c->pipe_resized_pending = true;
    } else if (c->pipe_resized_pending) {
      if (hipEventQuery(c->pipe_resized) == hipErrorNotReady) CRH_HIP(hipStreamWaitEvent(ln.stream, c->pipe_resized, 0));
      else c->pipe_resized_pending = false;
    }
    c->pending_n = 0;
    hipEvent_t e0 = get_event(c), e1 = get_event(c);          // crh_stats.seconds: device time of this frame (frames in flight overlap)
    hipEventRecord(e0, ln.stream);
    const hipEvent_t guard = c->rb_guard_pending ? c->rb_guard : nullptr; c->rb_guard_pending = false;      // later frames are ordered behind this one's accumulate
    const hipEvent_t after_reset = c->reset_pending ? c->reset_ev : nullptr; c->reset_pending = false;      // (it followed the joins of every earlier frame)
    rc = run_lane(c, ln, S, d_list, nt, nullptr, ns, 0, true, c->pipe_pending[prev] ? c->lane_join[prev] : nullptr, guard, after_reset); if (rc) return rc;
    hipEventRecord(e1, ln.stream);
    c->render_ev.emplace_back(e0, e1);
    CRH_HIP(hipEventRecord(c->lane_join[k], ln.stream));
    c->pipe_pending[k] = true; c->pipe_running[k] = true;
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
      rc = run_batch(c, S, d_list + t0, g, c->d_seeds + s0, std::min(spb, ns - s0));
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
  c->paths.stamp = next_stamp(c);
  Lane ln{cstream(c), c->paths, c->queues, (int)std::min<uint64_t>((uint64_t)c->grid, std::max<uint64_t>(512u, (uint64_t)most * tpp / 1024u)),
          (int)std::min<uint64_t>((uint64_t)c->grid_trace, std::max<uint64_t>(512u, (uint64_t)most * tpp / 2048u)), true};      // grids follow the batch (run_batch)
  ln.n_tiles_dev = c->d_adapt_n; ln.donate = c->donate;
  ln.frame = frame_ok(c, (uint64_t)most * tpp); ln.grid_frame = ln.frame ? std::min(frame_grid(c, S), (int)std::max<uint64_t>(64u, (uint64_t)most * tpp / 512u)) : 0;
  rc = run_lane(c, ln, S, c->d_tile_ids, most, c->d_seeds, 1, 1, true); if (rc) return rc;
  hipEventRecord(e1, cstream(c));
  c->render_ev.emplace_back(e0, e1);
  return trim_events(c);
}

}  // namespace

extern "C" {

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
  // the order the tiles are claimed in (crh_context.h TileOrder): most rays of the last accumulation first -- for a host whose frames start on an idle chip (six
  // calls in a row found nothing in flight: a viewer that waits for every frame; a drag loop with a gap now and then does not get there).  With frames in flight the drain overlaps the next frame anyway, the row-major
  // list keeps neighbouring tiles together (sorted: drag loop -4 % on C3 / C2) and a new list would wait for the frames that read the old one.
  {
    crh_ctx::TileOrder& to = c->tile_order;
    if (to.on && to.n == nt && n == 1u) {                  // (one sample per pixel: the frame kernel's regime; a wide batch keeps the row-major list -- its tile groups want neighbours)
      const bool busy = c->last_running != 0u;           // what the previous pipelined frame found in flight when it was submitted (no event queries of its own here:
                                                         // four more hipEventQuery calls per frame cost the drag loop 2 - 3 %)
      to.streak = busy ? 0u : std::min(to.streak + 1u, 1000u);
      if (to.streak >= 6u && to.pending && hipEventQuery(to.copied) == hipSuccess) {
        to.pending = false;
        uint64_t sum = 0; for (uint32_t t = 0; t < nt; ++t) sum += to.h_cost[t];
        if (sum) {
          const double mean = (double)sum / nt;
          std::vector<uint8_t> cls(nt);
          for (uint32_t t = 0; t < nt; ++t) { const double x = to.h_cost[t] / mean; cls[t] = x >= 2.0 ? 0 : x >= 1.25 ? 1 : x >= 0.75 ? 2 : x >= 0.4 ? 3 : 4; }
          if (cls != to.cls) {                                                  // (the classes only say WHEN to make a new list: noise between two frames of one view does not)
            // most rays first, row-major among equals: a counting sort over 256 levels of the cost (five classes in row-major order inside each class gain nothing
            // -- 2.27 against 2.30 ms on CAD1M -- the sorted list 2.16: profiles/r6/lone_frame.md 2c)
            uint32_t top = 1; for (uint32_t t = 0; t < nt; ++t) top = std::max(top, to.h_cost[t]);
            uint32_t at[258]; std::memset(at, 0, sizeof at);
            auto level = [&](uint32_t t) { return 255u - (uint32_t)(((uint64_t)to.h_cost[t] * 255u) / top); };      // 0 = the most expensive
            for (uint32_t t = 0; t < nt; ++t) ++at[level(t) + 1u];
            for (int k = 1; k < 258; ++k) at[k] += at[k - 1];
            to.order.resize(nt);
            for (uint32_t t = 0; t < nt; ++t) to.order[at[level(t)]++] = t;
            to.cls.swap(cls); ++to.reorders;
          }
        }
      }
      bool sorted = false;
      if (to.streak >= 6u && to.order.size() == nt) {
        while (!to.pend.empty() && hipEventQuery(to.pend.front().e1) != hipErrorNotReady) {      // frame kernels of the measurement that have finished
          const crh_ctx::TileOrder::Pend q = to.pend.front(); to.pend.pop_front();
          float ms = 0.f;
          if (q.which >= 0 && hipEventElapsedTime(&ms, q.e0, q.e1) == hipSuccess) { to.tms[q.which] += ms; ++to.tn[q.which]; }
          c->ev_pool.push_back(q.e0); c->ev_pool.push_back(q.e1);
        }
        if (!to.verdict && to.tn[0] >= 5u && to.tn[1] >= 5u) {
          to.verdict = to.tms[0] / to.tn[0] < 0.98 * (to.tms[1] / to.tn[1]) ? 1u : 2u;
          for (crh_ctx::TileOrder::Pend& q : to.pend) q.which = -1;
        }
        to.tag_next = -1;
        if (to.verdict) sorted = to.verdict == 1u;
        else if (c->feed_tune.on && !c->feed_tune.chosen) sorted = true;      // one measurement at a time: the feeder count first
        else { const uint32_t which = to.trials++ & 1u; sorted = which == 0u; to.tag_next = (int)which; }
      }
      if (sorted) { all = to.order; ++to.calls_sorted; } else ++to.calls_row_major;
    }
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
  if (!c || mode < CRH_SCHEDULE_AUTO || mode > CRH_SCHEDULE_STAGED) return fail(c, CRH_E_INVALID, "schedule must be CRH_SCHEDULE_AUTO / _WIDE / _SMALL / _STAGED");
  if (c->schedule == CRH_SCHEDULE_AUTO) { c->auto_lane_max_paths = c->lane_max_paths; c->auto_donate = c->donate; c->auto_pipeline = c->pipeline; c->auto_frame_kernel = c->frame_kernel; }
  c->schedule = mode; c->pending_n = 0;
  c->read_since_render = true;                           // the next frame is not pipelined behind frames of the other schedule
  // WIDE: no batch counts as small (run_batch: one stream, full grids, plain kernels; render_impl: no frame pipelining; adaptive
  // iterations: plain kernels).  SMALL: every batch up to the path budget does (the frame kernel takes those below 2^25 slots, the staged form the rest).
  // STAGED: like SMALL, in the staged form only (lanes / pipelined launches per stage and bounce, the donating kernels) -- the schedule the frame kernel replaced.
  c->lane_max_paths = mode == CRH_SCHEDULE_WIDE ? 0u : (mode == CRH_SCHEDULE_SMALL || mode == CRH_SCHEDULE_STAGED ? (1u << 30) : c->auto_lane_max_paths);
  c->frame_kernel = mode == CRH_SCHEDULE_STAGED ? false : c->auto_frame_kernel;
  c->donate = mode == CRH_SCHEDULE_WIDE ? false : c->auto_donate;
  c->pipeline = mode == CRH_SCHEDULE_WIDE ? false : c->auto_pipeline;
  return CRH_OK;
}

int crh_set_pipeline_depth(crh_ctx* c, uint32_t frames)
{
  if (!c || frames < 2u || frames > 8u) return fail(c, CRH_E_INVALID, "pipeline depth must be in 2 .. 8 frames");
  if (frames > pipeline_capacity()) {
    char b[320]; snprintf(b, sizeof b, "pipeline depth %u needs GPU_MAX_HW_QUEUES >= %u in the environment before the process's first HIP call: this process has %d "
                          "hardware queues, i.e. at most %u frames in flight (crh_query_pipeline_capacity)", frames, frames + 2u, hw_queues(), pipeline_capacity());
    return fail(c, CRH_E_INVALID, b);
  }
  CRH_HIP(hipSetDevice(c->device));
  CRH_HIP(hipStreamSynchronize(cstream(c)));            // the frames in flight own slices of the path state laid out for the old depth
  c->pipe_depth = frames; c->pipe_seq = 0; c->pipe_total = 0;
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

int crh_enable_counters(crh_ctx* c, int on) { if (!c) return CRH_E_INVALID; c->counters_on = on != 0; return CRH_OK; }

int crh_enable_kernel_timing(crh_ctx* c, int on) { if (!c) return CRH_E_INVALID; c->timing_on = on != 0; return CRH_OK; }

}  // extern "C"

// crh_readback.cpp -- HDR / LDR read-back (synchronous and asynchronous), accumulator checkpoints, statistics, kernel timing
// (one of the translation units behind include/cadrays_hip.h; the context, the shared helpers and the map of the files: crh_context.h)
#include "crh_context.h"

using namespace crh;
using namespace crh::api;

extern "C" {

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
  launch_tonemap(L, c->assembled_valid ? c->d_assembled : c->d_accum, (uint8_t*)c->d_scratch, n, c->par.tonemap_mode, c->par.exposure, c->par.white_point, c->spec.display_gamma22, d_mask, c->par.width, ts);
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
    launch_tonemap(L, src, c->d_rb[slot], n, c->par.tonemap_mode, c->par.exposure, c->par.white_point, c->spec.display_gamma22, overlay ? c->d_picked : nullptr, c->par.width, ts);
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
  return check_device_error(c);
}

int crh_get_packet_stats(crh_ctx* c, uint64_t* packet_rays, uint64_t* fallback_rays)
{
  if (!c) return CRH_E_INVALID;
  CRH_HIP(hipSetDevice(c->device));
  CRH_HIP(hipStreamSynchronize(cstream(c)));
  drain_events(c);
  DCounters h;
  CRH_HIP(hipMemcpyAsync(&h, c->d_counters, sizeof h, hipMemcpyDeviceToHost, cstream(c)));
  CRH_HIP(hipStreamSynchronize(cstream(c)));
  if (packet_rays) *packet_rays = h.packet_rays;
  if (fallback_rays) *fallback_rays = h.packet_fallback;
  return CRH_OK;
}

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

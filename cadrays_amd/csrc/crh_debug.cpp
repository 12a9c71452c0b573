// crh_debug.cpp -- kernel-level entry points: API ray tracing, micro-benchmark, math and BSDF test hooks
// (one of the translation units behind include/cadrays_hip.h; the context, the shared helpers and the map of the files: crh_context.h)
#include "crh_context.h"

using namespace crh;
using namespace crh::api;

extern "C" {

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

// Test hook (never called by a product path): crh_reduce's RCCL branch -- communicators kept on the root, one group of ncclReduce calls on the contexts' own
// streams, the assembled frame's lifetime -- on contexts that share ONE device, the only kind a 1-GPU pool has (crh_reduce.cpp explains what is replaced)
int crh_debug_reduce_fake_devices(crh_ctx* const* ctxs, uint32_t n, uint32_t root)
{
  if (!ctxs || n == 0 || root >= n || !ctxs[root]) return CRH_E_INVALID;
  return crh::api::reduce_fake_devices(ctxs, n, root);
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

}  // extern "C"

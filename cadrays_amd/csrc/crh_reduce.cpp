// crh_reduce.cpp -- crh_reduce: the tile-sharded frame of several contexts assembled on one (RCCL over xGMI, or peer copies)
// (one of the translation units behind include/cadrays_hip.h; the context, the shared helpers and the map of the files: crh_context.h)
#include "crh_context.h"
#include <dlfcn.h>

using namespace crh;
using namespace crh::api;

namespace {

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

}  // namespace

namespace crh {
namespace api {

void release_comms(crh_ctx* c)
{
  if (c->comms.empty()) return;
  if (RcclApi* R = rccl()) for (ncclComm_t m : c->comms) if (m) R->CommDestroy(m);
  c->comms.clear(); c->comm_ctxs.clear();
}

}  // namespace api
}  // namespace crh

namespace {

#define CRH_NCCL(call)                                                                                   \
  do { ncclResult_t r_ = (call); if (r_ != ncclSuccess) {                                                \
      char b_[512]; snprintf(b_, sizeof b_, "%s failed: %s", #call, R->GetErrorString ? R->GetErrorString(r_) : "rccl error"); \
      c->err = b_; return CRH_E_DEVICE; } } while (0)

// fake_devices: the test hook of crh_debug_reduce_fake_devices (crh_debug.cpp) -- never set by crh_reduce
int reduce_impl(crh_ctx* const* ctxs, uint32_t n, uint32_t root, bool fake_devices)
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
  RcclApi* R = ((distinct && (n > 1 || getenv("CRH_REDUCE_RCCL_SINGLE"))) || fake_devices) ? rccl() : nullptr;
  if (fake_devices && !R) return fail(c, CRH_E_DEVICE, "crh_debug_reduce_fake_devices: librccl could not be loaded");
  if (R && fake_devices) {
    // TEST HOOK (round-5 verdict, item 6): a 1-GPU pool can never run the branch below with n > 1.  Here the contexts -- all on ONE device -- take the same
    // steps with what one device allows: a communicator per context (each a single-rank group of its own: ncclCommInitAll over n copies of the same device
    // is refused), the same bookkeeping of communicators on the root, ONE group of n ncclReduce calls on the contexts' own streams (rank 0 of a
    // single-rank group: the root's accumulator lands in d_assembled, the others reduce in place), the same synchronisation -- and the sum over the
    // contexts, which the single-rank groups cannot do, by the add kernel of the same-device branch.
    if (c->comm_ctxs.size() != n || !std::equal(c->comm_ctxs.begin(), c->comm_ctxs.end(), ctxs)) {
      release_comms(c);
      c->comms.assign(n, nullptr);
      for (uint32_t i = 0; i < n; ++i) CRH_NCCL(R->CommInitAll(&c->comms[i], 1, &devs[i]));
      c->comm_ctxs.assign(ctxs, ctxs + n);
    }
    CRH_NCCL(R->GroupStart());
    for (uint32_t i = 0; i < n; ++i) {
      hipSetDevice(ctxs[i]->device);
      ncclResult_t r = R->Reduce(ctxs[i]->d_accum, i == root ? (void*)c->d_assembled : (void*)ctxs[i]->d_accum, 4 * n4, ncclFloat, ncclSum, 0, c->comms[i], cstream(ctxs[i]));
      if (r != ncclSuccess) { R->GroupEnd(); c->err = "ncclReduce failed"; return CRH_E_DEVICE; }
    }
    CRH_NCCL(R->GroupEnd());
    for (uint32_t i = 0; i < n; ++i) { CRH_HIP(hipSetDevice(ctxs[i]->device)); CRH_HIP(hipStreamSynchronize(cstream(ctxs[i]))); }
    CRH_HIP(hipSetDevice(c->device));
    Launch L{cstream(c), c->grid, false};
    for (uint32_t i = 0; i < n; ++i) if (i != root) launch_add4(L, c->d_assembled, ctxs[i]->d_accum, (uint32_t)n4);
    CRH_HIP(hipGetLastError());
    CRH_HIP(hipStreamSynchronize(cstream(c)));
  } else if (R) {
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

namespace crh {
namespace api {
int reduce_fake_devices(crh_ctx* const* ctxs, uint32_t n, uint32_t root) { return reduce_impl(ctxs, n, root, true); }
}  // namespace api
}  // namespace crh

extern "C" {

int crh_reduce(crh_ctx* const* ctxs, uint32_t n, uint32_t root)
{
  if (!ctxs || n == 0 || root >= n || !ctxs[root]) return CRH_E_INVALID;
  return reduce_impl(ctxs, n, root, false);
}

}  // extern "C"

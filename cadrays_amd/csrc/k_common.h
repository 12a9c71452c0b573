// k_common.h -- part of kernels.hip (ONE translation unit: included there inside namespace crh::(anonymous), in this order: k_common, k_traversal, k_packets, k_bsdf,
// k_lights_env, k_raygen, k_shade, k_accumulate).  Small device helpers: lane / wave utilities, streaming loads and stores, queue appends.

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

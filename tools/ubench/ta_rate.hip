// Micro-benchmark: cost of fully divergent per-lane loads (every lane its own 64-B record) as a function of load width and count.
// Prints CU-cycles per wave-level load instruction.  hipcc --offload-arch=gfx950 -O3 -o ta_rate ta_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int K, int W>   // K loads of W dwords (1,2,4) per record visit
__global__ __launch_bounds__(256) void k(const float4* __restrict__ tab, uint32_t mask, uint32_t iters, float* out, int coherent)
{
  uint32_t idx = (blockIdx.x * 256u + threadIdx.x) * 2654435761u;
  if (coherent) idx = blockIdx.x * 977u;
  float acc = 0.f;
  for (uint32_t i = 0; i < iters; ++i) {
    const uint32_t r = (idx >> 8) & mask;
    const float4* p = tab + 4u * r;
#pragma unroll
    for (int j = 0; j < K; ++j) {
      if (W == 4) { float4 v = p[j]; acc += v.x + v.w; idx += __float_as_uint(v.y); }
      if (W == 2) { float2 v = ((const float2*)(p + j))[0]; acc += v.x; idx += __float_as_uint(v.y); }
      if (W == 1) { float v = ((const float*)(p + j))[1]; idx += __float_as_uint(v); acc += v; }
    }
    idx = idx * 1664525u + 1013904223u;
  }
  out[blockIdx.x * 256u + threadIdx.x] = acc;
}

template <int K, int W> void run(const char* name, const float4* tab, uint32_t nrec, float* out, int cus, double ghz, int coherent)
{
  const uint32_t iters = 2000;
  const int grid = cus * 5;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<K, W>), dim3(grid), dim3(256), 0, 0, tab, nrec - 1, 200u, out, coherent);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<K, W>), dim3(grid), dim3(256), 0, 0, tab, nrec - 1, iters, out, coherent);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double wave_instr_per_cu = 5.0 * 4 * iters * K;         // waves per CU x iterations x loads
  const double cyc = ms * 1e-3 * ghz * 1e9;
  printf("%-34s records %8u  %8.3f ms  %7.1f CU-cycles per wave load instr  (%.1f per visit)\n", name, nrec, ms, cyc / wave_instr_per_cu, cyc / wave_instr_per_cu * K);
}

int main()
{
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount; const double ghz = p.clockRate * 1e-6;
  printf("CUs %d clock %.2f GHz\n", cus, ghz);
  for (uint32_t nrec : {128u, 16384u, 1u << 20}) {
    std::vector<float> h(16 * (size_t)nrec);
    for (size_t i = 0; i < h.size(); ++i) { uint32_t b = (uint32_t)(i * 2654435761u) >> 9; h[i] = *(float*)&b; }
    float4* tab; hipMalloc((void**)&tab, h.size() * 4); hipMemcpy(tab, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    float* out; hipMalloc((void**)&out, sizeof(float) * cus * 5 * 256);
    run<1, 4>("1 x dwordx4 divergent", tab, nrec, out, cus, ghz, 0);
    run<2, 4>("2 x dwordx4 divergent", tab, nrec, out, cus, ghz, 0);
    run<3, 4>("3 x dwordx4 divergent", tab, nrec, out, cus, ghz, 0);
    run<4, 4>("4 x dwordx4 divergent", tab, nrec, out, cus, ghz, 0);
    run<4, 2>("4 x dwordx2 divergent", tab, nrec, out, cus, ghz, 0);
    run<4, 1>("4 x dword   divergent", tab, nrec, out, cus, ghz, 0);
    run<4, 4>("4 x dwordx4 wave-uniform", tab, nrec, out, cus, ghz, 1);
    hipFree(tab); hipFree(out);
  }
  return 0;
}

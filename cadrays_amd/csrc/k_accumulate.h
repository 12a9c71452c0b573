// k_accumulate.h -- part of kernels.hip (ONE translation unit: included there inside namespace crh::(anonymous), in this order: k_common, k_traversal, k_packets, k_bsdf,
// k_lights_env, k_raygen, k_shade, k_accumulate).  K_accumulate, tone map / LDR, adaptive-sampling kernels, debug kernels.
// ================================================================== accumulate / display
__global__ __launch_bounds__(kBlock) void k_accumulate(DScene S, DPaths P, float4* __restrict__ accum, float* __restrict__ m2,
                                                        const uint32_t* __restrict__ tile_ids, uint32_t n_tiles,
                                                        uint32_t first_sample, uint32_t n_samples, uint32_t batch_samples, DCounters* C,
                                                        const uint32_t* __restrict__ n_tiles_dev, uint32_t* __restrict__ tile_cost)
{
  if (n_tiles_dev) n_tiles = *n_tiles_dev;
  // samples [first_sample, first_sample + n_samples) of the batch of batch_samples in the path buffer are folded in, in order
  const uint32_t per_sample = n_tiles * S.tile_size * S.tile_size;
  const uint32_t last = first_sample + n_samples;
  uint32_t done = 0;
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f); float q = 0.f;
  auto fold = [&](const float4 r) {
    const float w = 1.0f / (a.w + 1.0f);
    const bool written = __float_as_uint(r.w) == P.stamp;      // a path that never added anything left its record alone: zero radiance (DPaths::stamp)
    float v[3] = {written ? r.x : 0.f, written ? r.y : 0.f, written ? r.z : 0.f};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      if (!(v[k] == v[k])) v[k] = 0.f;
      if (S.clampv > 0.f && v[k] > S.clampv) v[k] = S.clampv;
    }
    a.x = CRH_FMA(v[0] - a.x, w, a.x);
    a.y = CRH_FMA(v[1] - a.y, w, a.y);
    a.z = CRH_FMA(v[2] - a.z, w, a.z);
    a.w = a.w + 1.0f;
    if (m2) {      // running mean of the squared luminance (adaptive sampling's variance estimate)
      const float l = CRH_FMA(0.0722f, v[2], CRH_FMA(0.7152f, v[1], 0.2126f * v[0]));
      q = CRH_FMA(l * l - q, w, q);
    }
    ++done;
  };
  if (sample_group(batch_samples) >= 8u) {
    // Wide batches: a pixel's samples sit in runs of >= 8 consecutive slots (one 128-B line), the pixels of a block far apart -- a
    // lane walking its own pixel would touch 64 lines per load.  A wavefront therefore takes one 8x8 block, fetches 64 pixels x 8
    // samples with eight coalesced loads (eight whole lines each) into LDS, and every lane folds ITS pixel's eight samples from
    // there, in sample order: the same arithmetic in the same order (4.05 -> ~1 ms per 128-spp step at 1080p).
    __shared__ float4 s_tile[4][64 * 9];
    const uint32_t wave = threadIdx.x >> 6, lane = lane_id();
    float4* tile = s_tile[wave];
    const uint32_t n_blocks = per_sample >> 6, stride_b = gridDim.x * 4u, rounds = (n_blocks + stride_b - 1u) / stride_b;
    for (uint32_t k = 0; k < rounds; ++k) {
      const uint32_t B = blockIdx.x * 4u + wave + k * stride_b;
      const bool live = B < n_blocks;                                  // uniform per wavefront; every wavefront keeps the barriers
      uint32_t px = 0, py = 0;
      const bool mine = live && slot_pixel(S, tile_ids, B * 64u + lane, px, py);
      const size_t pi = (size_t)py * S.width + px;
      if (mine) { a = accum[pi]; q = m2 ? m2[pi] : 0.f; }
      for (uint32_t s0 = first_sample & ~7u; s0 < last; s0 += 8u) {
        if (live) {
#pragma unroll
          for (uint32_t j = 0; j < 8u; ++j) {
            const uint32_t qp = 8u * j + (lane >> 3), ks = lane & 7u;
            tile[qp * 9u + ks] = P.rad[pixel_sample_to_slot(B * 64u + qp, s0 + ks, batch_samples)];
          }
        }
        __syncthreads();
        if (mine) {
#pragma unroll
          for (uint32_t ks = 0; ks < 8u; ++ks) { const uint32_t s = s0 + ks; if (s >= first_sample && s < last) fold(tile[lane * 9u + ks]); }
        }
        __syncthreads();
      }
      if (mine) { accum[pi] = a; if (m2) m2[pi] = q; }
    }
  } else {
    for (uint32_t local = blockIdx.x * kBlock + threadIdx.x; local < per_sample; local += gridDim.x * kBlock) {
      uint32_t px, py, bounces = 0u;
      if (slot_pixel(S, tile_ids, local, px, py)) {
        const size_t pi = (size_t)py * S.width + px;
        a = accum[pi]; q = m2 ? m2[pi] : 0.f;
        for (uint32_t s = first_sample; s < last; ++s) fold(P.rad[pixel_sample_to_slot(local, s, batch_samples)]);
        accum[pi] = a;
        if (m2) m2[pi] = q;
        // a frame-kernel frame leaves the index of each path's last ray in its slot (k_frame.h: ray_d.w = bounce << 26 | ...): rays traced by this pixel's path
        // (every pixel: a list made from one wavefront in eight lost the gain -- lone frame 2.20 instead of 2.13 ms on CAD1M, 2.92 instead of 2.85 on C3)
        if (tile_cost) bounces = (__float_as_uint(P.ray_d[0][pixel_sample_to_slot(local, first_sample, batch_samples)].w) >> kFrameBounceShift) + 1u;
      }
      // per-tile cost of the frame (round 6: the host claims the expensive tiles first in the next lone frame, crh_schedule.cpp): the 64 slots of a wavefront lie in
      // ONE tile (a tile is tile_size^2 consecutive slots, a multiple of 64), every lane of the wavefront runs the same number of rounds
      if (tile_cost) { const uint32_t sum = wave_sum(bounces); if (lane_id() == 0 && sum) atomicAdd(&tile_cost[tile_ids[local / (S.tile_size * S.tile_size)]], sum); }
    }
  }
  done = wave_sum(done);
  if (lane_id() == 0 && done) atomicAdd(&C->samples, (unsigned long long)done);
}

// Per-tile error estimate for the adaptive tile sampler (one workgroup per tile, fixed summation order so the
// CPU oracle reproduces every bit): pixel error = sqrt(max(E[l^2] - E[l]^2, 0) / n), unsampled or once-sampled
// pixels count as 1e3; lane j sums pixels j, j+256, ... of the row-major tile, then a stride-128..1 tree.
__global__ __launch_bounds__(kBlock) void k_tile_error(DScene S, const float4* __restrict__ accum, const float* __restrict__ m2,
                                                        float* __restrict__ tile_err, uint32_t* __restrict__ tile_min_count)
{
  __shared__ float s_e[kBlock];
  __shared__ float s_n[kBlock];
  __shared__ float s_c[kBlock];
  const uint32_t ts = S.tile_size, tx = (S.width + ts - 1u) / ts;
  const uint32_t tile = blockIdx.x, x0 = (tile % tx) * ts, y0 = (tile / tx) * ts;
  float e = 0.f, npx = 0.f, cmin = 3.0e38f;
  for (uint32_t i = threadIdx.x; i < ts * ts; i += kBlock) {
    const uint32_t px = x0 + i % ts, py = y0 + i / ts;
    if (px < S.width && py < S.height) {
      const size_t pi = (size_t)py * S.width + px;
      const float4 a = accum[pi];
      float pe = 1.0e3f;
      if (a.w >= 2.0f) {
        const float l = CRH_FMA(0.0722f, a.z, CRH_FMA(0.7152f, a.y, 0.2126f * a.x));
        pe = crh_sqrt(crh_max(m2[pi] - l * l, 0.f) / a.w);
      }
      e += pe; npx += 1.0f; cmin = crh_min(cmin, a.w);
    }
  }
  s_e[threadIdx.x] = e; s_n[threadIdx.x] = npx; s_c[threadIdx.x] = cmin;
  __syncthreads();
  for (uint32_t st = kBlock / 2; st > 0; st >>= 1) {
    if (threadIdx.x < st) {
      s_e[threadIdx.x] += s_e[threadIdx.x + st]; s_n[threadIdx.x] += s_n[threadIdx.x + st];
      s_c[threadIdx.x] = crh_min(s_c[threadIdx.x], s_c[threadIdx.x + st]);
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    tile_err[tile] = s_n[0] > 0.f ? s_e[0] / s_n[0] : 0.f;
    tile_min_count[tile] = s_n[0] > 0.f ? (uint32_t)s_c[0] : 0u;
  }
}

// Adaptive tile selection on the device (reference: AdaptiveScreenSampling / NbRayTracingTiles, SettingsWidget.cxx:427-477; the
// rule itself is DESIGN.md section 7 and the CPU oracle's adaptive_iteration): inverse-CDF draws driven by the base-2 radical
// inverse of a running pick counter, +1 sample on every distinct tile drawn, each at its own sample index.  ONE workgroup; the
// running sum of the errors is taken by one lane in tile order (the oracle's float summation order decides ties), through LDS
// in chunks; draws and the ordered compaction are parallel.  Nothing goes through the host: the tile list, its length and the
// per-tile frame seeds stay in HBM for k_raygen / k_accumulate.
constexpr uint32_t kPickChunk = 4096;
__global__ __launch_bounds__(kBlock) void k_adaptive_pick(const float* __restrict__ tile_err, const uint32_t* __restrict__ tile_cnt, uint32_t nt,
                                                           uint32_t pick0, uint32_t n_picks, uint32_t seed, float* __restrict__ cdf,
                                                           uint8_t* __restrict__ picked, uint32_t* __restrict__ tiles_out,
                                                           uint32_t* __restrict__ seeds_out, uint32_t* __restrict__ n_out)
{
  __shared__ float s_v[kPickChunk];
  __shared__ uint32_t s_part[kBlock];
  __shared__ float s_acc;
  if (threadIdx.x == 0) s_acc = 0.f;
  for (uint32_t i = threadIdx.x; i < nt; i += kBlock) picked[i] = 0;
  for (uint32_t c0 = 0; c0 < nt; c0 += kPickChunk) {
    const uint32_t m = min(kPickChunk, nt - c0);
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < m; i += kBlock) { const float e = tile_err[c0 + i]; s_v[i] = e > 0.f ? e : 0.f; }
    __syncthreads();
    if (threadIdx.x == 0) { float a = s_acc; for (uint32_t i = 0; i < m; ++i) { a += s_v[i]; s_v[i] = a; } s_acc = a; }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < m; i += kBlock) cdf[c0 + i] = s_v[i];
  }
  __threadfence_block();
  __syncthreads();
  const float acc = s_acc;
  for (uint32_t k = threadIdx.x; k < n_picks; k += kBlock) {
    const uint32_t v = __brev(pick0 + k);
    const float u = (float)(v >> 8) * 5.9604644775390625e-8f;
    uint32_t t;
    if (!(acc > 0.f)) t = (uint32_t)(u * (float)nt);                    // no estimate yet: uniform
    else {                                                              // first tile whose running sum exceeds x
      const float x = u * acc;
      uint32_t lo = 0, hi = nt;
      while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (cdf[mid] > x) hi = mid; else lo = mid + 1; }
      t = lo;
    }
    if (t >= nt) t = nt - 1u;
    picked[t] = 1;
  }
  __threadfence_block();
  __syncthreads();
  // ordered compaction: thread j owns tiles [j * per, (j + 1) * per)
  const uint32_t per = (nt + kBlock - 1) / kBlock, b0 = min(nt, threadIdx.x * per), b1 = min(nt, b0 + per);
  uint32_t mine = 0;
  for (uint32_t i = b0; i < b1; ++i) mine += picked[i];
  s_part[threadIdx.x] = mine;
  __syncthreads();
  if (threadIdx.x == 0) { uint32_t a = 0; for (int j = 0; j < kBlock; ++j) { const uint32_t v = s_part[j]; s_part[j] = a; a += v; } *n_out = a; }
  __syncthreads();
  uint32_t w = s_part[threadIdx.x];
  for (uint32_t i = b0; i < b1; ++i)
    if (picked[i]) {
      tiles_out[w] = i;
      // the tile's own sample index selects its frame seed: Bullard generator restarted at `seed`, frame n uses next() >> 2
      uint32_t hi = seed, lo = seed ^ 0x49616E42u, r = 0;
      const uint32_t n = tile_cnt[i];
      for (uint32_t j = 0; j <= n; ++j) { hi = (hi >> 2) + (hi << 2); hi += lo; lo += hi; r = hi; }
      seeds_out[w] = r >> 2;
      ++w;
    }
}

__device__ __forceinline__ float hable(float x)
{
  const float A = 0.22f, B = 0.30f, Cc = 0.10f, D = 0.20f, E = 0.01f, F = 0.30f;
  return (CRH_FMA(x, CRH_FMA(A, x, Cc * B), D * E) / CRH_FMA(x, CRH_FMA(A, x, B), D * F)) - E / F;
}
__global__ __launch_bounds__(kBlock) void k_tonemap(const float4* __restrict__ accum, uint8_t* __restrict__ out, uint32_t n,
                                                     int mode, float exposure, float white_point, int gamma22,
                                                     const uint8_t* __restrict__ tile_mask, uint32_t width, uint32_t tile_size)
{
  const uint32_t tiles_x = tile_mask ? (width + tile_size - 1u) / tile_size : 0u;
  const float gain = crh_exp(exposure * 0.69314718056f);
  const float wp = hable(white_point > 0.f ? white_point : 1.0f);
  for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
    const float4 a = accum[i];
    float v[3] = {a.x, a.y, a.z};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      float x = v[k];
      if (!(x == x) || x < 0.f) x = 0.f;
      x = x * gain;
      if (mode == 1) x = hable(x) / wp;
      x = crh_clamp(x, 0.f, 1.0f);
      x = gamma22 ? crh_pow(x, 1.0f / 2.2f) : crh_sqrt(x);      // crh_spec.h #15: gamma 2 is what the reference's icons show
      out[3u * i + k] = (uint8_t)(int)CRH_FMA(x, 255.0f, 0.5f);
    }
    if (tile_mask) {                                     // ShowSamplingTiles: red outline around the tiles just sampled
      const uint32_t px = i % width, py = i / width, lx = px % tile_size, ly = py % tile_size;
      if (tile_mask[(py / tile_size) * tiles_x + px / tile_size] && (lx == 0u || ly == 0u || lx == tile_size - 1u || ly == tile_size - 1u)) {
        out[3u * i] = 255; out[3u * i + 1u] = 0; out[3u * i + 2u] = 0;
      }
    }
  }
}
__global__ __launch_bounds__(kBlock) void k_hdr(const float4* __restrict__ accum, float* __restrict__ out, uint32_t n)
{
  for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
    const float4 a = accum[i];
    out[3u * i] = a.x; out[3u * i + 1u] = a.y; out[3u * i + 2u] = a.z;
  }
}

// dst += src over n float4 (the same-device leg of crh_reduce: disjoint tile support, so every pixel adds zeros to one value)
__global__ __launch_bounds__(kBlock) void k_add4(float4* __restrict__ dst, const float4* __restrict__ src, uint32_t n)
{
  for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
    float4 a = dst[i]; const float4 b = src[i];
    a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    dst[i] = a;
  }
}

// crh_set_transforms, static / moved split: overwrite the 48-B triangle records at the listed leaf positions (an object leaving the static
// tree: all-zero vertices, whose test yields NaN and rejects; coming back: the original vertices)
__global__ __launch_bounds__(kBlock) void k_scatter_tris(float4* __restrict__ tris, const uint32_t* __restrict__ pos, const float4* __restrict__ recs, uint32_t n)
{
  for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
    float4* d = tris + kTriStride * pos[i];
    d[0] = recs[3u * i]; d[1] = recs[3u * i + 1u]; d[2] = recs[3u * i + 2u];
  }
}

__global__ void k_debug_math(int fn, const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out,
                             float* __restrict__ out2, uint32_t n)
{
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    switch (fn) {
      case 0: crh_sincos2pi(a[i], &out[i], &out2[i]); break;
      case 1: out[i] = crh_exp(a[i]); break;
      case 2: out[i] = crh_log(a[i]); break;
      case 3: out[i] = crh_pow(a[i], b[i]); break;
      case 4: out[i] = crh_acos(a[i]); break;
      case 5: out[i] = crh_atan2(a[i], b[i]); break;
      case 6: crh_sincos(a[i], &out[i], &out2[i]); break;
      case 7: out[i] = crh_sqrt(a[i]); break;
      case 8: out[i] = a[i] / b[i]; break;
      case 9: { uint32_t s = crh_rng_seed(__float_as_uint(a[i]), __float_as_uint(b[i])); out[i] = crh_rng_next(&s); out2[i] = crh_rng_next(&s); } break;
      case 10: { const v3 x = crh_norm3(crh_mk3(a[i], b[i], a[i] * b[i])); out[i] = x.x; out2[i] = crh_dot3(x, crh_mk3(b[i], a[i], 1.0f)); } break;
      default: out[i] = 0.f;
    }
  }
}

// Test hook behind crh_debug_bsdf: the layered BSDF functions k_shade uses, evaluated on caller-supplied directions (local
// frame, z = shading normal) so that the analytic known-answer tests (pdf integrates to 1, sample weight = f cos / pdf,
// Fresnel limits, Snell) run on the gfx950 code itself and not only on the CPU oracle.
//   fn 0: out[3i..]   = eval_layered(wi, wo)            (f * cos)
//   fn 1: out[i]      = pdf_layered(wo, wi, W = 1)
//   fn 2: out[8i..]   = sample_layered with rng state bits(b[3i]), inside flag b[3i+1] != 0: wi.xyz, weight.xyz,
//                       flags (1 alive | 2 delta | 4 inside after), rng state after (uint bits)
//   fn 3: out[3i..]   = fresnel_media(a[3i], m.FresnelCoat)
__global__ void k_debug_bsdf(int fn, const float4* __restrict__ m, const float* __restrict__ a, const float* __restrict__ b,
                             float* __restrict__ out, uint32_t n, int two_sided)
{
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const v3 wo = crh_mk3(a[3u * i], a[3u * i + 1u], a[3u * i + 2u]);
    Bsdf bs;
    bs.Kc = xyz(m[0]); bs.Rc = m[0].w; bs.Kd = xyz(m[1]); bs.Ks = xyz(m[2]); bs.Rs = m[2].w; bs.Kt = xyz(m[3]); bs.Le = xyz(m[4]);
    bs.ab = m[5]; bs.fc = m[6]; bs.fb = m[7];
    bs.Fc = fresnel_media(wo.z, bs.fc);
    const v3 one = crh_mk3(1.0f, 1.0f, 1.0f);
    if (fn == 0) {
      const v3 r = eval_layered(bs, crh_mk3(b[3u * i], b[3u * i + 1u], b[3u * i + 2u]), wo, two_sided);
      out[3u * i] = r.x; out[3u * i + 1u] = r.y; out[3u * i + 2u] = r.z;
    } else if (fn == 1) {
      out[i] = pdf_layered(bs, wo, crh_mk3(b[3u * i], b[3u * i + 1u], b[3u * i + 2u]), one, two_sided);
    } else if (fn == 2) {
      uint32_t rng = __float_as_uint(b[3u * i]);
      bool inside = b[3u * i + 1u] != 0.f, delta = false;
      v3 W = one, wi = crh_mk3(0.f, 0.f, 0.f); int lobe;
      const bool alive = sample_layered(bs, wo, wi, W, inside, delta, rng, two_sided, SpecB{0, 1.0f}, lobe);
      float* o = out + 8u * i;
      o[0] = wi.x; o[1] = wi.y; o[2] = wi.z; o[3] = W.x; o[4] = W.y; o[5] = W.z;
      o[6] = (float)((alive ? 1 : 0) | (delta ? 2 : 0) | (inside ? 4 : 0)); o[7] = __uint_as_float(rng);
    } else {
      const v3 r = fresnel_media(wo.x, bs.fc);
      out[3u * i] = r.x; out[3u * i + 1u] = r.y; out[3u * i + 2u] = r.z;
    }
  }
}

// k_raygen.h -- part of kernels.hip (ONE translation unit: included there inside namespace crh::(anonymous), in this order: k_common, k_traversal, k_packets, k_bsdf,
// k_lights_env, k_raygen, k_shade, k_accumulate).  Path slot <-> pixel mapping of small and wide batches, k_raygen.
// ================================================================== path slot <-> pixel
// Slot layout inside one sample: tile-major, and inside a tile 8x8-pixel blocks so that one wavefront
// owns one 8x8 block (coherent primary rays, coalesced accumulator rows of 8 float4 = 128 B).
__device__ __forceinline__ bool slot_pixel(const DScene& S, const uint32_t* __restrict__ tile_ids, uint32_t local,
                                           uint32_t& px, uint32_t& py)
{
  const uint32_t ts = S.tile_size, tpp = ts * ts;
  const uint32_t ti = local / tpp, off = local - ti * tpp;
  const uint32_t tile = tile_ids ? tile_ids[ti] : ti;
  const uint32_t tx = (S.width + ts - 1u) / ts, ty = (S.height + ts - 1u) / ts;
  const uint32_t blk = off >> 6, l = off & 63u, bpr = ts >> 3;
  px = (tile % tx) * ts + (blk % bpr) * 8u + (l & 7u);
  py = (tile / tx) * ts + (blk / bpr) * 8u + (l >> 3);
  return tile < tx * ty && px < S.width && py < S.height;
}

// Path slot <-> (pixel slot, sample) inside one batch of `ns` samples.  An 8x8 pixel block owns 64 * ns consecutive slots = ns
// wavefronts.  With G = the largest power of two that divides ns (at most 64), a wavefront holds 64 / G pixels x G consecutive samples:
// ns = 1 (one Redraw): the 8x8 block, as ever; ns = 128 (the batch bench.py times): ONE pixel x 64 samples -- camera rays that differ
// only by their sub-pixel jitter walk the tree in lockstep and shade the same triangle (lane utilisation of the first launches), and
// the wavefronts in flight cover a few thousand pixels instead of a fifth of the image.  Only the ORDER of the slots changes: every
// path still owns (pixel, sample), seeds and per-pixel accumulation order are untouched, results are bit-identical.
#ifndef CRH_SAMPLE_GROUP_MAX
#define CRH_SAMPLE_GROUP_MAX 64
#endif
#ifndef CRH_SLOT_SAMPLE_MAJOR
#define CRH_SLOT_SAMPLE_MAJOR 0      // lane = pixel * G + sample (0) or sample * P + pixel (1) inside a wavefront's 64 slots
#endif
__device__ __forceinline__ uint32_t sample_group(uint32_t ns) { return min(ns & (0u - ns), (uint32_t)CRH_SAMPLE_GROUP_MAX); }
__device__ __forceinline__ void slot_to_pixel_sample(uint32_t pid, uint32_t ns, uint32_t& local, uint32_t& s)
{
  const uint32_t G = sample_group(ns), lg = 31u - (uint32_t)__clz((int)G);      // G is a power of two
  const uint32_t B = pid / (64u * ns), r = pid - B * 64u * ns, w = r >> 6, l = r & 63u;
  const uint32_t P = 64u >> lg;
  const uint32_t c = w >> lg, b = w & (G - 1u), pi_ = CRH_SLOT_SAMPLE_MAJOR ? l & (P - 1u) : l >> lg, si = CRH_SLOT_SAMPLE_MAJOR ? l / P : l & (G - 1u);
  local = B * 64u + b * P + pi_;
  s = (c << lg) + si;
}
__device__ __forceinline__ uint32_t pixel_sample_to_slot(uint32_t local, uint32_t s, uint32_t ns)
{
  const uint32_t G = sample_group(ns), lg = 31u - (uint32_t)__clz((int)G), P = 64u >> lg;
  const uint32_t B = local >> 6, p = local & 63u, b = p / P, pi_ = p - b * P, c = s >> lg, si = s & (G - 1u);
  return B * 64u * ns + (((c << lg) + b) << 6) + (CRH_SLOT_SAMPLE_MAJOR ? si * P + pi_ : (pi_ << lg) + si);
}

// The camera ray of pixel (px, py), sample s of the batch (a2 / a14): rng seed, sub-pixel jitter, pinhole / orthographic / frustum-corner ray, thin lens.  Shared by
// k_raygen and the frame kernel's own ray generation (k_frame.h).  fseed: the frame seed of the sample (whole-frame passes) or of the tile (adaptive passes).
__device__ __forceinline__ void camera_ray(const DScene& S, const uint32_t fseed, const uint32_t px, const uint32_t py, v3& o, v3& d, uint32_t& rng)
{
  const uint32_t pix = S.coherent ? ((py / 16u) * ((S.width + 15u) / 16u) + (px / 16u)) : (py * S.width + px);
  rng = crh_rng_seed(pix, fseed);
  const float jx = crh_rng_next_mode(&rng, S.spec_u32), jy = crh_rng_next_mode(&rng, S.spec_u32);
  const float nx = CRH_FMA(((float)px + jx) / (float)S.width, 2.0f, -1.0f);
  const float ny = CRH_FMA(((float)py + jy) / (float)S.height, -2.0f, 1.0f);
  if (S.is_ortho) {
    const float sx = (nx * S.ortho_scale) * S.aspect, sy = ny * S.ortho_scale;
    o = crh_madd3(crh_madd3(S.eye, S.right, sx), S.up, sy);
    d = S.fwd;
  } else if (S.spec_raygen) {
    // crh_spec.h #13 (SURVEY a2, Appendix A GenerateRay): blend of the four frustum-corner directions by the pixel's position in [0,1]^2
    const float u = ((float)px + jx) / (float)S.width, v = 1.0f - ((float)py + jy) / (float)S.height;
    o = S.eye;
    d = crh_norm3(crh_lerp3(crh_lerp3(S.corner[0], S.corner[1], u), crh_lerp3(S.corner[2], S.corner[3], u), v));
  } else {
    const float sx = (nx * S.tan_half) * S.aspect, sy = ny * S.tan_half;
    o = S.eye;
    d = crh_norm3(crh_madd3(crh_madd3(S.fwd, S.right, sx), S.up, sy));
  }
  if (S.aperture > 0.f) {
    const float k1 = crh_rng_next_mode(&rng, S.spec_u32), k2 = crh_rng_next_mode(&rng, S.spec_u32);
    const float ft = S.focal / crh_dot3(d, S.fwd);
    const v3 focus = crh_madd3(o, d, ft);
    const float r = S.aperture * crh_sqrt(k1); float sn, cs; crh_sincos2pi(k2, &sn, &cs);
    o = crh_madd3(crh_madd3(o, S.right, r * cs), S.up, r * sn);
    d = crh_norm3(crh_sub3(focus, o));
  }
}

// frame seeds by value: a small batch (<= 16 samples) whose seeds the host did not stage in HBM (`seeds` == nullptr) -- the batch's tracing then reads nothing that
// was uploaded for it and need not wait for the context's stream (the frame pipeline, crh_schedule.cpp)
struct SeedVals { uint32_t v[16]; };
// SPLIT: the instantiation for split scenes (static tree + moved objects) also lists the rays that touch a moved object; the plain one carries none of it
template <bool SPLIT>
__global__ __launch_bounds__(kBlock) void k_raygen(DScene S, DPaths P, uint32_t* __restrict__ q, uint32_t* __restrict__ count,
                                                    uint32_t* __restrict__ q2, uint32_t* __restrict__ count2,
                                                    uint32_t* __restrict__ cursors,
                                                    const uint32_t* __restrict__ tile_ids, uint32_t n_tiles,
                                                    const uint32_t* __restrict__ seeds, uint32_t n_samples, int seed_per_tile,
                                                    const uint32_t* __restrict__ n_tiles_dev, SeedVals sv)
{
  if (n_tiles_dev) n_tiles = *n_tiles_dev;          // the tile list was drawn on the device (adaptive sampling): its length lives there too
  // Queue space is reserved ONCE per chunk of kGenIters x 256 slots: pass 1 counts the slots that map to a pixel
  // inside the image (edge tiles are partial), one atomic reserves the range, pass 2 generates the rays and writes
  // their ids at exclusive-scan offsets.  (Per-workgroup appends were atomic-rate bound: 261 K atomics per 67 M paths.)
  __shared__ uint32_t s_cnt[kGenIters * 4];
  __shared__ uint32_t s_base;
  // split scenes: the camera rays that touch a moved object are listed for the second traversal pass (collected per chunk, one atomic per chunk)
  __shared__ uint32_t s_q2[SPLIT ? kGenIters * kBlock : 1];
  __shared__ uint32_t s_n2, s_b2;
  if (blockIdx.x == 0 && threadIdx.x == 0) { cursors[0] = 0u; cursors[1] = 0u; cursors[2] = 0u; cursors[4] = 0u; cursors[5] = 0u; cursors[7] = 0u; cursors[8] = 0u; }      // [7], [8]: count and cursor of the packet kernel's fall-back queue
  if (threadIdx.x == 0) s_n2 = 0u;
  const uint32_t per_sample = n_tiles * S.tile_size * S.tile_size;
  const uint32_t total = per_sample * n_samples;
  const uint32_t chunk = kGenIters * kBlock;
  const uint32_t lane = lane_id(), wv = threadIdx.x >> 6;
  for (uint32_t cbase = blockIdx.x * chunk; cbase < total; cbase += gridDim.x * chunk) {
    for (uint32_t it = 0; it < kGenIters; ++it) {
      const uint32_t pid = cbase + it * kBlock + threadIdx.x;
      bool valid = pid < total;
      uint32_t px, py;
      if (valid) { uint32_t local, s; slot_to_pixel_sample(pid, n_samples, local, s); valid = slot_pixel(S, tile_ids, local, px, py); }
      const unsigned long long m = __ballot(valid);
      if (lane == 0) s_cnt[it * 4u + wv] = (uint32_t)__popcll(m);
    }
    __syncthreads();
    if (wv == 0) {                                  // exclusive scan of the kGenIters*4 (= 128) counters by one wavefront
      const uint32_t a = s_cnt[2u * lane], b = s_cnt[2u * lane + 1u];
      uint32_t incl = a + b;
      for (int o = 1; o < 64; o <<= 1) { const uint32_t t = __shfl_up(incl, o); if ((int)lane >= o) incl += t; }
      const uint32_t excl = incl - (a + b);
      s_cnt[2u * lane] = excl; s_cnt[2u * lane + 1u] = excl + a;
      if (lane == 63) s_base = incl ? atomicAdd(count, incl) : 0u;
    }
    __syncthreads();
    for (uint32_t it = 0; it < kGenIters; ++it) {
    const uint32_t pid = cbase + it * kBlock + threadIdx.x;
    bool valid = pid < total, flagged = false;
    uint32_t px = 0, py = 0, s = 0, local = 0;
    if (valid) { slot_to_pixel_sample(pid, n_samples, local, s); valid = slot_pixel(S, tile_ids, local, px, py); }
    if (valid) {
      v3 o, d; uint32_t rng;
      // whole-frame passes share one frame seed per sample; adaptive passes give every tile its own sample index
      camera_ray(S, seeds ? (seed_per_tile ? seeds[local / (S.tile_size * S.tile_size)] : seeds[s]) : sv.v[s & 15u], px, py, o, d, rng);
      P.ray_o[0][pid] = mk4(o, __uint_as_float(rng));           // .w = rng state; position = path slot at bounce 0
      P.ray_d[0][pid] = mk4(d, __uint_as_float(pid << 1));      // .w = (path slot << 1) | inside-a-medium flag
      if (SPLIT) flagged = ray_touches_instances(S, o, d, CRH_MAXFLOAT);
      // throughput (1,1,1 | no pending pdf) and radiance (0) are NOT written here: every generated path goes through
      // the bounce-0 k_shade, which takes them as constants and writes the radiance record unconditionally
    }
    const unsigned long long m = __ballot(valid);
    if (valid) q[s_base + s_cnt[it * 4u + wv] + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = pid;
    if (SPLIT) lds_append(flagged, pid, s_q2, &s_n2);
    }
    __syncthreads();
    if (SPLIT) {
      if (threadIdx.x == 0) { s_b2 = s_n2 ? atomicAdd(count2, s_n2) : 0u; }
      __syncthreads();
      for (uint32_t j = threadIdx.x; j < s_n2; j += kBlock) q2[s_b2 + j] = s_q2[j];
      __syncthreads();
      if (threadIdx.x == 0) s_n2 = 0u;
    }
  }
}

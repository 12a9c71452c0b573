// k_lights_env.h -- part of kernels.hip (ONE translation unit: included there inside namespace crh::(anonymous), in this order: k_common, k_traversal, k_packets, k_bsdf,
// k_lights_env, k_raygen, k_shade, k_accumulate).  Shading frames, light sampling and intersection, the environment map.
// ================================================================== frames, lights, environment
struct Frame { v3 t, b, n; };
__device__ __forceinline__ Frame make_frame(v3 n)
{
  Frame f; f.n = n;
  const v3 t = (crh_abs(n.x) > crh_abs(n.z)) ? crh_mk3(-n.y, n.x, 0.f) : crh_mk3(0.f, -n.z, n.y);
  f.t = crh_norm3(t); f.b = crh_cross3(n, f.t);
  return f;
}
__device__ __forceinline__ v3 to_local(const Frame& f, v3 v) { return crh_mk3(crh_dot3(v, f.t), crh_dot3(v, f.b), crh_dot3(v, f.n)); }
__device__ __forceinline__ v3 from_local(const Frame& f, v3 l)
{
  return crh_mk3(CRH_FMA(f.n.x, l.z, CRH_FMA(f.b.x, l.y, f.t.x * l.x)),
                 CRH_FMA(f.n.y, l.z, CRH_FMA(f.b.y, l.y, f.t.y * l.x)),
                 CRH_FMA(f.n.z, l.z, CRH_FMA(f.b.z, l.y, f.t.z * l.x)));
}
__device__ __forceinline__ float lerpf(float a, float b, float t) { return CRH_FMA(t, b - a, a); }

__device__ v3 env_lookup(const DScene& S, v3 d)
{
  if (!S.env) return crh_mk3(S.bg[0], S.bg[1], S.bg[2]);
  float u = (crh_atan2(d.y, d.x) + CRH_PI) * CRH_INV_TWOPI;
  float v = crh_acos(d.z) * CRH_INV_PI;
  if (S.spec_env_orient) { u = crh_atan2(d.y, d.x) * CRH_INV_TWOPI; v = crh_acos(-d.z) * CRH_INV_PI; }      // crh_spec.h #14
  const float x = CRH_FMA(u, (float)S.env_w, -0.5f), y = CRH_FMA(v, (float)S.env_h, -0.5f);
  float xf = (float)(int)x; if (xf > x) xf -= 1.0f;
  float yf = (float)(int)y; if (yf > y) yf -= 1.0f;
  const float fx = x - xf, fy = y - yf;
  const int W = (int)S.env_w, H = (int)S.env_h;
  int x0 = (int)xf % W; if (x0 < 0) x0 += W;
  int x1 = x0 + 1; if (x1 >= W) x1 = 0;
  int y0 = (int)yf; int y1 = y0 + 1;
  if (y0 < 0) y0 = 0; if (y0 > H - 1) y0 = H - 1; if (y1 < 0) y1 = 0; if (y1 > H - 1) y1 = H - 1;
  const float4 p00 = S.env[y0 * W + x0], p10 = S.env[y0 * W + x1], p01 = S.env[y1 * W + x0], p11 = S.env[y1 * W + x1];
  v3 r = crh_mk3(lerpf(lerpf(p00.x, p10.x, fx), lerpf(p01.x, p11.x, fx), fy),
                 lerpf(lerpf(p00.y, p10.y, fx), lerpf(p01.y, p11.y, fx), fy),
                 lerpf(lerpf(p00.z, p10.z, fx), lerpf(p01.z, p11.z, fx), fy));
  if (S.spec_gamma2) r = crh_mul3(r, r);            // crh_spec.h #2: the filtered texel squared
  return r;
}

__device__ __forceinline__ float cone_pdf(float cosmax) { return 1.0f / (CRH_TWO_PI * (1.0f - cosmax)); }
__device__ __forceinline__ float sphere_cosmax(float radius, float dist)
{ const float q = radius / dist; return 1.0f / crh_sqrt(CRH_FMA(q, q, 1.0f)); }

__device__ v3 intersect_light(const DScene& S, v3 o, v3 d, uint32_t bounce, float hit_t, float& exp_pdf)
{
  v3 rad = crh_mk3(0.f, 0.f, 0.f); float pdf = 0.f; float hd = hit_t;
  const float sel = S.n_lights ? 1.0f / (float)S.n_lights : 0.f;
  for (uint32_t i = 0; i < S.n_lights; ++i) {
    const float4 l0 = S.lights[2u * i], l1 = S.lights[2u * i + 1u];
    if (l0.w != 0.f) {
      const v3 tl = crh_sub3(xyz(l0), o);
      const float dist = crh_len3(tl);
      if (dist < hd) {
        const float cm = sphere_cosmax(l1.w, dist);
        if (cm < 1.0f && crh_dot3(d, tl) * (1.0f / dist) >= cm) { hd = dist; rad = xyz(l1); pdf = sel * cone_pdf(cm); }
      }
    } else if (hd == CRH_MAXFLOAT) {
      const float cm = l1.w;
      if (cm < 1.0f && crh_dot3(d, xyz(l0)) >= cm) { rad = crh_add3(rad, xyz(l1)); pdf += sel * cone_pdf(cm); }
    }
  }
  if (pdf == 0.f && hd == CRH_MAXFLOAT) {
    if (bounce == 0u && !S.env_as_bg) rad = crh_mk3(S.bg[0], S.bg[1], S.bg[2]);
    else rad = env_lookup(S, d);
  }
  exp_pdf = pdf;
  return rad;
}

__device__ __forceinline__ v3 offset_origin(v3 p, v3 dir, v3 ng, float eps)
{
  const v3 o = crh_madd3(p, dir, eps);
  const float s = crh_dot3(ng, dir) >= 0.f ? eps : -eps;
  return crh_madd3(o, ng, s);
}

// Diffuse texture lookup: bilinear, repeat wrap, row 0 of the image = v 1.  The call site is behind
// a wave-uniform "any texture bound" test.
__device__ __forceinline__ float4 sample_texture(const DScene& S, uint32_t slot, uint32_t tri, float bu, float bv, float w0, float sc_s, float sc_t)
{
  if (slot >= S.n_tex || !S.uvs) return make_float4(1.f, 1.f, 1.f, 1.f);
  const uint4 td = S.tex_desc[slot];
  if (td.y == 0u) return make_float4(1.f, 1.f, 1.f, 1.f);
  const float4 ua = S.uvs[2u * tri], ub = S.uvs[2u * tri + 1u];
  const float ss = sc_s != 0.f ? sc_s : 1.0f, st_ = sc_t != 0.f ? sc_t : 1.0f;
  float us = CRH_FMA(ub.x, bv, CRH_FMA(ua.z, bu, ua.x * w0)) * ss;
  float vs = CRH_FMA(ub.y, bv, CRH_FMA(ua.w, bu, ua.y * w0)) * st_;
  // beyond 2^22 a float has no fraction left worth sampling and (int) would saturate (texel index out of range): wrap to 0
  if (!(crh_abs(us) < 4194304.0f)) us = 0.f;
  if (!(crh_abs(vs) < 4194304.0f)) vs = 0.f;
  float uf = (float)(int)us; if (uf > us) uf -= 1.0f;
  float vf = (float)(int)vs; if (vf > vs) vf -= 1.0f;
  const float x = CRH_FMA(us - uf, (float)td.y, -0.5f), y = CRH_FMA(1.0f - (vs - vf), (float)td.z, -0.5f);
  float xf = (float)(int)x; if (xf > x) xf -= 1.0f;
  float yf = (float)(int)y; if (yf > y) yf -= 1.0f;
  const float fx = x - xf, fy = y - yf;
  const int W = (int)td.y, H = (int)td.z;
  int x0 = (int)xf; if (x0 < 0) x0 += W; if (x0 >= W) x0 -= W;
  int x1 = x0 + 1; if (x1 >= W) x1 = 0;
  int y0 = (int)yf; if (y0 < 0) y0 += H; if (y0 >= H) y0 -= H;
  int y1 = y0 + 1; if (y1 >= H) y1 = 0;
  const float4* tb = S.texels + td.x;
  const float4 p00 = tb[y0 * W + x0], p10 = tb[y0 * W + x1], p01 = tb[y1 * W + x0], p11 = tb[y1 * W + x1];
  float4 r = make_float4(lerpf(lerpf(p00.x, p10.x, fx), lerpf(p01.x, p11.x, fx), fy),
                         lerpf(lerpf(p00.y, p10.y, fx), lerpf(p01.y, p11.y, fx), fy),
                         lerpf(lerpf(p00.z, p10.z, fx), lerpf(p01.z, p11.z, fx), fy),
                         lerpf(lerpf(p00.w, p10.w, fx), lerpf(p01.w, p11.w, fx), fy));   // RGB images are stored with alpha 1
  if (S.spec_gamma2) { r.x *= r.x; r.y *= r.y; r.z *= r.z; }      // crh_spec.h #2 (the alpha is a coverage, never squared)
  return r;
}

constexpr uint32_t kGenIters = 32;     // k_raygen: 32 x 256 = 8192 path slots per queue reservation
static_assert(kGenIters * 4 == 128, "k_raygen scans its 128 (iteration, wave) counters with one wavefront, two per lane");
#ifndef CRH_SHADE_ITERS
#define CRH_SHADE_ITERS 4
#endif
constexpr uint32_t kShadeIters = CRH_SHADE_ITERS;    // k_shade : 4 x 256 = 1024 paths per cursor fetch / queue reservation

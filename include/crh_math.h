/* crh_math.h -- the elementary arithmetic of the path-tracing hot path.
 *
 * This header is the *definition of arithmetic* for the renderer: every float
 * operation whose rounding could steer a path (ray/box, ray/triangle, BSDF lobes,
 * light sampling, environment lookup) is built from
 *   - IEEE-754 binary32 add / sub / mul / div / sqrt (correctly rounded),
 *   - explicit fused multiply-add (CRH_FMA) where written, and nowhere else,
 *   - the fixed polynomials below for sin/cos/exp/log/pow/acos/atan2.
 * It is compiled by gcc (CPU oracle, host code) and by hipcc (gfx950 kernels) with
 * -ffp-contract=off and no fast-math, so both sides execute the same rounding
 * sequence and take the same branches.  No libm / ocml transcendental is called on
 * the render path.
 *
 * The reference (CADRays) holds none of this arithmetic: it lives in OCCT's GLSL
 * (SURVEY.md section 0); the polynomials here are this project's frozen spec.
 */
#ifndef CRH_MATH_H
#define CRH_MATH_H

#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>   /* __host__ / __device__ / __forceinline__ */
#define CRH_HD __host__ __device__ __forceinline__
#else
#define CRH_HD static inline __attribute__((always_inline))
#endif

#define CRH_FMA(a, b, c) __builtin_fmaf((a), (b), (c))

#define CRH_PI        3.14159265358979323846f
#define CRH_TWO_PI    6.28318530717958647692f
#define CRH_INV_PI    0.31830988618379067154f
#define CRH_INV_TWOPI 0.15915494309189533577f
#define CRH_MAXFLOAT  1.0e15f   /* "infinite" distance / delta-pdf marker */

typedef struct { float x, y, z; } crh_v3;

/* ------------------------------------------------------------------ bits */
CRH_HD uint32_t crh_f2u(float f) { union { float f; uint32_t u; } c; c.f = f; return c.u; }
CRH_HD float    crh_u2f(uint32_t u) { union { float f; uint32_t u; } c; c.u = u; return c.f; }

CRH_HD float crh_abs(float a) { return crh_u2f(crh_f2u(a) & 0x7fffffffu); }
CRH_HD float crh_min(float a, float b) { return a < b ? a : b; }
CRH_HD float crh_max(float a, float b) { return a > b ? a : b; }
CRH_HD float crh_clamp(float a, float lo, float hi) { return crh_min(crh_max(a, lo), hi); }
CRH_HD float crh_sqrt(float a) { return __builtin_sqrtf(a); }
/* sign(x) in {-1, 0, +1} (GLSL semantics) */
CRH_HD float crh_sign(float a) { return a > 0.f ? 1.f : (a < 0.f ? -1.f : 0.f); }

/* ------------------------------------------------------------------ vec3 */
CRH_HD crh_v3 crh_mk3(float x, float y, float z) { crh_v3 r; r.x = x; r.y = y; r.z = z; return r; }
CRH_HD crh_v3 crh_add3(crh_v3 a, crh_v3 b) { return crh_mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
CRH_HD crh_v3 crh_sub3(crh_v3 a, crh_v3 b) { return crh_mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
CRH_HD crh_v3 crh_mul3(crh_v3 a, crh_v3 b) { return crh_mk3(a.x * b.x, a.y * b.y, a.z * b.z); }
CRH_HD crh_v3 crh_scale3(crh_v3 a, float s) { return crh_mk3(a.x * s, a.y * s, a.z * s); }
/* a + b*s, one fma per component */
CRH_HD crh_v3 crh_madd3(crh_v3 a, crh_v3 b, float s)
{ return crh_mk3(CRH_FMA(b.x, s, a.x), CRH_FMA(b.y, s, a.y), CRH_FMA(b.z, s, a.z)); }
/* dot = fma(az,bz, fma(ay,by, ax*bx)) */
CRH_HD float crh_dot3(crh_v3 a, crh_v3 b) { return CRH_FMA(a.z, b.z, CRH_FMA(a.y, b.y, a.x * b.x)); }
/* cross component = fma(p,q, -(r*s)) */
CRH_HD crh_v3 crh_cross3(crh_v3 a, crh_v3 b)
{
  return crh_mk3(CRH_FMA(a.y, b.z, -(a.z * b.y)),
                 CRH_FMA(a.z, b.x, -(a.x * b.z)),
                 CRH_FMA(a.x, b.y, -(a.y * b.x)));
}
CRH_HD float crh_len3(crh_v3 a) { return crh_sqrt(crh_dot3(a, a)); }
/* normalize = a * (1 / sqrt(dot)); a zero vector stays zero-ish (inf*0 avoided) */
CRH_HD crh_v3 crh_norm3(crh_v3 a)
{
  float l2 = crh_dot3(a, a);
  float inv = l2 > 0.f ? 1.0f / crh_sqrt(l2) : 0.f;
  return crh_scale3(a, inv);
}
CRH_HD float crh_maxcomp3(crh_v3 a) { return crh_max(a.x, crh_max(a.y, a.z)); }
/* a + (b - a) * t, one subtraction and one fma per component */
CRH_HD crh_v3 crh_lerp3(crh_v3 a, crh_v3 b, float t) { return crh_madd3(a, crh_sub3(b, a), t); }
/* crh_spec.h #13: direction through the frustum corner (sx, sy) in {-1, +1}^2 -- the expression the tan form of ray generation evaluates at ndc = (sx, sy);
 * unit != 0: normalised (what a host uploads when it hands the shader unit vectors) */
CRH_HD crh_v3 crh_frustum_corner(crh_v3 fwd, crh_v3 right, crh_v3 up, float tan_half, float aspect, float sx, float sy, int unit)
{
  const crh_v3 d = crh_madd3(crh_madd3(fwd, right, (sx * tan_half) * aspect), up, sy * tan_half);
  return unit ? crh_norm3(d) : d;
}

/* ---- split scenes (static tree + moved objects): "does the ray come near a moved object at all?"  (spec)
 * Every moved object's world box is wrapped in a sphere {centre, padded radius}; a ray is sent through the top-level tree only if the part
 * [0, tmax] of it comes within the sphere of at least one of them (first: within the sphere around ALL of them).  No reciprocal direction, no
 * division: the kernels that PRODUCE rays ask this for every ray they write.  The direction must be of unit length (path rays are; rays handed in
 * through the API are not asked).  Conservative by construction: triangles of a moved object lie inside its world box, the box inside the sphere;
 * the radius is padded for the rounding of the box itself and of the ray's transformation into object space (both scale with the coordinates'
 * magnitude), the comparison for the rounding of the closest-point evaluation (scales with the distance to the centre). */
CRH_HD void crh_box_sphere(const float* lo, const float* hi, float* s4)
{
  const float hx = (hi[0] - lo[0]) * 0.5f, hy = (hi[1] - lo[1]) * 0.5f, hz = (hi[2] - lo[2]) * 0.5f;
  const float far_ = crh_max(crh_max(crh_max(crh_abs(lo[0]), crh_abs(hi[0])), crh_max(crh_abs(lo[1]), crh_abs(hi[1]))), crh_max(crh_abs(lo[2]), crh_abs(hi[2])));
  const float r = crh_sqrt(CRH_FMA(hz, hz, CRH_FMA(hy, hy, hx * hx)));
  s4[0] = (lo[0] + hi[0]) * 0.5f; s4[1] = (lo[1] + hi[1]) * 0.5f; s4[2] = (lo[2] + hi[2]) * 0.5f;
  s4[3] = CRH_FMA(far_, 7.62939453125e-06f, r * 1.00000762939453125f);          /* r (1 + 2^-17) + 2^-17 * largest |coordinate| */
}
CRH_HD int crh_ray_near_sphere(crh_v3 o, crh_v3 d, float tmax, float cx, float cy, float cz, float r)
{
  const crh_v3 v = crh_mk3(cx - o.x, cy - o.y, cz - o.z);
  const float t = crh_min(crh_max(crh_dot3(v, d), 0.f), tmax);                  /* parameter of the point of [0, tmax] closest to the centre */
  const crh_v3 q = crh_mk3(CRH_FMA(-d.x, t, v.x), CRH_FMA(-d.y, t, v.y), CRH_FMA(-d.z, t, v.z));
  const float m = crh_max(crh_max(crh_abs(v.x), crh_abs(v.y)), crh_abs(v.z));
  const float rr = CRH_FMA(m, 3.814697265625e-06f, r);                          /* + 2^-18 of the distance to the centre */
  return crh_dot3(q, q) <= rr * rr;
}

/* ------------------------------------------------------------------ sin/cos */
/* core polynomials on [-pi/4, pi/4] */
CRH_HD float crh__sin_poly(float a)
{
  float z = a * a;
  float p = CRH_FMA(z, -1.9515295891e-4f, 8.3321608736e-3f);
  p = CRH_FMA(z, p, -1.6666654611e-1f);
  return CRH_FMA(a * z, p, a);
}
CRH_HD float crh__cos_poly(float a)
{
  float z = a * a;
  float p = CRH_FMA(z, 2.443315711809948e-5f, -1.388731625493765e-3f);
  p = CRH_FMA(z, p, 4.166664568298827e-2f);
  return CRH_FMA(z * z, p, CRH_FMA(z, -0.5f, 1.0f));
}
/* sin and cos of 2*pi*x for x in [0, 1] (the only form the samplers need). */
CRH_HD void crh_sincos2pi(float x, float* s, float* c)
{
  int   q = (int)CRH_FMA(x, 4.0f, 0.5f);          /* nearest quarter turn, 0..4 */
  float r = CRH_FMA((float)q, -0.25f, x);          /* exact, |r| <= 1/8         */
  float a = r * CRH_TWO_PI;
  float sp = crh__sin_poly(a), cp = crh__cos_poly(a);
  switch (q & 3)
  {
    case 0:  *s =  sp; *c =  cp; break;
    case 1:  *s =  cp; *c = -sp; break;
    case 2:  *s = -sp; *c = -cp; break;
    default: *s = -cp; *c =  sp; break;
  }
}
/* sin/cos of an angle in radians, |a| < ~1e4 (host-side parameter conversion:
 * light cone angle, camera fov) */
CRH_HD void crh_sincos(float a, float* s, float* c)
{
  float t = a * CRH_INV_TWOPI;
  float fl = (float)(int)t; if (fl > t) fl -= 1.0f;  /* floor */
  crh_sincos2pi(t - fl, s, c);
}

/* ------------------------------------------------------------------ exp/log/pow */
/* natural log, x > 0 finite normal; x <= 0 returns -CRH_MAXFLOAT */
CRH_HD float crh_log(float x)
{
  if (!(x > 0.f)) return -CRH_MAXFLOAT;
  uint32_t u = crh_f2u(x);
  int e = (int)(u >> 23) - 127;
  if (e == -127) { /* subnormal: scale up by 2^23 */
    u = crh_f2u(x * 8388608.0f); e = (int)(u >> 23) - 127 - 23;
  }
  float m = crh_u2f((u & 0x007fffffu) | 0x3f800000u);   /* [1,2) */
  if (m > 1.41421356237f) { m *= 0.5f; e += 1; }         /* [sqrt(.5), sqrt(2)) */
  float f = m - 1.0f;
  float z = f * f;
  float p = 7.0376836292e-2f;
  p = CRH_FMA(p, f, -1.1514610310e-1f);
  p = CRH_FMA(p, f,  1.1676998740e-1f);
  p = CRH_FMA(p, f, -1.2420140846e-1f);
  p = CRH_FMA(p, f,  1.4249322787e-1f);
  p = CRH_FMA(p, f, -1.6668057665e-1f);
  p = CRH_FMA(p, f,  2.0000714765e-1f);
  p = CRH_FMA(p, f, -2.4999993993e-1f);
  p = CRH_FMA(p, f,  3.3333331174e-1f);
  float y = (f * z) * p;
  float fe = (float)e;
  y = CRH_FMA(fe, -2.12194440e-4f, y);
  y = CRH_FMA(z, -0.5f, y);
  float r = f + y;
  return CRH_FMA(fe, 0.693359375f, r);
}
/* e^x; x clamped to [-87, 88] */
CRH_HD float crh_exp(float x)
{
  x = crh_clamp(x, -87.0f, 88.0f);
  float t = CRH_FMA(x, 1.44269504088896341f, 0.5f);
  float n = (float)(int)t; if (n > t) n -= 1.0f;          /* floor(x*log2e + .5) */
  x = CRH_FMA(n, -0.693359375f, x);
  x = CRH_FMA(n,  2.12194440e-4f, x);
  float z = x * x;
  float p = 1.9875691500e-4f;
  p = CRH_FMA(p, x, 1.3981999507e-3f);
  p = CRH_FMA(p, x, 8.3334519073e-3f);
  p = CRH_FMA(p, x, 4.1665795894e-2f);
  p = CRH_FMA(p, x, 1.6666665459e-1f);
  p = CRH_FMA(p, x, 5.0000001201e-1f);
  float y = CRH_FMA(p, z, x) + 1.0f;
  int in = (int)n;                                        /* [-126, 128] */
  /* scale by 2^n in two exact steps so that n = 128 and n = -126 stay finite */
  int h = in / 2;
  float s1 = crh_u2f((uint32_t)(h + 127) << 23);
  float s2 = crh_u2f((uint32_t)(in - h + 127) << 23);
  return (y * s1) * s2;
}
/* x^y for x >= 0.  0^y = 0 (y > 0), x^0 = 1. */
CRH_HD float crh_pow(float x, float y)
{
  if (y == 0.f) return 1.0f;
  if (!(x > 0.f)) return 0.f;
  return crh_exp(y * crh_log(x));
}

/* ------------------------------------------------------------------ acos / atan2 */
CRH_HD float crh__asin_core(float x, float z) /* x + x*z*P(z), |x| <= 0.5 */
{
  float p = 4.2163199048e-2f;
  p = CRH_FMA(p, z, 2.4181311049e-2f);
  p = CRH_FMA(p, z, 4.5470025998e-2f);
  p = CRH_FMA(p, z, 7.4953002686e-2f);
  p = CRH_FMA(p, z, 1.6666752422e-1f);
  return CRH_FMA(x * z, p, x);
}
/* acos on [-1, 1] (argument clamped) -> [0, pi] */
CRH_HD float crh_acos(float x)
{
  x = crh_clamp(x, -1.0f, 1.0f);
  if (x > 0.5f)  { float z = 0.5f * (1.0f - x); float s = crh_sqrt(z); return 2.0f * crh__asin_core(s, z); }
  if (x < -0.5f) { float z = 0.5f * (1.0f + x); float s = crh_sqrt(z); return CRH_FMA(-2.0f, crh__asin_core(s, z), CRH_PI); }
  return 1.57079632679489661923f - crh__asin_core(x, x * x);
}
CRH_HD float crh__atan_pos(float x) /* x >= 0 */
{
  float y0;
  if (x > 2.414213562373095f)       { y0 = 1.57079632679489661923f; x = -(1.0f / x); }
  else if (x > 0.4142135623730950f) { y0 = 0.78539816339744830962f; x = (x - 1.0f) / (x + 1.0f); }
  else                              { y0 = 0.f; }
  float z = x * x;
  float p = 8.05374449538e-2f;
  p = CRH_FMA(p, z, -1.38776856032e-1f);
  p = CRH_FMA(p, z,  1.99777106478e-1f);
  p = CRH_FMA(p, z, -3.33329491539e-1f);
  return y0 + CRH_FMA(p * z, x, x);
}
/* atan2(y, x) -> (-pi, pi]; atan2(0,0) = 0 */
CRH_HD float crh_atan2(float y, float x)
{
  if (x == 0.f && y == 0.f) return 0.f;
  float ax = crh_abs(x), ay = crh_abs(y);
  float a;
  if (ax == 0.f) a = 1.57079632679489661923f;
  else           a = crh__atan_pos(ay / ax);
  if (x < 0.f) a = CRH_PI - a;
  return y < 0.f ? -a : a;
}

/* ------------------------------------------------------------------ RNG */
/* Per-path stream: Wang hash of (pixel index + frame seed), then xorshift32
 * (13,17,5); uniform float = top 24 bits * 2^-24, in [0, 1).
 * (OCCT multiplies the full 32-bit state by 2^-32, which can round
 * to exactly 1.0 -- deliberately not reproduced.) */
CRH_HD uint32_t crh_wang_hash(uint32_t s)
{
  s = (s ^ 61u) ^ (s >> 16);
  s *= 9u;
  s = s ^ (s >> 4);
  s *= 0x27d4eb2du;
  s = s ^ (s >> 15);
  return s;
}
CRH_HD uint32_t crh_rng_seed(uint32_t pixel_index, uint32_t frame_seed)
{
  uint32_t s = crh_wang_hash(pixel_index + frame_seed);
  return s ? s : 0x9e3779b9u;   /* xorshift must not start at 0 */
}
CRH_HD float crh_rng_next(uint32_t* state)
{
  uint32_t s = *state;
  s ^= s << 13; s ^= s >> 17; s ^= s << 5;
  *state = s;
  return (float)(s >> 8) * 5.9604644775390625e-8f;  /* 2^-24 */
}
/* the same stream with crh_spec.uniform_32bit (crh_spec.h #1) selectable: full32 != 0 -> float(state) * 2^-32 (round to nearest;
 * the top 128 states give exactly 1.0), as recollected from OCCT's RandFloat() */
CRH_HD float crh_rng_next_mode(uint32_t* state, int full32)
{
  uint32_t s = *state;
  s ^= s << 13; s ^= s >> 17; s ^= s << 5;
  *state = s;
  return full32 ? (float)s * 2.3283064365386963e-10f : (float)(s >> 8) * 5.9604644775390625e-8f;
}

#endif /* CRH_MATH_H */

/* brute_force.c -- independent geometric ground truth for the traversal tests.  TEST INFRASTRUCTURE ONLY (loaded by tests/).
 *
 * No BVH, no shared headers: every ray is tested against every triangle.
 *   bf_nearest_f64   Moeller-Trumbore in double precision (a different formula from the product's n = (v0-v2) x (v1-v0) form):
 *                    the geometric truth, compared with a tolerance.
 *   bf_nearest_f32   the product's triangle test (DESIGN.md section 3 "Triangle (a7)"), restated here in plain float with the
 *                    same operation order (dot = fma(z,z, fma(y,y, x*x)), cross component = fma(p,q, -(r*s))), over ALL triangles in input order, first of equal t wins the tie like a front-to-back
 *                    walk cannot promise -- so the caller compares t/u/v bit-for-bit and the triangle index modulo exact t ties.
 *                    BVH traversal == this on every ray means the quantised boxes never cull a triangle the ray hits.
 * rays: 8 floats each {o.xyz, tmax, d.xyz, -}; tri_pos: 9 floats per triangle (world space); out: {t, u, v, index (as float bits)}.
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

#define BF_API __attribute__((visibility("default")))

BF_API void bf_nearest_f64(const float* tri_pos, uint32_t nT, const float* rays, uint32_t nR, double* out_tuv, int32_t* out_idx,
                           double* out_margin)
{
#pragma omp parallel for schedule(dynamic, 64)
  for (uint32_t r = 0; r < nR; ++r) {
    const double ox = rays[8 * r], oy = rays[8 * r + 1], oz = rays[8 * r + 2], tmax = rays[8 * r + 3];
    const double dx = rays[8 * r + 4], dy = rays[8 * r + 5], dz = rays[8 * r + 6];
    double best = tmax, bu = 0, bv = 0; int32_t bi = -1;
    for (uint32_t i = 0; i < nT; ++i) {
      const float* p = &tri_pos[9 * (size_t)i];
      const double e1x = (double)p[3] - p[0], e1y = (double)p[4] - p[1], e1z = (double)p[5] - p[2];
      const double e2x = (double)p[6] - p[0], e2y = (double)p[7] - p[1], e2z = (double)p[8] - p[2];
      const double px = dy * e2z - dz * e2y, py = dz * e2x - dx * e2z, pz = dx * e2y - dy * e2x;
      const double det = e1x * px + e1y * py + e1z * pz;
      if (det == 0.0) continue;
      const double inv = 1.0 / det;
      const double tx = ox - p[0], ty = oy - p[1], tz = oz - p[2];
      const double u = (tx * px + ty * py + tz * pz) * inv;
      if (u < 0.0 || u > 1.0) continue;
      const double qx = ty * e1z - tz * e1y, qy = tz * e1x - tx * e1z, qz = tx * e1y - ty * e1x;
      const double v = (dx * qx + dy * qy + dz * qz) * inv;
      if (v < 0.0 || u + v > 1.0) continue;
      const double t = (e2x * qx + e2y * qy + e2z * qz) * inv;
      if (t >= 0.0 && t < best) { best = t; bu = u; bv = v; bi = (int32_t)i; }
    }
    out_tuv[3 * (size_t)r] = best; out_tuv[3 * (size_t)r + 1] = bu; out_tuv[3 * (size_t)r + 2] = bv; out_idx[r] = bi;
    if (out_margin) {              /* distance of the hit from the triangle's edges in barycentric units (0 = on an edge) */
      double m = bu < bv ? bu : bv; const double w = 1.0 - bu - bv; if (w < m) m = w;
      out_margin[r] = bi >= 0 ? m : 1.0;
    }
  }
}

/* the product's float formula, operation by operation; built with -ffp-contract=off */
BF_API void bf_nearest_f32(const float* tri_pos, uint32_t nT, const float* rays, uint32_t nR, float* out)
{
#pragma omp parallel for schedule(dynamic, 64)
  for (uint32_t r = 0; r < nR; ++r) {
    const float ox = rays[8 * r], oy = rays[8 * r + 1], oz = rays[8 * r + 2];
    const float dx = rays[8 * r + 4], dy = rays[8 * r + 5], dz = rays[8 * r + 6];
    float best = rays[8 * r + 3], bu = 0.f, bv = 0.f; int32_t bi = -1;
    for (uint32_t i = 0; i < nT; ++i) {
      const float* p = &tri_pos[9 * (size_t)i];
      const float e0x = p[3] - p[0], e0y = p[4] - p[1], e0z = p[5] - p[2];          /* v1 - v0 */
      const float e1x = p[0] - p[6], e1y = p[1] - p[7], e1z = p[2] - p[8];          /* v0 - v2 */
      const float nx = fmaf(e1y, e0z, -(e1z * e0y)), ny = fmaf(e1z, e0x, -(e1x * e0z)), nz = fmaf(e1x, e0y, -(e1y * e0x));     /* e1 x e0 */
      const float tox = p[0] - ox, toy = p[1] - oy, toz = p[2] - oz;
      const float inv = 1.0f / fmaf(nz, dz, fmaf(ny, dy, nx * dx));
      const float cx = fmaf(dy, toz, -(dz * toy)), cy = fmaf(dz, tox, -(dx * toz)), cz = fmaf(dx, toy, -(dy * tox));           /* d x to */
      const float t = fmaf(nz, toz, fmaf(ny, toy, nx * tox)) * inv;
      const float u = fmaf(cz, e1z, fmaf(cy, e1y, cx * e1x)) * inv;
      const float v = fmaf(cz, e0z, fmaf(cy, e0y, cx * e0x)) * inv;
      if (t >= 0.f && u >= 0.f && v >= 0.f && (u + v) <= 1.0f && t < best) { best = t; bu = u; bv = v; bi = (int32_t)i; }
    }
    float* o = &out[4 * (size_t)r];
    o[0] = best; o[1] = bu; o[2] = bv; memcpy(&o[3], &bi, 4);
  }
}

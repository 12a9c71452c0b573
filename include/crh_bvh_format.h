/* crh_bvh_format.h -- the 4-wide BVH node (48 bytes of payload on a 64-byte stride) and its quantiser (data-format definition shared by the
 * host builder, the gfx950 traversal kernels and the CPU oracle, like crh_math.h for arithmetic).
 *
 * Node = 12 dwords = THREE dwordx4 fetches per visit (per-lane divergent loads cost ~0.7 TA cycles per lane and
 * instruction on MI355X -- tools/ubench/ta_rate.hip -- so the fourth fetch of the former explicit-reference node was
 * ~20 % of the address-processing time of a visit).  Nodes sit on a 64-B stride (dwords 12..15 are padding that is never
 * fetched): packed at 48 B half of them straddle two 64-B sectors, which costs more HBM traffic than the smaller array
 * saves (measured, Mrays/s at 64-B / 48-B stride: C3 2890 / 2836, C5 -- 10 M triangles, HBM-bound -- 2104 / 1955).
 *   [0..2]  origin.xyz (float)            minimum corner of the union of the child boxes
 *   [3]     kx | ky<<8 | kz<<16 | n_inner<<24 | n_children<<28
 *                                          per-axis grid step 2^k, k a SIGNED byte (two's complement, -126 .. 126): the traversal kernel scales
 *                                          the ray's reciprocal direction with one v_bfe_i32 + one v_ldexp_f32 per axis (it runs at the VALU issue
 *                                          limit; shift + mask + multiply for a biased exponent byte was three); child counts
 *   [4..6]  qlo_x, qlo_y, qlo_z            byte k = child k's lower bound on the grid:  origin + q * step
 *   [7..9]  qhi_x, qhi_y, qhi_z            byte k = child k's upper bound
 *   [10]    child_base                     slots 0 .. n_inner-1 are inner nodes child_base + slot (consecutive indices); 0 if n_inner = 0
 *   [11]    leaf_base                      slots n_inner .. n_children-1 are leaves with references leaf_base + (slot - n_inner):
 *                                          0x80000000 | triangle (ONE leaf-order triangle per leaf, consecutive triangles),
 *                                          or 0xF0000000 | position in the top-level tree's leaf-ordered instance list
 * Child references are therefore implicit: the builders number the inner children of a node consecutively and place
 * the triangles of its leaf children consecutively (inner children first, then leaves, each in collapse order).
 * Pair alignment: a block of >= 2 inner children starts on an EVEN node index (an all-zero slot is skipped when needed),
 * so the first two siblings share one 128-B L2 line.
 * The grid is conservative: origin + qlo*step <= true lower bound, origin + qhi*step >= true upper bound, so the
 * set of triangles a ray can reach is unchanged; only the number of visits grows slightly (<= 2/255 of the parent
 * extent per face).  Traversal evaluates a face as  t = fma((float)q, step * inv_d, fma(origin, inv_d, -o * inv_d)).
 */
#ifndef CRH_BVH_FORMAT_H
#define CRH_BVH_FORMAT_H

#include <stdint.h>
#include "crh_math.h"

#ifndef CRH_NODE_DWORDS
#define CRH_NODE_DWORDS 16          /* dwords between consecutive nodes (12 are used) */
#endif
/* triangles per leaf -- fixed by the format (a leaf reference is one triangle index).  Measured on MI355X with the
 * former explicit-reference node (C3, Mrays/s): 1 -> 2620, 2 -> 2521, 3 -> 2396, 4 -> 2295, 6 -> 2058: in a
 * triangle soup a multi-triangle leaf mostly buys failed tests; one extra inner level is cheaper. */
#define CRH_BVH_LEAF_SIZE 1
#define CRH_NODE_BYTES  (4 * CRH_NODE_DWORDS)
#define CRH_LEAF_TAG      0x80000000u
#define CRH_NODE_NINNER(w3)    (((w3) >> 24) & 7u)
#define CRH_NODE_NCHILDREN(w3) (((w3) >> 28) & 7u)
/* grid step exponent of axis a (0..2) as stored (signed byte) and as the biased exponent e = k + 127 the quantiser works with */
#define CRH_NODE_STEP_K(w3, a) ((int)(int8_t)(((w3) >> (8 * (a))) & 0xffu))
#define CRH_NODE_STEP_E(w3, a) ((uint32_t)(CRH_NODE_STEP_K(w3, a) + 127))

/* biased exponent byte E of the smallest power-of-two step with 255 * 2^(E-127) >= ext */
CRH_HD uint32_t crh_quant_exp(float ext)
{
  if (!(ext > 0.f)) return 1u;
  int e = (int)(crh_f2u(ext) >> 23) - 7;
  if (e < 1) e = 1;
  if (255.0f * crh_u2f((uint32_t)e << 23) < ext) e += 1;
  if (e > 253) e = 253;
  return (uint32_t)e;
}
CRH_HD float crh_quant_step(uint32_t e) { return crh_u2f(e << 23); }

/* lower bound: largest q in [0,255] with origin + q*step <= lo */
CRH_HD uint32_t crh_quant_lo(float lo, float origin, uint32_t e)
{
  const float step = crh_quant_step(e), inv = crh_u2f((254u - e) << 23);
  float t = (lo - origin) * inv;
  int q = t > 0.f ? (t < 255.0f ? (int)t : 255) : 0;
  while (q > 0 && (double)origin + (double)q * (double)step > (double)lo) --q;
  return (uint32_t)q;
}
/* upper bound: smallest q in [0,255] with origin + q*step >= hi */
CRH_HD uint32_t crh_quant_hi(float hi, float origin, uint32_t e)
{
  const float step = crh_quant_step(e), inv = crh_u2f((254u - e) << 23);
  float t = (hi - origin) * inv;
  int q = t > 0.f ? (t < 255.0f ? (int)t : 255) : 0;
  if ((float)q < t && q < 255) ++q;
  while (q < 255 && (double)origin + (double)q * (double)step < (double)hi) ++q;
  return (uint32_t)q;
}

/* Fill one node from <= 4 child boxes (cmin/cmax: [slot][axis], inner children first) and the two bases.  out: 12 dwords. */
CRH_HD void crh_pack_node(const float cmin[4][3], const float cmax[4][3], int n_inner, int n_children, uint32_t child_base,
                          uint32_t leaf_base, uint32_t out[CRH_NODE_DWORDS])
{
  float org[3], ext[3]; uint32_t e[3];
  for (int a = 0; a < 3; ++a) {
    float lo = 3.0e38f, hi = -3.0e38f;
    for (int k = 0; k < n_children; ++k) { if (cmin[k][a] < lo) lo = cmin[k][a]; if (cmax[k][a] > hi) hi = cmax[k][a]; }
    if (n_children == 0) { lo = 0.f; hi = 0.f; }
    org[a] = lo; ext[a] = hi - lo; e[a] = crh_quant_exp(ext[a]);
  }
  out[0] = crh_f2u(org[0]); out[1] = crh_f2u(org[1]); out[2] = crh_f2u(org[2]);
  out[3] = ((e[0] - 127u) & 0xffu) | (((e[1] - 127u) & 0xffu) << 8) | (((e[2] - 127u) & 0xffu) << 16) | ((uint32_t)n_inner << 24) | ((uint32_t)n_children << 28);
  for (int a = 0; a < 3; ++a) {
    uint32_t lo = 0u, hi = 0u;
    for (int k = 0; k < n_children; ++k) {
      lo |= crh_quant_lo(cmin[k][a], org[a], e[a]) << (8 * k);
      hi |= crh_quant_hi(cmax[k][a], org[a], e[a]) << (8 * k);
    }
    out[4 + a] = lo; out[7 + a] = hi;
  }
  out[10] = child_base; out[11] = leaf_base;
  for (int k = 12; k < CRH_NODE_DWORDS; ++k) out[k] = 0u;
}

/* reference of the child in `slot` (< n_children) of a packed node */
CRH_HD uint32_t crh_node_child_ref(const uint32_t w[CRH_NODE_DWORDS], uint32_t slot)
{
  const uint32_t ni = CRH_NODE_NINNER(w[3]);
  return slot < ni ? w[10] + slot : w[11] + (slot - ni);
}

#endif /* CRH_BVH_FORMAT_H */

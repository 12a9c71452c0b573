/* crh_xform.h -- per-object affine transforms of the two-level BVH (definition shared by the host code, the gfx950
 * kernels and the CPU oracle, like crh_math.h).  A transform is 12 floats, row-major 3x4: x' = M[0..2].x + M[3] ...
 * (the layout CADRays' per-object gp_Trsf locations take on the boundary, reference DataNode.cxx:239-242,
 * ImRaytraceControls.cxx:85-89).
 */
#ifndef CRH_XFORM_H
#define CRH_XFORM_H

#include "crh_math.h"

/* instance leaf of the top-level tree: 0xF in the top nibble (a triangle leaf never has count-1 == 7), index below */
#define CRH_REF_INSTANCE_TAG 0xF0000000u
#define CRH_REF_SENTINEL     0xFFFFFFFEu   /* stack marker: "leave object space" */

CRH_HD crh_v3 crh_xform_point(const float m[12], crh_v3 p)
{
  return crh_mk3(crh_dot3(crh_mk3(m[0], m[1], m[2]), p) + m[3],
                 crh_dot3(crh_mk3(m[4], m[5], m[6]), p) + m[7],
                 crh_dot3(crh_mk3(m[8], m[9], m[10]), p) + m[11]);
}
CRH_HD crh_v3 crh_xform_vector(const float m[12], crh_v3 v)
{
  return crh_mk3(crh_dot3(crh_mk3(m[0], m[1], m[2]), v), crh_dot3(crh_mk3(m[4], m[5], m[6]), v), crh_dot3(crh_mk3(m[8], m[9], m[10]), v));
}

/* inverse of an affine 3x4 (adjugate / determinant, fixed operation order); returns 0 for a singular matrix */
CRH_HD int crh_xform_inverse(const float m[12], float inv[12])
{
  const crh_v3 r0 = crh_mk3(m[0], m[1], m[2]), r1 = crh_mk3(m[4], m[5], m[6]), r2 = crh_mk3(m[8], m[9], m[10]);
  const crh_v3 c0 = crh_cross3(r1, r2), c1 = crh_cross3(r2, r0), c2 = crh_cross3(r0, r1);   /* columns of adj */
  const float det = crh_dot3(r0, c0);
  if (det == 0.f || !(det == det)) return 0;
  const float id = 1.0f / det;
  inv[0] = c0.x * id; inv[1] = c1.x * id; inv[2] = c2.x * id;
  inv[4] = c0.y * id; inv[5] = c1.y * id; inv[6] = c2.y * id;
  inv[8] = c0.z * id; inv[9] = c1.z * id; inv[10] = c2.z * id;
  const crh_v3 t = crh_mk3(m[3], m[7], m[11]);
  inv[3]  = -crh_dot3(crh_mk3(inv[0], inv[1], inv[2]), t);
  inv[7]  = -crh_dot3(crh_mk3(inv[4], inv[5], inv[6]), t);
  inv[11] = -crh_dot3(crh_mk3(inv[8], inv[9], inv[10]), t);
  return 1;
}

/* world-space box of an object-space box under m: bounds of the 8 transformed corners */
CRH_HD void crh_xform_box(const float m[12], const float bmin[3], const float bmax[3], float omin[3], float omax[3])
{
  for (int a = 0; a < 3; ++a) { omin[a] = 3.0e38f; omax[a] = -3.0e38f; }
  for (int k = 0; k < 8; ++k) {
    const crh_v3 p = crh_xform_point(m, crh_mk3((k & 1) ? bmax[0] : bmin[0], (k & 2) ? bmax[1] : bmin[1], (k & 4) ? bmax[2] : bmin[2]));
    const float q[3] = {p.x, p.y, p.z};
    for (int a = 0; a < 3; ++a) { if (q[a] < omin[a]) omin[a] = q[a]; if (q[a] > omax[a]) omax[a] = q[a]; }
  }
}

#endif /* CRH_XFORM_H */

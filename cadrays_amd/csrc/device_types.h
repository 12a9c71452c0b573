// device_types.h -- HBM-resident scene and path state shared by the host API and the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/crh_math.h"
#include "../../include/crh_spec.h"

namespace crh {

// ---- spec constants -----------------------------------------------------------------
constexpr float    kDirEps        = 1.0e-15f;
constexpr float    kSlabGuard     = 4.76837158203125e-7f;   // 2^-21: guard band of the slab test per unit of |1/d| x reach
constexpr float    kBsdfEps       = 1.0e-5f;
constexpr uint32_t kQEmpty        = 0xFFFFFFFFu;
constexpr uint32_t kQLeafBit      = 0x80000000u;

#ifndef CRH_TRI_STRIDE
#define CRH_TRI_STRIDE 4
#endif
static_assert(CRH_TRI_STRIDE == 4, "the fourth float4 of a triangle record holds the caller's triangle id");
constexpr uint32_t kTriStride = CRH_TRI_STRIDE;   // float4 between consecutive triangle records on the device (3 are fetched by the traversal): on a 64-B stride no
                                                  // record straddles two 64-B sectors (packed at 48 B half of them do): +2.8 % C3, +4 % C5
constexpr uint32_t kMaxIBox = 12;  // moved objects whose spheres the "does this ray come near a moved object at all" test looks at one by one (spec constant)
constexpr int kCounts     = 16;    // words per queue-counter block (DQueues::counts)
constexpr int kBlock      = 256;   // threads per workgroup (4 waves)
constexpr int kLdsStack   = 16;    // traversal stack entries per lane kept in LDS
constexpr int kOvfStack   = 112;   // spill entries per lane (scratch, rarely touched); 16 + 112 = 128 >= 63 (top-level tree: binary depth <= 40
                                   // -> 21 four-wide levels x 3 pending siblings) + 1 sentinel + 63 (object tree)

// Scene as the kernels see it.  All arrays are float4-granular so every fetch is one dwordx4.
struct DScene {
  const float4* nodes;    // 3 x float4 used per node (64-B stride): {origin.xyz, exps | child counts}, {qlo xyz, qhi x}, {qhi yz, child base, leaf base}  (crh_bvh_format.h)
  const float4* pnodes;   // packet nodes (k_trace_packets<true>): 8 x float4 per node, the quantised planes as floats (k_expand_packet_nodes); nullptr: none (two-level scenes)
  const float4* tris;     // 4 x float4 per triangle in leaf order (kTriStride apart; traversal fetches the first three): v0 | n.x, e0 = v1 - v0 | n.y, e1 = v0 - v2 | n.z
                          // (n = e1 x e0), {caller's triangle id, -, -, -}
  const float4* verts;    // two-level scenes: 3 x float4 per leaf position, the object-space vertices (shading transforms them); nullptr otherwise
  const float4* shade;    // 4 x float4 (one 64-B sector) per triangle in leaf order: n0|material, n1|instance, n2, geometric normal (single-level scenes)
  const float4* mats;     // 8 x float4 per material (crh_bsdf)
  const float4* lights;   // 2 x float4 per light: {vec.xyz (unit to-light dir | position), is_point}, {emission.rgb, cosmax | radius}
  const float4* env;      // W*H float4 texels, row 0 = zenith; nullptr -> constant background
  const float4* uvs;      // 2 x float4 per triangle in leaf order: {uv0, uv1}, {uv2, -, -}; nullptr -> no texture coordinates
  const float4* texels;   // all diffuse textures back to back
  const uint4*  tex_desc; // per slot: {first texel, width, height, 0}; width 0 = empty slot
  uint32_t n_tex;
  const float4* inst;     // two-level: 8 x float4 per OBJECT (valid for the objects rendered as instances): inverse rows (3), forward rows (3), {root, object, translation-only, -}, {object box centre.xyz, L1 half-extent}
  const float4* inst_leaf;  // the records of the instances in top-level leaf order (a top-level leaf reference is a position in this list)
  float4 guard_box;       // {centre.xyz, L1 half-extent} of the tree traversal starts in: scales the slab test's guard band (kSlabGuard)
  uint32_t root;          // node index traversal starts at: the (static) world-space tree; the top-level root when no static triangle is live
  uint32_t root2;         // static / moved split: top-level root walked AFTER the static tree (kQEmpty: none) ...
  float tlas_lo[3], tlas_hi[3];   // ... if the ray touches the bounds of all instances ...
  const float4* ibox;     // ... and, when there are at most kMaxIBox of them (n_ibox > 0), the sphere {centre, padded radius} around each one's world box (crh_box_sphere)
  uint32_t n_ibox;
  float4 usph;            // the sphere around the bounds of ALL instances
  int two_level;          // some object is rendered as an instance right now (object trees + top level exist)
  int split;              // render path of a split scene (root2 valid): two traversal passes -- the single-level kernels over the static tree, then the
                          // two-level ones over the top level for the rays their producers flagged (DQueues::q2 / q2_sh)
  uint32_t n_mats, n_lights, env_w, env_h;
  float bg[3]; int env_as_bg;
  // camera frame
  crh_v3 eye, fwd, right, up;
  float tan_half, aspect, ortho_scale, aperture, focal;
  int is_ortho;
  // params
  uint32_t width, height, max_depth, tile_size;
  float clampv, eps;
  int two_sided, coherent, rr;
  // crh_spec.h switches (wave-uniform; the defaults 0 / 0 / 0 / 1.0f are the frozen spec)
  int spec_u32, spec_gamma2, spec_mis1; float spec_eta_nd;
  // round 4 (#9 - #14; defaults 3 / 0.95 / 1e-2 / 1e-3 / 0 / 0)
  uint32_t spec_rr_start; float spec_rr_cap, spec_min_contrib, spec_min_thr; int spec_raygen, spec_env_orient;
  crh_v3 corner[4];       // #13: frustum-corner directions LB, RB, LT, RT (crh_frustum_corner; unit vectors when spec_raygen == 2)
};

// Wavefront path state, structure-of-arrays.  A queue entry is a POSITION in these arrays, not a pixel slot: k_raygen puts
// path slot s at position s; every k_shade chunk (1024 consecutive queue entries) writes its survivors, in order, over the
// positions of its own first entries in the OTHER ray buffer (ping-pong by bounce), so the live paths stay packed in runs of
// consecutive positions and every stage reads and writes whole cache lines however few paths survive.  The pixel slot
// travels in ray_d.w; only the radiance record is addressed by it.
struct DPaths {
  float4* ray_o[2];  // origin.xyz, rng state (uint bits)                                   [bounce parity][position]
  float4* ray_d[2];  // direction.xyz, (path slot << 1) | inside-a-medium flag (uint bits)
  float4* thr[2];    // throughput.rgb, implicit (BSDF) pdf of the ray that is in flight
  float4* hit;       // t, u, v, leaf-order triangle index (int bits; -1 = miss)             [position]
  float4* rad;       // radiance.rgb accumulated along the path, .w = `stamp` of the batch that wrote it [path slot]
  float4* sh_o;      // shadow ray origin.xyz, tmax                                         [position]
  float4* sh_d;      // shadow ray direction.xyz
  float4* sh_c;      // throughput * contribution to add when unoccluded, path slot (uint bits)
  // Batch stamp (never 0; the buffer is zeroed when it is allocated): a radiance record whose .w holds another value has not been written by THIS batch
  // and reads as zero.  Nobody initialises the records any more -- the bounce-0 shading launch used to write 16 B of zeros for every path that hit
  // something (3.2 GB per 256 M-path batch on the benchmark scene); k_accumulate and the read-modify-write adders check the stamp instead.
  uint32_t stamp;
};

struct DCounters {          // device-side mirror of crh_stats
  unsigned long long rays_nearest, rays_any, nodes_nearest, tris_nearest, nodes_any, tris_any, shaded_hits, samples;
  unsigned long long packet_rays, packet_fallback;      // device only (crh_get_packet_stats): camera rays walked as packets, those of them handed to the per-ray fall-back pass
};

struct DQueues {
  uint32_t* q[2];           // active path ids, ping-pong
  uint32_t* q_sh;           // path ids with a pending shadow ray
  uint32_t* q2;             // static / moved split: the entries of the current nearest-hit queue whose rays touch a moved object (second pass over the top level)
  uint32_t* q2_sh;          // ... and of the shadow queue
  uint32_t* counts;         // kCounts words: [0],[1] active counts (ping-pong), [2] shadow count, [4..6] work cursors (nearest, shade, any),
                            // [3],[10] second-pass nearest counts (ping-pong by bounce parity), [7] second-pass shadow count, [8],[9] second-pass cursors
};

}  // namespace crh

// bvh_builder.h -- host-side construction of the 4-wide BVH the gfx950 traversal kernels walk.
// Takes the place of OCCT's BVH_BinnedBuilder + BVH_Tree::CollapseToQuadTree, which CADRays reaches
// through AIS_InteractiveContext::Display -> OpenGl_View::updateRaytraceGeometry (SURVEY.md a3/a4;
// the only reference-side evidence is the TBB build option, reference CMakeLists.txt:79).
#pragma once
#include <cstdint>

#include "../../include/crh_bvh_format.h"
#include <vector>

namespace crh {

// spec constants (DESIGN.md "BVH"): must match what the traversal kernels assume
constexpr int      kBins      = 32;
constexpr uint32_t kLeafSize  = CRH_BVH_LEAF_SIZE;   // include/crh_bvh_format.h
constexpr int      kMaxDepth  = 40;
constexpr uint32_t kEmptyRef  = 0xFFFFFFFFu;
constexpr uint32_t kLeafBit   = 0x80000000u;

struct QNode { uint32_t w[CRH_NODE_DWORDS]; };   // 64-B stride, 48 B used: origin, grid exponents + child counts, 8-bit child bounds, child / leaf base (include/crh_bvh_format.h)

struct QBvh {
  std::vector<QNode>    nodes;       // root first; the inner children of a node are consecutive
  std::vector<uint32_t> prim_order;  // leaf order -> input triangle index
  float bbmin[3] = {0, 0, 0}, bbmax[3] = {0, 0, 0};
};

// pos: 3*nV floats (world space); tri: 4*nT ints {i0,i1,i2,material}.  threads <= 0: hardware concurrency.
void build_qbvh(const float* pos, const int32_t* tri, uint32_t n_tris, QBvh& out, int threads = 0);

// One tree over n axis-aligned boxes (6 floats each: min xyz, max xyz), appended to `nodes` (references are indices into
// `nodes`).  One box per leaf; leaf references are CRH_LEAF_TAG | (leaf0 + position) for an object's tree and
// CRH_REF_INSTANCE_TAG | (leaf0 + position) for the top-level tree of the two-level BVH.
// Returns the root's index; order[position] = box at that leaf position; bmin/bmax = bounds.
uint32_t build_tree(const float* boxes, uint32_t n, bool instance_leaves, uint32_t leaf0,
                    std::vector<QNode>& nodes, std::vector<uint32_t>& order, float bmin[3], float bmax[3], int threads = 0);

// threads the builder takes when asked for `threads <= 0`: the CPUs this process may run on at once (affinity mask, cgroup quota), its share of them
// when torchrun started several ranks on one host
int build_threads(int threads);

}  // namespace crh

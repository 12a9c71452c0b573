/* crh_oracle.c -- CPU restatement of the path-tracing hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this file's
 * library; the product (cadrays_amd/, libcadrays_hip.so) never links or calls it.
 *
 * PARITY UNPINNED against the reference renderer: the algorithm CADRays runs on this path
 * lives in Open CASCADE Technology's TKOpenGl (OpenGl_View_Raytrace.cxx, src/Shaders/
 * {RaytraceBase,PathtraceBase,Display}.fs, BVH_* package), an external, un-vendored,
 * version-unpinned dependency ("Current OCCT development snapshot", reference README.md:51;
 * found via CMakeLists.txt:64-71) that is absent from /root/reference and from this image.
 * The reference holds no golden image, known-answer vector or expected number for the path
 * (testing/CADRays_Testing.py compares against a user-supplied template folder).  What this
 * file follows is therefore (1) the reference's *input contract* -- cited per function -- and
 * (2) the published structure of OCCT's GLSL path tracer restated as this project's frozen
 * spec (DESIGN.md "Algorithm spec").  It is pinned by analytic known-answer tests
 * (tests/test_oracle_kat.py) and by self-generated golden vectors (tests/golden/).
 *
 * Plain C, scalar, AoS, recursive -- written independently of the HIP product; they share only
 * include/crh_math.h (the definition of the elementary arithmetic) and include/cadrays_hip.h
 * (the input structs of the boundary).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "../include/crh_math.h"
#include "../include/crh_bvh_format.h"
#include "../include/crh_xform.h"
#include "../include/cadrays_hip.h"

#define ORC_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------ spec constants */
#define BVH_NBINS      32          /* OCCT BVH_Constants_NbBinsOptimal [OCCT-ext]           */
#define BVH_LEAF       CRH_BVH_LEAF_SIZE   /* max triangles per leaf (include/crh_bvh_format.h) */
#define BVH_MAXDEPTH   40          /* binary depth bound (root = 0); median splits keep it  */
#define QBVH_EMPTY     0xFFFFFFFFu
#define QBVH_LEAFBIT   0x80000000u
#define STACK_MAX      128         /* >= 63 (top-level tree: 21 four-wide levels x 3 pending siblings) + 1 sentinel + 63 (object tree) */
#define DIR_EPS        1.0e-15f
#define SLAB_GUARD     4.76837158203125e-7f   /* 2^-21 */
#define BSDF_EPS       1.0e-5f     /* roughness / weight threshold ("FLT_EPSILON" in GLSL)  */
#define LUMA_R 0.2126f
#define LUMA_G 0.7152f
#define LUMA_B 0.0722f

typedef crh_v3 v3;

typedef struct { uint32_t w[CRH_NODE_DWORDS]; } qnode;   /* 64 B, layout in include/crh_bvh_format.h */
typedef struct { float f[12]; } qtri;      /* 48 B: v0.xyz,prim | v1.xyz,0 | v2.xyz,0 */

typedef struct { uint64_t nodes, tris, nodes_any, tris_any; } trav_counters;
/* ORC_FAST (oracle/Makefile, libcrh_oracle_fast.so): the honest CPU-baseline build of bench.py -- -O3 -march=native, contraction
 * allowed, node / triangle visit counters compiled out, 8x8-pixel work items.  Never used as the parity checker. */
#ifdef ORC_FAST
#define ORC_COUNT(x) ((void)0)
#else
#define ORC_COUNT(x) (x)
#endif
typedef struct orc_instance { float fwd[12], inv[12]; float bmin[3], bmax[3]; float wmin[3], wmax[3]; float sph[4]; uint32_t root, obj; } orc_instance;
#define MAX_IBOX 12u      /* moved objects whose spheres the "does the ray come near a moved object at all" test looks at one by one (kMaxIBox of the product) */
/* per object of a two-level scene ("static / moved split"): every object is BAKED into the static world-space tree with
 * the transform it has when the scene is built (static0 = 1 for every non-empty object); is_inst = rendered through its own object tree + the
 * top level right now, i.e. while its transform differs from the build-time one -- its triangles in the static tree are disabled meanwhile;
 * the object tree is built the first time it is needed and kept */
typedef struct orc_object { uint8_t static0, built, is_inst, in_static; uint32_t root, first, ntri; float bmin[3], bmax[3]; } orc_object;
/* round 6 (crh_set_visibility / crh_add_object): in_static = the object's records in the static tree are live right now (static0, visible, at its
 * build-time placement); an object ADDED to a built scene has static0 = 0 until the next full build bakes it: always an instance while visible */

typedef struct orc_ctx {
  /* inputs */
  uint32_t nV, nT, nM, nL;
  float* pos; float* nrm; float* uv; int32_t* tri;
  crh_bsdf* mats; crh_light* lights;
  float* env; uint32_t envW, envH;
  struct { float* rgb; uint32_t w, h, ch; } tex[64]; uint32_t nTex;   /* ch = 3 (RGB) or 4 (RGBA) floats per texel */
  crh_camera cam; crh_params par;
  crh_spec spec;                  /* include/crh_spec.h: the switchable departures from the recollected OCCT behaviour */
  /* derived */
  qnode* nodes; uint32_t nNodes; qtri* qtris; uint32_t nQT;
  /* two-level mode (per-object transforms): object-space BLAS per object + TLAS over instance boxes */
  int two_level; uint32_t nO; float* xf; int32_t* tri_obj;
  int flat;                        /* no object is rendered as an instance right now: the scene is ONE world-space tree */
  struct orc_instance* inst; uint32_t nInst; uint32_t nBlasNodes; uint32_t root;
  uint32_t* tlas_order;          /* instance at top-level leaf position i */
  orc_object* obj; uint32_t* obj_tris;   /* per object; its triangles (input order) at obj_tris[first .. first + ntri) */
  uint32_t* static_pos;          /* leaf position of triangle t in the static tree */
  float* xf0;                    /* the transforms the scene was built with (12 * nO) */
  uint8_t* hidden;               /* per object: 1 = erased from the view (crh_set_visibility); NULL = all displayed */
  float* pos_w; float* nrm_w;    /* per vertex: position / unit normal under its object's build-time transform = what the static tree holds */
  uint32_t n_static, n_static_live; float sbmin[3], sbmax[3];   /* triangles in the static tree, those not disabled; its bounds */
  uint32_t root2;                /* top-level root to walk AFTER the static tree (QBVH_EMPTY: none) and the bounds of the instances */
  float tlas_lo[3], tlas_hi[3], usph[4];      /* bounds of all instances, and the sphere around them (crh_box_sphere) */
  uint32_t capNodes, capQT;
  float bbmin[3], bbmax[3]; float eps;
  int built;
  uint32_t frames_done;          /* whole-frame iterations since the last restart: orc_render continues from here (like crh_render) */
  /* camera frame */
  v3 c_eye, c_fwd, c_right, c_up; float c_tanh, c_aspect;
  v3 c_corner[4];               /* crh_spec.h #13: frustum-corner directions LB, RB, LT, RT */
  /* lights prepared: vec = to-light dir (directional, normalized) or position; param = cosmax or radius */
  v3* l_vec; float* l_par;
  /* accumulator */
  float* accum;     /* W*H*4 */
  float* m2;        /* W*H: running mean of squared luminance (adaptive sampling) */
  int adaptive; uint32_t adaptive_tiles, adaptive_picks;
  int show_tiles; uint8_t* last_picked; uint32_t last_picked_n;   /* ShowSamplingTiles overlay */
  crh_stats st;
  char err[256];
} orc_ctx;

/* ------------------------------------------------------------------ Bullard frame seeds */
/* SURVEY.md a14 / Appendix A [OCCT-ext]: host generator reseeded when accumulation restarts;
 * frame n uses next() >> 2. */
static uint32_t frame_seed(uint32_t seed, uint32_t n)
{
  uint32_t hi = seed, lo = seed ^ 0x49616E42u, r = 0;
  for (uint32_t i = 0; i <= n; ++i) { hi = (hi >> 2) + (hi << 2); hi += lo; lo += hi; r = hi; }
  return r >> 2;
}

static void* dup_mem(const void* p, size_t n) { void* r = malloc(n ? n : 1); if (p && n) memcpy(r, p, n); return r; }

/* ================================================================== BVH build */
typedef struct { float mn[3], mx[3]; } aabb;
typedef struct { aabb box; int32_t left, right; uint32_t lo, hi; } bnode;   /* left < 0 => leaf [lo,hi) */

typedef struct {
  const aabb* pb; const float* cen; /* 3*n */
  uint32_t* idx; uint32_t* tmp;
  bnode* bn; uint32_t nbn, cap;
  uint32_t leaf_max;            /* 4 for triangle trees, 1 for the top-level tree over instances */
} builder;

static void aabb_empty(aabb* b) { for (int a = 0; a < 3; ++a) { b->mn[a] = 3.0e38f; b->mx[a] = -3.0e38f; } }
static void aabb_grow(aabb* b, const aabb* o)
{ for (int a = 0; a < 3; ++a) { if (o->mn[a] < b->mn[a]) b->mn[a] = o->mn[a]; if (o->mx[a] > b->mx[a]) b->mx[a] = o->mx[a]; } }
static float aabb_harea(const aabb* b)
{
  float dx = b->mx[0] - b->mn[0], dy = b->mx[1] - b->mn[1], dz = b->mx[2] - b->mn[2];
  return CRH_FMA(dx, dy, CRH_FMA(dy, dz, dz * dx));
}
static int ceil_log2_u32(uint32_t v) { int l = 0; uint32_t p = 1; while (p < v) { p <<= 1; ++l; } return l; }

/* sort helper for median splits: total order (key, prim index) */
typedef struct { float k; uint32_t i; } keyidx;
static int keyidx_cmp(const void* a, const void* b)
{
  const keyidx* x = (const keyidx*)a; const keyidx* y = (const keyidx*)b;
  if (x->k < y->k) return -1; if (x->k > y->k) return 1;
  return x->i < y->i ? -1 : (x->i > y->i ? 1 : 0);
}

/* the node array is allocated once (a binary tree over n primitives has at most 2n - 1 nodes): big subtrees are built by OpenMP tasks, and the
 * numbering of the BINARY nodes -- the only thing that depends on who allocates first -- does not reach the output (the collapse follows links) */
static uint32_t bn_new(builder* B) { return __atomic_fetch_add(&B->nbn, 1u, __ATOMIC_RELAXED); }

static uint32_t build_rec(builder* B, uint32_t lo, uint32_t hi, int depth)
{
  uint32_t me = bn_new(B);
  aabb box; aabb_empty(&box);
  for (uint32_t i = lo; i < hi; ++i) aabb_grow(&box, &B->pb[B->idx[i]]);
  uint32_t n = hi - lo;
  B->bn[me].box = box; B->bn[me].lo = lo; B->bn[me].hi = hi; B->bn[me].left = B->bn[me].right = -1;
  if (n <= B->leaf_max) return me;

  /* centroid bounds */
  float cmn[3] = {3.0e38f, 3.0e38f, 3.0e38f}, cmx[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
  for (uint32_t i = lo; i < hi; ++i) {
    const float* c = &B->cen[3 * B->idx[i]];
    for (int a = 0; a < 3; ++a) { if (c[a] < cmn[a]) cmn[a] = c[a]; if (c[a] > cmx[a]) cmx[a] = c[a]; }
  }
  uint32_t mid = 0; int done = 0;
  /* depth budget: once balanced median splits are needed to finish within BVH_MAXDEPTH, use them */
  int need = ceil_log2_u32((n + B->leaf_max - 1) / B->leaf_max);
  int force_median = (depth + need >= BVH_MAXDEPTH);

  if (!force_median) {
    float best = 3.0e38f; int baxis = -1, bsplit = -1;
    for (int a = 0; a < 3; ++a) {
      float ext = cmx[a] - cmn[a];
      if (!(ext > 0.f)) continue;
      uint32_t cnt[BVH_NBINS]; aabb bb[BVH_NBINS];
      for (int b = 0; b < BVH_NBINS; ++b) { cnt[b] = 0; aabb_empty(&bb[b]); }
      for (uint32_t i = lo; i < hi; ++i) {
        uint32_t p = B->idx[i];
        int b = (int)(((B->cen[3 * p + a] - cmn[a]) / ext) * (float)BVH_NBINS);
        if (b > BVH_NBINS - 1) b = BVH_NBINS - 1;
        cnt[b]++; aabb_grow(&bb[b], &B->pb[p]);
      }
      /* right-to-left suffix */
      float rarea[BVH_NBINS]; uint32_t rcnt[BVH_NBINS];
      aabb acc; aabb_empty(&acc); uint32_t c = 0;
      for (int b = BVH_NBINS - 1; b >= 1; --b) { if (cnt[b]) aabb_grow(&acc, &bb[b]); c += cnt[b]; rcnt[b] = c; rarea[b] = c ? aabb_harea(&acc) : 0.f; }
      aabb_empty(&acc); c = 0;
      for (int s = 0; s < BVH_NBINS - 1; ++s) {   /* split after bin s: left = bins [0..s] */
        if (cnt[s]) aabb_grow(&acc, &bb[s]); c += cnt[s];
        if (c == 0 || rcnt[s + 1] == 0) continue;
        float cost = CRH_FMA(aabb_harea(&acc), (float)c, rarea[s + 1] * (float)rcnt[s + 1]);
        if (cost < best) { best = cost; baxis = a; bsplit = s; }
      }
    }
    if (baxis >= 0) {
      /* stable partition by bin(prim) <= bsplit */
      float ext = cmx[baxis] - cmn[baxis];
      uint32_t nl = 0, nr = 0;
      for (uint32_t i = lo; i < hi; ++i) {
        uint32_t p = B->idx[i];
        int b = (int)(((B->cen[3 * p + baxis] - cmn[baxis]) / ext) * (float)BVH_NBINS);
        if (b > BVH_NBINS - 1) b = BVH_NBINS - 1;
        if (b <= bsplit) B->idx[lo + nl++] = p; else B->tmp[lo + nr++] = p;      /* scratch range [lo, hi): private to this subtree */
      }
      memcpy(&B->idx[lo + nl], B->tmp + lo, sizeof(uint32_t) * nr);
      mid = lo + nl; done = 1;
    }
  }
  if (!done) {
    /* median split along the widest centroid axis (ties: lowest axis); zero extent -> index order */
    int ax = 0; float e0 = cmx[0] - cmn[0], e1 = cmx[1] - cmn[1], e2 = cmx[2] - cmn[2];
    float em = e0; if (e1 > em) { em = e1; ax = 1; } if (e2 > em) { em = e2; ax = 2; }
    if (em > 0.f) {
      keyidx* ks = (keyidx*)malloc(sizeof(keyidx) * n);
      for (uint32_t i = 0; i < n; ++i) { ks[i].i = B->idx[lo + i]; ks[i].k = B->cen[3 * ks[i].i + ax]; }
      qsort(ks, n, sizeof(keyidx), keyidx_cmp);
      for (uint32_t i = 0; i < n; ++i) B->idx[lo + i] = ks[i].i;
      free(ks);
    }
    mid = lo + n / 2;
  }
  uint32_t l, r;
  if (n >= 32768u) {            /* the two halves own disjoint ranges of idx / tmp: same tree whoever builds them */
#pragma omp task shared(l) firstprivate(B, lo, mid, depth)
    l = build_rec(B, lo, mid, depth + 1);
    r = build_rec(B, mid, hi, depth + 1);
#pragma omp taskwait
  } else {
    l = build_rec(B, lo, mid, depth + 1);
    r = build_rec(B, mid, hi, depth + 1);
  }
  B->bn[me].left = (int32_t)l; B->bn[me].right = (int32_t)r;
  return me;
}

/* collapse binary -> 4-wide (OCCT BVH_Tree::CollapseToQuadTree idea: children := grandchildren).  Child references
 * are implicit in the 48-B node (crh_bvh_format.h): when a node is expanded, its inner children take the next free
 * node indices (one consecutive block) and the primitives of its leaf children the next free leaf positions; then the
 * inner children are expanded in slot order.  Slots: inner children first, then leaves, each in collapse order. */
typedef struct { const bnode* bn; qnode* qn; uint32_t nq, capq; int instances; uint32_t next_leaf, leaf0; const uint32_t* idx; uint32_t* order; } collapser;

static void collapse_rec(collapser* C, uint32_t bi, uint32_t me)
{
  uint32_t kids[4]; int nk = 0;
  const bnode* b = &C->bn[bi];
  if (b->left < 0) { kids[nk++] = bi; }
  else {
    uint32_t two[2] = {(uint32_t)b->left, (uint32_t)b->right};
    for (int k = 0; k < 2; ++k) {
      const bnode* c = &C->bn[two[k]];
      if (c->left < 0) kids[nk++] = two[k];
      else { kids[nk++] = (uint32_t)c->left; kids[nk++] = (uint32_t)c->right; }
    }
  }
  uint32_t slot[4]; int ni = 0, nc = 0;
  for (int k = 0; k < nk; ++k) if (C->bn[kids[k]].left >= 0) slot[nc++] = kids[k];
  ni = nc;
  for (int k = 0; k < nk; ++k) { const bnode* c = &C->bn[kids[k]]; if (c->left < 0 && c->hi > c->lo) slot[nc++] = kids[k]; }   /* an empty leaf (no primitives at all) is dropped */
  while (C->nq + (uint32_t)ni + 1u > C->capq) { C->capq *= 2; C->qn = (qnode*)realloc(C->qn, sizeof(qnode) * C->capq); }
  if (ni >= 2 && (C->nq & 1u)) { memset(&C->qn[C->nq], 0, sizeof(qnode)); C->nq++; }    /* pair alignment: a block of >= 2 inner children starts on an even index (one zero slot skipped) */
  const uint32_t child_base = C->nq;
  C->nq += (uint32_t)ni;
  const uint32_t leaf_base = (C->instances ? CRH_REF_INSTANCE_TAG : CRH_LEAF_TAG) | C->next_leaf;
  float cmin[4][3], cmax[4][3];
  for (int k = 0; k < nc; ++k) {
    const bnode* c = &C->bn[slot[k]];
    for (int a = 0; a < 3; ++a) { cmin[k][a] = c->box.mn[a]; cmax[k][a] = c->box.mx[a]; }
    if (k >= ni) C->order[C->next_leaf++ - C->leaf0] = C->idx[c->lo];      /* one primitive per leaf */
  }
  qnode q; memset(&q, 0, sizeof q);
  crh_pack_node(cmin, cmax, ni, nc, ni ? child_base : 0u, leaf_base, q.w);      /* no inner children: the field is 0 */
  C->qn[me] = q;
  for (int k = 0; k < ni; ++k) collapse_rec(C, slot[k], child_base + (uint32_t)k);
}

/* One tree over n primitives with boxes pb[0..n) (centre = box centre): binary build + 4-wide collapse appended to C.
 * Leaves are numbered from leaf0; returns the 4-wide root index; order[0..n) = primitive at leaf position leaf0 + i;
 * *rootbox = bounds. */
static uint32_t build_tree(collapser* C, const aabb* pb, uint32_t n, uint32_t leaf_max, int instances, uint32_t leaf0,
                           uint32_t* order, aabb* rootbox)
{
  float* cen = (float*)malloc(sizeof(float) * 3 * (n ? n : 1));
  for (uint32_t t = 0; t < n; ++t) for (int a = 0; a < 3; ++a) cen[3 * t + a] = (pb[t].mn[a] + pb[t].mx[a]) * 0.5f;
  builder B; B.pb = pb; B.cen = cen; B.leaf_max = leaf_max;
  B.idx = (uint32_t*)malloc(sizeof(uint32_t) * (n ? n : 1)); B.tmp = (uint32_t*)malloc(sizeof(uint32_t) * (n ? n : 1));
  for (uint32_t t = 0; t < n; ++t) B.idx[t] = t;
  B.cap = 2 * (n ? n : 1) + 1; B.nbn = 0; B.bn = (bnode*)malloc(sizeof(bnode) * B.cap);
  if (n >= 32768u) {
#pragma omp parallel
#pragma omp single
    build_rec(&B, 0, n, 0);
  } else build_rec(&B, 0, n, 0);
  if (rootbox) *rootbox = B.bn[0].box;
  C->bn = B.bn; C->instances = instances; C->next_leaf = leaf0; C->leaf0 = leaf0; C->idx = B.idx; C->order = order;
  if (C->nq == C->capq) { C->capq *= 2; C->qn = (qnode*)realloc(C->qn, sizeof(qnode) * C->capq); }
  uint32_t root = C->nq++;
  collapse_rec(C, 0, root);
  free(cen); free(B.tmp); free(B.idx); free(B.bn);
  return root;
}

static void tri_box_of(const orc_ctx* c, const float* pos, uint32_t t, aabb* b)
{
  aabb_empty(b);
  for (int k = 0; k < 3; ++k) {
    const float* p = &pos[3 * c->tri[4 * t + k]];
    for (int a = 0; a < 3; ++a) { if (p[a] < b->mn[a]) b->mn[a] = p[a]; if (p[a] > b->mx[a]) b->mx[a] = p[a]; }
  }
}
static void tri_box(const orc_ctx* c, uint32_t t, aabb* b) { tri_box_of(c, c->pos, t, b); }
static void emit_tri_of(const orc_ctx* c, const float* pos, qtri* q, uint32_t t)
{
  for (int k = 0; k < 3; ++k) {
    const float* p = &pos[3 * c->tri[4 * t + k]];
    q->f[4 * k + 0] = p[0]; q->f[4 * k + 1] = p[1]; q->f[4 * k + 2] = p[2]; q->f[4 * k + 3] = 0.f;
  }
  q->f[3] = crh_u2f(t);
}

static void emit_tri(const orc_ctx* c, qtri* q, uint32_t t) { emit_tri_of(c, c->pos, q, t); }

/* every object at the identity: the scene is one world-space tree -- no top level, no ray transforms (spec: same result as the same
 * triangles handed over without objects) */
static int is_identity(const float* m)
{
  static const float I[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
  for (int k = 0; k < 12; ++k) if (m[k] != I[k]) return 0;
  return 1;
}

/* (re)build the top-level tree over the world boxes of the objects that are rendered as instances right now (ascending object
 * index); keeps the static tree and the object trees [0, nBlasNodes).  Sets the walk's entry points: no instance -> the static tree
 * alone (flat); instances and live static triangles -> static tree first, then the top level (root2) if the ray touches the
 * instances' bounds; no live static triangle -> the top level alone. */
static void build_tlas(orc_ctx* c)
{
  uint32_t n = 0;
  for (uint32_t ob = 0; ob < c->nO; ++ob) if (c->obj[ob].is_inst) ++n;
  free(c->inst); c->inst = (orc_instance*)calloc(n ? n : 1, sizeof(orc_instance)); c->nInst = n;
  aabb* pb = (aabb*)malloc(sizeof(aabb) * (n ? n : 1));
  uint32_t* order = (uint32_t*)malloc(sizeof(uint32_t) * (n ? n : 1));
  aabb sb; aabb_empty(&sb);
  uint32_t i = 0;
  for (uint32_t ob = 0; ob < c->nO; ++ob) {
    if (!c->obj[ob].is_inst) continue;
    orc_instance* in = &c->inst[i];
    in->obj = ob; in->root = c->obj[ob].root;
    for (int a = 0; a < 3; ++a) { in->bmin[a] = c->obj[ob].bmin[a]; in->bmax[a] = c->obj[ob].bmax[a]; }
    memcpy(in->fwd, &c->xf[12 * ob], sizeof in->fwd);
    if (!crh_xform_inverse(in->fwd, in->inv)) memset(in->inv, 0, sizeof in->inv);
    crh_xform_box(in->fwd, in->bmin, in->bmax, pb[i].mn, pb[i].mx);
    for (int a = 0; a < 3; ++a) { in->wmin[a] = pb[i].mn[a]; in->wmax[a] = pb[i].mx[a]; }
    crh_box_sphere(in->wmin, in->wmax, in->sph);
    aabb_grow(&sb, &pb[i]);
    ++i;
  }
  c->flat = n == 0; c->root2 = QBVH_EMPTY;
  free(c->tlas_order); c->tlas_order = order;
  if (n == 0) {                                   /* one world-space tree */
    c->root = 0; c->nNodes = c->nBlasNodes;
    for (int a = 0; a < 3; ++a) { c->bbmin[a] = c->n_static ? c->sbmin[a] : 0.f; c->bbmax[a] = c->n_static ? c->sbmax[a] : 0.f; }
    free(pb);
    return;
  }
  collapser C; C.qn = c->nodes; C.nq = c->nBlasNodes; C.capq = c->capNodes;
  const uint32_t troot = build_tree(&C, pb, n, 1, 1, 0, order, NULL);
  c->nodes = C.qn; c->nNodes = C.nq; c->capNodes = C.capq;
  for (int a = 0; a < 3; ++a) { c->tlas_lo[a] = sb.mn[a]; c->tlas_hi[a] = sb.mx[a]; }
  crh_box_sphere(c->tlas_lo, c->tlas_hi, c->usph);
  if (c->n_static_live) {
    c->root = 0; c->root2 = troot;
    for (int a = 0; a < 3; ++a) { c->bbmin[a] = crh_min(sb.mn[a], c->sbmin[a]); c->bbmax[a] = crh_max(sb.mx[a], c->sbmax[a]); }
  } else {
    c->root = troot;
    for (int a = 0; a < 3; ++a) { c->bbmin[a] = sb.mn[a]; c->bbmax[a] = sb.mx[a]; }
  }
  free(pb);
}

/* the object-space tree of object ob, appended behind the trees built so far; its triangles take the next leaf positions */
static void build_object_tree(orc_ctx* c, collapser* C, uint32_t ob)
{
  orc_object* o = &c->obj[ob];
  const uint32_t m = o->ntri; const uint32_t* mem = &c->obj_tris[o->first];
  aabb* pb = (aabb*)malloc(sizeof(aabb) * m);
  uint32_t* order = (uint32_t*)malloc(sizeof(uint32_t) * m);
  for (uint32_t i = 0; i < m; ++i) tri_box(c, mem[i], &pb[i]);
  aabb box;
  if (c->nQT + m > c->capQT) { c->capQT = 2 * (c->nQT + m); c->qtris = (qtri*)realloc(c->qtris, sizeof(qtri) * c->capQT); }
  o->root = build_tree(C, pb, m, BVH_LEAF, 0, c->nQT, order, &box);
  for (int a = 0; a < 3; ++a) { o->bmin[a] = box.mn[a]; o->bmax[a] = box.mx[a]; }
  for (uint32_t i = 0; i < m; ++i) emit_tri(c, &c->qtris[c->nQT + i], mem[order[i]]);
  c->nQT += m; o->built = 1;
  free(pb); free(order);
}

/* a static0 object leaves / returns to the identity: its triangles in the static tree are disabled (all-zero vertices: the test yields
 * NaN and rejects) / restored; the tree's boxes do not change */
static void set_static_triangles(orc_ctx* c, uint32_t ob, int live)
{
  const orc_object* o = &c->obj[ob];
  for (uint32_t i = 0; i < o->ntri; ++i) {
    const uint32_t t = c->obj_tris[o->first + i];
    qtri* q = &c->qtris[c->static_pos[t]];
    if (live) emit_tri_of(c, c->pos_w, q, t);
    else { memset(q->f, 0, sizeof q->f); q->f[3] = crh_u2f(t); }
  }
  if (live) c->n_static_live += o->ntri; else c->n_static_live -= o->ntri;
}

/* Bring every object to the state its transform and its visibility ask for, touching nothing else: a displayed object at its build-time placement lives
 * in the static tree; a displayed object off it (or added after the build) is an instance with a tree of its own, built the first time; an erased object
 * is neither -- its records in the static tree are disabled exactly like a moved object's, and it stays out of the top level.  Then the top-level tree
 * of this moment.  (AIS_InteractiveContext::Display / Erase / SetLocation of src/ImportExport/DataNode.cxx:239-242, 304-344.) */
static void apply_objects(orc_ctx* c)
{
  collapser C; C.qn = c->nodes; C.nq = c->nBlasNodes; C.capq = c->capNodes;
  for (uint32_t ob = 0; ob < c->nO; ++ob) {
    orc_object* o = &c->obj[ob];
    if (!o->ntri) continue;
    const int shown = !(c->hidden && c->hidden[ob]);
    const int moved = !o->static0 || memcmp(&c->xf[12 * ob], &c->xf0[12 * ob], 12 * sizeof(float)) != 0;
    const int want_static = shown && !moved, want_inst = shown && moved;
    if (want_static != o->in_static) { set_static_triangles(c, ob, want_static); o->in_static = (uint8_t)want_static; }
    if (want_inst && !o->built) build_object_tree(c, &C, ob);
    o->is_inst = (uint8_t)want_inst;
  }
  c->nodes = C.qn; c->nBlasNodes = C.nq; c->capNodes = C.capq;
  build_tlas(c);
}

static int do_build(orc_ctx* c)
{
  uint32_t n = c->nT;
  free(c->nodes); free(c->qtris); free(c->inst); free(c->obj); free(c->obj_tris); free(c->static_pos);
  c->nodes = NULL; c->qtris = NULL; c->inst = NULL; c->obj = NULL; c->obj_tris = NULL; c->static_pos = NULL; c->nInst = 0; c->root = 0; c->root2 = QBVH_EMPTY;
  c->capQT = n ? n : 1;
  c->qtris = (qtri*)malloc(sizeof(qtri) * c->capQT); c->nQT = n;
  collapser C; C.capq = 1024; C.nq = 0; C.qn = (qnode*)malloc(sizeof(qnode) * C.capq);
  if (!c->two_level) {
    aabb* pb = (aabb*)malloc(sizeof(aabb) * (n ? n : 1));
    uint32_t* order = (uint32_t*)malloc(sizeof(uint32_t) * (n ? n : 1));
    for (uint32_t t = 0; t < n; ++t) tri_box(c, t, &pb[t]);
    aabb sb;
    build_tree(&C, pb, n, BVH_LEAF, 0, 0, order, &sb);
    for (uint32_t i = 0; i < n; ++i) emit_tri(c, &c->qtris[i], order[i]);
    for (int a = 0; a < 3; ++a) { c->bbmin[a] = n ? sb.mn[a] : 0.f; c->bbmax[a] = n ? sb.mx[a] : 0.f; }
    c->nodes = C.qn; c->nNodes = C.nq; c->nBlasNodes = C.nq; c->capNodes = C.capq; c->flat = 0;
    free(pb); free(order);
    return 0;
  }
  /* two-level: the objects at the identity share ONE world-space tree (the static tree, first in the node array, its triangles first in
   * leaf order); every other non-empty object gets an object-space tree (triangles in input order); then the top-level tree */
  c->obj = (orc_object*)calloc(c->nO ? c->nO : 1, sizeof(orc_object));
  c->obj_tris = (uint32_t*)malloc(sizeof(uint32_t) * (n ? n : 1));
  c->static_pos = (uint32_t*)malloc(sizeof(uint32_t) * (n ? n : 1));
  for (uint32_t t = 0; t < n; ++t) c->obj[c->tri_obj[t]].ntri++;
  { uint32_t acc = 0; for (uint32_t ob = 0; ob < c->nO; ++ob) { c->obj[ob].first = acc; acc += c->obj[ob].ntri; c->obj[ob].ntri = 0; c->obj[ob].static0 = 1; c->obj[ob].in_static = 1; } }
  for (uint32_t t = 0; t < n; ++t) { orc_object* o = &c->obj[c->tri_obj[t]]; c->obj_tris[o->first + o->ntri++] = t; }
  /* bake: every vertex under the transform its object has NOW (each vertex belongs to one object); an object at the identity keeps its bits */
  free(c->xf0); free(c->pos_w); free(c->nrm_w);
  c->xf0 = (float*)dup_mem(c->xf, sizeof(float) * 12 * c->nO);
  c->pos_w = (float*)dup_mem(c->pos, sizeof(float) * 3 * c->nV); c->nrm_w = (float*)dup_mem(c->nrm, sizeof(float) * 3 * c->nV);
  {
    uint8_t* done = (uint8_t*)calloc(c->nV ? c->nV : 1, 1);
    for (uint32_t t = 0; t < n; ++t) {
      const float* M = &c->xf0[12 * c->tri_obj[t]];
      if (is_identity(M)) continue;
      for (int k = 0; k < 3; ++k) {
        const int32_t vi = c->tri[4 * t + k];
        if (done[vi]) continue;
        done[vi] = 1;
        const v3 p = crh_xform_point(M, crh_mk3(c->pos[3 * vi], c->pos[3 * vi + 1], c->pos[3 * vi + 2]));
        v3 nn = crh_norm3(crh_xform_vector(M, crh_mk3(c->nrm[3 * vi], c->nrm[3 * vi + 1], c->nrm[3 * vi + 2])));
        if (!(crh_dot3(nn, nn) > 0.f)) nn = crh_mk3(c->nrm[3 * vi], c->nrm[3 * vi + 1], c->nrm[3 * vi + 2]);
        c->pos_w[3 * vi] = p.x; c->pos_w[3 * vi + 1] = p.y; c->pos_w[3 * vi + 2] = p.z;
        c->nrm_w[3 * vi] = nn.x; c->nrm_w[3 * vi + 1] = nn.y; c->nrm_w[3 * vi + 2] = nn.z;
      }
    }
    free(done);
  }
  uint32_t nS = 0;
  for (uint32_t t = 0; t < n; ++t) if (c->obj[c->tri_obj[t]].static0) ++nS;
  c->n_static = c->n_static_live = nS; c->nQT = 0;
  if (nS) {
    aabb* pb = (aabb*)malloc(sizeof(aabb) * nS);
    uint32_t* list = (uint32_t*)malloc(sizeof(uint32_t) * nS); uint32_t* order = (uint32_t*)malloc(sizeof(uint32_t) * nS);
    uint32_t k = 0;
    for (uint32_t t = 0; t < n; ++t) if (c->obj[c->tri_obj[t]].static0) { tri_box_of(c, c->pos_w, t, &pb[k]); list[k++] = t; }
    aabb sb;
    build_tree(&C, pb, nS, BVH_LEAF, 0, 0, order, &sb);
    for (uint32_t i = 0; i < nS; ++i) { emit_tri_of(c, c->pos_w, &c->qtris[i], list[order[i]]); c->static_pos[list[order[i]]] = i; }
    for (int a = 0; a < 3; ++a) { c->sbmin[a] = sb.mn[a]; c->sbmax[a] = sb.mx[a]; }
    c->nQT = nS;
    free(pb); free(list); free(order);
  }
  c->nodes = C.qn;
  for (uint32_t ob = 0; ob < c->nO; ++ob) {
    orc_object* o = &c->obj[ob];
    if (o->static0 || !o->ntri) continue;
    build_object_tree(c, &C, ob);
    o->is_inst = 1;
  }
  c->nodes = C.qn; c->nBlasNodes = C.nq; c->nNodes = C.nq; c->capNodes = C.capq;
  build_tlas(c);
  if (c->hidden) apply_objects(c);      /* objects erased before the build: baked like the rest, then disabled */
  return 0;
}

/* ================================================================== traversal */
typedef struct { float t, u, v; int32_t prim; } hit_t;

static inline float inv_dir(float d) { return 1.0f / (crh_abs(d) < DIR_EPS ? (d < 0.f ? -DIR_EPS : DIR_EPS) : d); }

/* Ray/triangle: n = (v0-v2) x (v1-v0); t = n.(v0-o) / n.d; u,v from (d x (v0-o)). */
static inline int tri_test(const qtri* q, v3 o, v3 d, float tmax, float* t, float* u, float* v)
{
  v3 v0 = crh_mk3(q->f[0], q->f[1], q->f[2]), v1 = crh_mk3(q->f[4], q->f[5], q->f[6]), v2 = crh_mk3(q->f[8], q->f[9], q->f[10]);
  v3 e0 = crh_sub3(v1, v0), e1 = crh_sub3(v0, v2);
  v3 n = crh_cross3(e1, e0);
  v3 to = crh_sub3(v0, o);
  float inv = 1.0f / crh_dot3(n, d);
  v3 vc = crh_cross3(d, to);
  float tt = crh_dot3(n, to) * inv;
  float uu = crh_dot3(vc, e1) * inv;
  float vv = crh_dot3(vc, e0) * inv;
  if (tt >= 0.f && uu >= 0.f && vv >= 0.f && (uu + vv) <= 1.0f && tt < tmax) { *t = tt; *u = uu; *v = vv; return 1; }
  return 0;
}

/* Ordered stack traversal of the 4-wide BVH.  any_hit: stop at the first accepted triangle. */
static int traverse_ex(const orc_ctx* c, v3 o, v3 d, float tmax, int any_hit, hit_t* h, trav_counters* cn, int ask);
static int traverse(const orc_ctx* c, v3 o, v3 d, float tmax, int any_hit, hit_t* h, trav_counters* cn) { return traverse_ex(c, o, d, tmax, any_hit, h, cn, 1); }
static int traverse_ex(const orc_ctx* c, v3 o, v3 d, float tmax, int any_hit, hit_t* h, trav_counters* cn, int ask)
{
  uint32_t stack[STACK_MAX]; int sp = 0;
  float ix = inv_dir(d.x), iy = inv_dir(d.y), iz = inv_dir(d.z);
  /* guard band of the slab test along each axis, in t: 2^-21 * |1/d| * (|o - c|_1 + 3 h) for the tree with box centre c and L1
   * half-extent h the walk is in (the scene; inside an instance: the object) */
  float gx, gy, gz;
#define SET_GUARD(LO, HI) do { \
    const float cx_ = ((LO)[0] + (HI)[0]) * 0.5f, cy_ = ((LO)[1] + (HI)[1]) * 0.5f, cz_ = ((LO)[2] + (HI)[2]) * 0.5f; \
    const float h_ = ((((HI)[0] - (LO)[0]) + ((HI)[1] - (LO)[1])) + ((HI)[2] - (LO)[2])) * 0.5f; \
    const float R_ = CRH_FMA(h_, 3.0f, (crh_abs(o.x - cx_) + crh_abs(o.y - cy_)) + crh_abs(o.z - cz_)) * SLAB_GUARD; \
    gx = crh_abs(ix) * R_; gy = crh_abs(iy) * R_; gz = crh_abs(iz) * R_; } while (0)
  SET_GUARD(c->bbmin, c->bbmax);
  float best = tmax; int found = 0;
  h->t = tmax; h->u = 0.f; h->v = 0.f; h->prim = -1;
  const v3 wo = o, wd = d;                       /* the world-space ray (restored when an object is left) */
  uint32_t cur = c->root;
  if (c->root2 != QBVH_EMPTY) {
    /* static tree first; the top-level tree over the moved objects waits at the bottom of the stack -- if the ray comes near a moved object at all
     * (include/crh_math.h, crh_ray_near_sphere): the sphere around the bounds of ALL instances, then, when there are at most MAX_IBOX of them, the
     * sphere of at least one (ONE instance: the two are the same numbers, one test).  Rays handed in through the API are not asked. */
    int touch = 1;
    if (ask) {
      if (c->nInst != 1u) touch = crh_ray_near_sphere(o, d, tmax, c->usph[0], c->usph[1], c->usph[2], c->usph[3]);
      if (touch && c->nInst <= MAX_IBOX) {
        touch = 0;
        for (uint32_t i = 0; i < c->nInst && !touch; ++i) touch = crh_ray_near_sphere(o, d, tmax, c->inst[i].sph[0], c->inst[i].sph[1], c->inst[i].sph[2], c->inst[i].sph[3]);
      }
    }
    if (touch) stack[sp++] = c->root2;
  }
  for (;;) {
    if ((cur & 0xF0000000u) == CRH_REF_INSTANCE_TAG) {
      /* top-level leaf: express the ray in the object's space (direction NOT renormalised, so t is unchanged),
       * mark the stack, continue at the object's root */
      const orc_instance* in = &c->inst[c->tlas_order[cur & 0x0FFFFFFFu]];
      o = crh_xform_point(in->inv, wo); d = crh_xform_vector(in->inv, wd);
      ix = inv_dir(d.x); iy = inv_dir(d.y); iz = inv_dir(d.z);
      SET_GUARD(in->bmin, in->bmax);
      stack[sp++] = CRH_REF_SENTINEL;
      cur = in->root;
      continue;
    }
    if (cur & QBVH_LEAFBIT) {
      const uint32_t off = cur & 0x0FFFFFFFu;                  /* one triangle per leaf */
      float t, u, v; ORC_COUNT(any_hit ? cn->tris_any++ : cn->tris++);
      if (tri_test(&c->qtris[off], o, d, best, &t, &u, &v)) {
        best = t; found = 1; h->t = t; h->u = u; h->v = v; h->prim = (int32_t)crh_f2u(c->qtris[off].f[3]);
        if (any_hit) return 1;
      }
    } else {
      const qnode* q = &c->nodes[cur]; ORC_COUNT(any_hit ? cn->nodes_any++ : cn->nodes++);
      /* decode the per-node grid: face t = fma(q, step * inv, fma(origin - o, inv, -+ guard)) */
      const uint32_t ew = q->w[3];
      const float ax = crh_quant_step(CRH_NODE_STEP_E(ew, 0)) * ix, ay = crh_quant_step(CRH_NODE_STEP_E(ew, 1)) * iy, az = crh_quant_step(CRH_NODE_STEP_E(ew, 2)) * iz;
      const float ddx = crh_u2f(q->w[0]) - o.x, ddy = crh_u2f(q->w[1]) - o.y, ddz = crh_u2f(q->w[2]) - o.z;
      const float bix = CRH_FMA(ddx, ix, -gx), box = CRH_FMA(ddx, ix, gx);
      const float biy = CRH_FMA(ddy, iy, -gy), boy = CRH_FMA(ddy, iy, gy);
      const float biz = CRH_FMA(ddz, iz, -gz), boz = CRH_FMA(ddz, iz, gz);
#if CRH_SPEC_ORDER_EXACT
      uint64_t key[4];           /* crh_spec.h #4: exact entry distance, ties by slot */
#else
      uint32_t key[4];
#endif
      uint32_t rf[4]; int nh = 0;
      const int nch = (int)CRH_NODE_NCHILDREN(ew);
      for (int k = 0; k < nch; ++k) {
        uint32_t r = crh_node_child_ref(q->w, (uint32_t)k);
        const int sh = 8 * k;
        /* along a negative direction the ray enters through the upper plane */
        const float qix = (float)(((ix < 0.f ? q->w[7] : q->w[4]) >> sh) & 0xffu), qox = (float)(((ix < 0.f ? q->w[4] : q->w[7]) >> sh) & 0xffu);
        const float qiy = (float)(((iy < 0.f ? q->w[8] : q->w[5]) >> sh) & 0xffu), qoy = (float)(((iy < 0.f ? q->w[5] : q->w[8]) >> sh) & 0xffu);
        const float qiz = (float)(((iz < 0.f ? q->w[9] : q->w[6]) >> sh) & 0xffu), qoz = (float)(((iz < 0.f ? q->w[6] : q->w[9]) >> sh) & 0xffu);
        float a0 = CRH_FMA(qix, ax, bix), a1 = CRH_FMA(qox, ax, box);
        float b0 = CRH_FMA(qiy, ay, biy), b1 = CRH_FMA(qoy, ay, boy);
        float c0 = CRH_FMA(qiz, az, biz), c1 = CRH_FMA(qoz, az, boz);
        float tmin = crh_max(crh_max(crh_max(a0, b0), c0), 0.f);
        float tmx  = crh_min(crh_min(crh_min(a1, b1), c1), best);
        if (tmin <= tmx) {
          /* order key: entry distance with the slot index in the two low mantissa bits -> unique keys,
           * ascending unsigned order == near-to-far, ties by slot */
          int32_t bits = (int32_t)crh_f2u(tmin); if (bits < 0) bits = 0;
#if CRH_SPEC_ORDER_EXACT
          uint64_t ky = ((uint64_t)(uint32_t)bits << 2) | (uint64_t)k;
#else
          uint32_t ky = ((uint32_t)bits & ~3u) | (uint32_t)k;
#endif
          if (any_hit && CRH_SPEC_ANYHIT_SLOT_ORDER) ky = (uint32_t)k;      /* crh_spec.h #8: occlusion queries take the hit children in slot order */
          int j = nh++;
          while (j > 0 && key[j - 1] > ky) { key[j] = key[j - 1]; rf[j] = rf[j - 1]; --j; }
          key[j] = ky; rf[j] = r;
        }
      }
      if (nh > 0) {
        for (int j = nh - 1; j >= 1; --j) stack[sp++] = rf[j];   /* far .. near */
        cur = rf[0];
        continue;
      }
    }
    if (sp == 0) break;
    cur = stack[--sp];
    if (cur == CRH_REF_SENTINEL) {                /* back to world space */
      o = wo; d = wd;
      ix = inv_dir(d.x); iy = inv_dir(d.y); iz = inv_dir(d.z);
      SET_GUARD(c->bbmin, c->bbmax);
      if (sp == 0) break;
      cur = stack[--sp];
    }
  }
  return found;
}

/* ================================================================== BSDF */
typedef struct { v3 Kc; float Rc; v3 Kd; v3 Ks; float Rs; v3 Kt; v3 Le; v3 Fc; float fc[4]; float fb[4]; float ab[4]; } bsdf_t;

static v3 fresnel_media(float cosI, const float f[4])
{
  if (f[0] > -0.5f) {                                 /* Schlick: F0 + (1-F0)(1-|cos|)^5 */
    float m = 1.0f - crh_abs(cosI); float m2 = m * m; float m5 = (m2 * m2) * m;
    return crh_mk3(CRH_FMA(1.0f - f[0], m5, f[0]), CRH_FMA(1.0f - f[1], m5, f[1]), CRH_FMA(1.0f - f[2], m5, f[2]));
  }
  if (f[0] > -1.5f) return crh_mk3(f[2], f[2], f[2]); /* constant */
  if (f[0] > -2.5f) {                                 /* conductor (n = y, k = z), unpolarised approx. */
    float ci = crh_abs(cosI), n = f[1], k = f[2];
    float tmp = (2.0f * n) * ci;
    float t1 = CRH_FMA(n, n, k * k);
    float ci2 = ci * ci;
    float sperp = ((t1 - tmp) + ci2) / ((t1 + tmp) + ci2);
    float t2 = t1 * ci2;
    float sparl = ((t2 - tmp) + 1.0f) / ((t2 + tmp) + 1.0f);
    float r = (sperp + sparl) * 0.5f;
    return crh_mk3(r, r, r);
  }
  {                                                   /* dielectric (n = y), signed cosine */
    float n = f[1];
    float etaI = cosI > 0.f ? 1.0f : n, etaT = cosI > 0.f ? n : 1.0f;
    float r = 1.0f;
    float ratio = etaI / etaT;
    float sinT2 = (ratio * ratio) * CRH_FMA(-cosI, cosI, 1.0f);
    if (sinT2 < 1.0f) {
      float ci = crh_abs(cosI), ct = crh_sqrt(1.0f - sinT2);
      float parl = (etaT * ci - etaI * ct) / (etaT * ci + etaI * ct);
      float perp = (etaI * ci - etaT * ct) / (etaI * ci + etaT * ct);
      r = (parl * parl + perp * perp) * 0.5f;
    }
    return crh_mk3(r, r, r);
  }
}

static float smith_g1(v3 w, v3 m, float rough)
{
  float r = 0.f;
  if (crh_dot3(w, m) * w.z > 0.f) {
    float tanT = crh_sqrt(crh_max(CRH_FMA(-w.z, w.z, 1.0f), 0.f)) / w.z;
    if (tanT == 0.f) r = 1.0f;
    else {
      float a = 1.0f / (rough * tanT);
      r = CRH_FMA(2.181f, a, 3.535f) / CRH_FMA(2.577f, a, 1.0f / a + 2.276f);
    }
  }
  return crh_min(r, 1.0f);
}

static float blinn_power(float rough) { return crh_max(2.0f / (rough * rough) - 2.0f, 0.f); }

/* f * cos(theta_i) of the Blinn microfacet lobe, local frame, wi.z, wo.z > 0 required */
static v3 eval_blinn(v3 wi, v3 wo, const float fr[4], float rough)
{
  if (wi.z <= 0.f || wo.z <= 0.f) return crh_mk3(0.f, 0.f, 0.f);
  v3 h = crh_norm3(crh_add3(wi, wo));
  float e = blinn_power(rough);
  float D = ((e + 2.0f) * CRH_INV_TWOPI) * crh_pow(h.z, e);
  float G = smith_g1(wo, h, rough) * smith_g1(wi, h, rough);
  v3 F = fresnel_media(crh_dot3(wo, h), fr);
  float s = (D * G) / (4.0f * wo.z);
  return crh_scale3(F, s);
}

static v3 eval_layered(const bsdf_t* b, v3 wi, v3 wo, int two_sided)
{
  if (two_sided) { wi.z *= crh_sign(wo.z); wo.z *= crh_sign(wo.z); }   /* same hemisphere test in the +z frame */
  float lam = (wi.z <= 0.f || wo.z <= 0.f) ? 0.f : wi.z * CRH_INV_PI;
  v3 r = crh_scale3(b->Kd, lam);
  if (b->Rs > BSDF_EPS) r = crh_add3(r, crh_mul3(b->Ks, eval_blinn(wi, wo, b->fb, b->Rs)));
  r = crh_mul3(r, crh_mk3(1.0f - b->Fc.x, 1.0f - b->Fc.y, 1.0f - b->Fc.z));
  if (b->Rc > BSDF_EPS) r = crh_add3(r, crh_mul3(b->Kc, eval_blinn(wi, wo, b->fc, b->Rc)));
  return r;
}

typedef struct { float pc, pd, ps, pt, total; v3 Tc; } lobes_t;
static void lobe_probs(const bsdf_t* b, v3 W, lobes_t* L)
{
  L->Tc = crh_mk3(1.0f - b->Fc.x, 1.0f - b->Fc.y, 1.0f - b->Fc.z);
  L->pc = crh_dot3(crh_mul3(b->Kc, b->Fc), W);
  L->pd = crh_dot3(crh_mul3(b->Kd, L->Tc), W);
  L->ps = crh_dot3(crh_mul3(b->Ks, L->Tc), W);
  L->pt = crh_dot3(crh_mul3(b->Kt, L->Tc), W);
  L->total = ((L->pc + L->pd) + L->ps) + L->pt;
}

static float blinn_pdf(float hz, float dotih, float rough)
{
  float e = blinn_power(rough);
  return (((e + 2.0f) * CRH_INV_TWOPI) * crh_pow(crh_abs(hz), e + 1.0f)) / (4.0f * crh_abs(dotih));
}

/* lobe < 0: mixture pdf of the non-delta lobes for direction wi (solid angle); lobe = 0 coat / 1 diffuse / 2 glossy: that lobe's pdf
 * times its selection probability only (crh_spec.h #3) */
static float pdf_layered(const bsdf_t* b, v3 wo, v3 wi, v3 W, int two_sided, int lobe)
{
  lobes_t L; lobe_probs(b, W, &L);
  if (!(L.total > BSDF_EPS)) return 0.f;
  if (two_sided) { wi.z *= crh_sign(wo.z); wo.z *= crh_sign(wo.z); }
  float pdf = 0.f;
  if (wi.z > 0.f && wo.z > 0.f) {
    v3 h = crh_norm3(crh_add3(wi, wo));
    float dih = crh_dot3(wi, h);
    if (lobe < 0 || lobe == 1) pdf = L.pd * (wi.z * CRH_INV_PI);
    if (b->Rc > BSDF_EPS && (lobe < 0 || lobe == 0)) pdf = CRH_FMA(L.pc, blinn_pdf(h.z, dih, b->Rc), pdf);
    if (b->Rs > BSDF_EPS && (lobe < 0 || lobe == 2)) pdf = CRH_FMA(L.ps, blinn_pdf(h.z, dih, b->Rs), pdf);
  }
  return pdf / L.total;
}

/* Blinn half-vector sampling; returns f*cos/pdf (without K), sets *ok = 0 for a failed sample */
static v3 sample_blinn(v3 wo, v3* wi, const float fr[4], float rough, uint32_t* rng, int two_sided, int* ok, int u32)
{
  float k1 = crh_rng_next_mode(rng, u32), k2 = crh_rng_next_mode(rng, u32);
  float e = blinn_power(rough);
  float cm = crh_pow(k1, 1.0f / (e + 2.0f));
  float s, c; crh_sincos2pi(k2, &s, &c);
  float sm = crh_sqrt(crh_max(CRH_FMA(-cm, cm, 1.0f), 0.f));
  v3 m = crh_mk3(c * sm, s * sm, cm);
  int flip = 0;
  if (two_sided && wo.z < 0.f) { flip = 1; wo.z = -wo.z; }
  float cd = crh_dot3(wo, m);
  *wi = crh_mk3(CRH_FMA(2.0f * cd, m.x, -wo.x), CRH_FMA(2.0f * cd, m.y, -wo.y), CRH_FMA(2.0f * cd, m.z, -wo.z));
  if (wi->z <= 0.f || wo.z <= 0.f || !(cd > 0.f)) { *ok = 0; return crh_mk3(0.f, 0.f, 0.f); }
  float G = smith_g1(wo, m, rough) * smith_g1(*wi, m, rough);
  v3 F = fresnel_media(cd, fr);
  float w = (G * cd) / (wo.z * m.z);
  if (flip) wi->z = -wi->z;
  *ok = 1;
  return crh_scale3(F, w);
}

/* Sample the layered BSDF.  In: wo, throughput *W (updated), *inside (toggled on transmission).
 * Out: wi (local), *delta = 1 when a delta lobe was chosen.  Returns 0 when the path dies. */
static int sample_layered(const bsdf_t* b, v3 wo, v3* wi, v3* W, int* inside, int* delta, uint32_t* rng, int two_sided,
                          const crh_spec* sp, int* lobe)
{
  const int u32 = sp->uniform_32bit;
  lobes_t L; lobe_probs(b, *W, &L);
  float ksi = L.total * crh_rng_next_mode(rng, u32);
  *delta = 0; *lobe = 0;
  if (!(L.total > BSDF_EPS)) { *W = crh_mk3(0.f, 0.f, 0.f); return 0; }
  v3 mirror = crh_mk3(-wo.x, -wo.y, wo.z);
  int ok = 1; v3 k;
  if (ksi < L.pc) {                                           /* coat reflection */
    k = crh_scale3(b->Kc, L.total / L.pc);
    if (b->Rc > BSDF_EPS) k = crh_mul3(k, sample_blinn(wo, wi, b->fc, b->Rc, rng, two_sided, &ok, u32));
    else { k = crh_mul3(k, b->Fc); *wi = mirror; *delta = 1; }
  } else if (ksi < L.pc + L.pd) {                             /* diffuse base */
    k = crh_scale3(crh_mul3(b->Kd, L.Tc), L.total / L.pd); *lobe = 1;
    float k1 = crh_rng_next_mode(rng, u32), k2 = crh_rng_next_mode(rng, u32);
    float s, c; crh_sincos2pi(k1, &s, &c);
    float r = crh_sqrt(k2);
    *wi = crh_mk3(c * r, s * r, crh_sqrt(1.0f - k2));
    if (two_sided) { if (wo.z < 0.f) wi->z = -wi->z; }
    else if (!(wo.z > 0.f)) ok = 0;
  } else if (ksi < (L.pc + L.pd) + L.ps) {                    /* glossy base */
    k = crh_scale3(crh_mul3(b->Ks, L.Tc), L.total / L.ps); *lobe = 2;
    if (b->Rs > BSDF_EPS) k = crh_mul3(k, sample_blinn(wo, wi, b->fb, b->Rs, rng, two_sided, &ok, u32));
    else { k = crh_mul3(k, fresnel_media(wo.z, b->fb)); *wi = mirror; *delta = 1; }
  } else {                                                    /* specular transmission */
    k = crh_scale3(crh_mul3(b->Kt, L.Tc), L.total / L.pt); *lobe = 3;
    float ior = b->fc[0] > -2.5f ? sp->eta_no_dielectric : b->fc[1];   /* no dielectric coat: crh_spec.h #7 (default 1: index-matched, straight through) */
    float eta = wo.z > 0.f ? 1.0f / ior : ior;
    float sinT2 = (eta * eta) * CRH_FMA(-wo.z, wo.z, 1.0f);
    if (!(sinT2 < 1.0f) || !(L.pt > 0.f)) ok = 0;
    else {
      float ct = crh_sqrt(1.0f - sinT2); if (wo.z > 0.f) ct = -ct;
      *wi = crh_norm3(crh_mk3(-(eta * wo.x), -(eta * wo.y), ct));
      *inside = !*inside; *delta = 1;
    }
  }
  if (!ok) { *W = crh_mk3(0.f, 0.f, 0.f); return 0; }
  *W = crh_mul3(*W, k);
  return 1;
}

static void load_bsdf(const orc_ctx* c, int32_t mat, bsdf_t* b)
{
  const crh_bsdf* m = &c->mats[(mat >= 0 && (uint32_t)mat < c->nM) ? mat : 0];
  b->Kc = crh_mk3(m->Kc[0], m->Kc[1], m->Kc[2]); b->Rc = m->Kc[3];
  b->Kd = crh_mk3(m->Kd[0], m->Kd[1], m->Kd[2]);
  b->Ks = crh_mk3(m->Ks[0], m->Ks[1], m->Ks[2]); b->Rs = m->Ks[3];
  b->Kt = crh_mk3(m->Kt[0], m->Kt[1], m->Kt[2]);
  b->Le = crh_mk3(m->Le[0], m->Le[1], m->Le[2]);
  memcpy(b->fc, m->FresnelCoat, 16); memcpy(b->fb, m->FresnelBase, 16); memcpy(b->ab, m->Absorption, 16);
}

/* diffuse texture (reference AisMesh.cxx:321-346, rttexture -scale ImportExportPlugin.cxx:679-727):
 * Kd *= bilinear texel at the repeat-wrapped, scaled, barycentrically interpolated uv; row 0 of the image is v = 1.
 * An RGBA texture's alpha a cuts the surface out: Kd *= a, Kt = (1 - a) + a * Kt  (alpha -> transmission) */
static float lerpf(float a, float b, float t);
static void apply_texture(const orc_ctx* c, const int32_t* ti, float w0, float u, float v, bsdf_t* b)
{
  const crh_bsdf* m = &c->mats[(ti[3] >= 0 && (uint32_t)ti[3] < c->nM) ? ti[3] : 0];
  int slot = (int)m->Kd[3] - 1;
  if (slot < 0 || (uint32_t)slot >= c->nTex || !c->uv || !c->tex[slot].rgb) return;
  int W = (int)c->tex[slot].w, H = (int)c->tex[slot].h;
  const float* t0 = &c->uv[2 * ti[0]]; const float* t1 = &c->uv[2 * ti[1]]; const float* t2 = &c->uv[2 * ti[2]];
  float ss = m->Kt[3] != 0.f ? m->Kt[3] : 1.0f, st = m->Le[3] != 0.f ? m->Le[3] : 1.0f;
  float us = CRH_FMA(t2[0], v, CRH_FMA(t1[0], u, t0[0] * w0)) * ss;
  float vs = CRH_FMA(t2[1], v, CRH_FMA(t1[1], u, t0[1] * w0)) * st;
  if (!(crh_abs(us) < 4194304.0f)) us = 0.f;      /* beyond 2^22: no fraction left, (int) would overflow -> coordinate 0 */
  if (!(crh_abs(vs) < 4194304.0f)) vs = 0.f;
  float uf = (float)(int)us; if (uf > us) uf -= 1.0f;
  float vf = (float)(int)vs; if (vf > vs) vf -= 1.0f;
  float x = CRH_FMA(us - uf, (float)W, -0.5f), y = CRH_FMA(1.0f - (vs - vf), (float)H, -0.5f);
  float xf = (float)(int)x; if (xf > x) xf -= 1.0f;
  float yf = (float)(int)y; if (yf > y) yf -= 1.0f;
  float fx = x - xf, fy = y - yf;
  int x0 = (int)xf; if (x0 < 0) x0 += W; if (x0 >= W) x0 -= W; int x1 = x0 + 1; if (x1 >= W) x1 = 0;
  int y0 = (int)yf; if (y0 < 0) y0 += H; if (y0 >= H) y0 -= H; int y1 = y0 + 1; if (y1 >= H) y1 = 0;
  const float* img = c->tex[slot].rgb; const int ch = (int)c->tex[slot].ch;
  const float* p00 = &img[ch * (y0 * W + x0)]; const float* p10 = &img[ch * (y0 * W + x1)];
  const float* p01 = &img[ch * (y1 * W + x0)]; const float* p11 = &img[ch * (y1 * W + x1)];
  v3 tx = crh_mk3(lerpf(lerpf(p00[0], p10[0], fx), lerpf(p01[0], p11[0], fx), fy),
                  lerpf(lerpf(p00[1], p10[1], fx), lerpf(p01[1], p11[1], fx), fy),
                  lerpf(lerpf(p00[2], p10[2], fx), lerpf(p01[2], p11[2], fx), fy));
  if (c->spec.texel_gamma2) tx = crh_mul3(tx, tx);      /* crh_spec.h #2: the filtered texel squared (never the alpha) */
  b->Kd = crh_mul3(b->Kd, tx);
  if (ch == 4) {
    float a = lerpf(lerpf(p00[3], p10[3], fx), lerpf(p01[3], p11[3], fx), fy);
    if (a != 1.0f) {
      b->Kd = crh_scale3(b->Kd, a);
      float ia = 1.0f - a;
      b->Kt = crh_mk3(CRH_FMA(a, b->Kt.x, ia), CRH_FMA(a, b->Kt.y, ia), CRH_FMA(a, b->Kt.z, ia));
    }
  }
}

/* ================================================================== lights / env (a11, a12) */
typedef struct { v3 t, b, n; } frame_t;
static frame_t make_frame(v3 n)
{
  frame_t f; f.n = n;
  v3 t = (crh_abs(n.x) > crh_abs(n.z)) ? crh_mk3(-n.y, n.x, 0.f) : crh_mk3(0.f, -n.z, n.y);
  f.t = crh_norm3(t); f.b = crh_cross3(n, f.t);
  return f;
}
static v3 to_local(const frame_t* f, v3 v) { return crh_mk3(crh_dot3(v, f->t), crh_dot3(v, f->b), crh_dot3(v, f->n)); }
static v3 from_local(const frame_t* f, v3 l)
{
  return crh_mk3(CRH_FMA(f->n.x, l.z, CRH_FMA(f->b.x, l.y, f->t.x * l.x)),
                 CRH_FMA(f->n.y, l.z, CRH_FMA(f->b.y, l.y, f->t.y * l.x)),
                 CRH_FMA(f->n.z, l.z, CRH_FMA(f->b.z, l.y, f->t.z * l.x)));
}

static float lerpf(float a, float b, float t) { return CRH_FMA(t, b - a, a); }

static v3 env_lookup(const orc_ctx* c, v3 d)
{
  if (!c->env) return crh_mk3(c->par.background[0], c->par.background[1], c->par.background[2]);
  float u = (crh_atan2(d.y, d.x) + CRH_PI) * CRH_INV_TWOPI;
  float v = crh_acos(d.z) * CRH_INV_PI;
  if (c->spec.env_orientation) { u = crh_atan2(d.y, d.x) * CRH_INV_TWOPI; v = crh_acos(-d.z) * CRH_INV_PI; }      /* crh_spec.h #14 */
  float x = CRH_FMA(u, (float)c->envW, -0.5f), y = CRH_FMA(v, (float)c->envH, -0.5f);
  float xf = (float)(int)x; if (xf > x) xf -= 1.0f;
  float yf = (float)(int)y; if (yf > y) yf -= 1.0f;
  float fx = x - xf, fy = y - yf;
  int W = (int)c->envW, H = (int)c->envH;
  int x0 = (int)xf % W; if (x0 < 0) x0 += W; int x1 = x0 + 1; if (x1 >= W) x1 = 0;
  int y0 = (int)yf; int y1 = y0 + 1;
  if (y0 < 0) y0 = 0; if (y0 > H - 1) y0 = H - 1; if (y1 < 0) y1 = 0; if (y1 > H - 1) y1 = H - 1;
  const float* p00 = &c->env[3 * (y0 * W + x0)]; const float* p10 = &c->env[3 * (y0 * W + x1)];
  const float* p01 = &c->env[3 * (y1 * W + x0)]; const float* p11 = &c->env[3 * (y1 * W + x1)];
  v3 r = crh_mk3(lerpf(lerpf(p00[0], p10[0], fx), lerpf(p01[0], p11[0], fx), fy),
                 lerpf(lerpf(p00[1], p10[1], fx), lerpf(p01[1], p11[1], fx), fy),
                 lerpf(lerpf(p00[2], p10[2], fx), lerpf(p01[2], p11[2], fx), fy));
  if (c->spec.texel_gamma2) r = crh_mul3(r, r);         /* crh_spec.h #2 */
  return r;
}

static float cone_pdf(float cosmax) { return 1.0f / (CRH_TWO_PI * (1.0f - cosmax)); }
static float sphere_cosmax(float radius, float dist) { return 1.0f / crh_sqrt(CRH_FMA(radius / dist, radius / dist, 1.0f)); }

/* Radiance arriving along a BSDF-sampled (or camera) ray from the analytic lights in front of the
 * surface hit, or from the environment on a miss; *exp_pdf = pdf NEE would have had. */
static v3 intersect_light(const orc_ctx* c, v3 o, v3 d, int bounce, float hit_t_, float* exp_pdf)
{
  v3 rad = crh_mk3(0.f, 0.f, 0.f); float pdf = 0.f; float hd = hit_t_;
  float sel = c->nL ? 1.0f / (float)c->nL : 0.f;
  for (uint32_t i = 0; i < c->nL; ++i) {
    const crh_light* l = &c->lights[i];
    if (l->is_point != 0.f) {
      v3 tl = crh_sub3(c->l_vec[i], o);
      float dist = crh_len3(tl);
      if (dist < hd) {
        float cm = sphere_cosmax(c->l_par[i], dist);
        if (cm < 1.0f && crh_dot3(d, tl) * (1.0f / dist) >= cm) {
          hd = dist; rad = crh_mk3(l->emission[0], l->emission[1], l->emission[2]); pdf = sel * cone_pdf(cm);
        }
      }
    } else if (hd == CRH_MAXFLOAT) {
      float cm = c->l_par[i];
      if (cm < 1.0f && crh_dot3(d, c->l_vec[i]) >= cm) {
        rad = crh_add3(rad, crh_mk3(l->emission[0], l->emission[1], l->emission[2]));
        pdf += sel * cone_pdf(cm);
      }
    }
  }
  if (pdf == 0.f && hd == CRH_MAXFLOAT) {
    if (bounce == 0 && !c->par.env_as_background)
      rad = crh_mk3(c->par.background[0], c->par.background[1], c->par.background[2]);
    else rad = env_lookup(c, d);
  }
  *exp_pdf = pdf;
  return rad;
}

/* ================================================================== path integrator (a13) */
static void gen_camera_ray(const orc_ctx* c, uint32_t px, uint32_t py, uint32_t* rng, v3* o, v3* d)
{
  const int u32 = c->spec.uniform_32bit;
  float jx = crh_rng_next_mode(rng, u32), jy = crh_rng_next_mode(rng, u32);
  float W = (float)c->par.width, H = (float)c->par.height;
  float nx = CRH_FMA(((float)px + jx) / W, 2.0f, -1.0f);
  float ny = CRH_FMA(((float)py + jy) / H, -2.0f, 1.0f);
  if (c->cam.is_ortho) {
    float sx = (nx * c->cam.ortho_scale) * c->c_aspect, sy = ny * c->cam.ortho_scale;
    *o = crh_madd3(crh_madd3(c->c_eye, c->c_right, sx), c->c_up, sy);
    *d = c->c_fwd;
  } else if (c->spec.raygen_bilinear) {
    /* crh_spec.h #13 (SURVEY a2, Appendix A GenerateRay): blend of the four frustum-corner directions by the pixel's position in [0,1]^2 */
    const float u = ((float)px + jx) / W, v = 1.0f - ((float)py + jy) / H;
    *o = c->c_eye;
    *d = crh_norm3(crh_lerp3(crh_lerp3(c->c_corner[0], c->c_corner[1], u), crh_lerp3(c->c_corner[2], c->c_corner[3], u), v));
  } else {
    float sx = (nx * c->c_tanh) * c->c_aspect, sy = ny * c->c_tanh;
    *o = c->c_eye;
    *d = crh_norm3(crh_madd3(crh_madd3(c->c_fwd, c->c_right, sx), c->c_up, sy));
  }
  if (c->cam.aperture_radius > 0.f) {
    float k1 = crh_rng_next_mode(rng, u32), k2 = crh_rng_next_mode(rng, u32);
    float ft = c->cam.focal_dist / crh_dot3(*d, c->c_fwd);
    v3 focus = crh_madd3(*o, *d, ft);
    float r = c->cam.aperture_radius * crh_sqrt(k1); float s, cc; crh_sincos2pi(k2, &s, &cc);
    *o = crh_madd3(crh_madd3(*o, c->c_right, r * cc), c->c_up, r * s);
    *d = crh_norm3(crh_sub3(focus, *o));
  }
}

static v3 offset_origin(v3 p, v3 dir, v3 ng, float eps)
{
  v3 o = crh_madd3(p, dir, eps);
  float s = crh_dot3(ng, dir) >= 0.f ? eps : -eps;
  return crh_madd3(o, ng, s);
}

static v3 path_trace(const orc_ctx* c, uint32_t px, uint32_t py, uint32_t fseed, crh_stats* st, trav_counters* cn)
{
  uint32_t pix = c->par.coherent_rng ? ((py / 16u) * ((c->par.width + 15u) / 16u) + (px / 16u)) : (py * c->par.width + px);
  uint32_t rng = crh_rng_seed(pix, fseed);
  v3 o, d; gen_camera_ray(c, px, py, &rng, &o, &d);
  v3 rad = crh_mk3(0.f, 0.f, 0.f), W = crh_mk3(1.0f, 1.0f, 1.0f);
  int inside = 0; float imp_pdf = CRH_MAXFLOAT;
  int two = c->par.two_sided;
  const int u32 = c->spec.uniform_32bit;
  for (uint32_t bounce = 0; bounce < c->par.max_depth; ++bounce) {
    hit_t h; st->rays_nearest++;
    int found = traverse(c, o, d, CRH_MAXFLOAT, 0, &h, cn);
    float exp_pdf;
    v3 le = intersect_light(c, o, d, (int)bounce, found ? h.t : CRH_MAXFLOAT, &exp_pdf);
    if (le.x > 0.f || le.y > 0.f || le.z > 0.f || !found) {
      float mis = (bounce == 0 || imp_pdf == CRH_MAXFLOAT) ? 1.0f : (imp_pdf * imp_pdf) / CRH_FMA(exp_pdf, exp_pdf, imp_pdf * imp_pdf);
      rad = crh_add3(rad, crh_scale3(crh_mul3(W, le), mis));
      break;
    }
    st->shaded_hits++;
    const int32_t* ti = &c->tri[4 * h.prim];
    const float* M = (c->two_level && c->obj[c->tri_obj[h.prim]].is_inst) ? &c->xf[12 * c->tri_obj[h.prim]] : NULL;      /* object -> world (an object in the static tree is in world space) */
    /* a hit in the static tree of a two-level scene sees the BAKED vertices and normals (build-time transform applied per vertex) */
    const float* VP = (c->two_level && !M) ? c->pos_w : c->pos; const float* VN = (c->two_level && !M) ? c->nrm_w : c->nrm;
    v3 p0 = crh_mk3(VP[3 * ti[0]], VP[3 * ti[0] + 1], VP[3 * ti[0] + 2]);
    v3 p1 = crh_mk3(VP[3 * ti[1]], VP[3 * ti[1] + 1], VP[3 * ti[1] + 2]);
    v3 p2 = crh_mk3(VP[3 * ti[2]], VP[3 * ti[2] + 1], VP[3 * ti[2] + 2]);
    if (M) { p0 = crh_xform_point(M, p0); p1 = crh_xform_point(M, p1); p2 = crh_xform_point(M, p2); }
    v3 ng = crh_norm3(crh_cross3(crh_sub3(p0, p2), crh_sub3(p1, p0)));
    float w0 = (1.0f - h.u) - h.v;
    const float* n0 = &VN[3 * ti[0]]; const float* n1 = &VN[3 * ti[1]]; const float* n2 = &VN[3 * ti[2]];
    v3 ns = crh_norm3(crh_mk3(CRH_FMA(n2[0], h.v, CRH_FMA(n1[0], h.u, n0[0] * w0)),
                               CRH_FMA(n2[1], h.v, CRH_FMA(n1[1], h.u, n0[1] * w0)),
                               CRH_FMA(n2[2], h.v, CRH_FMA(n1[2], h.u, n0[2] * w0))));
    if (M) ns = crh_norm3(crh_xform_vector(M, ns));          /* rigid + uniform scale (gp_Trsf): the 3x3 part carries normals */
    if (!(crh_dot3(ns, ns) > 0.f)) ns = ng;
    v3 p = crh_madd3(o, d, h.t);
    bsdf_t b; load_bsdf(c, ti[3], &b);
    apply_texture(c, ti, w0, h.u, h.v, &b);
    frame_t fr = make_frame(ns);
    v3 wo = to_local(&fr, crh_mk3(-d.x, -d.y, -d.z));
    b.Fc = fresnel_media(wo.z, b.fc);
    if (inside) {                                           /* Beer-Lambert along the segment just travelled */
      float k = -(h.t * b.ab[3]);
      W = crh_mul3(W, crh_mk3(crh_exp(k * (1.0f - b.ab[0])), crh_exp(k * (1.0f - b.ab[1])), crh_exp(k * (1.0f - b.ab[2]))));
    }
    rad = crh_add3(rad, crh_mul3(W, b.Le));
    /* next event estimation */
    {
      v3 nd = crh_add3(b.Kd, crh_add3(b.Rc > BSDF_EPS ? b.Kc : crh_mk3(0.f, 0.f, 0.f), b.Rs > BSDF_EPS ? b.Ks : crh_mk3(0.f, 0.f, 0.f)));
      if (c->nL > 0 && crh_dot3(nd, W) > BSDF_EPS) {
        float fl = crh_rng_next_mode(&rng, u32) * (float)c->nL;
        uint32_t li = (uint32_t)fl; if (li > c->nL - 1u) li = c->nL - 1u;
        float k1 = crh_rng_next_mode(&rng, u32), k2 = crh_rng_next_mode(&rng, u32);
        const crh_light* l = &c->lights[li];
        v3 axis; float dist, cm;
        if (l->is_point != 0.f) { v3 tl = crh_sub3(c->l_vec[li], p); dist = crh_len3(tl); axis = crh_scale3(tl, 1.0f / dist); cm = sphere_cosmax(c->l_par[li], dist); }
        else { axis = c->l_vec[li]; dist = CRH_MAXFLOAT; cm = c->l_par[li]; }
        frame_t lf = make_frame(axis);
        float ct = CRH_FMA(-k2, 1.0f - cm, 1.0f);
        float s, cc; crh_sincos2pi(k1, &s, &cc);
        float sn = crh_sqrt(crh_max(CRH_FMA(-ct, ct, 1.0f), 0.f));
        v3 ld = crh_norm3(from_local(&lf, crh_mk3(cc * sn, s * sn, ct)));
        float e_pdf = (cm < 1.0f) ? (1.0f / (float)c->nL) * cone_pdf(cm) : CRH_MAXFLOAT;
        v3 wi = to_local(&fr, ld);
        float i_pdf = pdf_layered(&b, wo, wi, W, two, -1);
        float mis = (e_pdf == CRH_MAXFLOAT) ? 1.0f : e_pdf / CRH_FMA(e_pdf, e_pdf, i_pdf * i_pdf);
        v3 contrib = crh_scale3(crh_mul3(crh_mk3(l->emission[0], l->emission[1], l->emission[2]), eval_layered(&b, wi, wo, two)), mis);
        v3 wc = crh_mul3(W, contrib);
        const float mc = c->spec.min_contribution;      /* crh_spec.h #11 */
        if (contrib.x > mc || contrib.y > mc || contrib.z > mc) {
          hit_t sh; st->rays_any++;
          v3 so = offset_origin(p, ld, ng, c->eps);
          if (!traverse(c, so, ld, dist, 1, &sh, cn)) rad = crh_add3(rad, wc);
        }
      }
    }
    /* BSDF sampling */
    v3 wi; int delta, lobe; v3 Wsel = W;      /* lobe-selection weights = throughput before the bounce */
    int alive = sample_layered(&b, wo, &wi, &W, &inside, &delta, &rng, two, &c->spec, &lobe);
    if (alive) imp_pdf = delta ? CRH_MAXFLOAT : pdf_layered(&b, wo, wi, Wsel, two, c->spec.mis_single_lobe ? lobe : -1);
    const float mt = c->spec.min_throughput;            /* crh_spec.h #12, #9, #10 */
    const int roulette = c->par.russian_roulette && bounce >= (uint32_t)c->spec.rr_start_bounce;
    float survive = (W.x > mt || W.y > mt || W.z > mt) ? 1.0f : 0.f;
    if (roulette)
      survive = crh_min(CRH_FMA(LUMA_B, W.z, CRH_FMA(LUMA_G, W.y, LUMA_R * W.x)), c->spec.rr_survival_cap) * survive;
    float kr = crh_rng_next_mode(&rng, u32);
    if (!alive || !(kr < survive)) break;
    if (roulette) W = crh_mk3(W.x / survive, W.y / survive, W.z / survive);
    v3 nd2 = crh_norm3(from_local(&fr, wi));
    o = offset_origin(p, nd2, ng, c->eps);
    d = nd2;
  }
  return rad;
}

/* ================================================================== accumulate / render (a15) */
static void prepare(orc_ctx* c)
{
  c->c_eye = crh_mk3(c->cam.eye[0], c->cam.eye[1], c->cam.eye[2]);
  c->c_fwd = crh_norm3(crh_mk3(c->cam.dir[0], c->cam.dir[1], c->cam.dir[2]));
  c->c_right = crh_norm3(crh_cross3(c->c_fwd, crh_mk3(c->cam.up[0], c->cam.up[1], c->cam.up[2])));
  c->c_up = crh_cross3(c->c_right, c->c_fwd);
  float s, cs; crh_sincos((c->cam.fovy_deg * 0.5f) * (CRH_PI / 180.0f), &s, &cs);
  c->c_tanh = s / cs;
  c->c_aspect = c->cam.aspect > 0.f ? c->cam.aspect : (float)c->par.width / (float)c->par.height;
  for (int k = 0; k < 4; ++k)
    c->c_corner[k] = crh_frustum_corner(c->c_fwd, c->c_right, c->c_up, c->c_tanh, c->c_aspect, (k & 1) ? 1.0f : -1.0f, (k & 2) ? 1.0f : -1.0f, c->spec.raygen_bilinear == 2);
  free(c->l_vec); free(c->l_par);
  c->l_vec = (v3*)malloc(sizeof(v3) * (c->nL ? c->nL : 1)); c->l_par = (float*)malloc(sizeof(float) * (c->nL ? c->nL : 1));
  for (uint32_t i = 0; i < c->nL; ++i) {
    const crh_light* l = &c->lights[i];
    if (l->is_point != 0.f) { c->l_vec[i] = crh_mk3(l->vec[0], l->vec[1], l->vec[2]); c->l_par[i] = l->smoothness; }
    else {
      c->l_vec[i] = crh_norm3(crh_mk3(-l->vec[0], -l->vec[1], -l->vec[2]));
      float sn, cn; crh_sincos(l->smoothness, &sn, &cn); c->l_par[i] = l->smoothness > 0.f ? cn : 1.0f;
    }
  }
  v3 dg = crh_mk3(c->bbmax[0] - c->bbmin[0], c->bbmax[1] - c->bbmin[1], c->bbmax[2] - c->bbmin[2]);
  c->eps = c->par.scene_epsilon > 0.f ? c->par.scene_epsilon
         : (c->spec.eps_rule ? crh_max(1.0e-6f, 1.0e-4f * (crh_len3(dg) * 0.5f)) : crh_max(1.0e-6f, 1.0e-5f * crh_len3(dg)));   /* crh_spec.h #6 */
}

static void accumulate_px(const orc_ctx* c, float* a, v3 s, float* m2)
{
  float clampv = c->par.radiance_clamp;
  float r[3] = {s.x, s.y, s.z};
  float n = a[3];
  float w = 1.0f / (n + 1.0f);
  for (int k = 0; k < 3; ++k) {
    float v = r[k];
    if (!(v == v)) v = 0.f;                 /* NaN -> 0 */
    if (clampv > 0.f && v > clampv) v = clampv;
    r[k] = v;
    a[k] = CRH_FMA(v - a[k], w, a[k]);
  }
  if (m2) {
    float l = CRH_FMA(LUMA_B, r[2], CRH_FMA(LUMA_G, r[1], LUMA_R * r[0]));
    *m2 = CRH_FMA(l * l - *m2, w, *m2);
  }
  a[3] = n + 1.0f;
}

static int render_tiles(orc_ctx* c, const uint32_t* tiles, uint32_t nt, uint32_t first, uint32_t ns, const uint32_t* tile_seeds)
{
  if (!c->built) return CRH_E_NOTBUILT;
  uint32_t ts = c->par.tile_size ? c->par.tile_size : 32u;
  uint32_t tx = (c->par.width + ts - 1) / ts, ty = (c->par.height + ts - 1) / ts;
  uint32_t* seeds = (uint32_t*)malloc(sizeof(uint32_t) * (ns ? ns : 1));
  for (uint32_t s = 0; s < ns; ++s) seeds[s] = frame_seed(c->par.seed, first + s);
  struct timespec t0, t1; clock_gettime(CLOCK_MONOTONIC, &t0);
  uint64_t rn = 0, ra = 0, nn = 0, tt = 0, na = 0, ta = 0, hh = 0, sm = 0;
  uint32_t total = tiles ? nt : tx * ty;
  /* work item = one 8x8 sub-block of a tile, so that the host threads are not left waiting for the last of a few tiles each (a parity
   * gate of 2 - 4 tiles x 10 240 samples would otherwise run on 2 - 4 threads) -- pixels are independent, so the image does not depend on it */
  const uint32_t sub = 8u;
  const uint32_t spr = (ts + sub - 1) / sub, nsub = spr * spr;
#pragma omp parallel for schedule(dynamic, 1) reduction(+ : rn, ra, nn, tt, na, ta, hh, sm)
  for (uint32_t wi = 0; wi < total * nsub; ++wi) {
    const uint32_t ti = wi / nsub, sb = wi - ti * nsub;
    uint32_t t = tiles ? tiles[ti] : ti;
    if (t >= tx * ty) continue;
    uint32_t x0 = (t % tx) * ts + (sb % spr) * sub, y0 = (t / tx) * ts + (sb / spr) * sub;
    uint32_t x1 = x0 + sub, y1 = y0 + sub;
    if (x1 > (t % tx) * ts + ts) x1 = (t % tx) * ts + ts;
    if (y1 > (t / tx) * ts + ts) y1 = (t / tx) * ts + ts;
    crh_stats st; memset(&st, 0, sizeof st); trav_counters cn = {0, 0, 0, 0};
    for (uint32_t s = 0; s < ns; ++s)
      for (uint32_t y = y0; y < y1 && y < c->par.height; ++y)
        for (uint32_t x = x0; x < x1 && x < c->par.width; ++x) {
          v3 r = path_trace(c, x, y, tile_seeds ? tile_seeds[ti] : seeds[s], &st, &cn);
          accumulate_px(c, &c->accum[4 * ((size_t)y * c->par.width + x)], r, c->adaptive ? &c->m2[(size_t)y * c->par.width + x] : NULL);
          st.samples++;
        }
    rn += st.rays_nearest; ra += st.rays_any; nn += cn.nodes; tt += cn.tris; na += cn.nodes_any; ta += cn.tris_any; hh += st.shaded_hits; sm += st.samples;
  }
  clock_gettime(CLOCK_MONOTONIC, &t1);
  c->st.rays_nearest += rn; c->st.rays_any += ra; c->st.nodes_nearest += nn; c->st.tris_nearest += tt; c->st.nodes_any += na; c->st.tris_any += ta;
  c->st.shaded_hits += hh; c->st.samples += sm;
  c->st.seconds += (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
  free(seeds);
  return 0;
}

/* ================================================================== tone map (a17) */
static float hable(float x)
{
  const float A = 0.22f, B = 0.30f, C = 0.10f, D = 0.20f, E = 0.01f, F = 0.30f;
  return (CRH_FMA(x, CRH_FMA(A, x, C * B), D * E) / CRH_FMA(x, CRH_FMA(A, x, B), D * F)) - E / F;
}
static uint8_t to_ldr(const crh_params* p, int gamma22, float v)
{
  if (!(v == v) || v < 0.f) v = 0.f;
  v = v * crh_exp(p->exposure * 0.69314718056f);
  if (p->tonemap_mode == 1) v = hable(v) / hable(p->white_point > 0.f ? p->white_point : 1.0f);
  v = crh_clamp(v, 0.f, 1.0f);
  v = gamma22 ? crh_pow(v, 1.0f / 2.2f) : crh_sqrt(v);      /* crh_spec.h #15 */
  return (uint8_t)(int)CRH_FMA(v, 255.0f, 0.5f);
}

/* ================================================================== C API (mirrors crh_*) */
ORC_API orc_ctx* orc_create(void)
{
  orc_ctx* c = (orc_ctx*)calloc(1, sizeof(orc_ctx));
  c->par.width = 64; c->par.height = 64; c->par.max_depth = 5; c->par.radiance_clamp = 0.f; c->par.two_sided = 1;
  c->par.seed = 1; c->par.tile_size = 32; c->par.white_point = 1.0f; c->par.russian_roulette = 1; c->par.env_as_background = 1;
  c->cam.dir[1] = 1.0f; c->cam.up[2] = 1.0f; c->cam.fovy_deg = 45.0f;
  { const crh_spec d = CRH_SPEC_DEFAULTS; c->spec = d; }
  return c;
}
ORC_API void orc_destroy(orc_ctx* c)
{
  if (!c) return;
  free(c->pos); free(c->nrm); free(c->uv); free(c->tri); free(c->mats); free(c->lights); free(c->env);
  free(c->nodes); free(c->qtris); free(c->l_vec); free(c->l_par); free(c->accum); free(c->m2); free(c->last_picked); free(c->xf); free(c->tri_obj); free(c->inst); free(c->tlas_order); free(c->obj); free(c->obj_tris); free(c->static_pos); free(c->xf0); free(c->pos_w); free(c->nrm_w); free(c->hidden); free(c);
}
ORC_API const char* orc_last_error(orc_ctx* c) { return c ? c->err : "null ctx"; }

/* the boundary takes finite numbers only (same rule as the product: include/cadrays_hip.h) */
static int all_finite(const float* v, size_t n, float limit)
{
  for (size_t i = 0; i < n; ++i) if (!(v[i] >= -limit && v[i] <= limit)) return 0;
  return 1;
}

ORC_API int orc_set_geometry(orc_ctx* c, const float* pos, const float* nrm, const float* uv, uint32_t nV,
                             const int32_t* tri, uint32_t nT, const int32_t* tri_obj, const float* xf, uint32_t nO)
{
  if (!c || (nV && (!pos || !nrm)) || (nT && !tri)) return CRH_E_INVALID;
  for (uint32_t t = 0; t < nT; ++t) for (int k = 0; k < 3; ++k) if (tri[4 * t + k] < 0 || (uint32_t)tri[4 * t + k] >= nV) { snprintf(c->err, sizeof c->err, "triangle %u index out of range", t); return CRH_E_INVALID; }
  if (!all_finite(pos, 3 * (size_t)nV, 1.0e30f) || !all_finite(nrm, 3 * (size_t)nV, 3.0e38f) || (uv && !all_finite(uv, 2 * (size_t)nV, 3.0e38f)) ||
      (xf && !all_finite(xf, 12 * (size_t)nO, 1.0e30f))) { snprintf(c->err, sizeof c->err, "geometry holds a NaN / Inf (or a coordinate beyond 1e30)"); return CRH_E_INVALID; }
  free(c->pos); free(c->nrm); free(c->xf); free(c->tri_obj); c->xf = NULL; c->tri_obj = NULL; c->two_level = 0; c->nO = 0;
  free(c->hidden); c->hidden = NULL;                /* a new scene: everything displayed */
  c->pos = (float*)dup_mem(pos, sizeof(float) * 3 * nV); c->nrm = (float*)dup_mem(nrm, sizeof(float) * 3 * nV);
  if (tri_obj && xf && nO) {
    /* two-level mode: vertices stay in object space; each object gets its own tree, instances carry the transforms */
    for (uint32_t t = 0; t < nT; ++t) if (tri_obj[t] < 0 || (uint32_t)tri_obj[t] >= nO) { snprintf(c->err, sizeof c->err, "triangle %u object id out of range", t); return CRH_E_INVALID; }
    /* every vertex belongs to one object (the bake of do_build applies one transform per vertex): same rule, same refusal as the product */
    int32_t* owner = (int32_t*)malloc(sizeof(int32_t) * (nV ? nV : 1));
    for (uint32_t v = 0; v < nV; ++v) owner[v] = -1;
    for (uint32_t t = 0; t < nT; ++t) for (int k = 0; k < 3; ++k) {
      int32_t* o = &owner[tri[4 * t + k]];
      if (*o < 0) *o = tri_obj[t];
      else if (*o != tri_obj[t]) { snprintf(c->err, sizeof c->err, "vertex %d is shared by objects %d and %d: each vertex belongs to one object (duplicate it)", tri[4 * t + k], *o, tri_obj[t]); free(owner); return CRH_E_INVALID; }
    }
    free(owner);
    c->xf = (float*)dup_mem(xf, sizeof(float) * 12 * nO); c->tri_obj = (int32_t*)dup_mem(tri_obj, sizeof(int32_t) * nT);
    c->two_level = 1; c->nO = nO;
  }
  free(c->uv); c->uv = uv ? (float*)dup_mem(uv, sizeof(float) * 2 * nV) : NULL;
  free(c->tri); c->tri = (int32_t*)dup_mem(tri, sizeof(int32_t) * 4 * nT);
  c->nV = nV; c->nT = nT; c->built = 0;
  return 0;
}
ORC_API int orc_reset(orc_ctx* c);
ORC_API int orc_set_transforms(orc_ctx* c, const float* xf, uint32_t nO)
{
  if (!c || !xf || !c->two_level || nO != c->nO || !all_finite(xf, 12 * (size_t)nO, 1.0e30f)) return CRH_E_INVALID;
  memcpy(c->xf, xf, sizeof(float) * 12 * nO);
  if (c->built) {
    /* static / moved split: the static tree and the object trees built so far are never rebuilt.  An object of the static tree that
     * leaves its build-time placement gets its triangles there disabled and (the first time) an object tree of its own; back there, its
     * triangles are restored and the instance is dropped.  Then the top-level tree over the instances of this moment. */
    apply_objects(c);
  }
  return orc_reset(c);
}
/* crh_set_visibility: AIS_InteractiveContext::Display / Erase of one or more objects (DataNode.cxx:304-344; the GUI's eye icons, `rtdisplay` / `rterase`
 * of ImportExportPlugin.cxx:373-425) without a rebuild */
ORC_API int orc_set_visibility(orc_ctx* c, const uint8_t* visible, uint32_t nO)
{
  if (!c || !visible || !c->two_level || nO != c->nO) return CRH_E_INVALID;
  if (!c->hidden) c->hidden = (uint8_t*)calloc(nO ? nO : 1, 1);
  for (uint32_t ob = 0; ob < nO; ++ob) c->hidden[ob] = visible[ob] ? 0 : 1;
  if (c->built) apply_objects(c);
  return orc_reset(c);
}
/* crh_add_object: AIS_InteractiveContext::Display of a NEW object in a built scene (ImportExportPlugin.cxx:132-354 `rtmeshread` into a running viewer):
 * the arrays grow, the object gets an object-space tree and enters the top level as an instance; the next full build bakes it like the rest */
ORC_API int orc_add_object(orc_ctx* c, const float* pos, const float* nrm, const float* uv, uint32_t nV, const int32_t* tri, uint32_t nT,
                           const float* xform, uint32_t* object_out)
{
  if (!c || !pos || !nrm || !tri || !xform || !nV || !nT) return CRH_E_INVALID;
  if (!c->built) return CRH_E_NOTBUILT;
  if (!c->two_level) { snprintf(c->err, sizeof c->err, "crh_add_object needs a scene handed over with objects"); return CRH_E_INVALID; }
  for (uint32_t t = 0; t < nT; ++t) for (int k = 0; k < 3; ++k) if (tri[4 * t + k] < 0 || (uint32_t)tri[4 * t + k] >= nV) { snprintf(c->err, sizeof c->err, "triangle %u index out of range", t); return CRH_E_INVALID; }
  if (!all_finite(pos, 3 * (size_t)nV, 1.0e30f) || !all_finite(nrm, 3 * (size_t)nV, 3.0e38f) || (uv && !all_finite(uv, 2 * (size_t)nV, 3.0e38f)) || !all_finite(xform, 12, 1.0e30f)) {
    snprintf(c->err, sizeof c->err, "geometry holds a NaN / Inf (or a coordinate beyond 1e30)"); return CRH_E_INVALID; }
  if ((uint64_t)c->nT + nT >= (1u << 28)) return CRH_E_INVALID;
  const uint32_t V0 = c->nV, T0 = c->nT, ob = c->nO;
  c->pos = (float*)realloc(c->pos, sizeof(float) * 3 * (V0 + nV)); memcpy(c->pos + 3 * (size_t)V0, pos, sizeof(float) * 3 * nV);
  c->nrm = (float*)realloc(c->nrm, sizeof(float) * 3 * (V0 + nV)); memcpy(c->nrm + 3 * (size_t)V0, nrm, sizeof(float) * 3 * nV);
  c->pos_w = (float*)realloc(c->pos_w, sizeof(float) * 3 * (V0 + nV)); memcpy(c->pos_w + 3 * (size_t)V0, pos, sizeof(float) * 3 * nV);
  c->nrm_w = (float*)realloc(c->nrm_w, sizeof(float) * 3 * (V0 + nV)); memcpy(c->nrm_w + 3 * (size_t)V0, nrm, sizeof(float) * 3 * nV);
  if (c->uv) {                                   /* a scene with texture coordinates keeps one pair per vertex (zeros when the new object brings none) */
    c->uv = (float*)realloc(c->uv, sizeof(float) * 2 * (V0 + nV));
    if (uv) memcpy(c->uv + 2 * (size_t)V0, uv, sizeof(float) * 2 * nV); else memset(c->uv + 2 * (size_t)V0, 0, sizeof(float) * 2 * nV);
  }
  c->tri = (int32_t*)realloc(c->tri, sizeof(int32_t) * 4 * (T0 + nT));
  c->tri_obj = (int32_t*)realloc(c->tri_obj, sizeof(int32_t) * (T0 + nT));
  c->obj_tris = (uint32_t*)realloc(c->obj_tris, sizeof(uint32_t) * (T0 + nT));
  c->static_pos = (uint32_t*)realloc(c->static_pos, sizeof(uint32_t) * (T0 + nT));
  for (uint32_t t = 0; t < nT; ++t) {
    for (int k = 0; k < 3; ++k) c->tri[4 * (size_t)(T0 + t) + k] = tri[4 * t + k] + (int32_t)V0;
    c->tri[4 * (size_t)(T0 + t) + 3] = tri[4 * t + 3];
    c->tri_obj[T0 + t] = (int32_t)ob; c->obj_tris[T0 + t] = T0 + t; c->static_pos[T0 + t] = 0;
  }
  c->xf = (float*)realloc(c->xf, sizeof(float) * 12 * (ob + 1)); memcpy(c->xf + 12 * (size_t)ob, xform, sizeof(float) * 12);
  c->xf0 = (float*)realloc(c->xf0, sizeof(float) * 12 * (ob + 1)); memcpy(c->xf0 + 12 * (size_t)ob, xform, sizeof(float) * 12);
  c->obj = (orc_object*)realloc(c->obj, sizeof(orc_object) * (ob + 1)); memset(&c->obj[ob], 0, sizeof(orc_object));
  c->obj[ob].first = T0; c->obj[ob].ntri = nT;
  if (c->hidden) { c->hidden = (uint8_t*)realloc(c->hidden, ob + 1); c->hidden[ob] = 0; }
  c->nV = V0 + nV; c->nT = T0 + nT; c->nO = ob + 1;
  apply_objects(c);
  if (object_out) *object_out = ob;
  return orc_reset(c);
}
ORC_API int orc_get_tlas(orc_ctx* c, uint32_t* root, uint32_t* n_instances, uint32_t* n_blas_nodes)
{
  if (!c || !c->built) return CRH_E_NOTBUILT;
  if (root) *root = c->nInst ? (c->root2 != QBVH_EMPTY ? c->root2 : c->root) : 0u; if (n_instances) *n_instances = c->nInst; if (n_blas_nodes) *n_blas_nodes = c->nBlasNodes;
  return 0;
}
ORC_API int orc_set_materials(orc_ctx* c, const crh_bsdf* m, uint32_t n)
{ if (!c || (n && !m) || !all_finite((const float*)m, 32 * (size_t)n, 3.0e38f)) return CRH_E_INVALID; free(c->mats); c->mats = (crh_bsdf*)dup_mem(m, sizeof(crh_bsdf) * n); c->nM = n; return 0; }
ORC_API int orc_set_lights(orc_ctx* c, const crh_light* l, uint32_t n)
{ if (!c || (n && !l) || !all_finite((const float*)l, 8 * (size_t)n, 1.0e30f)) return CRH_E_INVALID; free(c->lights); c->lights = (crh_light*)dup_mem(l, sizeof(crh_light) * n); c->nL = n; return 0; }
ORC_API int orc_set_envmap(orc_ctx* c, const float* rgb, uint32_t w, uint32_t h)
{
  if (!c || (rgb && w && h && !all_finite(rgb, 3 * (size_t)w * h, 3.0e38f))) return CRH_E_INVALID;
  free(c->env); c->env = NULL; c->envW = c->envH = 0;
  if (rgb && w && h) { c->env = (float*)dup_mem(rgb, sizeof(float) * 3 * (size_t)w * h); c->envW = w; c->envH = h; }
  return 0;
}
ORC_API int orc_set_texture(orc_ctx* c, uint32_t slot, const float* rgb, uint32_t w, uint32_t h, uint32_t channels)
{
  if (!c || slot >= 64u || (rgb && channels != 3u && channels != 4u)) return CRH_E_INVALID;
  free(c->tex[slot].rgb); c->tex[slot].rgb = NULL; c->tex[slot].w = c->tex[slot].h = 0;
  if (rgb && w && h) { c->tex[slot].rgb = (float*)dup_mem(rgb, sizeof(float) * channels * (size_t)w * h); c->tex[slot].w = w; c->tex[slot].h = h; c->tex[slot].ch = channels; }
  if (slot + 1 > c->nTex) c->nTex = slot + 1;
  return 0;
}
ORC_API int orc_set_camera(orc_ctx* c, const crh_camera* cam)
{
  if (!c || !cam) return CRH_E_INVALID;
  const float f[] = {cam->eye[0], cam->eye[1], cam->eye[2], cam->dir[0], cam->dir[1], cam->dir[2], cam->up[0], cam->up[1], cam->up[2],
                     cam->fovy_deg, cam->aspect, cam->ortho_scale, cam->aperture_radius, cam->focal_dist};
  if (!all_finite(f, sizeof f / sizeof f[0], 1.0e30f)) return CRH_E_INVALID;
  c->cam = *cam; return 0;
}
ORC_API int orc_reset(orc_ctx* c)
{
  if (!c) return CRH_E_INVALID;
  c->frames_done = 0;
  free(c->accum); c->accum = (float*)calloc((size_t)c->par.width * c->par.height * 4, sizeof(float));
  free(c->m2); c->m2 = (float*)calloc((size_t)c->par.width * c->par.height, sizeof(float)); c->adaptive_picks = 0;
  free(c->last_picked); c->last_picked = NULL; c->last_picked_n = 0;
  memset(&c->st, 0, sizeof c->st);
  return 0;
}
ORC_API int orc_set_params(orc_ctx* c, const crh_params* p)
{
  if (!c || !p || !p->width || !p->height || p->max_depth < 1 || p->max_depth > 32) return CRH_E_INVALID;
  { const float f[] = {p->radiance_clamp, p->exposure, p->white_point, p->background[0], p->background[1], p->background[2], p->scene_epsilon};
    if (!all_finite(f, sizeof f / sizeof f[0], 3.0e38f)) return CRH_E_INVALID; }
  c->par = *p; return orc_reset(c);
}
ORC_API int orc_set_spec(orc_ctx* c, const crh_spec* sp)
{
  if (!c) return CRH_E_INVALID;
  crh_spec n; const char* why = "";
  if (crh_spec_normalise(sp, &n, &why)) { snprintf(c->err, sizeof c->err, "%s", why); return CRH_E_INVALID; }
  c->spec = n;
  return orc_reset(c);
}
ORC_API int orc_get_spec(orc_ctx* c, crh_spec* out) { if (!c || !out) return CRH_E_INVALID; return crh_spec_export(&c->spec, out, NULL) ? CRH_E_INVALID : 0; }
ORC_API int orc_spec_order_exact(void) { return CRH_SPEC_ORDER_EXACT; }
ORC_API int orc_spec_anyhit_slot_order(void) { return CRH_SPEC_ANYHIT_SLOT_ORDER; }
ORC_API int orc_build(orc_ctx* c)
{
  if (!c) return CRH_E_INVALID;
  if (c->nT && !c->nM) { snprintf(c->err, sizeof c->err, "no materials"); return CRH_E_INVALID; }
  do_build(c); c->built = 1; return orc_reset(c);
}
ORC_API int orc_render_tiles(orc_ctx* c, const uint32_t* tiles, uint32_t nt, uint32_t first, uint32_t ns)
{ if (!c) return CRH_E_INVALID; if (!c->built) return CRH_E_NOTBUILT;
  for (uint32_t i = 0; i < nt; ++i) for (uint32_t j = 0; j < i; ++j) if (nt <= 4096 && tiles[i] == tiles[j]) { snprintf(c->err, sizeof c->err, "duplicate tile id"); return CRH_E_INVALID; } prepare(c); return render_tiles(c, tiles, nt, first, ns, NULL); }
/* --- adaptive screen sampling (reference controls at SettingsWidget.cxx:427-477) ---------------------
 * tile error = mean over the tile's pixels of sqrt(max(E[l^2] - E[l]^2, 0) / n) (1e3 for pixels with n < 2), summed in
 * the fixed order "lane j of 256 takes pixels j, j+256, .. of the row-major tile, then a stride 128..1 tree". */
static void tile_stats(const orc_ctx* c, float* err, uint32_t* cnt)
{
  uint32_t ts = c->par.tile_size, tx = (c->par.width + ts - 1) / ts, ty = (c->par.height + ts - 1) / ts;
  for (uint32_t t = 0; t < tx * ty; ++t) {
    float e[256], np_[256], cm[256];
    uint32_t x0 = (t % tx) * ts, y0 = (t / tx) * ts;
    for (uint32_t j = 0; j < 256; ++j) {
      e[j] = 0.f; np_[j] = 0.f; cm[j] = 3.0e38f;
      for (uint32_t i = j; i < ts * ts; i += 256) {
        uint32_t px = x0 + i % ts, py = y0 + i / ts;
        if (px < c->par.width && py < c->par.height) {
          const float* a = &c->accum[4 * ((size_t)py * c->par.width + px)];
          float pe = 1.0e3f;
          if (a[3] >= 2.0f) {
            float l = CRH_FMA(LUMA_B, a[2], CRH_FMA(LUMA_G, a[1], LUMA_R * a[0]));
            pe = crh_sqrt(crh_max(c->m2[(size_t)py * c->par.width + px] - l * l, 0.f) / a[3]);
          }
          e[j] += pe; np_[j] += 1.0f; cm[j] = crh_min(cm[j], a[3]);
        }
      }
    }
    for (uint32_t st = 128; st > 0; st >>= 1)
      for (uint32_t j = 0; j < st; ++j) { e[j] += e[j + st]; np_[j] += np_[j + st]; cm[j] = crh_min(cm[j], cm[j + st]); }
    err[t] = np_[0] > 0.f ? e[0] / np_[0] : 0.f;
    cnt[t] = np_[0] > 0.f ? (uint32_t)cm[0] : 0u;
  }
}

static uint32_t radical_inverse2(uint32_t v)
{
  uint32_t r = 0;
  for (int b = 0; b < 32; ++b) { r = (r << 1) | (v & 1u); v >>= 1; }
  return r;
}

static int adaptive_iteration(orc_ctx* c)
{
  uint32_t ts = c->par.tile_size, nt = ((c->par.width + ts - 1) / ts) * ((c->par.height + ts - 1) / ts);
  float* err = (float*)malloc(sizeof(float) * nt); uint32_t* cnt = (uint32_t*)malloc(sizeof(uint32_t) * nt);
  float* cdf = (float*)malloc(sizeof(float) * nt); uint8_t* picked = (uint8_t*)calloc(nt, 1);
  tile_stats(c, err, cnt);
  float acc = 0.f;
  for (uint32_t i = 0; i < nt; ++i) { acc += err[i] > 0.f ? err[i] : 0.f; cdf[i] = acc; }
  for (uint32_t k = 0; k < c->adaptive_tiles; ++k) {
    float u = (float)(radical_inverse2(c->adaptive_picks++) >> 8) * 5.9604644775390625e-8f;
    uint32_t t = 0;
    if (!(acc > 0.f)) t = (uint32_t)(u * (float)nt);
    else { float x = u * acc; while (t < nt && !(cdf[t] > x)) ++t; }     /* first tile whose cdf exceeds x */
    if (t >= nt) t = nt - 1;
    picked[t] = 1;
  }
  uint32_t n = 0; for (uint32_t i = 0; i < nt; ++i) n += picked[i];
  uint32_t* tiles = (uint32_t*)malloc(sizeof(uint32_t) * (n ? n : 1)); uint32_t* seeds = (uint32_t*)malloc(sizeof(uint32_t) * (n ? n : 1));
  n = 0;
  for (uint32_t i = 0; i < nt; ++i) if (picked[i]) { tiles[n] = i; seeds[n] = frame_seed(c->par.seed, cnt[i]); ++n; }
  int rc = render_tiles(c, tiles, n, 0, 1, seeds);
  free(c->last_picked); c->last_picked = picked; c->last_picked_n = nt;
  free(err); free(cnt); free(cdf); free(tiles); free(seeds);
  return rc;
}

ORC_API int orc_set_adaptive(orc_ctx* c, int on, uint32_t tiles_per_iteration)
{
  if (!c || (on && !tiles_per_iteration)) return CRH_E_INVALID;
  c->adaptive = on != 0; if (on) c->adaptive_tiles = tiles_per_iteration;
  return orc_reset(c);
}
ORC_API int orc_set_show_tiles(orc_ctx* c, int on) { if (!c) return CRH_E_INVALID; c->show_tiles = on != 0; return 0; }
ORC_API int orc_get_tile_stats(orc_ctx* c, float* err, uint32_t* counts, uint32_t* n_tiles)
{
  if (!c || !c->accum) return CRH_E_INVALID;
  uint32_t ts = c->par.tile_size, nt = ((c->par.width + ts - 1) / ts) * ((c->par.height + ts - 1) / ts);
  if (n_tiles) *n_tiles = nt;
  if (err && counts) tile_stats(c, err, counts);
  return 0;
}

ORC_API int orc_render(orc_ctx* c, uint32_t n)
{
  if (!c) return CRH_E_INVALID; if (!c->built) return CRH_E_NOTBUILT;
  prepare(c);
  if (c->adaptive) { for (uint32_t i = 0; i < n; ++i) { int rc = adaptive_iteration(c); if (rc) return rc; } c->frames_done += n; return 0; }
  /* whole frames continue from the iteration counter, whatever orc_render_tiles did to individual tiles in between */
  uint32_t first = c->frames_done;
  c->frames_done += n;
  return render_tiles(c, NULL, 0, first, n, NULL);
}
ORC_API int orc_read_accum(orc_ctx* c, float* out) { if (!c || !c->accum) return CRH_E_INVALID; memcpy(out, c->accum, sizeof(float) * 4 * (size_t)c->par.width * c->par.height); return 0; }
ORC_API int orc_read_hdr(orc_ctx* c, float* out)
{
  if (!c || !c->accum) return CRH_E_INVALID;
  size_t n = (size_t)c->par.width * c->par.height;
  for (size_t i = 0; i < n; ++i) { out[3 * i] = c->accum[4 * i]; out[3 * i + 1] = c->accum[4 * i + 1]; out[3 * i + 2] = c->accum[4 * i + 2]; }
  return 0;
}
ORC_API int orc_read_ldr(orc_ctx* c, uint8_t* out)
{
  if (!c || !c->accum) return CRH_E_INVALID;
  size_t n = (size_t)c->par.width * c->par.height;
  for (size_t i = 0; i < n; ++i) for (int k = 0; k < 3; ++k) out[3 * i + k] = to_ldr(&c->par, c->spec.display_gamma22, c->accum[4 * i + k]);
  uint32_t ts = c->par.tile_size, tx = (c->par.width + ts - 1) / ts, ty = (c->par.height + ts - 1) / ts;
  if (c->show_tiles && c->adaptive && c->last_picked && c->last_picked_n == tx * ty)
    for (uint32_t y = 0; y < c->par.height; ++y) for (uint32_t x = 0; x < c->par.width; ++x) {
      uint32_t lx = x % ts, ly = y % ts;
      if (c->last_picked[(y / ts) * tx + x / ts] && (lx == 0 || ly == 0 || lx == ts - 1 || ly == ts - 1)) {
        uint8_t* o = &out[3 * ((size_t)y * c->par.width + x)]; o[0] = 255; o[1] = 0; o[2] = 0;
      }
    }
  return 0;
}
ORC_API int orc_get_stats(orc_ctx* c, crh_stats* s) { if (!c || !s) return CRH_E_INVALID; *s = c->st; return 0; }
ORC_API int orc_get_bvh(orc_ctx* c, float* nodes, uint32_t* nn, float* tris, uint32_t* nt)
{
  if (!c || !c->built) return CRH_E_NOTBUILT;
  if (nn) *nn = c->nNodes; if (nt) *nt = c->nQT;
  if (nodes) memcpy(nodes, c->nodes, sizeof(qnode) * c->nNodes);
  if (tris) memcpy(tris, c->qtris, sizeof(qtri) * c->nQT);
  return 0;
}
ORC_API int orc_trace_nearest(orc_ctx* c, const float* rays, uint32_t n, float* out)
{
  if (!c || !c->built) return CRH_E_NOTBUILT;
  uint64_t nn = 0, tt = 0;
#pragma omp parallel for schedule(dynamic, 1024) reduction(+ : nn, tt)
  for (uint32_t i = 0; i < n; ++i) {
    const float* r = &rays[8 * i]; hit_t h; trav_counters cn = {0, 0, 0, 0};
    traverse_ex(c, crh_mk3(r[0], r[1], r[2]), crh_mk3(r[4], r[5], r[6]), r[3], 0, &h, &cn, 0);
    out[4 * i] = h.t; out[4 * i + 1] = h.u; out[4 * i + 2] = h.v; out[4 * i + 3] = crh_u2f((uint32_t)h.prim);
    nn += cn.nodes; tt += cn.tris;
  }
  c->st.rays_nearest += n; c->st.nodes_nearest += nn; c->st.tris_nearest += tt;
  return 0;
}
ORC_API int orc_trace_any(orc_ctx* c, const float* rays, uint32_t n, uint32_t* vis)
{
  if (!c || !c->built) return CRH_E_NOTBUILT;
  uint64_t nn = 0, tt = 0;
#pragma omp parallel for schedule(dynamic, 1024) reduction(+ : nn, tt)
  for (uint32_t i = 0; i < n; ++i) {
    const float* r = &rays[8 * i]; hit_t h; trav_counters cn = {0, 0, 0, 0};
    vis[i] = traverse_ex(c, crh_mk3(r[0], r[1], r[2]), crh_mk3(r[4], r[5], r[6]), r[3], 1, &h, &cn, 0) ? 0u : 1u;
    nn += cn.nodes_any; tt += cn.tris_any;
  }
  c->st.rays_any += n; c->st.nodes_any += nn; c->st.tris_any += tt;
  return 0;
}
ORC_API int orc_set_threads(int n)
{
#ifdef _OPENMP
  if (n > 0) omp_set_num_threads(n);
  return omp_get_max_threads();
#else
  (void)n; return 1;
#endif
}

/* --- unit entry points for the known-answer tests -------------------------------------------- */
static void bsdf_from_abi(const crh_bsdf* m, const float wo[3], bsdf_t* b)
{
  orc_ctx tmp; memset(&tmp, 0, sizeof tmp); tmp.mats = (crh_bsdf*)m; tmp.nM = 1;
  load_bsdf(&tmp, 0, b);
  b->Fc = fresnel_media(wo[2], b->fc);
}
ORC_API void orc_fresnel(float cosI, const float f[4], float out[3]) { v3 r = fresnel_media(cosI, f); out[0] = r.x; out[1] = r.y; out[2] = r.z; }
ORC_API void orc_bsdf_eval(const crh_bsdf* m, const float wo[3], const float wi[3], int two_sided, float out[3])
{ bsdf_t b; bsdf_from_abi(m, wo, &b); v3 r = eval_layered(&b, crh_mk3(wi[0], wi[1], wi[2]), crh_mk3(wo[0], wo[1], wo[2]), two_sided); out[0] = r.x; out[1] = r.y; out[2] = r.z; }
ORC_API float orc_bsdf_pdf(const crh_bsdf* m, const float wo[3], const float wi[3], const float W[3], int two_sided)
{ bsdf_t b; bsdf_from_abi(m, wo, &b); return pdf_layered(&b, crh_mk3(wo[0], wo[1], wo[2]), crh_mk3(wi[0], wi[1], wi[2]), crh_mk3(W[0], W[1], W[2]), two_sided, -1); }
/* returns alive; weight_io in/out, rng in/out, flags_out bit0 = delta, bit1 = inside after */
ORC_API int orc_bsdf_sample(const crh_bsdf* m, const float wo[3], float weight_io[3], uint32_t* rng, int two_sided, int inside_in, float wi_out[3], int* flags_out)
{
  bsdf_t b; bsdf_from_abi(m, wo, &b);
  v3 W = crh_mk3(weight_io[0], weight_io[1], weight_io[2]), wi = crh_mk3(0.f, 0.f, 0.f); int inside = inside_in, delta = 0;
  const crh_spec sp = CRH_SPEC_DEFAULTS; int lobe;
  int alive = sample_layered(&b, crh_mk3(wo[0], wo[1], wo[2]), &wi, &W, &inside, &delta, rng, two_sided, &sp, &lobe);
  weight_io[0] = W.x; weight_io[1] = W.y; weight_io[2] = W.z; wi_out[0] = wi.x; wi_out[1] = wi.y; wi_out[2] = wi.z;
  *flags_out = (delta ? 1 : 0) | (inside ? 2 : 0);
  return alive;
}
/* elementary functions, vectorised over n, for checking crh_math.h against libm in the tests */
ORC_API void orc_math(int fn, const float* a, const float* b, float* out, float* out2, uint32_t n)
{
  for (uint32_t i = 0; i < n; ++i) switch (fn) {
    case 0: crh_sincos2pi(a[i], &out[i], &out2[i]); break;
    case 1: out[i] = crh_exp(a[i]); break;
    case 2: out[i] = crh_log(a[i]); break;
    case 3: out[i] = crh_pow(a[i], b[i]); break;
    case 4: out[i] = crh_acos(a[i]); break;
    case 5: out[i] = crh_atan2(a[i], b[i]); break;
    case 6: crh_sincos(a[i], &out[i], &out2[i]); break;
    default: out[i] = 0.f;
  }
}
ORC_API void orc_rng_stream(uint32_t pixel, uint32_t fseed, float* out, uint32_t n)
{ uint32_t s = crh_rng_seed(pixel, fseed); for (uint32_t i = 0; i < n; ++i) out[i] = crh_rng_next(&s); }
ORC_API uint32_t orc_frame_seed(uint32_t seed, uint32_t n) { return frame_seed(seed, n); }
/* the uniform drawn from xorshift state `s_after` under either setting of crh_spec.uniform_32bit (the conversion crh_rng_next_mode applies) */
/* test hooks for the split-scene pre-test (include/crh_math.h): the padded sphere around a box, and "does [0, tmax] of the ray come within it" */
ORC_API void orc_box_sphere(const float* lo, const float* hi, float* s4) { crh_box_sphere(lo, hi, s4); }
ORC_API int orc_ray_near_sphere(const float* o, const float* d, float tmax, const float* s4)
{ return crh_ray_near_sphere(crh_mk3(o[0], o[1], o[2]), crh_mk3(d[0], d[1], d[2]), tmax, s4[0], s4[1], s4[2], s4[3]); }
ORC_API float orc_rng_float(uint32_t s_after, int full32) { return full32 ? (float)s_after * 2.3283064365386963e-10f : (float)(s_after >> 8) * 5.9604644775390625e-8f; }

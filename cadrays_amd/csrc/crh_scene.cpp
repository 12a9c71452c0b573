// crh_scene.cpp -- scene inputs, BVH build, HBM residency of the scene, and the DScene the kernels see
// (one of the translation units behind include/cadrays_hip.h; the context, the shared helpers and the map of the files: crh_context.h)
#include "crh_context.h"

using namespace crh;
using namespace crh::api;

namespace crh {
namespace api {

void fill_scene(const crh_ctx* c, DScene& S)
{
  std::memset(&S, 0, sizeof S);
  S.nodes = c->d_nodes; S.pnodes = c->d_pnodes; S.tris = c->d_tris; S.verts = c->d_verts; S.shade = c->d_shade; S.mats = c->d_mats; S.lights = c->d_lights; S.env = (c->envW && c->envH) ? c->d_env : nullptr;
  S.inst = c->d_inst; S.inst_leaf = c->d_inst ? c->d_inst + 8 * (size_t)c->nO : nullptr; S.root = c->root; S.two_level = c->inst.empty() ? 0 : 1;
  S.root2 = c->inst.empty() ? kQEmpty : c->root2;
  // render path of a split scene: with at most kMaxIBox moved objects the producers can tell precisely which rays come near one -- two traversal passes
  // (the plain single-level kernels over everything, the two-level ones over the few flagged rays); with more, most rays would be flagged: one walk
  // "static tree, then top level" in the two-level kernels, like the API-level tracers.  Same hits, same counters either way (CRH_SPLIT_PASSES=0/1 forces one).
  S.split = (S.root2 != kQEmpty && (c->split_passes < 0 ? c->inst.size() <= kMaxIBox : c->split_passes != 0)) ? 1 : 0;
  for (int a = 0; a < 3; ++a) { S.tlas_lo[a] = c->tlas_lo[a]; S.tlas_hi[a] = c->tlas_hi[a]; }
  S.ibox = c->d_ibox; S.n_ibox = (c->d_ibox && c->inst.size() <= kMaxIBox) ? (uint32_t)c->inst.size() : 0u;
  S.usph = make_float4(c->usph[0], c->usph[1], c->usph[2], c->usph[3]);
  {
    const float* lo = c->bvh.bbmin; const float* hi = c->bvh.bbmax;      // bounds of the tree the walk starts in (the world box of a two-level scene)
    S.guard_box = make_float4((lo[0] + hi[0]) * 0.5f, (lo[1] + hi[1]) * 0.5f, (lo[2] + hi[2]) * 0.5f, (((hi[0] - lo[0]) + (hi[1] - lo[1])) + (hi[2] - lo[2])) * 0.5f);
  }
  S.uvs = c->d_uvs; S.texels = c->d_texels; S.tex_desc = c->d_tex_desc; S.n_tex = c->d_tex_desc ? (uint32_t)c->textures.size() : 0u;
  S.n_mats = (uint32_t)c->mats.size(); S.n_lights = (uint32_t)c->lights.size(); S.env_w = c->envW; S.env_h = c->envH;
  for (int k = 0; k < 3; ++k) S.bg[k] = c->par.background[k];
  S.env_as_bg = c->par.env_as_background;
  S.eye = crh_mk3(c->cam.eye[0], c->cam.eye[1], c->cam.eye[2]);
  S.fwd = crh_norm3(crh_mk3(c->cam.dir[0], c->cam.dir[1], c->cam.dir[2]));
  S.right = crh_norm3(crh_cross3(S.fwd, crh_mk3(c->cam.up[0], c->cam.up[1], c->cam.up[2])));
  S.up = crh_cross3(S.right, S.fwd);
  float s, cs; crh_sincos((c->cam.fovy_deg * 0.5f) * (CRH_PI / 180.0f), &s, &cs);
  S.tan_half = s / cs;
  S.aspect = c->cam.aspect > 0.f ? c->cam.aspect : (float)c->par.width / (float)c->par.height;
  S.ortho_scale = c->cam.ortho_scale; S.aperture = c->cam.aperture_radius; S.focal = c->cam.focal_dist; S.is_ortho = c->cam.is_ortho;
  S.width = c->par.width; S.height = c->par.height; S.max_depth = c->par.max_depth; S.tile_size = c->par.tile_size;
  S.clampv = c->par.radiance_clamp;
  const crh_v3 dg = crh_mk3(c->bvh.bbmax[0] - c->bvh.bbmin[0], c->bvh.bbmax[1] - c->bvh.bbmin[1], c->bvh.bbmax[2] - c->bvh.bbmin[2]);
  S.eps = c->par.scene_epsilon > 0.f ? c->par.scene_epsilon
        : (c->spec.eps_rule ? crh_max(1.0e-6f, 1.0e-4f * (crh_len3(dg) * 0.5f)) : crh_max(1.0e-6f, 1.0e-5f * crh_len3(dg)));      // crh_spec.h #6
  S.two_sided = c->par.two_sided; S.coherent = c->par.coherent_rng; S.rr = c->par.russian_roulette;
  S.spec_u32 = c->spec.uniform_32bit; S.spec_gamma2 = c->spec.texel_gamma2; S.spec_mis1 = c->spec.mis_single_lobe; S.spec_eta_nd = c->spec.eta_no_dielectric;
  S.spec_rr_start = (uint32_t)c->spec.rr_start_bounce; S.spec_rr_cap = c->spec.rr_survival_cap; S.spec_min_contrib = c->spec.min_contribution;
  S.spec_min_thr = c->spec.min_throughput; S.spec_raygen = c->spec.raygen_bilinear; S.spec_env_orient = c->spec.env_orientation;
  for (int k = 0; k < 4; ++k)      // crh_spec.h #13: frustum-corner directions LB, RB, LT, RT
    S.corner[k] = crh_frustum_corner(S.fwd, S.right, S.up, S.tan_half, S.aspect, (k & 1) ? 1.0f : -1.0f, (k & 2) ? 1.0f : -1.0f, c->spec.raygen_bilinear == 2);
}

int upload_textures(crh_ctx* c)
{
  if (!c->textures_dirty) return CRH_OK;
  std::vector<float> all; std::vector<uint32_t> desc(4 * std::max<size_t>(c->textures.size(), 1), 0u);
  for (size_t i = 0; i < c->textures.size(); ++i) {
    desc[4 * i] = (uint32_t)(all.size() / 4); desc[4 * i + 1] = c->textures[i].w; desc[4 * i + 2] = c->textures[i].h;
    all.insert(all.end(), c->textures[i].rgba.begin(), c->textures[i].rgba.end());
  }
  if (all.empty()) all.assign(4, 0.f);
  int rc;
  if ((rc = dev_upload(c, c->d_texels, all.data(), all.size() * sizeof(float)))) return rc;
  if ((rc = dev_upload(c, c->d_tex_desc, desc.data(), desc.size() * sizeof(uint32_t)))) return rc;
  c->textures_dirty = false;
  return CRH_OK;
}

}  // namespace api
}  // namespace crh

namespace {

int upload_lights(crh_ctx* c)
{
  std::vector<float> l(8 * std::max<size_t>(c->lights.size(), 1), 0.f);
  for (size_t i = 0; i < c->lights.size(); ++i) {
    const crh_light& s = c->lights[i]; float* o = &l[8 * i];
    if (s.is_point != 0.f) { o[0] = s.vec[0]; o[1] = s.vec[1]; o[2] = s.vec[2]; o[3] = 1.f; o[7] = s.smoothness; }
    else {
      const crh_v3 d = crh_norm3(crh_mk3(-s.vec[0], -s.vec[1], -s.vec[2]));
      float sn, cn; crh_sincos(s.smoothness, &sn, &cn);
      o[0] = d.x; o[1] = d.y; o[2] = d.z; o[3] = 0.f; o[7] = s.smoothness > 0.f ? cn : 1.0f;
    }
    o[4] = s.emission[0]; o[5] = s.emission[1]; o[6] = s.emission[2];
  }
  return dev_put(c, c->d_lights, c->cap_lights, l.data(), l.size() * sizeof(float), 8 * 32);
}

static bool is_identity(const float* m)
{
  static const float I[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
  for (int k = 0; k < 12; ++k) if (m[k] != I[k]) return false;
  return true;
}

// (Re)build the top-level tree over the world boxes of the objects rendered as instances right now, behind the static tree and the object
// trees [0, n_blas_nodes), and refresh the instance table on the device.  Sets the walk's entry points (fill_scene): no instance -> the static
// tree alone; instances + live static triangles -> the static tree, then the top level (root2); no live static triangle -> the top level.
int build_tlas(crh_ctx* c)
{
  c->bvh.nodes.resize(c->n_blas_nodes);
  c->inst.clear();
  for (uint32_t ob = 0; ob < c->nO; ++ob) {
    const TwoLevelState::Obj& o = c->objs[ob];
    if (!o.is_inst) continue;
    TwoLevelState::Inst in{}; in.obj = ob; in.root = o.root;
    std::memcpy(in.bmin, o.bmin, sizeof in.bmin); std::memcpy(in.bmax, o.bmax, sizeof in.bmax);
    c->inst.push_back(in);
  }
  const uint32_t n = (uint32_t)c->inst.size();
  c->root2 = kQEmpty;
  if (n == 0) {                                 // one world-space tree
    c->root = 0;
    for (int a = 0; a < 3; ++a) { c->bvh.bbmin[a] = c->n_static ? c->sbmin[a] : 0.f; c->bvh.bbmax[a] = c->n_static ? c->sbmax[a] : 0.f; }
    return CRH_OK;
  }
  // instance table: one record per OBJECT (shading looks the hit triangle's object up; only the instances' records are filled) followed
  // by the instances' records in top-level LEAF order (a top-level leaf reference is a position in that order)
  std::vector<float> boxes(6 * (size_t)n, 0.f), table(32 * ((size_t)c->nO + n), 0.f);
  for (uint32_t i = 0; i < n; ++i) {
    TwoLevelState::Inst& in = c->inst[i];
    float* rec = &table[32 * (size_t)in.obj];
    std::memcpy(in.fwd, &c->xf[12 * (size_t)in.obj], sizeof in.fwd);
    if (!crh_xform_inverse(in.fwd, in.inv)) std::memset(in.inv, 0, sizeof in.inv);
    crh_xform_box(in.fwd, in.bmin, in.bmax, &boxes[6 * (size_t)i], &boxes[6 * (size_t)i + 3]);
    std::memcpy(rec, in.inv, 48); std::memcpy(rec + 12, in.fwd, 48);
    std::memcpy(rec + 24, &in.root, 4); std::memcpy(rec + 25, &in.obj, 4);
    const float* iv = in.inv;                    // meta.z = 1: the inverse's 3x3 part is exactly the identity (translation only)
    const uint32_t pure_translation = (iv[0] == 1.f && iv[5] == 1.f && iv[10] == 1.f && iv[1] == 0.f && iv[2] == 0.f && iv[4] == 0.f &&
                                       iv[6] == 0.f && iv[8] == 0.f && iv[9] == 0.f) ? 1u : 0u;
    std::memcpy(rec + 26, &pure_translation, 4);
    // the object's own box {centre, L1 half-extent}: the guard band of the slab test inside the object
    for (int a = 0; a < 3; ++a) rec[28 + a] = (in.bmin[a] + in.bmax[a]) * 0.5f;
    rec[31] = (((in.bmax[0] - in.bmin[0]) + (in.bmax[1] - in.bmin[1])) + (in.bmax[2] - in.bmin[2])) * 0.5f;
  }
  std::vector<uint32_t> order;
  const uint32_t troot = build_tree(boxes.data(), n, true, 0, c->bvh.nodes, order, c->tlas_lo, c->tlas_hi, n >= 4096 ? 0 : 1);   // small trees: one thread beats the hand-off
  for (uint32_t p = 0; p < n; ++p) std::memcpy(&table[32 * ((size_t)c->nO + p)], &table[32 * (size_t)c->inst[order[p]].obj], 128);
  if (c->n_static_live) {
    c->root = 0; c->root2 = troot;
    for (int a = 0; a < 3; ++a) { c->bvh.bbmin[a] = crh_min(c->tlas_lo[a], c->sbmin[a]); c->bvh.bbmax[a] = crh_max(c->tlas_hi[a], c->sbmax[a]); }
  } else {
    c->root = troot;
    for (int a = 0; a < 3; ++a) { c->bvh.bbmin[a] = c->tlas_lo[a]; c->bvh.bbmax[a] = c->tlas_hi[a]; }
  }
  if (!c->d_ibox) CRH_HIP(hipMalloc((void**)&c->d_ibox, sizeof(float4) * kMaxIBox));
  crh_box_sphere(c->tlas_lo, c->tlas_hi, c->usph);
  if (n <= kMaxIBox) {
    float ib[4 * kMaxIBox] = {0};
    for (uint32_t i = 0; i < n; ++i) crh_box_sphere(&boxes[6 * (size_t)i], &boxes[6 * (size_t)i + 3], &ib[4 * i]);
    int rc_b = stage_copy(c, c->d_ibox, ib, sizeof(float) * 4 * n); if (rc_b) return rc_b;
  }
  return dev_put(c, c->d_inst, c->cap_inst, table.data(), table.size() * sizeof(float), 32 * sizeof(float) * ((size_t)c->nO + 64));
}

// the object-space tree of object ob, appended behind the trees built so far; its triangles take the next leaf positions
void build_object_tree(crh_ctx* c, uint32_t ob, int threads)
{
  TwoLevelState::Obj& o = c->objs[ob];
  const uint32_t m = o.ntri; const uint32_t* mem = &c->obj_tris[o.first];
  std::vector<float> boxes(6 * (size_t)m); std::vector<uint32_t> order;
  for (uint32_t i = 0; i < m; ++i)
    for (int a = 0; a < 3; ++a) {
      const uint32_t t = mem[i];
      const float v0 = c->pos[3 * c->tri[4 * t + 0] + a], v1 = c->pos[3 * c->tri[4 * t + 1] + a], v2 = c->pos[3 * c->tri[4 * t + 2] + a];
      boxes[6 * (size_t)i + a] = std::min(v0, std::min(v1, v2)); boxes[6 * (size_t)i + 3 + a] = std::max(v0, std::max(v1, v2));
    }
  o.root = build_tree(boxes.data(), m, false, c->n_pos, c->bvh.nodes, order, o.bmin, o.bmax, threads);
  for (uint32_t i = 0; i < m; ++i) { c->bvh.prim_order.push_back(mem[order[i]]); c->pos_obj.push_back(ob); }
  c->n_pos += m; o.built = true;
}

// Leaf-ordered device records of positions [p0, p1): 16 floats of triangle (48 B used), 16 floats of shading record, 8 floats of uv.
// Also refreshes the host copy h_tris (12 floats per position, crh_get_bvh).
// device form of one triangle record from its host form q = {v0 | id, v1, v2}: {v0 | n.x}, {e0 | n.y}, {e1 | n.z}, {id} with e0 = v1 - v0, e1 = v0 - v2,
// n = e1 x e0 -- the expressions the traversal kernel used to evaluate per test, evaluated here with the same inline arithmetic (same bits).  A
// disabled triangle (all-zero vertices) stays all zero: n . d = 0, the test yields NaN and rejects.
static void device_tri_record(const float* q, float* d)
{
  const crh_v3 v0 = crh_mk3(q[0], q[1], q[2]), v1 = crh_mk3(q[4], q[5], q[6]), v2 = crh_mk3(q[8], q[9], q[10]);
  const crh_v3 e0 = crh_sub3(v1, v0), e1 = crh_sub3(v0, v2), n = crh_cross3(e1, e0);
  d[0] = v0.x; d[1] = v0.y; d[2] = v0.z; d[3] = n.x;
  d[4] = e0.x; d[5] = e0.y; d[6] = e0.z; d[7] = n.y;
  d[8] = e1.x; d[9] = e1.y; d[10] = e1.z; d[11] = n.z;
  std::memcpy(&d[12], &q[3], 4); d[13] = d[14] = d[15] = 0.f;
}

void fill_records(crh_ctx* c, uint32_t p0, uint32_t p1, std::vector<float>& tr, std::vector<float>& sh, std::vector<float>& uvr, std::vector<float>* verts = nullptr)
{
  const size_t n = p1 - p0;
  tr.assign(4 * (size_t)kTriStride * std::max<size_t>(n, 1), 0.f); sh.assign(16 * std::max<size_t>(n, 1), 0.f);
  if (verts) verts->assign(12 * std::max<size_t>(n, 1), 0.f);
  if (!c->uv.empty()) uvr.assign(8 * std::max<size_t>(n, 1), 0.f); else uvr.clear();
  c->h_tris.resize(12 * (size_t)std::max(p1, 1u), 0.f);
  // every position is independent (gathers from the vertex arrays, writes its own records): ranges of positions on the builder's threads -- at
  // 10 M triangles this loop was 2 s of the 9 s a scene hand-over takes
  auto fill = [&](uint32_t q0, uint32_t q1) {
  for (uint32_t p = q0; p < q1; ++p) {
    const uint32_t t = c->bvh.prim_order[p]; const size_t i = p - p0;
    float* q = &c->h_tris[12 * (size_t)p]; float* s_ = &sh[16 * i];
    // a position of the static tree of a two-level scene holds the BAKED vertex (its object's build-time transform applied); an object tree the object-space one
    const bool baked = c->two_level && p < c->n_static;
    const float* VP = baked ? c->pos_w.data() : c->pos.data(); const float* VN = baked ? c->nrm_w.data() : c->nrm.data();
    for (int k = 0; k < 3; ++k) {
      const int32_t vi = c->tri[4 * t + k];
      for (int a = 0; a < 3; ++a) { q[4 * k + a] = VP[3 * vi + a]; s_[4 * k + a] = VN[3 * vi + a]; }
      q[4 * k + 3] = 0.f;
      if (!uvr.empty()) { uvr[8 * i + 2 * k] = c->uv[2 * vi]; uvr[8 * i + 2 * k + 1] = c->uv[2 * vi + 1]; }
    }
    {
      // the kernel's former expression on the three vertices, evaluated once here with the same inline arithmetic (same bits)
      const crh_v3 a0 = crh_mk3(q[0], q[1], q[2]), a1 = crh_mk3(q[4], q[5], q[6]), a2 = crh_mk3(q[8], q[9], q[10]);
      const crh_v3 ng = crh_norm3(crh_cross3(crh_sub3(a0, a2), crh_sub3(a1, a0)));
      s_[12] = ng.x; s_[13] = ng.y; s_[14] = ng.z;
    }
    std::memcpy(&q[3], &t, 4);
    const int32_t mat = c->tri[4 * t + 3];
    std::memcpy(&s_[3], &mat, 4);
    const int32_t ob = (c->two_level && p >= c->n_static) ? (int32_t)c->pos_obj[p - c->n_static] : -1;      // n1.w: the object whose transform shading applies (-1: world space)
    std::memcpy(&s_[7], &ob, 4);
    device_tri_record(q, &tr[4 * (size_t)kTriStride * i]);
    if (verts) std::memcpy(&(*verts)[12 * i], q, 48);
  }
  };
  const int threads = build_threads_env();
  const uint32_t nth = n >= 65536u ? (uint32_t)std::min<size_t>((size_t)build_threads(threads), n / 32768u) : 1u;
  if (nth <= 1u) fill(p0, p1);
  else {
    std::vector<std::thread> pool;
    for (uint32_t k = 0; k < nth; ++k) pool.emplace_back(fill, p0 + (uint32_t)((uint64_t)n * k / nth), p0 + (uint32_t)((uint64_t)n * (k + 1) / nth));
    for (auto& th : pool) th.join();
  }
}

}  // namespace

extern "C" {

int crh_set_geometry(crh_ctx* c, const float* pos, const float* nrm, const float* uv, uint32_t nV, const int32_t* tri, uint32_t nT,
                     const int32_t* tri_obj, const float* xf, uint32_t nO)
{
  if (!c) return CRH_E_INVALID;
  if ((nV && (!pos || !nrm)) || (nT && !tri)) return fail(c, CRH_E_INVALID, "null geometry array");
  if (nT >= (1u << 28)) return fail(c, CRH_E_INVALID, "too many triangles (limit 2^28)");
  if (!all_finite(pos, 3 * (size_t)nV, 1.0e30f) || !all_finite(nrm, 3 * (size_t)nV) || (uv && !all_finite(uv, 2 * (size_t)nV)) ||
      (xf && !all_finite(xf, 12 * (size_t)nO, 1.0e30f)))
    return fail(c, CRH_E_INVALID, "geometry holds a NaN / Inf (or a coordinate beyond 1e30)");
  for (uint32_t t = 0; t < nT; ++t)
    for (int k = 0; k < 3; ++k)
      if (tri[4 * t + k] < 0 || (uint32_t)tri[4 * t + k] >= nV) { char b[96]; snprintf(b, sizeof b, "triangle %u index out of range", t); return fail(c, CRH_E_INVALID, b); }
  if (tri_obj && xf && nO) {
    for (uint32_t t = 0; t < nT; ++t)
      if (tri_obj[t] < 0 || (uint32_t)tri_obj[t] >= nO) return fail(c, CRH_E_INVALID, "triangle object id out of range");
    // every vertex belongs to ONE object: crh_build bakes a vertex once, under the build-time transform of its object -- a vertex shared by triangles
    // of two objects would silently take the placement of whichever comes first.  AisMesh.cxx:372-413 emits one vertex array per object.
    std::vector<int32_t> owner(nV, -1);
    for (uint32_t t = 0; t < nT; ++t)
      for (int k = 0; k < 3; ++k) {
        int32_t& o = owner[tri[4 * t + k]];
        if (o < 0) o = tri_obj[t];
        else if (o != tri_obj[t]) { char b[160]; snprintf(b, sizeof b, "vertex %d is shared by objects %d and %d: each vertex belongs to one object (duplicate it)", tri[4 * t + k], o, tri_obj[t]); return fail(c, CRH_E_INVALID, b); }
      }
  }
  // every check passed: only now is the context's state replaced
  c->pos.assign(pos, pos + 3 * (size_t)nV); c->nrm.assign(nrm, nrm + 3 * (size_t)nV);
  if (uv) c->uv.assign(uv, uv + 2 * (size_t)nV); else c->uv.clear();
  c->tri.assign(tri, tri + 4 * (size_t)nT);
  c->two_level = false; c->nO = 0; c->xf.clear(); c->tri_obj.clear(); c->hidden.clear();      // a new scene: everything displayed
  if (tri_obj && xf && nO) {
    // two-level mode: vertices stay in object space; every object gets its own tree (crh_build), the top-level tree
    // over the instances carries the transforms (crh_set_transforms rebuilds only that)
    c->two_level = true; c->nO = nO;
    c->xf.assign(xf, xf + 12 * (size_t)nO); c->tri_obj.assign(tri_obj, tri_obj + nT);
  }
  c->built = false; c->pending_n = 0;
  return CRH_OK;
}

// Bring every object to the state its transform and its visibility ask for, touching nothing else (the body crh_set_transforms had up to round 5, now
// shared with crh_set_visibility and crh_add_object).  A displayed object at its build-time placement lives in the static tree; a displayed object off it
// -- or added after the build (static0 = false) -- is an instance with an object tree of its own, built the first time and appended behind the trees built
// so far; an erased object is neither: its records in the static tree are disabled exactly like a moved object's (a scatter of all-zero records) and it
// stays out of the top level.  Then the top-level tree over the instances of this moment is rebuilt on the host and only the new nodes, the patched
// records and the instance table travel, stream-ordered.  Nothing big is ever rebuilt here.
// `xf` (may be null: keep the transforms) and `hidden` (may be null: keep the flags) are the caller's new values; they are taken over only after the
// capacity check, so that a refusal leaves the context as it was.
static int apply_objects(crh_ctx* c, const float* xf, const uint8_t* visible)
{
  const uint32_t nO = c->nO;
  const int threads = build_threads_env();
  const uint32_t old_nodes = c->n_blas_nodes, old_pos = c->n_pos;
  const bool had_instances = !c->inst.empty();
  auto moved_now = [&](uint32_t ob) { const float* m = xf ? &xf[12 * (size_t)ob] : &c->xf[12 * (size_t)ob]; return !c->objs[ob].static0 || std::memcmp(m, &c->xf0[12 * (size_t)ob], 12 * sizeof(float)) != 0; };
  auto shown_now = [&](uint32_t ob) { return visible ? visible[ob] != 0 : (c->hidden.empty() || !c->hidden[ob]); };
  {
    uint64_t extra = 0;
    for (uint32_t ob = 0; ob < nO; ++ob) {
      const crh_ctx::Obj& o = c->objs[ob];
      if (o.ntri && !o.built && shown_now(ob) && moved_now(ob)) extra += o.ntri;
    }
    if ((uint64_t)c->n_pos + extra > c->cap_pos || (uint64_t)c->n_pos + extra >= (1ull << 28)) return fail(c, CRH_E_NOMEM, "leaf positions exhausted (object trees of moved objects)");
  }
  if (xf) c->xf.assign(xf, xf + 12 * (size_t)nO);
  if (visible) { c->hidden.resize(nO); for (uint32_t ob = 0; ob < nO; ++ob) c->hidden[ob] = visible[ob] ? 0 : 1; }
  c->bvh.nodes.resize(c->n_blas_nodes);
  std::vector<uint32_t> ppos; std::vector<float> prec;
  for (uint32_t ob = 0; ob < nO; ++ob) {
    TwoLevelState::Obj& o = c->objs[ob];
    if (!o.ntri) continue;
    const bool shown = shown_now(ob), moved = moved_now(ob);
    const bool want_static = shown && !moved, want_inst = shown && moved;
    if (want_static != o.in_static) {
      // the object's records in the static tree die / come back
      for (uint32_t i = 0; i < o.ntri; ++i) {
        const uint32_t t = c->obj_tris[o.first + i], p = c->static_pos[t];
        float* q = &c->h_tris[12 * (size_t)p];
        if (!want_static) { std::memset(q, 0, 48); std::memcpy(&q[3], &t, 4); }
        else {
          for (int k = 0; k < 3; ++k) { const int32_t vi = c->tri[4 * t + k]; for (int a = 0; a < 3; ++a) q[4 * k + a] = c->pos_w[3 * vi + a]; q[4 * k + 3] = 0.f; }
          std::memcpy(&q[3], &t, 4);
        }
        float d[16]; device_tri_record(q, d);
        ppos.push_back(p); prec.insert(prec.end(), d, d + 12);
      }
      if (want_static) c->n_static_live += o.ntri; else c->n_static_live -= o.ntri;
      o.in_static = want_static;
    }
    if (want_inst && !o.built) build_object_tree(c, ob, threads);
    o.is_inst = want_inst;
  }
  c->n_blas_nodes = (uint32_t)c->bvh.nodes.size();
  int rc;
  if (c->n_pos > old_pos) {                               // records of the object trees just built
    std::vector<float> tr, sh, uvr, vt;
    fill_records(c, old_pos, c->n_pos, tr, sh, uvr, &vt);
    const size_t n = c->n_pos - old_pos;
    auto put = [&](void* dst, const std::vector<float>& v, size_t rec_floats) -> int {
      const size_t bytes = n * rec_floats * sizeof(float);
      if (bytes <= (4u << 20)) return stage_copy(c, dst, v.data(), bytes);
      CRH_HIP(hipMemcpyAsync(dst, v.data(), bytes, hipMemcpyHostToDevice, cstream(c))); CRH_HIP(hipStreamSynchronize(cstream(c))); return CRH_OK;
    };
    if ((rc = put(c->d_tris + (size_t)kTriStride * old_pos, tr, 4 * kTriStride))) return rc;
    if ((rc = put(c->d_shade + 4 * (size_t)old_pos, sh, 16))) return rc;
    if (c->d_uvs && !uvr.empty() && (rc = put(c->d_uvs + 2 * (size_t)old_pos, uvr, 8))) return rc;
    if ((rc = put(c->d_verts + 3 * (size_t)old_pos, vt, 12))) return rc;
  }
  if (!ppos.empty()) {
    const size_t nb = ppos.size() * 4, rb = prec.size() * 4, need = ((nb + 255) & ~(size_t)255) + rb;
    if (need > c->cap_patch) {
      CRH_HIP(hipStreamSynchronize(cstream(c)));
      if (c->d_patch) { CRH_HIP(hipFree(c->d_patch)); c->d_patch = nullptr; c->cap_patch = 0; }
      CRH_HIP(hipMalloc(&c->d_patch, need + need / 2)); c->cap_patch = need + need / 2;
    }
    char* base = (char*)c->d_patch; char* recs = base + ((nb + 255) & ~(size_t)255);
    auto put = [&](void* dst, const void* src, size_t bytes) -> int {
      if (bytes <= (4u << 20)) return stage_copy(c, dst, src, bytes);
      CRH_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, cstream(c))); CRH_HIP(hipStreamSynchronize(cstream(c))); return CRH_OK;
    };
    if ((rc = put(base, ppos.data(), nb))) return rc;
    if ((rc = put(recs, prec.data(), rb))) return rc;
    Launch L{cstream(c), c->grid, false};
    launch_scatter_tris(L, c->d_tris, (const uint32_t*)base, (const float4*)recs, (uint32_t)ppos.size());
    CRH_HIP(hipGetLastError());
  }
  if ((rc = build_tlas(c))) return rc;
  if (had_instances != !c->inst.empty()) c->feed_tune.restart();      // the frame kernel's other instantiation (one walk / two levels): its feeder count is measured again
  const size_t tail = c->bvh.nodes.size() - old_nodes;
  if (c->bvh.nodes.size() * sizeof(QNode) > c->cap_nodes) {          // more object trees than crh_build left room for: the whole array again, with head-room
    if ((rc = dev_put(c, c->d_nodes, c->cap_nodes, c->bvh.nodes.data(), c->bvh.nodes.size() * sizeof(QNode), c->bvh.nodes.size() * sizeof(QNode) / 2 + (size_t)(4 * c->nO + 64) * sizeof(QNode)))) return rc;
  } else if (tail) {
    const size_t bytes = tail * sizeof(QNode); void* dst = (char*)c->d_nodes + (size_t)old_nodes * sizeof(QNode);
    if (bytes <= (4u << 20)) { if ((rc = stage_copy(c, dst, c->bvh.nodes.data() + old_nodes, bytes))) return rc; }
    else { CRH_HIP(hipMemcpyAsync(dst, c->bvh.nodes.data() + old_nodes, bytes, hipMemcpyHostToDevice, cstream(c))); CRH_HIP(hipStreamSynchronize(cstream(c))); }
  }
  return do_reset(c);
}

int crh_set_transforms(crh_ctx* c, const float* xf, uint32_t nO)
{
  if (!c || !xf) return fail(c, CRH_E_INVALID, "null transforms");
  if (!c->two_level || nO != c->nO) return fail(c, CRH_E_INVALID, "crh_set_transforms needs a two-level scene with the same object count");
  if (!all_finite(xf, 12 * (size_t)nO, 1.0e30f)) return fail(c, CRH_E_INVALID, "transform holds a NaN / Inf");
  CRH_HIP(hipSetDevice(c->device));
  if (!c->built) { c->xf.assign(xf, xf + 12 * (size_t)nO); return do_reset(c); }
  // The manipulator calls this every frame (ImRaytraceControls.cxx:58-89).
  return apply_objects(c, xf, nullptr);
}

int crh_set_visibility(crh_ctx* c, const uint8_t* visible, uint32_t nO)
{
  if (!c || !visible) return fail(c, CRH_E_INVALID, "null visibility flags");
  if (!c->two_level || nO != c->nO) return fail(c, CRH_E_INVALID, "crh_set_visibility needs a scene handed over with objects, and one flag per object");
  CRH_HIP(hipSetDevice(c->device));
  if (!c->built) { c->hidden.resize(nO); for (uint32_t ob = 0; ob < nO; ++ob) c->hidden[ob] = visible[ob] ? 0 : 1; return do_reset(c); }
  return apply_objects(c, nullptr, visible);
}

// Grow a leaf-ordered device array to `cap` positions, keeping the first `keep`: a new allocation, a device-to-device copy, the old one freed.
static int grow_positions(crh_ctx* c, float4*& d, size_t rec_float4, size_t keep, size_t cap)
{
  float4* n = nullptr;
  CRH_HIP(hipMalloc((void**)&n, cap * rec_float4 * sizeof(float4)));
  if (d && keep) CRH_HIP(hipMemcpyAsync(n, d, keep * rec_float4 * sizeof(float4), hipMemcpyDeviceToDevice, cstream(c)));
  CRH_HIP(hipStreamSynchronize(cstream(c)));
  if (d) CRH_HIP(hipFree(d));
  d = n;
  return CRH_OK;
}

int crh_add_object(crh_ctx* c, const float* pos, const float* nrm, const float* uv, uint32_t nV, const int32_t* tri, uint32_t nT,
                   const float* xform, uint32_t* object_out)
{
  if (!c) return CRH_E_INVALID;
  if (!pos || !nrm || !tri || !xform || !nV || !nT) return fail(c, CRH_E_INVALID, "crh_add_object: null or empty geometry");
  if (!c->built) return fail(c, CRH_E_NOTBUILT, "crh_build has not been called");
  if (!c->two_level) return fail(c, CRH_E_INVALID, "crh_add_object needs a scene handed over with objects");
  for (uint32_t t = 0; t < nT; ++t)
    for (int k = 0; k < 3; ++k)
      if (tri[4 * t + k] < 0 || (uint32_t)tri[4 * t + k] >= nV) { char b[96]; snprintf(b, sizeof b, "triangle %u index out of range", t); return fail(c, CRH_E_INVALID, b); }
  if (!all_finite(pos, 3 * (size_t)nV, 1.0e30f) || !all_finite(nrm, 3 * (size_t)nV) || (uv && !all_finite(uv, 2 * (size_t)nV)) || !all_finite(xform, 12, 1.0e30f))
    return fail(c, CRH_E_INVALID, "geometry holds a NaN / Inf (or a coordinate beyond 1e30)");
  const uint32_t T0 = (uint32_t)(c->tri.size() / 4), V0 = (uint32_t)(c->pos.size() / 3), ob = c->nO;
  if ((uint64_t)T0 + nT >= (1u << 28) || (uint64_t)c->n_pos + nT >= (1u << 28)) return fail(c, CRH_E_INVALID, "too many triangles (limit 2^28)");
  CRH_HIP(hipSetDevice(c->device));
  // room for the new object's leaf positions BEHIND the room crh_build left for the object trees of the scene's own objects that have not moved yet (the new tree
  // must not live on their reservation: a later crh_set_transforms would find the positions exhausted -- tests/hunts/visibility_walks.py found that), and as
  // many again: the next additions should not reallocate 64 B x 3 per triangle of the whole scene each time
  uint64_t reserved = 0;
  for (const TwoLevelState::Obj& o : c->objs) if (!o.built) reserved += o.ntri;
  if ((uint64_t)c->n_pos + reserved + nT >= (1ull << 28)) return fail(c, CRH_E_INVALID, "too many triangles (limit 2^28)");
  if ((size_t)c->n_pos + reserved + nT > c->cap_pos) {
    const size_t cap = (size_t)c->n_pos + (size_t)reserved + 2 * (size_t)nT + c->cap_pos / 4;
    int rc;
    CRH_HIP(hipStreamSynchronize(cstream(c)));
    for (int k = 0; k < 8; ++k) if (c->pipe_pending[k]) CRH_HIP(hipEventSynchronize(c->lane_join[k]));      // frames in flight still read the old arrays
    if ((rc = grow_positions(c, c->d_tris, kTriStride, c->n_pos, cap))) return rc;
    if ((rc = grow_positions(c, c->d_shade, 4, c->n_pos, cap))) return rc;
    if (c->d_uvs && (rc = grow_positions(c, c->d_uvs, 2, c->n_pos, cap))) return rc;
    if ((rc = grow_positions(c, c->d_verts, 3, c->n_pos, cap))) return rc;
    c->cap_pos = cap;
  }
  // host arrays: the new object's vertices and triangles behind the scene's own (vertex indices shifted), one more object
  c->pos.insert(c->pos.end(), pos, pos + 3 * (size_t)nV); c->nrm.insert(c->nrm.end(), nrm, nrm + 3 * (size_t)nV);
  c->pos_w.insert(c->pos_w.end(), pos, pos + 3 * (size_t)nV); c->nrm_w.insert(c->nrm_w.end(), nrm, nrm + 3 * (size_t)nV);
  if (!c->uv.empty()) { if (uv) c->uv.insert(c->uv.end(), uv, uv + 2 * (size_t)nV); else c->uv.resize(c->uv.size() + 2 * (size_t)nV, 0.f); }
  c->tri.resize(4 * ((size_t)T0 + nT)); c->tri_obj.resize((size_t)T0 + nT); c->obj_tris.resize((size_t)T0 + nT); c->static_pos.resize((size_t)T0 + nT, 0u);
  for (uint32_t t = 0; t < nT; ++t) {
    for (int k = 0; k < 3; ++k) c->tri[4 * ((size_t)T0 + t) + k] = tri[4 * t + k] + (int32_t)V0;
    c->tri[4 * ((size_t)T0 + t) + 3] = tri[4 * t + 3];
    c->tri_obj[T0 + t] = (int32_t)ob; c->obj_tris[T0 + t] = T0 + t;
  }
  c->xf.insert(c->xf.end(), xform, xform + 12); c->xf0.insert(c->xf0.end(), xform, xform + 12);
  TwoLevelState::Obj o{}; o.first = T0; o.ntri = nT; o.static0 = false; o.in_static = false;
  c->objs.push_back(o);
  if (!c->hidden.empty()) c->hidden.push_back(0);
  c->nO = ob + 1;
  if (object_out) *object_out = ob;
  return apply_objects(c, nullptr, nullptr);
}

int crh_get_tlas(crh_ctx* c, uint32_t* root, uint32_t* n_inst, uint32_t* n_blas)
{
  if (!c) return CRH_E_INVALID;
  if (!c->built) return fail(c, CRH_E_NOTBUILT, "crh_build has not been called");
  if (root) *root = c->inst.empty() ? 0u : (c->root2 != kQEmpty ? c->root2 : c->root);
  if (n_inst) *n_inst = (uint32_t)c->inst.size(); if (n_blas) *n_blas = c->n_blas_nodes;
  return CRH_OK;
}

int crh_set_materials(crh_ctx* c, const crh_bsdf* m, uint32_t n)
{
  if (!c || (n && !m)) return fail(c, CRH_E_INVALID, "null materials");
  if (!all_finite((const float*)m, 32 * (size_t)n)) return fail(c, CRH_E_INVALID, "material holds a NaN / Inf");
  CRH_HIP(hipSetDevice(c->device));
  c->mats.assign(m, m + n); c->pending_n = 0;
  return dev_put(c, c->d_mats, c->cap_mats, c->mats.data(), sizeof(crh_bsdf) * n, 16 * sizeof(crh_bsdf));
}

int crh_set_lights(crh_ctx* c, const crh_light* l, uint32_t n)
{
  if (!c || (n && !l)) return fail(c, CRH_E_INVALID, "null lights");
  if (!all_finite((const float*)l, 8 * (size_t)n, 1.0e30f)) return fail(c, CRH_E_INVALID, "light holds a NaN / Inf");
  CRH_HIP(hipSetDevice(c->device));
  c->lights.assign(l, l + n); c->pending_n = 0;
  return upload_lights(c);
}

int crh_set_envmap(crh_ctx* c, const float* rgb, uint32_t w, uint32_t h)
{
  if (!c) return CRH_E_INVALID;
  if (rgb && w && h && !all_finite(rgb, 3 * (size_t)w * h)) return fail(c, CRH_E_INVALID, "environment map holds a NaN / Inf");
  CRH_HIP(hipSetDevice(c->device));
  c->envW = c->envH = 0; c->pending_n = 0;
  if (rgb && w && h) {
    std::vector<float> t(4 * (size_t)w * h);
    for (size_t i = 0; i < (size_t)w * h; ++i) { t[4 * i] = rgb[3 * i]; t[4 * i + 1] = rgb[3 * i + 1]; t[4 * i + 2] = rgb[3 * i + 2]; t[4 * i + 3] = 0.f; }
    int rc = dev_put(c, c->d_env, c->cap_env, t.data(), t.size() * sizeof(float)); if (rc) return rc;
    c->envW = w; c->envH = h;
  }
  return CRH_OK;      // without a map the kernels take the background colour (fill_scene hands them a null pointer); the allocation is kept
}

int crh_set_texture(crh_ctx* c, uint32_t slot, const float* rgb, uint32_t w, uint32_t h, uint32_t channels)
{
  if (!c || slot >= 4096u) return fail(c, CRH_E_INVALID, "texture slot out of range");
  if (rgb && channels != 3u && channels != 4u) return fail(c, CRH_E_INVALID, "texture channels must be 3 or 4");
  if (rgb && w && h && !all_finite(rgb, (size_t)channels * w * h)) return fail(c, CRH_E_INVALID, "texture holds a NaN / Inf");
  CRH_HIP(hipSetDevice(c->device));
  CRH_HIP(hipStreamSynchronize(cstream(c)));
  if (c->textures.size() <= slot) c->textures.resize(slot + 1);
  HostInputs::HostTex& t = c->textures[slot];
  t.rgba.clear(); t.w = t.h = 0;
  if (rgb && w && h) {
    t.rgba.resize(4 * (size_t)w * h);
    for (size_t i = 0; i < (size_t)w * h; ++i) {
      t.rgba[4 * i] = rgb[channels * i]; t.rgba[4 * i + 1] = rgb[channels * i + 1]; t.rgba[4 * i + 2] = rgb[channels * i + 2];
      t.rgba[4 * i + 3] = channels == 4u ? rgb[4 * i + 3] : 1.f;
    }
    t.w = w; t.h = h;
  }
  c->textures_dirty = true; c->pending_n = 0;
  return do_reset(c);
}

int crh_set_camera(crh_ctx* c, const crh_camera* cam)
{
  if (!c || !cam) return fail(c, CRH_E_INVALID, "null camera");
  const float f[] = {cam->eye[0], cam->eye[1], cam->eye[2], cam->dir[0], cam->dir[1], cam->dir[2], cam->up[0], cam->up[1], cam->up[2],
                     cam->fovy_deg, cam->aspect, cam->ortho_scale, cam->aperture_radius, cam->focal_dist};
  if (!all_finite(f, sizeof f / sizeof f[0], 1.0e30f)) return fail(c, CRH_E_INVALID, "camera holds a NaN / Inf");
  c->cam = *cam; c->pending_n = 0;                      // samples traced ahead with the old camera are dropped
  c->read_since_render = true;
  return CRH_OK;
}

int crh_set_params(crh_ctx* c, const crh_params* p)
{
  if (!c || !p) return fail(c, CRH_E_INVALID, "null params");
  if (!p->width || !p->height || p->max_depth < 1 || p->max_depth > 32) return fail(c, CRH_E_INVALID, "width/height must be > 0 and max_depth in 1..32");
  if (p->width > 32768u || p->height > 32768u || (uint64_t)p->width * p->height > (1ull << 28)) return fail(c, CRH_E_INVALID, "render target too large (limit 32768 per side, 2^28 pixels)");
  if (p->tile_size < 8 || (p->tile_size & 7u) || p->tile_size > 1024) return fail(c, CRH_E_INVALID, "tile_size must be a multiple of 8 in 8..1024");
  { const float f[] = {p->radiance_clamp, p->exposure, p->white_point, p->background[0], p->background[1], p->background[2], p->scene_epsilon};
    if (!all_finite(f, sizeof f / sizeof f[0])) return fail(c, CRH_E_INVALID, "params hold a NaN / Inf"); }
  if (p->width != c->par.width || p->height != c->par.height || p->max_depth != c->par.max_depth) c->feed_tune.restart();      // another frame: measured again
  if (p->width != c->par.width || p->height != c->par.height || p->tile_size != c->par.tile_size) { c->tile_order.order.clear(); c->tile_order.cls.clear(); c->tile_order.pending = false; c->tile_order.remeasure(); }      // other tiles
  c->par = *p;
  return do_reset(c);
}

int crh_set_spec(crh_ctx* c, const crh_spec* sp)
{
  if (!c) return CRH_E_INVALID;
  crh_spec n; const char* why = "";
  if (crh_spec_normalise(sp, &n, &why)) return fail(c, CRH_E_INVALID, why);      // an older, shorter struct is accepted: the fields it lacks take their defaults
  c->spec = n;
  return do_reset(c);                                     // like every rendering-parameter change (pending look-ahead samples are dropped there)
}

int crh_get_spec(crh_ctx* c, crh_spec* out)
{
  if (!c || !out) return fail(c, CRH_E_INVALID, "null spec");
  const char* why = "";
  if (crh_spec_export(&c->spec, out, &why)) return fail(c, CRH_E_INVALID, why);      // out->size is the CALLER's struct size: an older, shorter struct gets only its own bytes
  return CRH_OK;
}

int crh_spec_order_exact(void) { return CRH_SPEC_ORDER_EXACT; }
int crh_spec_anyhit_slot_order(void) { return CRH_SPEC_ANYHIT_SLOT_ORDER; }

// crh_build and crh_build_prebuilt: `pre_nodes` / `pre_order` non-null = the single-level tree handed over instead of built
static int build_scene(crh_ctx* c, const QNode* pre_nodes, uint32_t pre_n_nodes, const uint32_t* pre_order)
{
  const uint32_t nT = (uint32_t)(c->tri.size() / 4);
  if (nT && c->mats.empty()) return fail(c, CRH_E_INVALID, "no materials");
  CRH_HIP(hipSetDevice(c->device));
  CRH_HIP(hipStreamSynchronize(cstream(c)));
  const int threads = build_threads_env();
  const bool verbose = getenv("CRH_BUILD_VERBOSE") != nullptr; auto tp = std::chrono::steady_clock::now();
  auto phase = [&](const char* what) { if (verbose) { const auto now = std::chrono::steady_clock::now(); fprintf(stderr, "crh_build: %-28s %.3f s\n", what, std::chrono::duration<double>(now - tp).count()); tp = now; } };
  c->inst.clear(); c->root = 0; c->root2 = kQEmpty; c->objs.clear(); c->obj_tris.clear(); c->static_pos.clear(); c->pos_obj.clear();
  c->bvh.nodes.clear(); c->bvh.prim_order.clear();
  if (!c->two_level && pre_nodes) {
    // the tree of another context / process for the SAME geometry (crh_build_prebuilt validated it): nodes and leaf order are copied, the bounds
    // are what build_tree takes them from -- the extremes of the triangles' vertices
    c->bvh.nodes.assign(pre_nodes, pre_nodes + pre_n_nodes);
    c->bvh.prim_order.assign(pre_order, pre_order + nT);
    float lo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, hi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
    for (uint32_t t = 0; t < nT; ++t)
      for (int k = 0; k < 3; ++k) { const float* v = &c->pos[3 * (size_t)c->tri[4 * (size_t)t + k]]; for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], v[a]); hi[a] = std::max(hi[a], v[a]); } }
    for (int a = 0; a < 3; ++a) { c->bvh.bbmin[a] = nT ? lo[a] : 0.f; c->bvh.bbmax[a] = nT ? hi[a] : 0.f; }
    c->n_static = c->n_static_live = c->n_pos = nT;
    for (int a = 0; a < 3; ++a) { c->sbmin[a] = c->bvh.bbmin[a]; c->sbmax[a] = c->bvh.bbmax[a]; }
  } else if (!c->two_level) {
    build_qbvh(c->pos.data(), c->tri.data(), nT, c->bvh, threads);
    c->n_static = c->n_static_live = c->n_pos = nT;
    for (int a = 0; a < 3; ++a) { c->sbmin[a] = c->bvh.bbmin[a]; c->sbmax[a] = c->bvh.bbmax[a]; }
  } else {
    // static / moved split: the objects at the identity share ONE world-space tree (first in the node array, its triangles first in leaf
    // order); every other non-empty object gets an object-space tree (triangles in input order); then the top-level tree over the instances
    c->objs.assign(c->nO, TwoLevelState::Obj{}); c->obj_tris.resize(nT ? nT : 1); c->static_pos.assign(nT ? nT : 1, 0u);
    for (uint32_t t = 0; t < nT; ++t) c->objs[c->tri_obj[t]].ntri++;
    { uint32_t acc = 0; for (uint32_t ob = 0; ob < c->nO; ++ob) { TwoLevelState::Obj& o = c->objs[ob]; o.first = acc; acc += o.ntri; o.ntri = 0; o.static0 = true; o.in_static = true; } }
    for (uint32_t t = 0; t < nT; ++t) { TwoLevelState::Obj& o = c->objs[c->tri_obj[t]]; c->obj_tris[o.first + o.ntri++] = t; }
    // bake: every vertex under the transform its object has NOW (each vertex belongs to one object; an object at the identity keeps its bits) -- a
    // loaded scene whose objects all carry a location (vlocation lines of model.tcl) renders as ONE tree at the single-level rate until one is dragged
    c->xf0 = c->xf; c->pos_w = c->pos; c->nrm_w = c->nrm;
    {
      std::vector<uint8_t> done(c->pos.size() / 3 + 1, 0);
      for (uint32_t t = 0; t < nT; ++t) {
        const float* M = &c->xf0[12 * (size_t)c->tri_obj[t]];
        if (is_identity(M)) continue;
        for (int k = 0; k < 3; ++k) {
          const int32_t vi = c->tri[4 * t + k];
          if (done[vi]) continue;
          done[vi] = 1;
          const crh_v3 pw = crh_xform_point(M, crh_mk3(c->pos[3 * vi], c->pos[3 * vi + 1], c->pos[3 * vi + 2]));
          crh_v3 nn = crh_norm3(crh_xform_vector(M, crh_mk3(c->nrm[3 * vi], c->nrm[3 * vi + 1], c->nrm[3 * vi + 2])));
          if (!(crh_dot3(nn, nn) > 0.f)) nn = crh_mk3(c->nrm[3 * vi], c->nrm[3 * vi + 1], c->nrm[3 * vi + 2]);
          c->pos_w[3 * vi] = pw.x; c->pos_w[3 * vi + 1] = pw.y; c->pos_w[3 * vi + 2] = pw.z;
          c->nrm_w[3 * vi] = nn.x; c->nrm_w[3 * vi + 1] = nn.y; c->nrm_w[3 * vi + 2] = nn.z;
        }
      }
    }
    std::vector<uint32_t> list; list.reserve(nT);
    for (uint32_t t = 0; t < nT; ++t) if (c->objs[c->tri_obj[t]].static0) list.push_back(t);
    const uint32_t nS = (uint32_t)list.size();
    c->n_static = c->n_static_live = nS; c->n_pos = nS;
    if (nS) {
      std::vector<float> boxes(6 * (size_t)nS); std::vector<uint32_t> order;
      for (uint32_t i = 0; i < nS; ++i)
        for (int a = 0; a < 3; ++a) {
          const uint32_t t = list[i];
          const float v0 = c->pos_w[3 * c->tri[4 * t + 0] + a], v1 = c->pos_w[3 * c->tri[4 * t + 1] + a], v2 = c->pos_w[3 * c->tri[4 * t + 2] + a];
          boxes[6 * (size_t)i + a] = std::min(v0, std::min(v1, v2)); boxes[6 * (size_t)i + 3 + a] = std::max(v0, std::max(v1, v2));
        }
      c->bvh.nodes.reserve(nS / 2 + 16);
      build_tree(boxes.data(), nS, false, 0, c->bvh.nodes, order, c->sbmin, c->sbmax, threads);
      c->bvh.prim_order.resize(nS);
      for (uint32_t i = 0; i < nS; ++i) { c->bvh.prim_order[i] = list[order[i]]; c->static_pos[list[order[i]]] = i; }
    }
    for (uint32_t ob = 0; ob < c->nO; ++ob) {
      TwoLevelState::Obj& o = c->objs[ob];
      if (o.static0 || !o.ntri) continue;
      build_object_tree(c, ob, threads);
      o.is_inst = true;
    }
  }
  phase("trees");
  c->n_blas_nodes = (uint32_t)c->bvh.nodes.size();
  if (c->n_pos >= (1u << 28)) return fail(c, CRH_E_INVALID, "too many leaf positions (limit 2^28)");
  if (c->two_level) {
    // room for the instance table of ANY later placement (one record per object + one per instance), so that the first crh_set_transforms --
    // the user has just grabbed the gizmo -- allocates nothing
    const size_t want_inst = 128 * (2 * (size_t)c->nO + 64);
    if (c->cap_inst < want_inst) {
      if (c->d_inst) { CRH_HIP(hipFree(c->d_inst)); c->d_inst = nullptr; c->cap_inst = 0; }
      CRH_HIP(hipMalloc((void**)&c->d_inst, want_inst)); c->cap_inst = want_inst;
      CRH_HIP(hipMemsetAsync(c->d_inst, 0, want_inst, cstream(c)));
    }
  }
  { int rc_t = build_tlas(c); if (rc_t) return rc_t; }
  // leaf-ordered triangle, shading and uv records.  A two-level scene keeps room for an object tree of every object of the static tree
  // (each may be dragged away once; the copies cost 2 x 64 B per triangle of HBM, nothing at run time)
  std::vector<float> tr, sh, uvr, vt;
  c->h_tris.clear();
  fill_records(c, 0, c->n_pos, tr, sh, uvr, c->two_level ? &vt : nullptr);
  phase("leaf-ordered records");
  c->cap_pos = (size_t)std::max(c->n_pos, 1u) + (c->two_level ? c->n_static : 0u);
  if (c->two_level) {
    // host arrays that grow when an object tree is built later: reserve now -- the first growth of a 33 MB node vector or a 48 MB record vector is a
    // reallocation + copy of 10-20 ms, which used to land in the first dragged frame
    c->bvh.nodes.reserve(c->bvh.nodes.size() + 2 * (size_t)c->n_static + 4 * (size_t)c->nO + 64);
    c->h_tris.reserve(12 * c->cap_pos); c->bvh.prim_order.reserve(c->cap_pos); c->pos_obj.reserve(c->cap_pos);
  }
  int rc;
  // head-room behind the node array: object trees built later by crh_set_transforms (<= ~1.5 nodes per triangle incl. alignment holes) and the top-level tree
  if ((rc = dev_put(c, c->d_nodes, c->cap_nodes, c->bvh.nodes.data(), c->bvh.nodes.size() * sizeof(QNode),
                    (size_t)((c->two_level ? 2 * (size_t)c->n_static : 0) + 4 * (size_t)c->nO + 64) * sizeof(QNode)))) return rc;
  auto alloc_put = [&](float4*& dptr, const std::vector<float>& v, size_t rec_floats) -> int {
    CRH_HIP(hipStreamSynchronize(cstream(c)));
    if (dptr) { CRH_HIP(hipFree(dptr)); dptr = nullptr; }
    CRH_HIP(hipMalloc((void**)&dptr, c->cap_pos * rec_floats * sizeof(float)));
    CRH_HIP(hipMemcpyAsync(dptr, v.data(), (size_t)std::max(c->n_pos, 1u) * rec_floats * sizeof(float), hipMemcpyHostToDevice, cstream(c)));
    CRH_HIP(hipStreamSynchronize(cstream(c)));
    return CRH_OK;
  };
  if ((rc = alloc_put(c->d_tris, tr, 4 * kTriStride))) return rc;
  if ((rc = alloc_put(c->d_shade, sh, 16))) return rc;
  if (!c->uv.empty()) { if ((rc = alloc_put(c->d_uvs, uvr, 8))) return rc; }
  else if (c->d_uvs) { CRH_HIP(hipFree(c->d_uvs)); c->d_uvs = nullptr; }
  if (c->two_level) { if ((rc = alloc_put(c->d_verts, vt, 12))) return rc; }
  else if (c->d_verts) { CRH_HIP(hipFree(c->d_verts)); c->d_verts = nullptr; }
  // packet nodes for the camera rays of wide batches (k_trace_packets<true>): the node array again with its quantised planes as floats, 128 B per node.
  // Single-level scenes only (a scene with objects grows its node array while the user drags; its packets read the 64-B nodes)
  if (!c->two_level && c->packets > 0) {
    const size_t nn = c->bvh.nodes.size();
    if (nn > c->cap_pnodes) { if (c->d_pnodes) { CRH_HIP(hipFree(c->d_pnodes)); c->d_pnodes = nullptr; c->cap_pnodes = 0; } CRH_HIP(hipMalloc((void**)&c->d_pnodes, nn * 128)); c->cap_pnodes = nn; }
    Launch Lx{cstream(c), 64, false};
    launch_expand_packet_nodes(Lx, c->d_nodes, c->d_pnodes, (uint32_t)nn);
    CRH_HIP(hipGetLastError());
  } else if (c->d_pnodes) { CRH_HIP(hipFree(c->d_pnodes)); c->d_pnodes = nullptr; c->cap_pnodes = 0; }
  phase("upload");
  if (c->two_level) {
    // what the FIRST crh_set_transforms would otherwise allocate while the user is dragging: the staging of the triangle patches of the largest object
    uint32_t biggest = 0; for (const TwoLevelState::Obj& o : c->objs) biggest = std::max(biggest, o.ntri);
    // ... and the pinned staging buffers the records of one object travel through (stage_copy grows them lazily: four hipHostMalloc of a few
    // milliseconds each would otherwise land in the first dragged frame)
    const size_t want_stage = std::min<size_t>(4u << 20, (size_t)biggest * 64 + (size_t)c->nO * 160 + 65536);
    for (BuiltScene::Stage& st : c->stage)
      if (st.cap < want_stage) {
        if (st.used) CRH_HIP(hipEventSynchronize(st.ev));
        if (st.p) { CRH_HIP(hipHostFree(st.p)); st.p = nullptr; st.cap = 0; }
        CRH_HIP(hipHostMalloc(&st.p, want_stage, hipHostMallocDefault)); st.cap = want_stage;
      }
    const size_t want_patch = 2 * ((size_t)biggest * 52 + 512);
    if (c->cap_patch < want_patch) {
      if (c->d_patch) { CRH_HIP(hipFree(c->d_patch)); c->d_patch = nullptr; c->cap_patch = 0; }
      CRH_HIP(hipMalloc(&c->d_patch, want_patch)); c->cap_patch = want_patch;
    }
  }
  c->built = true; c->feed_tune.restart(); c->tile_order.remeasure();
  if (c->two_level && c->inst.empty()) {
    // Every object sits at the identity: the single-level kernels render this scene.  The first crh_set_transforms (the user has just grabbed the
    // gizmo) switches to the two-level instantiations and the record scatter -- launch each of them once now, on empty queues, so that their
    // first-launch cost (function lookup, code upload: ~20 ms for the set) is paid while the scene loads and not in the first dragged frame.
    if ((rc = ensure_paths(c, 4096))) return rc;
    CRH_HIP(hipMemsetAsync(c->queues.counts, 0, kCounts * sizeof(uint32_t), cstream(c)));
    DScene S; fill_scene(c, S); S.two_level = 1; S.root2 = 0;
    for (int don = 0; don < 2; ++don) {
      Launch LT{cstream(c), 64, false, c->clamp_grid ? c->cus : 0, don != 0};      // with the occupancy query of resident_grid<>
      S.split = 0; launch_trace_nearest(LT, S, c->paths, c->queues, 0, 0, c->d_counters); launch_trace_any(LT, S, c->paths, c->queues, c->d_counters);      // the instantiations of an all-moved scene
      S.split = 1; launch_trace_nearest(LT, S, c->paths, c->queues, 0, 0, c->d_counters); launch_trace_any(LT, S, c->paths, c->queues, c->d_counters);      // and the second-pass ones
    }
    Launch L{cstream(c), 64, false};
    launch_scatter_tris(L, c->d_tris, (const uint32_t*)c->d_patch, (const float4*)c->d_patch, 0);
    CRH_HIP(hipGetLastError());
  }
  if (c->two_level && !c->hidden.empty()) { if (c->hidden.size() != c->nO) c->hidden.assign(c->nO, 0); if ((rc = apply_objects(c, nullptr, nullptr))) return rc; }      // objects erased before the build: baked like the rest, then disabled
  rc = do_reset(c); if (rc) return rc;
  CRH_HIP(hipStreamSynchronize(cstream(c)));
  return CRH_OK;
}

int crh_build(crh_ctx* c)
{
  if (!c) return CRH_E_INVALID;
  return build_scene(c, nullptr, 0, nullptr);
}

int crh_build_prebuilt(crh_ctx* c, const float* nodes, uint32_t n_nodes, const uint32_t* prim_order, uint32_t n_tris)
{
  if (!c) return CRH_E_INVALID;
  if (c->two_level) return fail(c, CRH_E_INVALID, "crh_build_prebuilt takes single-level scenes (no per-object transforms)");
  const uint32_t nT = (uint32_t)(c->tri.size() / 4);
  if (!nodes || !prim_order || n_tris != nT || n_nodes == 0 || n_nodes > 2u * std::max(nT, 1u) + 16u) return fail(c, CRH_E_INVALID, "prebuilt tree does not fit the geometry (triangle / node count)");
  // the kernels trust the tree: every reference must stay inside it, every triangle must sit at exactly one leaf position
  {
    std::vector<uint8_t> seen(std::max(nT, 1u), 0);
    for (uint32_t p = 0; p < nT; ++p) { const uint32_t t = prim_order[p]; if (t >= nT || seen[t]) return fail(c, CRH_E_INVALID, "prebuilt leaf order is not a permutation of the triangles"); seen[t] = 1; }
    const QNode* q = (const QNode*)nodes;
    for (uint32_t i = 0; i < n_nodes; ++i) {
      const uint32_t w3 = q[i].w[3], ni = CRH_NODE_NINNER(w3), nc = CRH_NODE_NCHILDREN(w3);
      if (nc == 0 && ni == 0) continue;                                     // an alignment hole (all zero) or the root of an empty scene
      if (nc > 4 || ni > nc) return fail(c, CRH_E_INVALID, "prebuilt node with a bad child count");
      if (ni && ((uint64_t)q[i].w[10] + ni > n_nodes || q[i].w[10] <= i)) return fail(c, CRH_E_INVALID, "prebuilt node refers outside the node array");
      if (nc > ni) {
        const uint32_t lb = q[i].w[11];
        if ((lb & 0xF0000000u) != CRH_LEAF_TAG || (uint64_t)(lb & 0x0FFFFFFFu) + (nc - ni) > nT) return fail(c, CRH_E_INVALID, "prebuilt node refers outside the leaf positions");
      }
    }
    // Depth: the per-lane traversal stack holds kLdsStack + kOvfStack entries and is not bounds-checked on the device.  A walk that arrives at a
    // node has at most `pend` entries waiting (every ancestor left <= n_children - 1 siblings behind) and pushes <= n_children - 1 more.  Links only point
    // forward (checked above), so one pass in index order sees every parent before its children.
    {
      std::vector<uint16_t> pend(n_nodes, 0); std::vector<uint8_t> reached(n_nodes, 0);
      reached[0] = 1;
      for (uint32_t i = 0; i < n_nodes; ++i) {
        const uint32_t w3 = q[i].w[3], ni = CRH_NODE_NINNER(w3), nc = CRH_NODE_NCHILDREN(w3);
        if (!reached[i] || nc == 0) continue;
        if ((uint32_t)pend[i] + (nc - 1u) > (uint32_t)(kLdsStack + kOvfStack)) return fail(c, CRH_E_INVALID, "prebuilt tree is deeper than the traversal stack (kLdsStack + kOvfStack pending entries)");
        for (uint32_t k = 0; k < ni; ++k) {
          const uint32_t ch = q[i].w[10] + k;
          if (reached[ch]) return fail(c, CRH_E_INVALID, "prebuilt node has two parents");
          reached[ch] = 1; pend[ch] = (uint16_t)(pend[i] + (nc - 1u));
        }
      }
      // Containment: a child box that does not hold what hangs below it gives silently wrong images.  The TRUE bounds of every subtree follow bottom-up from the
      // triangles (children have larger indices than their parent: one pass from the last node to the first); every slot's box -- origin + q * 2^k on the node's
      // grid -- must contain them.  The quantiser is conservative, so a tree this library built passes exactly.
      std::vector<float> tb(6 * (size_t)n_nodes);
      auto plane = [&](const QNode& n, int a, uint32_t qv) { return (double)crh_u2f(n.w[a]) + (double)qv * (double)crh_quant_step(CRH_NODE_STEP_E(n.w[3], a)); };
      for (uint32_t i = n_nodes; i-- > 0;) {
        const uint32_t w3 = q[i].w[3], ni = CRH_NODE_NINNER(w3), nc = CRH_NODE_NCHILDREN(w3);
        float* me = &tb[6 * (size_t)i];
        for (int a = 0; a < 3; ++a) { me[a] = 3.0e38f; me[3 + a] = -3.0e38f; }
        if (!reached[i]) continue;
        for (uint32_t k = 0; k < nc; ++k) {
          float lo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, hi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
          if (k < ni) { const float* ch = &tb[6 * (size_t)(q[i].w[10] + k)]; for (int a = 0; a < 3; ++a) { lo[a] = ch[a]; hi[a] = ch[3 + a]; } }
          else {
            const uint32_t t = prim_order[(q[i].w[11] & 0x0FFFFFFFu) + (k - ni)];
            for (int v = 0; v < 3; ++v) {
              const float* pv = &c->pos[3 * (size_t)c->tri[4 * (size_t)t + v]];
              for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], pv[a]); hi[a] = std::max(hi[a], pv[a]); }
            }
          }
          for (int a = 0; a < 3; ++a) {
            if (lo[a] > hi[a]) continue;                                   // an empty subtree holds nothing to miss
            if (plane(q[i], a, (q[i].w[4 + a] >> (8 * k)) & 0xffu) > (double)lo[a] || plane(q[i], a, (q[i].w[7 + a] >> (8 * k)) & 0xffu) < (double)hi[a])
              return fail(c, CRH_E_INVALID, "prebuilt child box does not contain what hangs below it");
            me[a] = std::min(me[a], lo[a]); me[3 + a] = std::max(me[3 + a], hi[a]);
          }
        }
      }
    }
  }
  return build_scene(c, (const QNode*)nodes, n_nodes, prim_order);
}

int crh_get_bvh(crh_ctx* c, float* nodes, uint32_t* nn, float* tris, uint32_t* nt)
{
  if (!c) return CRH_E_INVALID;
  if (!c->built) return fail(c, CRH_E_NOTBUILT, "crh_build has not been called");
  const uint32_t nP = c->n_pos;                         // leaf positions in use: the triangles, + the object-tree copies of objects dragged out of the static tree
  if (nn) *nn = (uint32_t)c->bvh.nodes.size();
  if (nt) *nt = nP;
  if (nodes) std::memcpy(nodes, c->bvh.nodes.data(), c->bvh.nodes.size() * sizeof(QNode));
  if (tris && nP) std::memcpy(tris, c->h_tris.data(), 48 * (size_t)nP);
  return CRH_OK;
}

int crh_build_bvh_host(const float* pos, uint32_t nV, const int32_t* tri, uint32_t nT, int threads, float* nodes, uint32_t* n_nodes,
                       uint32_t* prim_order)
{
  if ((nT && (!pos || !tri)) || !n_nodes) return CRH_E_INVALID;
  for (uint32_t t = 0; t < nT; ++t) for (int k = 0; k < 3; ++k) if (tri[4 * t + k] < 0 || (uint32_t)tri[4 * t + k] >= nV) return CRH_E_INVALID;
  if (nV && !all_finite(pos, 3 * (size_t)nV, 1.0e30f)) return CRH_E_INVALID;
  QBvh b; build_qbvh(pos, tri, nT, b, threads);
  *n_nodes = (uint32_t)b.nodes.size();
  if (nodes) std::memcpy(nodes, b.nodes.data(), b.nodes.size() * sizeof(QNode));
  if (prim_order && nT) std::memcpy(prim_order, b.prim_order.data(), sizeof(uint32_t) * nT);
  return CRH_OK;
}

}  // extern "C"

// tools/experiments/wide8_gpu.hip -- GPU A/B of the "one 128-B line" 8-wide BVH node against the 4-wide 64-B node
// (VERDICT round 1, item 4).  NOT part of the product: a self-contained experiment with its own binary SAH builder, one
// greedy collapse used for both widths, and ONE persistent-wave engine (pool + lane refill + LDS stack, the product's shape)
// templated on the node width, so that the only thing that differs between the two measurements is the node.
//
//   hipcc -O3 --offload-arch=gfx950 -o wide8_gpu tools/experiments/wide8_gpu.hip
//   ./wide8_gpu <triangles> <rays> [waves_per_simd_w4] [waves_per_simd_w8]
//
// Node formats (quantiser of include/crh_bvh_format.h in both):
//   W=4:  the product's node, 12 dwords on a 64-B stride, 3 x dwordx4 per visit
//   W=8:  18 dwords on a 128-B stride: origin, exponents | counts, 8 x 6 bound bytes, two bases; 4 x dwordx4 + 1 x dwordx2 per visit
// Children of a node: inner children first (consecutive node indices), then leaves (consecutive leaf-order triangles), one
// triangle per leaf -- the product's implicit references.  Child order: sorted by entry distance (4: 5-comparator network,
// 8: 19-comparator network), or for W=8 optionally octant-ordered slots walked without a sort (mode "oct").
// Round 6 (round-5 verdict, item 2): MODE 4 / 5 of the 4-wide walk -- the child planes as FP8 (E4M3) numbers converted TWO per instruction
// (v_cvt_pk_f32_fp8) instead of bytes converted one at a time (v_cvt_f32_ubyteN): 12 converts per visit instead of 24.  MODE 5: both planes as offsets from the
// node's minimum corner (lower planes rounded down, upper planes rounded up to the next E4M3 number); MODE 4: corner-relative -- lower planes from the minimum
// corner, upper planes from the MAXIMUM corner, both rounded down, so that the planes near either corner are exact (tools/experiments/fp8_planes_visits.py is
// the CPU estimate of this variant; it needs the maximum corner in the node's fourth quarter: 64 B fetched per visit instead of 48).
// The program checks that both widths return bit-identical hit distances for every ray, then prints rate, visits per ray and
// (under rocprofv3 --pmc FETCH_SIZE) lets the memory-side bytes be read per kernel name.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/crh_bvh_format.h"

#define HIPCHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

// ------------------------------------------------------------------------------------------------ host: scene + binary SAH tree
static uint64_t g_sm = 0x9E3779B97F4A7C15ull;
static uint64_t splitmix() { uint64_t z = (g_sm += 0x9E3779B97F4A7C15ull); z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }
static float urand() { return (float)((splitmix() >> 40) * (1.0 / 16777216.0)); }

struct BNode { float mn[3], mx[3]; int left, right; uint32_t prim; };
struct Build {
  const float* pb; const float* cen; std::vector<uint32_t> idx; std::vector<BNode> bn;
  int rec(uint32_t lo, uint32_t hi) {
    const int me = (int)bn.size(); bn.push_back(BNode());
    float mn[3] = {3e38f, 3e38f, 3e38f}, mx[3] = {-3e38f, -3e38f, -3e38f}, cmn[3] = {3e38f, 3e38f, 3e38f}, cmx[3] = {-3e38f, -3e38f, -3e38f};
    for (uint32_t i = lo; i < hi; ++i) { const float* b = pb + 6 * idx[i]; const float* c = cen + 3 * idx[i];
      for (int a = 0; a < 3; ++a) { mn[a] = std::min(mn[a], b[a]); mx[a] = std::max(mx[a], b[3 + a]); cmn[a] = std::min(cmn[a], c[a]); cmx[a] = std::max(cmx[a], c[a]); } }
    for (int a = 0; a < 3; ++a) { bn[me].mn[a] = mn[a]; bn[me].mx[a] = mx[a]; }
    if (hi - lo == 1) { bn[me].left = bn[me].right = -1; bn[me].prim = idx[lo]; return me; }
    constexpr int NB = 16;
    int best_axis = -1, best_bin = 0; float best_cost = 3e38f;
    for (int a = 0; a < 3; ++a) {
      const float ext = cmx[a] - cmn[a]; if (!(ext > 0.f)) continue;
      const float sc = NB / ext;
      float bmn[NB][3], bmx[NB][3]; uint32_t cnt[NB];
      for (int b = 0; b < NB; ++b) { cnt[b] = 0; for (int k = 0; k < 3; ++k) { bmn[b][k] = 3e38f; bmx[b][k] = -3e38f; } }
      for (uint32_t i = lo; i < hi; ++i) { const float* bx = pb + 6 * idx[i]; int b = (int)((cen[3 * idx[i] + a] - cmn[a]) * sc); b = b < 0 ? 0 : (b >= NB ? NB - 1 : b);
        ++cnt[b]; for (int k = 0; k < 3; ++k) { bmn[b][k] = std::min(bmn[b][k], bx[k]); bmx[b][k] = std::max(bmx[b][k], bx[3 + k]); } }
      float ra[NB]; uint32_t rc[NB]; float m[3] = {3e38f, 3e38f, 3e38f}, M[3] = {-3e38f, -3e38f, -3e38f}; uint32_t c = 0;
      for (int b = NB - 1; b >= 1; --b) { for (int k = 0; k < 3; ++k) { m[k] = std::min(m[k], bmn[b][k]); M[k] = std::max(M[k], bmx[b][k]); } c += cnt[b];
        const float dx = M[0] - m[0], dy = M[1] - m[1], dz = M[2] - m[2]; ra[b] = c ? dx * dy + dy * dz + dz * dx : 0.f; rc[b] = c; }
      for (int k = 0; k < 3; ++k) { m[k] = 3e38f; M[k] = -3e38f; } c = 0;
      for (int b = 0; b < NB - 1; ++b) { for (int k = 0; k < 3; ++k) { m[k] = std::min(m[k], bmn[b][k]); M[k] = std::max(M[k], bmx[b][k]); } c += cnt[b];
        if (c == 0 || rc[b + 1] == 0) continue;
        const float dx = M[0] - m[0], dy = M[1] - m[1], dz = M[2] - m[2]; const float cost = (dx * dy + dy * dz + dz * dx) * c + ra[b + 1] * rc[b + 1];
        if (cost < best_cost) { best_cost = cost; best_axis = a; best_bin = b; } }
    }
    uint32_t mid;
    if (best_axis < 0) mid = (lo + hi) / 2;
    else {
      const int a = best_axis; const float sc = NB / (cmx[a] - cmn[a]);
      auto it = std::partition(idx.begin() + lo, idx.begin() + hi, [&](uint32_t p) { int b = (int)((cen[3 * p + a] - cmn[a]) * sc); b = b < 0 ? 0 : (b >= NB ? NB - 1 : b); return b <= best_bin; });
      mid = (uint32_t)(it - idx.begin());
      if (mid == lo || mid == hi) mid = (lo + hi) / 2;
    }
    const int l = rec(lo, mid); const int r = rec(mid, hi);
    bn[me].left = l; bn[me].right = r; bn[me].prim = 0;
    return me;
  }
};

// ------------------------------------------------------------------------------------------------ host: collapse to W-wide
struct Wide { std::vector<uint32_t> words; uint32_t n_nodes = 0, stride = 0; std::vector<uint32_t> leaf_prims; double fill = 0; };

static float harea(const BNode& b) { const float dx = b.mx[0] - b.mn[0], dy = b.mx[1] - b.mn[1], dz = b.mx[2] - b.mn[2]; return dx * dy + dy * dz + dz * dx; }

// octant = 1: children are put into slots so that slot index bit a says "upper half along axis a" (greedy assignment by centre
// offset); empty slots allowed; inner/leaf kind per slot comes from a mask
static Wide collapse(const std::vector<BNode>& bn, int W, bool octant)
{
  Wide out; out.stride = W == 4 ? 16u : 32u;
  std::vector<int> owner; owner.push_back(0);                       // wide node -> binary node (-1: alignment pad)
  std::vector<uint32_t>& w = out.words;
  uint64_t kids_total = 0, real_nodes = 0;
  for (size_t i = 0; i < owner.size(); ++i) {
    w.resize((i + 1) * out.stride, 0u);
    if (owner[i] < 0) continue;
    int kids[8]; int nk = 0; const BNode& b = bn[owner[i]];
    if (b.left < 0) kids[nk++] = owner[i];
    else {
      kids[nk++] = b.left; kids[nk++] = b.right;
      while (nk < W) { int best = -1; float ba = -1.f;
        for (int k = 0; k < nk; ++k) if (bn[kids[k]].left >= 0) { const float a = harea(bn[kids[k]]); if (a > ba) { ba = a; best = k; } }
        if (best < 0) break;
        const int l = bn[kids[best]].left, r = bn[kids[best]].right;
        for (int k = nk; k > best + 1; --k) kids[k] = kids[k - 1];
        kids[best] = l; kids[best + 1] = r; ++nk; }
    }
    kids_total += nk; ++real_nodes;
    int slot[8]; int ns = 0;
    if (!octant) {                                                 // inner children first, then leaves, each in collapse order
      for (int k = 0; k < nk; ++k) if (bn[kids[k]].left >= 0) slot[ns++] = kids[k];
      for (int k = 0; k < nk; ++k) if (bn[kids[k]].left < 0) slot[ns++] = kids[k];
    } else {
      float cen[3]; for (int a = 0; a < 3; ++a) cen[a] = 0.5f * (b.mn[a] + b.mx[a]);
      bool used[8] = {false}, done[8] = {false}; for (int s = 0; s < 8; ++s) slot[s] = -1;
      for (int it = 0; it < nk; ++it) { float bc = -3e38f; int bk = -1, bs = -1;
        for (int k = 0; k < nk; ++k) if (!done[k]) for (int s = 0; s < 8; ++s) if (!used[s]) { float c = 0.f;
          for (int a = 0; a < 3; ++a) { const float dd = 0.5f * (bn[kids[k]].mn[a] + bn[kids[k]].mx[a]) - cen[a]; c += ((s >> a) & 1) ? dd : -dd; }
          if (c > bc) { bc = c; bk = k; bs = s; } }
        done[bk] = true; used[bs] = true; slot[bs] = kids[bk]; }
      ns = 8;
    }
    int ni = 0; for (int k = 0; k < ns; ++k) if (slot[k] >= 0 && bn[slot[k]].left >= 0) ++ni;
    if (W == 4 && ni >= 2 && (owner.size() & 1)) owner.push_back(-1);          // pair alignment (product rule)
    const uint32_t child_base = ni ? (uint32_t)owner.size() : 0u, leaf_base = (uint32_t)out.leaf_prims.size() | 0x80000000u;
    uint32_t inner_mask = 0, valid_mask = 0;
    for (int k = 0; k < ns; ++k) if (slot[k] >= 0) { valid_mask |= 1u << k;
      if (bn[slot[k]].left >= 0) { owner.push_back(slot[k]); inner_mask |= 1u << k; } else out.leaf_prims.push_back(bn[slot[k]].prim); }
    float org[3]; uint32_t e[3];
    for (int a = 0; a < 3; ++a) { float lo = 3e38f, hi = -3e38f;
      for (int k = 0; k < ns; ++k) if (slot[k] >= 0) { lo = std::min(lo, bn[slot[k]].mn[a]); hi = std::max(hi, bn[slot[k]].mx[a]); }
      org[a] = lo; e[a] = crh_quant_exp(hi - lo); }
    uint32_t* o = &w[i * out.stride];
    memcpy(o, org, 12);
    const int per = W == 4 ? 1 : 2;                                 // bound words per axis and side
    for (int a = 0; a < 3; ++a) for (int k = 0; k < ns; ++k) {
      uint32_t ql = 255u, qh = 0u;                                  // empty slot: inverted box, never hit
      if (slot[k] >= 0) { ql = crh_quant_lo(bn[slot[k]].mn[a], org[a], e[a]); qh = crh_quant_hi(bn[slot[k]].mx[a], org[a], e[a]); }
      o[4 + per * a + (k >> 2)] |= ql << (8 * (k & 3)); o[4 + per * (3 + a) + (k >> 2)] |= qh << (8 * (k & 3)); }
    if (W == 4) { o[3] = e[0] | (e[1] << 8) | (e[2] << 16) | ((uint32_t)ni << 24) | ((uint32_t)ns << 28); o[10] = child_base; o[11] = leaf_base; }
    else if (!octant) { o[3] = e[0] | (e[1] << 8) | (e[2] << 16) | ((uint32_t)ni << 24) | ((uint32_t)ns << 28); o[16] = child_base; o[17] = leaf_base; }
    else { o[3] = e[0] | (e[1] << 8) | (e[2] << 16) | (inner_mask << 24); o[16] = child_base; o[17] = leaf_base; o[18] = valid_mask; }
  }
  out.n_nodes = (uint32_t)owner.size(); out.fill = (double)kids_total / (double)real_nodes;
  return out;
}

// W = 4, "treelet pair" numbering: a 128-B line (two 64-B node slots, even index first) holds a HEAD node and, in its second half,
// the head's FAVOURITE inner child (largest surface area = most likely to be descended into right after the head, i.e. while
// the line is still in flight).  Every other inner child is a head of its own line; sibling heads are consecutive lines.
//   head (bit 27 of word 3 set when it has a favourite):  slot 0 -> self + 1,  slot k >= 1 (inner) -> child_base + 2 (k - 1)
//   non-head / head without inner children:                slot k (inner) -> child_base + 2 k
static Wide collapse4_pairs(const std::vector<BNode>& bn)
{
  Wide out; out.stride = 16u;
  struct Item { uint32_t index; int b; bool head; };
  std::vector<Item> queue; queue.push_back({0u, 0, true});
  uint32_t next_line = 1;                                            // in units of lines (2 slots)
  std::vector<uint32_t>& w = out.words;
  uint64_t kids_total = 0, favs = 0;
  for (size_t qi = 0; qi < queue.size(); ++qi) {
    const Item it = queue[qi];
    int kids[4]; int nk = 0; const BNode& b = bn[it.b];
    if (b.left < 0) kids[nk++] = it.b;
    else { kids[nk++] = b.left; kids[nk++] = b.right;
      while (nk < 4) { int best = -1; float ba = -1.f;
        for (int k = 0; k < nk; ++k) if (bn[kids[k]].left >= 0) { const float a = harea(bn[kids[k]]); if (a > ba) { ba = a; best = k; } }
        if (best < 0) break;
        const int l = bn[kids[best]].left, r = bn[kids[best]].right;
        for (int k = nk; k > best + 1; --k) kids[k] = kids[k - 1];
        kids[best] = l; kids[best + 1] = r; ++nk; } }
    kids_total += nk;
    int slot[4]; int ns = 0, ni = 0;
    for (int k = 0; k < nk; ++k) if (bn[kids[k]].left >= 0) slot[ns++] = kids[k];
    ni = ns;
    for (int k = 0; k < nk; ++k) if (bn[kids[k]].left < 0) slot[ns++] = kids[k];
    const bool fav = it.head && ni > 0;
    if (fav) { int bk = 0; for (int k = 1; k < ni; ++k) if (harea(bn[slot[k]]) > harea(bn[slot[bk]])) bk = k; std::swap(slot[0], slot[bk]); ++favs; }
    const int n_heads = ni - (fav ? 1 : 0);
    const uint32_t child_base = n_heads ? 2u * next_line : 0u, leaf_base = (uint32_t)out.leaf_prims.size() | 0x80000000u;
    if (fav) queue.push_back({it.index + 1u, slot[0], false});
    for (int k = fav ? 1 : 0; k < ni; ++k) queue.push_back({2u * next_line++, slot[k], true});
    for (int k = ni; k < ns; ++k) out.leaf_prims.push_back(bn[slot[k]].prim);
    if (w.size() < (size_t)(it.index + 2u) * 16u) w.resize((size_t)(it.index + 2u) * 16u, 0u);
    if (w.size() < (size_t)2u * next_line * 16u) w.resize((size_t)2u * next_line * 16u, 0u);
    float org[3]; uint32_t e[3];
    for (int a = 0; a < 3; ++a) { float lo = 3e38f, hi = -3e38f;
      for (int k = 0; k < ns; ++k) { lo = std::min(lo, bn[slot[k]].mn[a]); hi = std::max(hi, bn[slot[k]].mx[a]); }
      org[a] = lo; e[a] = crh_quant_exp(hi - lo); }
    uint32_t* o = &w[(size_t)it.index * 16u];
    memcpy(o, org, 12);
    for (int a = 0; a < 3; ++a) for (int k = 0; k < ns; ++k) {
      o[4 + a] |= crh_quant_lo(bn[slot[k]].mn[a], org[a], e[a]) << (8 * k); o[7 + a] |= crh_quant_hi(bn[slot[k]].mx[a], org[a], e[a]) << (8 * k); }
    o[3] = e[0] | (e[1] << 8) | (e[2] << 16) | ((uint32_t)ni << 24) | (fav ? 1u << 27 : 0u) | ((uint32_t)ns << 28); o[10] = child_base; o[11] = leaf_base;
  }
  out.n_nodes = 2u * next_line; out.fill = (double)kids_total / (double)queue.size();
  fprintf(stderr, "pair layout: %zu nodes in %u lines, %llu with a favourite child in the line\n", queue.size(), next_line, (unsigned long long)favs);
  return out;
}

// ------------------------------------------------------------------------------------------------ device
constexpr int kBlock = 256;
constexpr uint32_t kDone = 0xFFFFFFFFu, kLeaf = 0x80000000u;
constexpr float kDirEps = 1e-30f, kSlabGuard = 4.76837158203125e-07f;   // 2^-21
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ld_stream(const float4* p) { const f32x4 v = __builtin_nontemporal_load((const f32x4*)p); return make_float4(v.x, v.y, v.z, v.w); }
__device__ __forceinline__ void st_stream(float4* p, float4 v) { const f32x4 w = {v.x, v.y, v.z, v.w}; __builtin_nontemporal_store(w, (f32x4*)p); }

__device__ __forceinline__ float inv_dir(float d) { return 1.0f / (fabsf(d) < kDirEps ? (d < 0.f ? -kDirEps : kDirEps) : d); }
#define CE(a, b) { const uint32_t lo_ = min(a, b); const uint32_t hi_ = max(a, b); a = lo_; b = hi_; }

// what the hardware makes of the 256 FP8 codes (gfx950: OCP E4M3): the host encoder rounds against THIS table, so it is right whatever the format is
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__global__ void k_fp8_table(float* out)
{
  const uint32_t c = threadIdx.x;                              // 256 threads
  const f32x2_t v = __builtin_amdgcn_cvt_pk_f32_fp8((int)(c | (c << 8)), false);
  out[c] = v.x;
}

// MODE 4 / 5 (W = 4 only): FP8 planes, two per convert (see the head of the file);
// MODE 0: W children sorted by entry distance;  MODE 1 (W = 8 only): octant slots, walked in order of (slot ^ ray octant), no sort;
// MODE 2 (W = 4 only): MODE 0 on the treelet-pair numbering of collapse4_pairs
template <int W, int MODE, bool COUNT, int LDS_STACK>
__global__ __launch_bounds__(kBlock) void k_trace(const float4* __restrict__ nodes, const float4* __restrict__ tris, const float4* __restrict__ rays,
                                                    float4* __restrict__ hits, uint32_t* __restrict__ cursor, uint32_t n, float4 gbox,
                                                    unsigned long long* __restrict__ counters)
{
  __shared__ uint32_t stk[LDS_STACK * kBlock];
  uint32_t* lds = &stk[threadIdx.x];
  uint32_t ovf[64];
  const uint32_t lane = threadIdx.x & 63u;
  const unsigned long long lt_mask = (1ull << lane) - 1ull;
  bool have = false; uint32_t cur = kDone, tag = 0; int sp = 0;
  float ox = 0, oy = 0, oz = 0, dx = 0, dy = 0, dz = 0, ix = 0, iy = 0, iz = 0, gx = 0, gy = 0, gz = 0, best = 0;
  float4 hit = make_float4(0, 0, 0, __int_as_float(-1));
  uint32_t oct = 0;
  uint32_t n_nodes = 0, n_tris = 0;
  uint32_t pool_next = 0, pool_end = 0; bool exhausted = false;
  constexpr uint32_t chunk = 256u;
  for (;;) {
    unsigned long long idle = __ballot(!have);
    if (!exhausted && (uint32_t)__popcll(idle) >= 12u) {
      for (int round = 0; round < 2 && idle != 0ull; ++round) {
        if (pool_next == pool_end) {
          uint32_t base = 0; if (lane == 0) base = atomicAdd(cursor, chunk); base = __shfl(base, 0);
          if (base >= n) { exhausted = true; break; }
          pool_next = base; pool_end = min(base + chunk, n);
        }
        const uint32_t take = min(pool_end - pool_next, (uint32_t)__popcll(idle));
        const uint32_t rank = (uint32_t)__popcll(idle & lt_mask);
        const bool mine = !have && ((idle >> lane) & 1ull) && rank < take;
        if (mine) {
          tag = pool_next + rank;
          const float4 a = ld_stream(rays + 2u * tag), b = ld_stream(rays + 2u * tag + 1u);
          ox = a.x; oy = a.y; oz = a.z; dx = b.x; dy = b.y; dz = b.z; best = a.w;
          ix = inv_dir(dx); iy = inv_dir(dy); iz = inv_dir(dz);
          const float R = fmaf(gbox.w, 3.0f, (fabsf(ox - gbox.x) + fabsf(oy - gbox.y)) + fabsf(oz - gbox.z)) * kSlabGuard;
          gx = fabsf(ix) * R; gy = fabsf(iy) * R; gz = fabsf(iz) * R;
          oct = (dx < 0.f ? 1u : 0u) | (dy < 0.f ? 2u : 0u) | (dz < 0.f ? 4u : 0u);
          sp = 0; cur = 0; have = true; hit = make_float4(a.w, 0.f, 0.f, __int_as_float(-1));
        }
        pool_next += take; idle &= ~__ballot(mine);
      }
    }
    if (__ballot(have) == 0ull) { if (exhausted) break; else continue; }

    auto pop = [&]() {
      if (sp == 0) { cur = kDone; return; }
      --sp; cur = sp < LDS_STACK ? lds[sp * kBlock] : ovf[sp - LDS_STACK];
    };
    auto push = [&](uint32_t v) { if (sp < LDS_STACK) lds[sp * kBlock] = v; else ovf[sp - LDS_STACK] = v; ++sp; };
#define QB(Wd, K) ((float)(((Wd) >> (8 * (K))) & 0xffu))
    auto inner_step = [&]() {
      const float4* np = nodes + (W == 4 ? 4u : 8u) * cur;
      const bool deep = MODE == 3 && cur >= 65536u;                   // MODE 3: nodes below the first 4 MB of the (breadth-first) array are loaded non-temporally
      const float4 n0 = deep ? ld_stream(np) : np[0];
      if (COUNT) ++n_nodes;
      const uint32_t ew = __float_as_uint(n0.w);
      const float ax = __uint_as_float((ew & 0xffu) << 23) * ix, ay = __uint_as_float(((ew >> 8) & 0xffu) << 23) * iy, az = __uint_as_float(((ew >> 16) & 0xffu) << 23) * iz;
      const float ddx = n0.x - ox, ddy = n0.y - oy, ddz = n0.z - oz;
      const bool sx = ix < 0.f, sy = iy < 0.f, sz = iz < 0.f;
      const f32x2 ax2 = {ax, ax}, ay2 = {ay, ay}, az2 = {az, az};
      const f32x2 bx2 = __builtin_elementwise_fma((f32x2){ddx, ddx}, (f32x2){ix, ix}, (f32x2){-gx, gx});
      const f32x2 by2 = __builtin_elementwise_fma((f32x2){ddy, ddy}, (f32x2){iy, iy}, (f32x2){-gy, gy});
      const f32x2 bz2 = __builtin_elementwise_fma((f32x2){ddz, ddz}, (f32x2){iz, iz}, (f32x2){-gz, gz});
#define CHILD(K, LX, HX, LY, HY, LZ, HZ, KEYBITS, VALID)                                                     \
      {                                                                                                     \
        const f32x2 tx = __builtin_elementwise_fma((f32x2){QB(LX, (K) & 3), QB(HX, (K) & 3)}, ax2, bx2);     \
        const f32x2 ty = __builtin_elementwise_fma((f32x2){QB(LY, (K) & 3), QB(HY, (K) & 3)}, ay2, by2);     \
        const f32x2 tz = __builtin_elementwise_fma((f32x2){QB(LZ, (K) & 3), QB(HZ, (K) & 3)}, az2, bz2);     \
        const float tmin = fmaxf(fmaxf(fmaxf(tx.x, ty.x), tz.x), 0.f);                                     \
        const float tmx = fminf(fminf(fminf(tx.y, ty.y), tz.y), best);                                     \
        const int bits = max(__float_as_int(tmin), 0);                                                     \
        key[K] = ((VALID) && tmin <= tmx) ? (((uint32_t)bits & ~(uint32_t)(KEYBITS)) | (uint32_t)(K)) : 0xFFFFFFFFu; \
      }
      if constexpr (W == 4 && (MODE == 4 || MODE == 5)) {
        // node: {origin | exponents, counts}, {x01 x23 y01 y23}, {z01 z23 child_base leaf_base}, MODE 4: {maximum corner | -}; a plane word holds
        // [lo_k, hi_k, lo_k+1, hi_k+1] as FP8 codes.  A ray that runs DOWN an axis swaps the bytes of every half word (one v_perm_b32 per word, where the byte
        // walk has one v_cndmask per word), so that a half word is always {near plane, far plane} and one packed convert feeds one packed fma.
        const float4 n1 = np[1], n2 = np[2];
        const uint32_t ni = (ew >> 24) & 7u, nch = (ew >> 28) & 7u;
        const uint32_t base_inner = __float_as_uint(n2.z), base_leaf = __float_as_uint(n2.w) - ni;
        const uint32_t px = sx ? 0x02030001u : 0x03020100u, py = sy ? 0x02030001u : 0x03020100u, pz = sz ? 0x02030001u : 0x03020100u;
        const uint32_t wx0 = __builtin_amdgcn_perm(0u, __float_as_uint(n1.x), px), wx1 = __builtin_amdgcn_perm(0u, __float_as_uint(n1.y), px);
        const uint32_t wy0 = __builtin_amdgcn_perm(0u, __float_as_uint(n1.z), py), wy1 = __builtin_amdgcn_perm(0u, __float_as_uint(n1.w), py);
        const uint32_t wz0 = __builtin_amdgcn_perm(0u, __float_as_uint(n2.x), pz), wz1 = __builtin_amdgcn_perm(0u, __float_as_uint(n2.y), pz);
        f32x2 Ax, Ay, Az, Bx, By, Bz;
        if constexpr (MODE == 5) { Ax = ax2; Ay = ay2; Az = az2; Bx = bx2; By = by2; Bz = bz2; }
        else {
          const float4 n3 = np[3];
          const float elx = ddx * ix, ely = ddy * iy, elz = ddz * iz, ehx = (n3.x - ox) * ix, ehy = (n3.y - oy) * iy, ehz = (n3.z - oz) * iz;
          // up an axis: {lower offset * A + entry at the minimum corner, upper offset * (-A) + exit at the maximum corner}; down an axis the pair is
          // {upper offset, lower offset}: {upper * (-A) + entry at the maximum corner, lower * A + exit at the minimum corner} -- A < 0 there
          Ax = sx ? (f32x2){-ax, ax} : (f32x2){ax, -ax}; Ay = sy ? (f32x2){-ay, ay} : (f32x2){ay, -ay}; Az = sz ? (f32x2){-az, az} : (f32x2){az, -az};
          Bx = sx ? (f32x2){ehx - gx, elx + gx} : (f32x2){elx - gx, ehx + gx};
          By = sy ? (f32x2){ehy - gy, ely + gy} : (f32x2){ely - gy, ehy + gy};
          Bz = sz ? (f32x2){ehz - gz, elz + gz} : (f32x2){elz - gz, ehz + gz};
        }
        uint32_t key[4];
#define CHILDP(K, WX, WY, WZ)                                                                               \
        {                                                                                                   \
          const f32x2 tx = __builtin_elementwise_fma(__builtin_amdgcn_cvt_pk_f32_fp8((int)(WX), ((K) & 1) != 0), Ax, Bx); \
          const f32x2 ty = __builtin_elementwise_fma(__builtin_amdgcn_cvt_pk_f32_fp8((int)(WY), ((K) & 1) != 0), Ay, By); \
          const f32x2 tz = __builtin_elementwise_fma(__builtin_amdgcn_cvt_pk_f32_fp8((int)(WZ), ((K) & 1) != 0), Az, Bz); \
          const float tmin = fmaxf(fmaxf(fmaxf(tx.x, ty.x), tz.x), 0.f);                                   \
          const float tmx = fminf(fminf(fminf(tx.y, ty.y), tz.y), best);                                   \
          const int bits = max(__float_as_int(tmin), 0);                                                   \
          key[K] = (((uint32_t)(K) < nch) && tmin <= tmx) ? (((uint32_t)bits & ~3u) | (uint32_t)(K)) : 0xFFFFFFFFu; \
        }
        CHILDP(0, wx0, wy0, wz0) CHILDP(1, wx0, wy0, wz0) CHILDP(2, wx1, wy1, wz1) CHILDP(3, wx1, wy1, wz1)
#undef CHILDP
        CE(key[0], key[1]) CE(key[2], key[3]) CE(key[0], key[2]) CE(key[1], key[3]) CE(key[1], key[2])
#define REF(KEY) (((((KEY) & 3u) < ni) ? base_inner : base_leaf) + ((KEY) & 3u))
        const uint32_t r0 = REF(key[0]), r1 = REF(key[1]), r2 = REF(key[2]), r3 = REF(key[3]);
#undef REF
        const int nh = 4 + ((((int)key[0] >> 31) + ((int)key[1] >> 31)) + (((int)key[2] >> 31) + ((int)key[3] >> 31)));
        if (__builtin_expect(sp <= LDS_STACK - 3, 1)) {
          uint32_t* top = lds + sp * kBlock;
          const int p1 = max(nh, 2) - 2, p2 = (nh == 3) ? 0 : 1, p3 = (nh == 4) ? 0 : 2;
          top[p3 * kBlock] = r3; top[p2 * kBlock] = r2; top[p1 * kBlock] = r1;
          sp += max(nh, 1) - 1;
        } else { if (nh == 4) push(r3); if (nh >= 3) push(r2); if (nh >= 2) push(r1); }
        if (nh >= 1) cur = r0; else pop();
      } else if constexpr (W == 4) {
        const float4 n1 = deep ? ld_stream(np + 1) : np[1], n2 = deep ? ld_stream(np + 2) : np[2];
        const uint32_t ni = (ew >> 24) & 7u, nch = (ew >> 28) & 7u;
        const uint32_t fav = (ew >> 27) & 1u;                                     // MODE 2: slot 0 lives in the other half of this node's line
        const uint32_t base_inner = MODE == 2 ? __float_as_uint(n2.z) - 2u * fav : __float_as_uint(n2.z), base_leaf = __float_as_uint(n2.w) - ni;
        const uint32_t self1 = cur + 1u;
        const uint32_t lx = __float_as_uint(sx ? n1.w : n1.x), ly = __float_as_uint(sy ? n2.x : n1.y), lz = __float_as_uint(sz ? n2.y : n1.z);
        const uint32_t hx = __float_as_uint(sx ? n1.x : n1.w), hy = __float_as_uint(sy ? n1.y : n2.x), hz = __float_as_uint(sz ? n1.z : n2.y);
        uint32_t key[4];
        CHILD(0, lx, hx, ly, hy, lz, hz, 3, 0u < nch) CHILD(1, lx, hx, ly, hy, lz, hz, 3, 1u < nch)
        CHILD(2, lx, hx, ly, hy, lz, hz, 3, 2u < nch) CHILD(3, lx, hx, ly, hy, lz, hz, 3, 3u < nch)
        CE(key[0], key[1]) CE(key[2], key[3]) CE(key[0], key[2]) CE(key[1], key[3]) CE(key[1], key[2])
#define REF(KEY) (MODE == 2 ? ((((KEY) & 3u) < ni) ? (((KEY) & 3u) < fav ? self1 : base_inner + 2u * ((KEY) & 3u)) : base_leaf + ((KEY) & 3u)) \
                           : (((((KEY) & 3u) < ni) ? base_inner : base_leaf) + ((KEY) & 3u)))
        const uint32_t r0 = REF(key[0]), r1 = REF(key[1]), r2 = REF(key[2]), r3 = REF(key[3]);
#undef REF
        const int nh = 4 + ((((int)key[0] >> 31) + ((int)key[1] >> 31)) + (((int)key[2] >> 31) + ((int)key[3] >> 31)));
        if (__builtin_expect(sp <= LDS_STACK - 3, 1)) {
          uint32_t* top = lds + sp * kBlock;
          const int p1 = max(nh, 2) - 2, p2 = (nh == 3) ? 0 : 1, p3 = (nh == 4) ? 0 : 2;
          top[p3 * kBlock] = r3; top[p2 * kBlock] = r2; top[p1 * kBlock] = r1;
          sp += max(nh, 1) - 1;
        } else { if (nh == 4) push(r3); if (nh >= 3) push(r2); if (nh >= 2) push(r1); }
        if (nh >= 1) cur = r0; else pop();
      } else {
        const float4 n1 = np[1], n2 = np[2], n3 = np[3], n4 = np[4];
        // n1 = lox0 lox1 loy0 loy1, n2 = loz0 loz1 hix0 hix1, n3 = hiy0 hiy1 hiz0 hiz1, n4 = base_inner base_leaf valid_mask -
        const uint32_t lx0 = __float_as_uint(sx ? n2.z : n1.x), lx1 = __float_as_uint(sx ? n2.w : n1.y), hx0 = __float_as_uint(sx ? n1.x : n2.z), hx1 = __float_as_uint(sx ? n1.y : n2.w);
        const uint32_t ly0 = __float_as_uint(sy ? n3.x : n1.z), ly1 = __float_as_uint(sy ? n3.y : n1.w), hy0 = __float_as_uint(sy ? n1.z : n3.x), hy1 = __float_as_uint(sy ? n1.w : n3.y);
        const uint32_t lz0 = __float_as_uint(sz ? n3.z : n2.x), lz1 = __float_as_uint(sz ? n3.w : n2.y), hz0 = __float_as_uint(sz ? n2.x : n3.z), hz1 = __float_as_uint(sz ? n2.y : n3.w);
        uint32_t key[8];
        if constexpr (MODE == 0) {
          const uint32_t ni = (ew >> 24) & 15u, nch = ew >> 28;
          const uint32_t base_inner = __float_as_uint(n4.x), base_leaf = __float_as_uint(n4.y) - ni;
          CHILD(0, lx0, hx0, ly0, hy0, lz0, hz0, 7, 0u < nch) CHILD(1, lx0, hx0, ly0, hy0, lz0, hz0, 7, 1u < nch)
          CHILD(2, lx0, hx0, ly0, hy0, lz0, hz0, 7, 2u < nch) CHILD(3, lx0, hx0, ly0, hy0, lz0, hz0, 7, 3u < nch)
          CHILD(4, lx1, hx1, ly1, hy1, lz1, hz1, 7, 4u < nch) CHILD(5, lx1, hx1, ly1, hy1, lz1, hz1, 7, 5u < nch)
          CHILD(6, lx1, hx1, ly1, hy1, lz1, hz1, 7, 6u < nch) CHILD(7, lx1, hx1, ly1, hy1, lz1, hz1, 7, 7u < nch)
          CE(key[0], key[1]) CE(key[2], key[3]) CE(key[4], key[5]) CE(key[6], key[7])
          CE(key[0], key[2]) CE(key[1], key[3]) CE(key[4], key[6]) CE(key[5], key[7])
          CE(key[1], key[2]) CE(key[5], key[6])
          CE(key[0], key[4]) CE(key[1], key[5]) CE(key[2], key[6]) CE(key[3], key[7])
          CE(key[2], key[4]) CE(key[3], key[5])
          CE(key[1], key[2]) CE(key[3], key[4]) CE(key[5], key[6])
          int nh = 8;
#pragma unroll
          for (int j = 0; j < 8; ++j) nh += (int)key[j] >> 31;
#define REF(KEY) (((((KEY) & 7u) < ni) ? base_inner : base_leaf) + ((KEY) & 7u))
          if (__builtin_expect(sp <= LDS_STACK - 7, 1)) {
            // seven unconditional stores: hit child j (1 <= j < nh) lands at sp + nh-1-j, the others in the dead slots above the new top
            uint32_t* top = lds + sp * kBlock;
#pragma unroll
            for (int j = 7; j >= 1; --j) top[((nh - 1 - j) & 7) * kBlock] = REF(key[j]);
            sp += max(nh, 1) - 1;
          } else {
#pragma unroll
            for (int j = 7; j >= 1; --j) if (j < nh) push(REF(key[j]));
          }
          if (nh >= 1) cur = REF(key[0]); else pop();
#undef REF
        } else {
          // octant slots: visiting order = increasing (slot ^ oct') where oct' makes the near corner slot 0; no distance sort.
          // rank of an inner slot among the inner slots / of a leaf among the leaves gives the implicit reference.
          const uint32_t inner_mask = ew >> 24, valid = __float_as_uint(n4.z);
          const uint32_t base_inner = __float_as_uint(n4.x), base_leaf = __float_as_uint(n4.y);
          CHILD(0, lx0, hx0, ly0, hy0, lz0, hz0, 0, true) CHILD(1, lx0, hx0, ly0, hy0, lz0, hz0, 0, true)
          CHILD(2, lx0, hx0, ly0, hy0, lz0, hz0, 0, true) CHILD(3, lx0, hx0, ly0, hy0, lz0, hz0, 0, true)
          CHILD(4, lx1, hx1, ly1, hy1, lz1, hz1, 0, true) CHILD(5, lx1, hx1, ly1, hy1, lz1, hz1, 0, true)
          CHILD(6, lx1, hx1, ly1, hy1, lz1, hz1, 0, true) CHILD(7, lx1, hx1, ly1, hy1, lz1, hz1, 0, true)
          uint32_t hitm = 0;
#pragma unroll
          for (int k = 0; k < 8; ++k) hitm |= (key[k] != 0xFFFFFFFFu ? 1u : 0u) << k;
          hitm &= valid;
          // far .. near: order index q = 7 .. 0, slot = q ^ oct
          uint32_t first = kDone; int cnt = 0;
          const int nh = __popc(hitm);
#pragma unroll
          for (int q = 7; q >= 0; --q) {
            const uint32_t s = (uint32_t)q ^ oct;
            if ((hitm >> s) & 1u) {
              const uint32_t below = (1u << s) - 1u;
              const uint32_t ref = ((inner_mask >> s) & 1u) ? base_inner + (uint32_t)__popc(inner_mask & below) : base_leaf + (uint32_t)__popc(valid & ~inner_mask & below);
              ++cnt;
              if (cnt == nh) first = ref; else push(ref);
            }
          }
          if (nh >= 1) cur = first; else pop();
        }
      }
#undef CHILD
    };
#pragma unroll 1
    for (int step_ = 0; step_ < 2 && have && !(cur & kLeaf); ++step_) inner_step();
    if (have && (cur & kLeaf) && cur != kDone) {
      const uint32_t ti = cur & 0x0FFFFFFFu;
      const float4* tp = tris + 4u * ti;
      const float4 a = tp[0], b = tp[1], c = tp[2];
      if (COUNT) ++n_tris;
      const float e0x = b.x - a.x, e0y = b.y - a.y, e0z = b.z - a.z, e1x = a.x - c.x, e1y = a.y - c.y, e1z = a.z - c.z;
      const float nx = e1y * e0z - e1z * e0y, ny = e1z * e0x - e1x * e0z, nz = e1x * e0y - e1y * e0x;
      const float tox = a.x - ox, toy = a.y - oy, toz = a.z - oz;
      const float inv = 1.0f / (nx * dx + ny * dy + nz * dz);
      const float vx = dy * toz - dz * toy, vy = dz * tox - dx * toz, vz = dx * toy - dy * tox;
      const float tt = (nx * tox + ny * toy + nz * toz) * inv, uu = (vx * e1x + vy * e1y + vz * e1z) * inv, vv = (vx * e0x + vy * e0y + vz * e0z) * inv;
      if (tt >= 0.f && uu >= 0.f && vv >= 0.f && (uu + vv) <= 1.0f && tt < best) { best = tt; hit = make_float4(tt, uu, vv, __int_as_float((int)ti)); }
      pop();
    }
    if (have && cur == kDone) { st_stream(hits + tag, hit); have = false; }
  }
  if (COUNT) { atomicAdd(&counters[0], (unsigned long long)n_nodes); atomicAdd(&counters[1], (unsigned long long)n_tris); }
}


// the 4-wide byte-plane tree with its planes re-coded as FP8 (decode table `tab` read from the device): corner = true -> MODE 4, false -> MODE 5
static Wide to_fp8(const Wide& w4, const float* tab, bool corner, double* grow_out)
{
  // every finite non-negative code, ascending by value
  std::vector<std::pair<float, uint32_t>> codes;
  for (uint32_t c = 0; c < 128; ++c) if (std::isfinite(tab[c]) && tab[c] >= 0.f) codes.push_back({tab[c], c});
  std::sort(codes.begin(), codes.end());
  auto enc_floor = [&](float x) { uint32_t best = codes[0].second; for (auto& pc : codes) { if (pc.first <= x) best = pc.second; else break; } return best; };
  auto enc_ceil = [&](float x) { for (auto& pc : codes) if (pc.first >= x) return pc.second; fprintf(stderr, "fp8: %g has no code above it\n", x); exit(1); return 0u; };
  Wide out = w4;
  double grow = 0; uint64_t boxes = 0;
  for (uint32_t i = 0; i < w4.n_nodes; ++i) {
    const uint32_t* o = &w4.words[(size_t)i * 16]; uint32_t* q = &out.words[(size_t)i * 16];
    const uint32_t ns = (o[3] >> 28) & 7u;
    if (ns == 0) continue;
    float org[3]; memcpy(org, o, 12);
    uint32_t planes[6] = {0, 0, 0, 0, 0, 0};
    for (int a = 0; a < 3; ++a) {
      const float step = crh_quant_step((o[3] >> (8 * a)) & 0xffu);
      uint32_t ext = 0; for (uint32_t k = 0; k < ns; ++k) ext = std::max(ext, (o[7 + a] >> (8 * k)) & 0xffu);
      if (corner) { float hc = org[a] + (float)ext * step; q[12 + a] = __builtin_bit_cast(uint32_t, hc); }
      for (uint32_t k = 0; k < 4; ++k) {
        uint32_t cl = 0, ch = 0;
        if (k < ns) {
          const uint32_t ql = (o[4 + a] >> (8 * k)) & 0xffu, qh = (o[7 + a] >> (8 * k)) & 0xffu;
          cl = enc_floor((float)ql);
          ch = corner ? enc_floor((float)(ext - qh)) : enc_ceil((float)qh);
          const float lo = tab[cl], hi = corner ? (float)ext - tab[ch] : tab[ch];
          if (qh > ql) { grow += (double)(hi - lo) / (double)(qh - ql); ++boxes; }
        }
        planes[2 * a + (k >> 1)] |= (cl | (ch << 8)) << (16 * (k & 1));
      }
    }
    for (int w = 0; w < 6; ++w) q[4 + w] = planes[w];
  }
  if (grow_out) *grow_out = boxes ? grow / (double)boxes : 0.0;
  return out;
}

// bounds of (triangle ∩ box): Sutherland-Hodgman against the six planes (double precision; experiment only)
static bool clip_bounds(const float* tri, const float* blo, const float* bhi, float* olo, float* ohi)
{
  double poly[16][3], tmp[16][3]; int np = 3;
  for (int k = 0; k < 3; ++k) for (int a = 0; a < 3; ++a) poly[k][a] = tri[3 * k + a];
  for (int a = 0; a < 3; ++a) for (int side = 0; side < 2; ++side) {
    const double pl = side ? bhi[a] : blo[a]; int nq = 0;
    for (int k = 0; k < np; ++k) { const double* A = poly[k]; const double* Bq = poly[(k + 1) % np];
      const bool ia = side ? A[a] <= pl : A[a] >= pl, ib = side ? Bq[a] <= pl : Bq[a] >= pl;
      if (ia) { for (int c = 0; c < 3; ++c) tmp[nq][c] = A[c]; ++nq; }
      if (ia != ib) { const double t = (pl - A[a]) / (Bq[a] - A[a]); for (int c = 0; c < 3; ++c) tmp[nq][c] = A[c] + t * (Bq[c] - A[c]); tmp[nq][a] = pl; ++nq; } }
    np = nq; if (np == 0) return false;
    for (int k = 0; k < np; ++k) for (int c = 0; c < 3; ++c) poly[k][c] = tmp[k][c];
  }
  for (int a = 0; a < 3; ++a) { double lo = 1e300, hi = -1e300; for (int k = 0; k < np; ++k) { lo = std::min(lo, poly[k][a]); hi = std::max(hi, poly[k][a]); }
    olo[a] = std::max((float)blo[a], std::nextafterf((float)lo, -3e38f)); ohi[a] = std::min((float)bhi[a], std::nextafterf((float)hi, 3e38f)); }
  return true;
}

// ------------------------------------------------------------------------------------------------ main
template <int W, int MODE, int LDS_STACK>
static void run(const char* name, const Wide& wd, const float4* d_tris, const float4* d_rays, uint32_t n_rays, float4 gbox, int waves, std::vector<float>& t_out)
{
  float4* d_nodes; float4* d_hits; uint32_t* d_cursor; unsigned long long* d_cnt;
  HIPCHECK(hipMalloc(&d_nodes, wd.words.size() * 4)); HIPCHECK(hipMemcpy(d_nodes, wd.words.data(), wd.words.size() * 4, hipMemcpyHostToDevice));
  HIPCHECK(hipMalloc(&d_hits, (size_t)n_rays * 16)); HIPCHECK(hipMalloc(&d_cursor, 4)); HIPCHECK(hipMalloc(&d_cnt, 16)); HIPCHECK(hipMemset(d_cnt, 0, 16));
  hipDeviceProp_t prop; HIPCHECK(hipGetDeviceProperties(&prop, 0));
  int per_cu = 0; HIPCHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_trace<W, MODE, false, LDS_STACK>, kBlock, 0));
  const int wg_per_cu = std::min(per_cu, waves);
  const uint32_t grid = (uint32_t)(prop.multiProcessorCount * wg_per_cu);
  hipEvent_t e0, e1; HIPCHECK(hipEventCreate(&e0)); HIPCHECK(hipEventCreate(&e1));
  float best_ms = 1e30f, sum_ms = 0.f; const int reps = 5;
  for (int r = -1; r < reps; ++r) {
    HIPCHECK(hipMemset(d_cursor, 0, 4));
    HIPCHECK(hipEventRecord(e0));
    k_trace<W, MODE, false, LDS_STACK><<<grid, kBlock>>>(d_nodes, d_tris, d_rays, d_hits, d_cursor, n_rays, gbox, d_cnt);
    HIPCHECK(hipEventRecord(e1)); HIPCHECK(hipEventSynchronize(e1));
    float ms; HIPCHECK(hipEventElapsedTime(&ms, e0, e1));
    if (r >= 0) { best_ms = std::min(best_ms, ms); sum_ms += ms; }
  }
  std::vector<float4> h(n_rays); HIPCHECK(hipMemcpy(h.data(), d_hits, (size_t)n_rays * 16, hipMemcpyDeviceToHost));
  t_out.resize(n_rays); uint64_t nhit = 0; for (uint32_t i = 0; i < n_rays; ++i) { t_out[i] = h[i].x; nhit += __builtin_bit_cast(int, h[i].w) >= 0; }
  HIPCHECK(hipMemset(d_cursor, 0, 4));
  k_trace<W, MODE, true, LDS_STACK><<<grid, kBlock>>>(d_nodes, d_tris, d_rays, d_hits, d_cursor, n_rays, gbox, d_cnt);
  unsigned long long cnt[2]; HIPCHECK(hipMemcpy(cnt, d_cnt, 16, hipMemcpyDeviceToHost));
  const double avg = sum_ms / reps, vis = (double)cnt[0] / n_rays, tr = (double)cnt[1] / n_rays;
  const double node_b = W == 4 ? (MODE == 4 ? 64.0 : 48.0) : 72.0, line_b = W == 4 ? 64.0 : 128.0;
  printf("%-34s nodes %9u (%.1f MB, fill %.2f/%d)  occupancy %d WG/CU (max %d)  avg %.3f ms  best %.3f ms  %.0f Mrays/s  visits/ray %.2f  tris/ray %.2f  hit %.3f  "
         "fetched B/ray %.0f  sectors B/ray %.0f\n", name, wd.n_nodes, wd.words.size() * 4 / 1e6, wd.fill, W, wg_per_cu, per_cu, avg, best_ms, n_rays / avg * 1e-3, vis, tr,
         (double)nhit / n_rays, vis * node_b + tr * 48.0 + 48.0, vis * line_b + tr * 64.0 + 48.0);
  fflush(stdout);
  HIPCHECK(hipFree(d_nodes)); HIPCHECK(hipFree(d_hits)); HIPCHECK(hipFree(d_cursor)); HIPCHECK(hipFree(d_cnt));
}

int main(int argc, char** argv)
{
  const uint32_t n = argc > 1 ? (uint32_t)atoi(argv[1]) : 1000000u;
  const uint32_t n_rays = argc > 2 ? (uint32_t)atoi(argv[2]) : (16u << 20);
  const int waves4 = argc > 3 ? atoi(argv[3]) : 6, waves8 = argc > 4 ? atoi(argv[4]) : 6, presplit = argc > 5 ? atoi(argv[5]) : 0;
  { // 0-1 principle check of the 19-comparator network
    for (uint32_t m = 0; m < 256; ++m) { uint32_t key[8]; for (int k = 0; k < 8; ++k) key[k] = (m >> k) & 1u;
#define HCE(a, b) { uint32_t lo_ = std::min(a, b), hi_ = std::max(a, b); a = lo_; b = hi_; }
      HCE(key[0], key[1]) HCE(key[2], key[3]) HCE(key[4], key[5]) HCE(key[6], key[7]) HCE(key[0], key[2]) HCE(key[1], key[3]) HCE(key[4], key[6]) HCE(key[5], key[7])
      HCE(key[1], key[2]) HCE(key[5], key[6]) HCE(key[0], key[4]) HCE(key[1], key[5]) HCE(key[2], key[6]) HCE(key[3], key[7]) HCE(key[2], key[4]) HCE(key[3], key[5])
      HCE(key[1], key[2]) HCE(key[3], key[4]) HCE(key[5], key[6])
      for (int k = 0; k < 7; ++k) if (key[k] > key[k + 1]) { fprintf(stderr, "sort network wrong\n"); return 1; } }
  }
  // the benchmark's triangle soup (centre U([-1,1]^3), edges U([-1,1]^3) * 1.5 n^(-1/3)); not the same random stream
  std::vector<float> pos((size_t)9 * n), pb((size_t)6 * n), cen((size_t)3 * n);
  const float r = 1.5f * powf((float)n, -1.f / 3.f);
  for (uint32_t t = 0; t < n; ++t) {
    float* p = &pos[(size_t)9 * t];
    for (int a = 0; a < 3; ++a) p[a] = urand() * 2.f - 1.f;
    for (int k = 1; k < 3; ++k) for (int a = 0; a < 3; ++a) p[3 * k + a] = p[a] + (urand() * 2.f - 1.f) * r;
    for (int a = 0; a < 3; ++a) { const float lo = std::min(p[a], std::min(p[3 + a], p[6 + a])), hi = std::max(p[a], std::max(p[3 + a], p[6 + a]));
      pb[(size_t)6 * t + a] = lo; pb[(size_t)6 * t + 3 + a] = hi; cen[(size_t)3 * t + a] = 0.5f * (lo + hi); }
  }
  auto t0 = std::chrono::steady_clock::now();
  Build B; B.pb = pb.data(); B.cen = cen.data(); B.idx.resize(n); for (uint32_t t = 0; t < n; ++t) B.idx[t] = t; B.bn.reserve((size_t)2 * n);
  B.rec(0, n);
  fprintf(stderr, "binary tree: %zu nodes, %.1f s\n", B.bn.size(), std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
  const Wide w4 = collapse(B.bn, 4, false), w8 = collapse(B.bn, 8, false), w8o = collapse(B.bn, 8, true);
  // rays: origin just off a random triangle, uniformly random direction (what bounces >= 2 of the path tracer look like)
  std::vector<float> rays((size_t)8 * n_rays);
  for (uint32_t i = 0; i < n_rays; ++i) {
    const uint32_t t = (uint32_t)(splitmix() % n); float dx, dy, dz, l;
    do { dx = urand() * 2.f - 1.f; dy = urand() * 2.f - 1.f; dz = urand() * 2.f - 1.f; l = dx * dx + dy * dy + dz * dz; } while (l > 1.f || l < 1e-4f);
    l = sqrtf(l); float* q = &rays[(size_t)8 * i];
    q[0] = pos[(size_t)9 * t] + 1e-4f * dx; q[1] = pos[(size_t)9 * t + 1] + 1e-4f * dy; q[2] = pos[(size_t)9 * t + 2] + 1e-4f * dz; q[3] = 3.0e38f;
    q[4] = dx / l; q[5] = dy / l; q[6] = dz / l; q[7] = 0.f;
  }
  float4* d_rays; HIPCHECK(hipMalloc(&d_rays, rays.size() * 4)); HIPCHECK(hipMemcpy(d_rays, rays.data(), rays.size() * 4, hipMemcpyHostToDevice));
  float4 gbox = make_float4(0.5f * (B.bn[0].mn[0] + B.bn[0].mx[0]), 0.5f * (B.bn[0].mn[1] + B.bn[0].mx[1]), 0.5f * (B.bn[0].mn[2] + B.bn[0].mx[2]),
                            0.5f * ((B.bn[0].mx[0] - B.bn[0].mn[0]) + (B.bn[0].mx[1] - B.bn[0].mn[1]) + (B.bn[0].mx[2] - B.bn[0].mn[2])));
  const std::vector<uint32_t>* piece_tri = nullptr;                  // pre-split run: leaf primitive = piece of a triangle
  auto upload_tris = [&](const Wide& w) { const size_t nl = w.leaf_prims.size(); std::vector<float> tr((size_t)16 * nl, 0.f);
    for (size_t k = 0; k < nl; ++k) { const uint32_t t_ = piece_tri ? (*piece_tri)[w.leaf_prims[k]] : w.leaf_prims[k];
      const float* p = &pos[(size_t)9 * t_]; float* o = &tr[(size_t)16 * k];
      o[0] = p[0]; o[1] = p[1]; o[2] = p[2]; o[4] = p[3]; o[5] = p[4]; o[6] = p[5]; o[8] = p[6]; o[9] = p[7]; o[10] = p[8]; }
    float4* d; HIPCHECK(hipMalloc(&d, tr.size() * 4)); HIPCHECK(hipMemcpy(d, tr.data(), tr.size() * 4, hipMemcpyHostToDevice)); return d; };
  printf("%u triangles, %u rays (origin on a random triangle, random direction)\n", n, n_rays);
  std::vector<float> t4, t8, t8o;
  float4* d_t = upload_tris(w4);  run<4, 0, 16>("4-wide 64-B node, sorted", w4, d_t, d_rays, n_rays, gbox, waves4, t4); HIPCHECK(hipFree(d_t));
  {
    float* d_tab; float tab[256]; HIPCHECK(hipMalloc(&d_tab, 1024)); k_fp8_table<<<1, 256>>>(d_tab); HIPCHECK(hipMemcpy(tab, d_tab, 1024, hipMemcpyDeviceToHost)); HIPCHECK(hipFree(d_tab));
    printf("FP8 decode as this GPU does it: code 0x08 -> %g, 0x38 -> %g, 0x40 -> %g, 0x78 -> %g, largest finite %g\n", tab[0x08], tab[0x38], tab[0x40], tab[0x78],
           *std::max_element(tab, tab + 128, [](float a, float b) { return (std::isfinite(a) ? a : -1.f) < (std::isfinite(b) ? b : -1.f); }));
    double g5 = 0, g4 = 0;
    const Wide w5 = to_fp8(w4, tab, false, &g5), w4c = to_fp8(w4, tab, true, &g4);
    printf("mean child-box edge against the byte planes: FP8 from the minimum corner x %.4f, corner-relative x %.4f\n", g5, g4);
    std::vector<float> t5, t4c;
    d_t = upload_tris(w5);  run<4, 5, 16>("4-wide, FP8 planes (min corner)", w5, d_t, d_rays, n_rays, gbox, waves4, t5); HIPCHECK(hipFree(d_t));
    d_t = upload_tris(w4c); run<4, 4, 16>("4-wide, FP8 planes (corner-relative)", w4c, d_t, d_rays, n_rays, gbox, waves4, t4c); HIPCHECK(hipFree(d_t));
    uint64_t b5 = 0, b4 = 0; for (uint32_t i = 0; i < n_rays; ++i) { b5 += memcmp(&t4[i], &t5[i], 4) != 0; b4 += memcmp(&t4[i], &t4c[i], 4) != 0; }
    printf("hit distances differing from the byte-plane walk: FP8 min-corner %llu, FP8 corner-relative %llu\n", (unsigned long long)b5, (unsigned long long)b4);
    if (b5 || b4) return 2;
    // the same walk on the byte planes once more, so that drift between the first and the last measurement of this process shows
    std::vector<float> t4b; d_t = upload_tris(w4); run<4, 0, 16>("4-wide 64-B node, sorted (again)", w4, d_t, d_rays, n_rays, gbox, waves4, t4b); HIPCHECK(hipFree(d_t));
  }
  if (getenv("WIDE8_ONLY_FP8")) return 0;
  d_t = upload_tris(w8);          run<8, 0, 24>("8-wide 128-B node, sorted", w8, d_t, d_rays, n_rays, gbox, waves8, t8); HIPCHECK(hipFree(d_t));
  d_t = upload_tris(w8o);         run<8, 1, 24>("8-wide 128-B node, octant order", w8o, d_t, d_rays, n_rays, gbox, waves8, t8o); HIPCHECK(hipFree(d_t));
  if (presplit > 0) {
    // early split clipping: every triangle's box is cut `presplit` times at the middle of its longest axis; a piece keeps the
    // bounds of the clipped triangle.  Leaves = pieces (the triangle record is duplicated); traversal unchanged.
    std::vector<float> ppb, pcen; std::vector<uint32_t> ptri;
    struct Piece { float lo[3], hi[3]; };
    for (uint32_t t = 0; t < n; ++t) {
      std::vector<Piece> cur(1), nxt; for (int a = 0; a < 3; ++a) { cur[0].lo[a] = pb[(size_t)6 * t + a]; cur[0].hi[a] = pb[(size_t)6 * t + 3 + a]; }
      for (int lv = 0; lv < presplit; ++lv) { nxt.clear();
        for (const Piece& pc : cur) { int ax = 0; for (int a = 1; a < 3; ++a) if (pc.hi[a] - pc.lo[a] > pc.hi[ax] - pc.lo[ax]) ax = a;
          const float mid = 0.5f * (pc.lo[ax] + pc.hi[ax]);
          Piece l = pc, r2 = pc; l.hi[ax] = mid; r2.lo[ax] = mid; Piece o;
          if (clip_bounds(&pos[(size_t)9 * t], l.lo, l.hi, o.lo, o.hi)) nxt.push_back(o);
          if (clip_bounds(&pos[(size_t)9 * t], r2.lo, r2.hi, o.lo, o.hi)) nxt.push_back(o); }
        cur.swap(nxt); }
      for (const Piece& pc : cur) { for (int a = 0; a < 3; ++a) ppb.push_back(pc.lo[a]); for (int a = 0; a < 3; ++a) ppb.push_back(pc.hi[a]);
        for (int a = 0; a < 3; ++a) pcen.push_back(0.5f * (pc.lo[a] + pc.hi[a])); ptri.push_back(t); }
    }
    const uint32_t n2 = (uint32_t)ptri.size();
    Build B2; B2.pb = ppb.data(); B2.cen = pcen.data(); B2.idx.resize(n2); for (uint32_t t = 0; t < n2; ++t) B2.idx[t] = t; B2.bn.reserve((size_t)2 * n2);
    B2.rec(0, n2);
    const Wide w4s = collapse(B2.bn, 4, false); std::vector<float> t4s; piece_tri = &ptri;
    char nm[64]; snprintf(nm, sizeof nm, "4-wide, %d x pre-split (%.2f refs/tri)", presplit, (double)n2 / n);
    d_t = upload_tris(w4s); run<4, 0, 16>(nm, w4s, d_t, d_rays, n_rays, gbox, waves4, t4s); HIPCHECK(hipFree(d_t)); piece_tri = nullptr;
    uint64_t bad = 0; for (uint32_t i = 0; i < n_rays; ++i) bad += memcmp(&t4[i], &t4s[i], 4) != 0;
    printf("hit distances differing from the 4-wide walk: pre-split %llu\n", (unsigned long long)bad);
  }
  { std::vector<float> t4n; d_t = upload_tris(w4); run<4, 3, 16>("4-wide, deep nodes non-temporal", w4, d_t, d_rays, n_rays, gbox, waves4, t4n); HIPCHECK(hipFree(d_t)); }
  { const Wide w4p = collapse4_pairs(B.bn); std::vector<float> t4p;
    d_t = upload_tris(w4p); run<4, 2, 16>("4-wide, parent + favourite child / line", w4p, d_t, d_rays, n_rays, gbox, waves4, t4p); HIPCHECK(hipFree(d_t));
    uint64_t bad = 0; for (uint32_t i = 0; i < n_rays; ++i) bad += memcmp(&t4[i], &t4p[i], 4) != 0;
    printf("hit distances differing from the 4-wide walk: pair layout %llu\n", (unsigned long long)bad); if (bad) return 2; }
  uint64_t bad8 = 0, bad8o = 0;
  for (uint32_t i = 0; i < n_rays; ++i) { bad8 += memcmp(&t4[i], &t8[i], 4) != 0; bad8o += memcmp(&t4[i], &t8o[i], 4) != 0; }
  printf("hit distances differing from the 4-wide walk: 8-wide sorted %llu, 8-wide octant %llu of %u\n", (unsigned long long)bad8, (unsigned long long)bad8o, n_rays);
  return bad8 || bad8o ? 2 : 0;
}

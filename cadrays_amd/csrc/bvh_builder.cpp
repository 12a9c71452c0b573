// bvh_builder.cpp -- parallel binned-SAH builder + 4-wide collapse (host, C++17).
//
// Build rules (DESIGN.md "BVH"): triangle box centre = (min+max)*0.5; at each node all three axes are
// binned in one pass over SoA primitive arrays (32 bins, bin = int(((c - cmin) / extent) * 32), clamped);
// the cheapest split by  area(L)*n(L) + area(R)*n(R)  wins, ties to the lower axis, then the lower bin;
// the partition is stable; leaves hold ONE triangle; when the remaining depth budget is only enough
// for balanced splitting, split at the object median of the widest centroid axis (total order by
// (centre, index)).  The binary tree is then collapsed by replacing children with grandchildren; the inner
// children of a node get consecutive indices and the triangles of its leaf children consecutive positions, so
// every 4-wide node packs into 48 bytes with 8-bit child bounds and two base references (crh_bvh_format.h).
// Large subtrees are built by separate threads; the result does not depend on the thread count because every
// subtree owns a disjoint index range and numbering happens afterwards.
#include "bvh_builder.h"
#include <sched.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../../include/crh_bvh_format.h"
#include "../../include/crh_xform.h"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <atomic>
#include <cstring>
#include <future>
#include <thread>

namespace crh {
namespace {

inline float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

struct Box {
  float mn[3], mx[3];
  void clear() { for (int a = 0; a < 3; ++a) { mn[a] = 3.0e38f; mx[a] = -3.0e38f; } }
  void grow(const Box& o) { for (int a = 0; a < 3; ++a) { mn[a] = std::min(mn[a], o.mn[a]); mx[a] = std::max(mx[a], o.mx[a]); } }
  float half_area() const {
    float dx = mx[0] - mn[0], dy = mx[1] - mn[1], dz = mx[2] - mn[2];
    return fma_(dx, dy, fma_(dy, dz, dz * dx));
  }
};

struct BNode { Box box; uint32_t lo, hi; int32_t left, right; };

struct Builder {
  // SoA primitive data
  std::vector<float> pmn[3], pmx[3], cen[3];
  std::vector<uint32_t> idx, tmp;
  std::vector<BNode> nodes;
  std::atomic<uint32_t> next{0};
  std::atomic<int> spare_threads{0};
  uint32_t leaf_max = kLeafSize;       // one primitive per leaf (triangle trees and the top-level tree over instances alike)
  uint32_t task_min = 32768;           // a left half larger than this may be handed to another thread (smaller for small trees: build_tree)

  uint32_t alloc() { return next.fetch_add(1, std::memory_order_relaxed); }

  Box prim_box(uint32_t p) const {
    Box b; for (int a = 0; a < 3; ++a) { b.mn[a] = pmn[a][p]; b.mx[a] = pmx[a][p]; } return b;
  }

  static int ceil_log2(uint32_t v) { int l = 0; uint32_t p = 1; while (p < v) { p <<= 1; ++l; } return l; }

  // ---- parallelism inside one large node (the top levels of the tree would otherwise be serial) -------------------
  static constexpr uint32_t kParMin = 65536;      // primitives below which a node is processed by one thread
  static constexpr int kMaxChunks = 64;
  int grab_threads(uint32_t n) {                  // extra threads for a node of n primitives (0 = stay serial)
    if (n < kParMin) return 0;
    const int want = (int)std::min<uint32_t>(n / (kParMin / 2), kMaxChunks) - 1;
    int have = spare_threads.load();
    while (have > 0) {
      const int take = std::min(have, want);
      if (spare_threads.compare_exchange_weak(have, have - take)) return take;
    }
    return 0;
  }
  template <class F> static void run_chunks(int chunks, F f) {   // f(chunk) on `chunks` threads, caller included
    std::vector<std::thread> th;
    for (int c = 1; c < chunks; ++c) th.emplace_back(f, c);
    f(0);
    for (auto& t : th) t.join();
  }

  // returns the split position, reorders idx[lo,hi)
  uint32_t split(uint32_t lo, uint32_t hi, int depth, const float cmn[3], const float cmx[3], int extra) {
    const uint32_t n = hi - lo;
    const int chunks = extra + 1;
    auto c_lo = [&](int c) { return lo + (uint32_t)((uint64_t)n * c / chunks); };
    const int need = ceil_log2((n + leaf_max - 1) / leaf_max);
    const bool force_median = depth + need >= kMaxDepth;
    if (!force_median && chunks == 1) {
      // One thread, typically a small node (most nodes of a tree hold a handful of primitives): bins are cleared on first touch
      // and the sweep visits only OCCUPIED bins.  Same result as the full sweep below: a split after an empty bin s has the
      // same two sides -- hence bit-identical cost -- as the split after the previous bin, and only a strictly lower cost
      // replaces the best one, so an empty bin never wins.  (1 000 boxes: 0.92 -> 0.3 ms; this is crh_set_transforms' per-frame cost.)
      float ext[3]; bool ok[3];
      for (int a = 0; a < 3; ++a) { ext[a] = cmx[a] - cmn[a]; ok[a] = ext[a] > 0.f; }
      uint32_t cnt[3][kBins]; Box bb[3][kBins]; uint32_t occ[3] = {0u, 0u, 0u};
      static_assert(kBins == 32, "occupancy masks are one 32-bit word per axis");
      for (uint32_t i = lo; i < hi; ++i) {
        const uint32_t p = idx[i];
        const Box pb = prim_box(p);
        for (int a = 0; a < 3; ++a) {
          if (!ok[a]) continue;
          int b = (int)(((cen[a][p] - cmn[a]) / ext[a]) * (float)kBins);
          if (b > kBins - 1) b = kBins - 1;
          if (!(occ[a] >> b & 1u)) { occ[a] |= 1u << b; cnt[a][b] = 0; bb[a][b].clear(); }
          cnt[a][b]++; bb[a][b].grow(pb);
        }
      }
      float best = 3.0e38f; int baxis = -1, bsplit = -1;
      for (int a = 0; a < 3; ++a) {
        if (!ok[a]) continue;
        // suffix (area, count) at every occupied bin, highest first
        float rarea[kBins]; uint32_t rcnt[kBins]; int next_occ[kBins];
        Box acc; acc.clear(); uint32_t c = 0; int above = -1;
        for (uint32_t m = occ[a]; m;) {
          const int b = 31 - __builtin_clz(m); m &= ~(1u << b);
          next_occ[b] = above;
          acc.grow(bb[a][b]); c += cnt[a][b]; rcnt[b] = c; rarea[b] = acc.half_area();
          above = b;
        }
        acc.clear(); c = 0;
        for (uint32_t m = occ[a]; m;) {
          const int sbin = __builtin_ctz(m); m &= m - 1u;
          acc.grow(bb[a][sbin]); c += cnt[a][sbin];
          const int nb = next_occ[sbin];
          if (nb < 0 || sbin >= kBins - 1) continue;            // nothing on the right of this split
          const float cost = fma_(acc.half_area(), (float)c, rarea[nb] * (float)rcnt[nb]);
          if (cost < best) { best = cost; baxis = a; bsplit = sbin; }
        }
      }
      if (baxis >= 0) {
        const std::vector<float>& cc = cen[baxis];
        const float c0 = cmn[baxis], e = ext[baxis];
        uint32_t nl = 0, nr = 0;
        uint32_t* right = tmp.data() + lo;          // this subtree's private scratch range
        for (uint32_t i = lo; i < hi; ++i) {
          const uint32_t p = idx[i];
          int b = (int)(((cc[p] - c0) / e) * (float)kBins);
          if (b > kBins - 1) b = kBins - 1;
          if (b <= bsplit) idx[lo + nl++] = p; else right[nr++] = p;
        }
        std::memcpy(&idx[lo + nl], right, sizeof(uint32_t) * nr);
        return lo + nl;
      }
    } else if (!force_median) {
      float ext[3], inv_ok[3];
      for (int a = 0; a < 3; ++a) { ext[a] = cmx[a] - cmn[a]; inv_ok[a] = ext[a] > 0.f ? 1.f : 0.f; }
      struct Bins { uint32_t cnt[3][kBins]; Box bb[3][kBins]; };
      std::vector<Bins> part(chunks);
      run_chunks(chunks, [&](int c) {
        Bins& B = part[c];
        for (int a = 0; a < 3; ++a) for (int b = 0; b < kBins; ++b) { B.cnt[a][b] = 0; B.bb[a][b].clear(); }
        for (uint32_t i = c_lo(c), e = c_lo(c + 1); i < e; ++i) {
          const uint32_t p = idx[i];
          const Box pb = prim_box(p);
          for (int a = 0; a < 3; ++a) {
            if (inv_ok[a] == 0.f) continue;
            int b = (int)(((cen[a][p] - cmn[a]) / ext[a]) * (float)kBins);
            if (b > kBins - 1) b = kBins - 1;
            B.cnt[a][b]++; B.bb[a][b].grow(pb);
          }
        }
      });
      Bins& M = part[0];                               // merge (min / max / integer sums: order-independent)
      for (int c = 1; c < chunks; ++c)
        for (int a = 0; a < 3; ++a) for (int b = 0; b < kBins; ++b) if (part[c].cnt[a][b]) { M.cnt[a][b] += part[c].cnt[a][b]; M.bb[a][b].grow(part[c].bb[a][b]); }
      auto& cnt = M.cnt; auto& bb = M.bb;
      float best = 3.0e38f; int baxis = -1, bsplit = -1;
      for (int a = 0; a < 3; ++a) {
        if (inv_ok[a] == 0.f) continue;
        float rarea[kBins]; uint32_t rcnt[kBins];
        Box acc; acc.clear(); uint32_t c = 0;
        for (int b = kBins - 1; b >= 1; --b) {
          if (cnt[a][b]) acc.grow(bb[a][b]);
          c += cnt[a][b]; rcnt[b] = c; rarea[b] = c ? acc.half_area() : 0.f;
        }
        acc.clear(); c = 0;
        for (int s = 0; s < kBins - 1; ++s) {
          if (cnt[a][s]) acc.grow(bb[a][s]);
          c += cnt[a][s];
          if (c == 0 || rcnt[s + 1] == 0) continue;
          const float cost = fma_(acc.half_area(), (float)c, rarea[s + 1] * (float)rcnt[s + 1]);
          if (cost < best) { best = cost; baxis = a; bsplit = s; }
        }
      }
      if (baxis >= 0) {
        const std::vector<float>& cc = cen[baxis];
        const float c0 = cmn[baxis], e = ext[baxis];
        auto goes_left = [&](uint32_t p) {
          int b = (int)(((cc[p] - c0) / e) * (float)kBins);
          if (b > kBins - 1) b = kBins - 1;
          return b <= bsplit;
        };
        if (chunks == 1) {
          uint32_t nl = 0, nr = 0;
          uint32_t* right = tmp.data() + lo;          // this subtree's private scratch range
          for (uint32_t i = lo; i < hi; ++i) { const uint32_t p = idx[i]; if (goes_left(p)) idx[lo + nl++] = p; else right[nr++] = p; }
          std::memcpy(&idx[lo + nl], right, sizeof(uint32_t) * nr);
          return lo + nl;
        }
        // chunked stable partition: count, exclusive prefix, scatter into tmp[lo, hi) at final positions, copy back
        std::vector<uint32_t> nlc(chunks + 1, 0);
        run_chunks(chunks, [&](int c) { uint32_t k = 0; for (uint32_t i = c_lo(c), e2 = c_lo(c + 1); i < e2; ++i) k += goes_left(idx[i]); nlc[c + 1] = k; });
        for (int c = 0; c < chunks; ++c) nlc[c + 1] += nlc[c];
        const uint32_t NL = nlc[chunks];
        run_chunks(chunks, [&](int c) {
          uint32_t l = lo + nlc[c], r = lo + NL + ((c_lo(c) - lo) - nlc[c]);
          for (uint32_t i = c_lo(c), e2 = c_lo(c + 1); i < e2; ++i) { const uint32_t p = idx[i]; if (goes_left(p)) tmp[l++] = p; else tmp[r++] = p; }
        });
        run_chunks(chunks, [&](int c) { std::memcpy(&idx[c_lo(c)], &tmp[c_lo(c)], sizeof(uint32_t) * (c_lo(c + 1) - c_lo(c))); });
        return lo + NL;
      }
    }
    // object median on the widest centroid axis
    int ax = 0; float em = cmx[0] - cmn[0];
    if (cmx[1] - cmn[1] > em) { em = cmx[1] - cmn[1]; ax = 1; }
    if (cmx[2] - cmn[2] > em) { em = cmx[2] - cmn[2]; ax = 2; }
    if (em > 0.f) {
      const std::vector<float>& cc = cen[ax];
      std::sort(idx.begin() + lo, idx.begin() + hi, [&](uint32_t x, uint32_t y) {
        const float kx = cc[x], ky = cc[y];
        return kx < ky || (kx == ky && x < y);
      });
    }
    return lo + n / 2;
  }

  void build(uint32_t me, uint32_t lo, uint32_t hi, int depth) {
    for (;;) {
      BNode& nd = nodes[me];
      nd.lo = lo; nd.hi = hi; nd.left = nd.right = -1;
      const int extra = grab_threads(hi - lo);         // > 0: this node is processed in extra+1 chunks
      const int chunks = extra + 1;
      struct Bounds { Box box; float cmn[3], cmx[3]; };
      Box box; float cmn[3], cmx[3];
      if (chunks == 1) {                               // the common case: no heap traffic per node
        box.clear();
        for (int a = 0; a < 3; ++a) { cmn[a] = 3.0e38f; cmx[a] = -3.0e38f; }
        for (uint32_t i = lo; i < hi; ++i) {
          const uint32_t p = idx[i];
          for (int a = 0; a < 3; ++a) {
            box.mn[a] = std::min(box.mn[a], pmn[a][p]); box.mx[a] = std::max(box.mx[a], pmx[a][p]);
            cmn[a] = std::min(cmn[a], cen[a][p]); cmx[a] = std::max(cmx[a], cen[a][p]);
          }
        }
      } else {
        std::vector<Bounds> bp(chunks);
        run_chunks(chunks, [&](int c) {
          Bounds& b = bp[c]; b.box.clear();
          for (int a = 0; a < 3; ++a) { b.cmn[a] = 3.0e38f; b.cmx[a] = -3.0e38f; }
          const uint32_t n_ = hi - lo;
          for (uint32_t i = lo + (uint32_t)((uint64_t)n_ * c / chunks), e = lo + (uint32_t)((uint64_t)n_ * (c + 1) / chunks); i < e; ++i) {
            const uint32_t p = idx[i];
            for (int a = 0; a < 3; ++a) {
              b.box.mn[a] = std::min(b.box.mn[a], pmn[a][p]); b.box.mx[a] = std::max(b.box.mx[a], pmx[a][p]);
              b.cmn[a] = std::min(b.cmn[a], cen[a][p]); b.cmx[a] = std::max(b.cmx[a], cen[a][p]);
            }
          }
        });
        box = bp[0].box;
        for (int a = 0; a < 3; ++a) { cmn[a] = bp[0].cmn[a]; cmx[a] = bp[0].cmx[a]; }
        for (int c = 1; c < chunks; ++c) {
          box.grow(bp[c].box);
          for (int a = 0; a < 3; ++a) { cmn[a] = std::min(cmn[a], bp[c].cmn[a]); cmx[a] = std::max(cmx[a], bp[c].cmx[a]); }
        }
      }
      nd.box = box;
      if (hi - lo <= leaf_max) { spare_threads.fetch_add(extra); return; }
      const uint32_t mid = split(lo, hi, depth, cmn, cmx, extra);
      spare_threads.fetch_add(extra);                  // the children may use them (subtree tasks or their own chunks)
      const uint32_t l = alloc(), r = alloc();
      nodes[me].left = (int32_t)l; nodes[me].right = (int32_t)r;
      // hand the left half to another thread when it is big enough and one is free
      if (mid - lo > task_min && spare_threads.fetch_sub(1) > 0) {
        auto fut = std::async(std::launch::async, [this, l, lo, mid, depth] { build(l, lo, mid, depth + 1); spare_threads.fetch_add(1); });
        build(r, mid, hi, depth + 1);
        fut.get();
        return;
      }
      else if (mid - lo > task_min) spare_threads.fetch_add(1);
      build(l, lo, mid, depth + 1);
      me = r; lo = mid; ++depth;   // tail-iterate on the right half
    }
  }
};

struct Collapser {
  const std::vector<BNode>& bn;
  std::vector<QNode>& qn;
  bool instances; uint32_t leaf0; const std::vector<uint32_t>& idx; std::vector<uint32_t>& order;

  // 4-wide children of binary node bi (its grandchildren, a leaf child stays) in SLOT order: inner children first, then
  // the leaves, each group in collapse order; a leaf without primitives (empty scene) is dropped.  Returns the child
  // count, ni = number of inner children.
  int slots_of(uint32_t bi, uint32_t slot[4], int& ni) const {
    uint32_t kids[4]; int nk = 0;
    const BNode& b = bn[bi];
    if (b.left < 0) kids[nk++] = bi;
    else {
      for (uint32_t ch : {(uint32_t)b.left, (uint32_t)b.right}) {
        const BNode& c = bn[ch];
        if (c.left < 0) kids[nk++] = ch;
        else { kids[nk++] = (uint32_t)c.left; kids[nk++] = (uint32_t)c.right; }
      }
    }
    int nc = 0;
    for (int k = 0; k < nk; ++k) if (bn[kids[k]].left >= 0) slot[nc++] = kids[k];
    ni = nc;
    for (int k = 0; k < nk; ++k) if (bn[kids[k]].left < 0 && bn[kids[k]].hi > bn[kids[k]].lo) slot[nc++] = kids[k];
    return nc;
  }

  // pass 1: number of 4-wide nodes below (and including) the node made from binary node bi.  With it -- and one primitive
  // per leaf, so a subtree's leaf count is hi - lo -- every node index and leaf position follows from prefix sums, before
  // any node is packed: subtrees can be packed by different threads and the numbering equals the sequential "take the
  // next free block when a node is expanded" rule of the format (crh_bvh_format.h).
  std::vector<uint32_t> qcnt;
  uint32_t count(uint32_t bi) {
    uint32_t slot[4]; int ni; slots_of(bi, slot, ni);
    uint32_t c = 1;
    for (int k = 0; k < ni; ++k) c += count(slot[k]);
    return qcnt[bi] = c;
  }

  std::atomic<int>* spare = nullptr;
  // pass 2: pack node `me` from binary node bi; nb = first node index of the block holding its descendants,
  // lb = first leaf position of its subtree
  void pack(uint32_t bi, uint32_t me, uint32_t nb, uint32_t lb) {
    uint32_t slot[4]; int ni; const int nc = slots_of(bi, slot, ni);
    QNode q; std::memset(&q, 0, sizeof q);
    float cmin[4][3], cmax[4][3];
    for (int k = 0; k < nc; ++k) {
      const BNode& c = bn[slot[k]];
      for (int a = 0; a < 3; ++a) { cmin[k][a] = c.box.mn[a]; cmax[k][a] = c.box.mx[a]; }
      if (k >= ni) order[lb + (uint32_t)(k - ni) - leaf0] = idx[c.lo];          // one primitive per leaf
    }
    crh_pack_node(cmin, cmax, ni, nc, ni ? nb : 0u, (instances ? CRH_REF_INSTANCE_TAG : CRH_LEAF_TAG) | lb, q.w);   // 8-bit child bounds on the node's power-of-two grid
    qn[me] = q;
    std::vector<std::future<void>> tasks;
    uint32_t next_nb = nb + (uint32_t)ni, next_lb = lb + (uint32_t)(nc - ni);
    for (int k = 0; k < ni; ++k) {
      const uint32_t child = slot[k], at = nb + (uint32_t)k, cnb = next_nb, clb = next_lb;
      next_nb += qcnt[child] - 1u;
      next_lb += bn[child].hi - bn[child].lo;
      if (qcnt[child] > 16384u && spare && spare->fetch_sub(1) > 0)
        tasks.push_back(std::async(std::launch::async, [this, child, at, cnb, clb] { pack(child, at, cnb, clb); spare->fetch_add(1); }));
      else {
        if (qcnt[child] > 16384u && spare) spare->fetch_add(1);
        pack(child, at, cnb, clb);
      }
    }
    for (auto& t : tasks) t.get();
  }

  uint32_t run(uint32_t bi, std::atomic<int>* spare_threads) {
    qcnt.assign(bn.size(), 0u);
    const uint32_t total = count(bi);
    const uint32_t me = (uint32_t)qn.size();
    qn.reserve((size_t)me + total + total / 6u + 16u);      // head-room for the pair-alignment holes (build_tree): no second allocation of the array
    qn.resize((size_t)me + total);
    spare = spare_threads;
    pack(bi, me, me + 1u, leaf0);
    return me;
  }
};

}  // namespace

// CPUs this process can run on at once: hardware threads, cut down to the scheduler affinity mask and to the cgroup CPU quota (a
// container throttled to 16 CPUs of time on a 256-thread host builds FASTER with 16 threads than with 256)
static int usable_cpus()
{
  int n = (int)std::thread::hardware_concurrency(); if (n < 1) n = 1;
  cpu_set_t set; CPU_ZERO(&set);
  if (sched_getaffinity(0, sizeof set, &set) == 0) { const int a = CPU_COUNT(&set); if (a > 0 && a < n) n = a; }
  double quota = 0.0;
  if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) { char q[64]; double per = 0; if (fscanf(f, "%63s %lf", q, &per) == 2 && strcmp(q, "max") != 0 && per > 0) quota = atof(q) / per; fclose(f); }
  else if (FILE* g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {
    double q = 0, per = 0; if (fscanf(g, "%lf", &q) != 1) q = 0; fclose(g);
    if (FILE* h = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(h, "%lf", &per) != 1) per = 0; fclose(h); }
    if (q > 0 && per > 0) quota = q / per;
  }
  if (quota > 0.0) { const int qn = (int)(quota + 0.999); if (qn >= 1 && qn < n) n = qn; }
  // one process per GPU on one host (torchrun sets LOCAL_WORLD_SIZE): every rank builds the same tree at the same time -- each takes its share of the
  // CPUs instead of all of them (8 ranks x all threads on the 16 usable CPUs of a GPU box was an 8-fold oversubscription)
  if (const char* e = getenv("LOCAL_WORLD_SIZE")) { const int w = atoi(e); if (w > 1) n = n / w > 1 ? n / w : 1; }
  return n;
}

int build_threads(int threads) { if (threads <= 0) threads = usable_cpus(); return threads < 1 ? 1 : threads; }

uint32_t build_tree(const float* boxes, uint32_t n, bool instance_leaves, uint32_t leaf0,
                    std::vector<QNode>& nodes, std::vector<uint32_t>& order, float bmin[3], float bmax[3], int threads) {
  Builder B;
  B.leaf_max = 1;                      // the node format holds one primitive per leaf
  if (n < 524288u) B.task_min = std::max(1024u, n / 16u);       // small trees (a top-level tree over instances is rebuilt every frame) still split across threads
  const uint32_t cap = n ? n : 1;
  for (int a = 0; a < 3; ++a) { B.pmn[a].resize(cap); B.pmx[a].resize(cap); B.cen[a].resize(cap); }
  B.idx.resize(cap); B.tmp.resize(cap);
  Box scene; scene.clear();
  for (uint32_t t = 0; t < n; ++t) {
    for (int a = 0; a < 3; ++a) {
      const float lo = boxes[6 * (size_t)t + a], hi = boxes[6 * (size_t)t + 3 + a];
      B.pmn[a][t] = lo; B.pmx[a][t] = hi; B.cen[a][t] = (lo + hi) * 0.5f;
      scene.mn[a] = std::min(scene.mn[a], lo); scene.mx[a] = std::max(scene.mx[a], hi);
    }
    B.idx[t] = t;
  }
  B.nodes.resize(2 * (size_t)cap + 1);
  if (threads <= 0) threads = usable_cpus();
  if (threads < 1) threads = 1;
  B.spare_threads.store(threads - 1);
  const auto t0_ = std::chrono::steady_clock::now();
  const uint32_t root = B.alloc();
  B.build(root, 0, n, 0);
  const auto t1_ = std::chrono::steady_clock::now();
  order.assign(n, 0u);
  Collapser C{B.nodes, nodes, instance_leaves, leaf0, B.idx, order, {}, nullptr};
  B.spare_threads.store(threads - 1);
  const uint32_t qroot = C.run(root, &B.spare_threads);
  const auto t2_ = std::chrono::steady_clock::now();
  {
    // Pair alignment (format rule, crh_bvh_format.h): a block of >= 2 inner children starts on an EVEN node index, so that
    // the first two siblings share one 128-B L2 line (two 64-B node slots per line); the skipped slot stays zero.  The
    // blocks were numbered by prefix sums above (parallel); the holes depend on the running parity, so they are inserted
    // by one sequential pass in block-allocation order (= depth-first expansion order), +1.7 % C3 / +1.9 % C5.
    const uint32_t base = qroot, cnt = (uint32_t)nodes.size() - base;
    std::vector<uint32_t> nid(cnt, 0u), ncbs(cnt, 0u), stack; stack.push_back(0u);
    uint32_t shift = 0;
    while (!stack.empty()) {
      const uint32_t n = stack.back(); stack.pop_back();
      const QNode& q = nodes[base + n];
      const uint32_t ni = CRH_NODE_NINNER(q.w[3]);
      if (!ni) continue;
      const uint32_t cb = q.w[10] - base;
      uint32_t ncb = cb + shift;
      if (ni >= 2 && ((base + ncb) & 1u)) { ++shift; ++ncb; }
      ncbs[n] = base + ncb;
      for (uint32_t k = 0; k < ni; ++k) nid[cb + k] = ncb + k;
      for (uint32_t k = ni; k-- > 0;) stack.push_back(cb + k);
    }
    if (shift) {
      // nid is monotonic (blocks are numbered in the order this walk visits them, the shift only grows) and nid[n] >= n: the nodes move IN PLACE, from the
      // last one down, and the skipped slots are zeroed on the way -- no second copy of the array (at 10 M triangles: 350 MB allocated, zeroed and
      // copied back, seconds of first-touch page faults)
      nodes.resize((size_t)base + cnt + shift);
      uint32_t above = cnt + shift;                            // first slot already settled
      for (uint32_t n = cnt; n-- > 0;) {
        QNode q = nodes[base + n]; if (CRH_NODE_NINNER(q.w[3])) q.w[10] = ncbs[n];
        const uint32_t dst = nid[n];
        for (uint32_t h = dst + 1u; h < above; ++h) std::memset(&nodes[base + h], 0, sizeof(QNode));      // a hole (at most one per block)
        nodes[base + dst] = q; above = dst;
      }
    }
  }
  if (getenv("CRH_BUILD_VERBOSE")) fprintf(stderr, "build_tree n=%u: binary %.3f s, collapse + pack %.3f s, pair alignment %.3f s\n", n, std::chrono::duration<double>(t1_ - t0_).count(), std::chrono::duration<double>(t2_ - t1_).count(), std::chrono::duration<double>(std::chrono::steady_clock::now() - t2_).count());
  for (int a = 0; a < 3; ++a) { bmin[a] = n ? scene.mn[a] : 0.f; bmax[a] = n ? scene.mx[a] : 0.f; }
  return qroot;
}

void build_qbvh(const float* pos, const int32_t* tri, uint32_t n, QBvh& out, int threads) {
  std::vector<float> boxes(6 * (size_t)(n ? n : 1));
  for (uint32_t t = 0; t < n; ++t)
    for (int a = 0; a < 3; ++a) {
      const float v0 = pos[3 * tri[4 * t + 0] + a], v1 = pos[3 * tri[4 * t + 1] + a], v2 = pos[3 * tri[4 * t + 2] + a];
      boxes[6 * (size_t)t + a] = std::min(v0, std::min(v1, v2)); boxes[6 * (size_t)t + 3 + a] = std::max(v0, std::max(v1, v2));
    }
  out.nodes.clear();
  out.nodes.reserve(n / 2 + 16);
  build_tree(boxes.data(), n, false, 0, out.nodes, out.prim_order, out.bbmin, out.bbmax, threads);
}

}  // namespace crh

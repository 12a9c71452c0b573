// sanitize_host.cpp -- drives the product's host-side BVH builder (cadrays_amd/csrc/bvh_builder.cpp: worker threads, atomics,
// futures) under AddressSanitizer / UBSan / ThreadSanitizer.  CPU only; built by `make -C oracle asan|ubsan|tsan`, run by
// tests/hunts/run_sanitizers.sh.  Checks on top of what the sanitizer reports: the tree bytes do not depend on the thread count, and
// a top-level tree over instance boxes builds on the same code path.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../cadrays_amd/csrc/bvh_builder.h"

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static float urand()
{
  rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17;
  return (float)((rng_state >> 40) & 0xFFFFFF) / 16777216.0f;
}

int main(int argc, char** argv)
{
  const uint32_t sizes[] = {0, 1, 2, 5, 63, 1000, argc > 1 ? (uint32_t)atoi(argv[1]) : 60000u};
  int bad = 0;
  for (uint32_t n : sizes) {
    std::vector<float> pos(9 * (size_t)(n ? n : 1));
    std::vector<int32_t> tri(4 * (size_t)(n ? n : 1));
    const float r = n ? 1.5f / std::max(1.0f, (float)cbrt((double)n)) : 1.0f;
    for (uint32_t t = 0; t < n; ++t) {
      const float c[3] = {urand() * 2 - 1, urand() * 2 - 1, urand() * 2 - 1};
      for (int k = 0; k < 3; ++k) for (int a = 0; a < 3; ++a) pos[9 * (size_t)t + 3 * k + a] = c[a] + (k ? (urand() * 2 - 1) * r : 0.f);
      tri[4 * (size_t)t] = 3 * t; tri[4 * (size_t)t + 1] = 3 * t + 1; tri[4 * (size_t)t + 2] = 3 * t + 2; tri[4 * (size_t)t + 3] = 0;
    }
    if (n > 100) for (uint32_t t = 10; t < 30; ++t) memcpy(&pos[9 * (size_t)t], &pos[9 * 10], 36);      // coincident triangles: median splits
    crh::QBvh ref;
    const int threads[] = {1, 2, 4, 8, 0};
    for (int th : threads) {
      crh::QBvh b;
      crh::build_qbvh(pos.data(), tri.data(), n, b, th);
      if (th == 1) { ref = b; continue; }
      if (b.nodes.size() != ref.nodes.size() || (b.nodes.size() && memcmp(b.nodes.data(), ref.nodes.data(), b.nodes.size() * sizeof(crh::QNode))) ||
          b.prim_order != ref.prim_order) {
        fprintf(stderr, "sanitize_host: n=%u threads=%d: tree differs from the single-thread build\n", n, th); ++bad;
      }
    }
    // top-level tree over boxes (the crh_set_transforms path), appended behind the object tree like crh_api.cpp does
    if (n) {
      const uint32_t nb = n < 4096 ? n : 4096;
      std::vector<float> boxes(6 * (size_t)nb);
      for (uint32_t i = 0; i < nb; ++i) for (int a = 0; a < 3; ++a) {
        float lo = pos[9 * (size_t)i + a], hi = lo;
        for (int k = 1; k < 3; ++k) { lo = std::min(lo, pos[9 * (size_t)i + 3 * k + a]); hi = std::max(hi, pos[9 * (size_t)i + 3 * k + a]); }
        boxes[6 * (size_t)i + a] = lo; boxes[6 * (size_t)i + 3 + a] = hi;
      }
      std::vector<crh::QNode> nodes = ref.nodes; std::vector<uint32_t> order; float bmin[3], bmax[3];
      for (int th : {1, 8}) {
        nodes.resize(ref.nodes.size());
        const uint32_t root = crh::build_tree(boxes.data(), nb, true, 0, nodes, order, bmin, bmax, th);
        if (root < ref.nodes.size() || order.size() != nb) { fprintf(stderr, "sanitize_host: top-level tree malformed\n"); ++bad; }
      }
    }
    printf("sanitize_host: n=%u nodes=%zu ok\n", n, ref.nodes.size());
  }
  return bad ? 1 : 0;
}

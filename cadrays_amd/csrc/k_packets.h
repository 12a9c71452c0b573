// k_packets.h -- part of kernels.hip (ONE translation unit: included there inside namespace crh::(anonymous), in this order: k_common, k_traversal, k_packets, k_bsdf,
// k_lights_env, k_raygen, k_shade, k_accumulate).  The camera rays of wide batches as wavefront packets: packet nodes, packet_walk, k_trace_packets; API-level ray tracing kernels.
// ================================================================== camera-ray packets
// Bounce 0 of a wide batch: the 64 consecutive queue entries a wavefront takes are 64 samples of ONE pixel (or of 2 - 4 neighbouring ones: slot_to_pixel_sample)
// -- rays that differ by a sub-pixel jitter.  Walked one by one (trace_engine) they fetch the same nodes 64 times through the address path and still run at
// 0.71 of the lanes, because they reach their leaves in different rounds; it is the launch that costs most (20 % of the traversal time, VALU-issue bound:
// profiles/r4/per_bounce_counters.txt).  Here the WAVEFRONT walks the tree once for all of them:
//   * one stack per wavefront, held in three VGPRs (lane i = entry i: a select on push, v_readlane on pop), each entry a node reference + the 64-bit mask of the lanes
//     whose ray entered that child's box; node and triangle records are fetched with SCALAR loads (one request per wavefront instead of 64 lane requests);
//   * a lane takes part in a node / triangle test iff its bit is set -- exactly the rays that would get there in a walk of their own (the box test, its guard
//     band and the pruning against the lane's own `best` are the per-ray ones, bit for bit) -- so every lane computes something useful in every instruction;
//   * children are taken near to far as the FIRST participating lane sees them (its keys, sorted on the scalar unit); children only other lanes hit follow in
//     slot order.
// The hit a ray ends with does not depend on the order triangles are tested in -- the nearest one wins -- EXCEPT among triangles at exactly the same distance,
// where the spec says "the first in the ray's own near-to-far walk" (strict t < best).  A lane that meets such a tie (a valid hit at t == best) is flagged and
// written to the fall-back queue instead of the hit buffer; k_trace_nearest<.., FB> walks those rays (a handful per million in scenes with shared edges, none
// in a triangle soup) in the prescribed order afterwards.  Hits are therefore the spec's, bit for bit; node visits are not counted here -- the counting
// instantiations never use packets.
constexpr uint32_t kPacketChunk = 1024;      // queue entries (16 packets) per cursor fetch: one counter word sustains ~88 atomics / us
__device__ __forceinline__ uint32_t sgpr(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }

// lane `lane` (uniform) of three registers <- three uniform values: v_writelane with the lane in M0 (two different scalar registers in one VALU instruction
// exceed the constant bus; the compiler's builtin for it is not declared by this toolchain)
__device__ __forceinline__ void stack_put(uint32_t& a, uint32_t& b, uint32_t& c, uint32_t va, uint32_t vb, uint32_t vc, uint32_t lane)
{
  asm("s_mov_b32 m0, %6\n\tv_writelane_b32 %0, %3, m0\n\tv_writelane_b32 %1, %4, m0\n\tv_writelane_b32 %2, %5, m0"
      : "+v"(a), "+v"(b), "+v"(c) : "s"(va), "s"(vb), "s"(vc), "s"(lane) : "m0");
}
// one of four uniform masks by a uniform index: three scalar selects
__device__ __forceinline__ unsigned long long pick_mask(unsigned long long m0, unsigned long long m1, unsigned long long m2, unsigned long long m3, uint32_t i)
{
  unsigned long long m = m0;
  m = i == 1u ? m1 : m; m = i == 2u ? m2 : m; m = i == 3u ? m3 : m;
  return m;
}

// (t & ~3) | K as ONE vector instruction that the compiler may not move behind the v_readlane that follows it (it would: two scalar instructions on values that
// are uniform by then -- but the packet walk is bound by the scalar unit, k_trace_packets)
template <int K> __device__ __forceinline__ uint32_t key_bits(uint32_t t)
{
  uint32_t r;
  asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(r) : "v"(t), "s"(0x7FFFFFFCu), "n"(K));
  return r;
}

// Packet nodes (k_trace_packets<true>): node i of the tree as 8 x float4 -- {origin.xyz | exponents, counts}, x planes {lo0, hi0, lo1, hi1} {lo2, hi2, lo3, hi3},
// y planes, z planes, {first inner child, first leaf reference, -, -} -- the quantised bytes of the 64-B node converted once per scene instead of once per visit
__global__ __launch_bounds__(kBlock) void k_expand_packet_nodes(const float4* __restrict__ nodes, float4* __restrict__ pn, uint32_t n)
{
  const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  const float4* np = nodes + (uint32_t)(CRH_NODE_DWORDS / 4) * i;
  const float4 n0 = np[0], n1 = np[1], n2 = np[2];
  const uint32_t lo[3] = {__float_as_uint(n1.x), __float_as_uint(n1.y), __float_as_uint(n1.z)}, hi[3] = {__float_as_uint(n1.w), __float_as_uint(n2.x), __float_as_uint(n2.y)};
  const uint32_t nch = __float_as_uint(n0.w) >> 28;
  float4* out = pn + 8u * i;
  out[0] = n0;
  for (int a = 0; a < 3; ++a) {
    float f[8];
    for (int k = 0; k < 4; ++k) {
      f[2 * k] = (float)((lo[a] >> (8 * k)) & 0xffu); f[2 * k + 1] = (float)((hi[a] >> (8 * k)) & 0xffu);
      // a slot without a child: lower plane +inf, upper plane -inf -- the entry distance comes out +inf and the exit distance -inf on every axis whatever the
      // direction (inf x finite scale; a 0 x inf = NaN on ONE axis is dropped by max / min, and a unit direction cannot scale all three axes to zero), so no
      // ray enters it and the walk needs no child-count test
      if ((uint32_t)k >= nch) { f[2 * k] = __builtin_inff(); f[2 * k + 1] = -__builtin_inff(); }
    }
    out[1 + 2 * a] = make_float4(f[0], f[1], f[2], f[3]); out[2 + 2 * a] = make_float4(f[4], f[5], f[6], f[7]);
  }
  out[7] = make_float4(n2.z, n2.w, 0.f, 0.f);
}

// PN: the node is read from the PACKET-NODE array (k_expand_packet_nodes: the eight quantised planes of the four children as FLOATS, 128 B per node): a packed
// multiply-add takes a child's {lower, upper} plane pair straight from the scalar registers the node was loaded into, and the 24 byte-to-float conversions of a
// visit are gone.  The values are the same floats (0 .. 255), so is every result.  Only with uniform direction signs (OCT < 8).
template <int OCT, bool PN>
__device__ __forceinline__ void packet_walk(const float4* __restrict__ nodes, const float4* __restrict__ tris, uint32_t root, uint32_t lane, bool act,
                                            v3 o, v3 d, float ix, float iy, float iz, float gx, float gy, float gz, float4& hit, bool& amb)
{
  float best = CRH_MAXFLOAT; bool found = false;
  // OCT < 8: the direction signs of the whole packet (bit 0 / 1 / 2 = x / y / z negative), known at compile time -- the lower / upper byte words of a node are
  // then picked by REGISTER CHOICE (scalar operands of v_cvt_f32_ubyte) instead of six selects per visit; OCT = 8: mixed signs, per-lane selects
  const bool sx = OCT < 8 ? (OCT & 1) != 0 : ix < 0.f, sy = OCT < 8 ? (OCT & 2) != 0 : iy < 0.f, sz = OCT < 8 ? (OCT & 4) != 0 : iz < 0.f;
  // the wavefront's stack: lane i of these three registers is entry i
  uint32_t st_ref = 0, st_mlo = 0, st_mhi = 0; uint32_t sp = 0; bool ovf = false;
  unsigned long long cm = __ballot(act);
  uint32_t cur = root;
  while (cm != 0ull) {
    const bool in = (cm >> lane) & 1ull;
    if (!(cur & kQLeafBit)) {
      // uniform address: scalar loads
      const float4* np = PN ? nodes + 8u * cur : nodes + (uint32_t)(CRH_NODE_DWORDS / 4) * cur;
      const float4 n0 = np[0], n1 = np[1], n2 = np[2];
      float4 n3 = n0, n4 = n0, n5 = n0, n6 = n0, n7 = n0;
      if (PN) { n3 = np[3]; n4 = np[4]; n5 = np[5]; n6 = np[6]; n7 = np[7]; }
      const uint32_t ew = __float_as_uint(n0.w);
      const uint32_t ni = (ew >> 24) & 7u, nch = ew >> 28;
      const uint32_t base_inner = __float_as_uint(PN ? n7.x : n2.z), base_leaf = __float_as_uint(PN ? n7.y : n2.w) - ni;
      const float ax = __builtin_amdgcn_ldexpf(ix, (int)(ew << 24) >> 24), ay = __builtin_amdgcn_ldexpf(iy, (int)(ew << 16) >> 24), az = __builtin_amdgcn_ldexpf(iz, (int)(ew << 8) >> 24);
      const float ddx = n0.x - o.x, ddy = n0.y - o.y, ddz = n0.z - o.z;
      const uint32_t lx = __float_as_uint(sx ? n1.w : n1.x), ly = __float_as_uint(sy ? n2.x : n1.y), lz = __float_as_uint(sz ? n2.y : n1.z);
      const uint32_t hx = __float_as_uint(sx ? n1.x : n1.w), hy = __float_as_uint(sy ? n1.y : n2.x), hz = __float_as_uint(sz ? n1.z : n2.y);
      const f32x2 ax2 = {ax, ax}, ay2 = {ay, ay}, az2 = {az, az};
      // PN: component 0 of a pair belongs to the LOWER plane whatever the direction, so the guard band changes sides with the sign instead of the planes
      const f32x2 bx2 = __builtin_elementwise_fma((f32x2){ddx, ddx}, (f32x2){ix, ix}, (PN && sx) ? (f32x2){gx, -gx} : (f32x2){-gx, gx});
      const f32x2 by2 = __builtin_elementwise_fma((f32x2){ddy, ddy}, (f32x2){iy, iy}, (PN && sy) ? (f32x2){gy, -gy} : (f32x2){-gy, gy});
      const f32x2 bz2 = __builtin_elementwise_fma((f32x2){ddz, ddz}, (f32x2){iz, iz}, (PN && sz) ? (f32x2){gz, -gz} : (f32x2){-gz, gz});
      const uint32_t L = (uint32_t)__builtin_ctzll(cm);                      // the lane whose keys order the children
      const float best_in = in ? best : -1.0f;                               // PN: entry distances are >= 0, so a lane that is not in this node's mask enters no child
      unsigned long long mk[4] = {0ull, 0ull, 0ull, 0ull};
      uint32_t key[4] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
#define CRH_QB(W, K) ((float)(((W) >> (8 * (K))) & 0xffu))
#define CRH_PP(A, B, K) ((K) == 0 ? (f32x2){(A).x, (A).y} : (K) == 1 ? (f32x2){(A).z, (A).w} : (K) == 2 ? (f32x2){(B).x, (B).y} : (f32x2){(B).z, (B).w})
#define CRH_PCHILD(K)                                                                                          \
      if (PN || (uint32_t)K < nch) {      /* PN: all four slots, an absent child is masked out below (no branch: the node's loads stay one batch) */ \
        f32x2 tx, ty, tz;                                                                                  \
        if (PN) {                                                                                          \
          const f32x2 px = __builtin_elementwise_fma(CRH_PP(n1, n2, K), ax2, bx2), py = __builtin_elementwise_fma(CRH_PP(n3, n4, K), ay2, by2); \
          const f32x2 pz = __builtin_elementwise_fma(CRH_PP(n5, n6, K), az2, bz2);                         \
          tx = sx ? (f32x2){px.y, px.x} : px; ty = sy ? (f32x2){py.y, py.x} : py; tz = sz ? (f32x2){pz.y, pz.x} : pz;      /* {near, far}: a choice of registers */ \
        } else {                                                                                           \
          tx = __builtin_elementwise_fma((f32x2){CRH_QB(lx, K), CRH_QB(hx, K)}, ax2, bx2);                \
          ty = __builtin_elementwise_fma((f32x2){CRH_QB(ly, K), CRH_QB(hy, K)}, ay2, by2);                \
          tz = __builtin_elementwise_fma((f32x2){CRH_QB(lz, K), CRH_QB(hz, K)}, az2, bz2);                \
        }                                                                                                  \
        const float tmin = fmaxf(fmaxf(fmaxf(tx.x, ty.x), tz.x), 0.f);                                     \
        const float tmx  = fminf(fminf(fminf(tx.y, ty.y), tz.y), PN ? best_in : best);                     \
        const bool hitk = tmin <= tmx;                                                                     \
        /* every lane of the wavefront runs this loop: the vote is a plain compare into a scalar pair.  PN: an empty slot admits nobody (k_expand_packet_nodes), a lane \
           outside the node's mask prunes against -1 (best_in), and the ordering key is lane L's entry distance whether or not L itself enters the child */ \
        mk[K] = PN ? __builtin_amdgcn_ballot_w64(hitk) : (__builtin_amdgcn_ballot_w64(hitk) & cm);         \
        const uint32_t tv = __float_as_uint((PN || hitk) ? tmin : __builtin_inff());      /* !PN: lane L is in cm: +inf when it misses this child */ \
        const uint32_t tb = PN ? (uint32_t)__builtin_amdgcn_readlane((int)key_bits<K>(tv), (int)L)      /* PN: the key is finished on the vector unit (one v_and_or), the scalar one is the busier */ \
                               : (((uint32_t)__builtin_amdgcn_readlane((int)tv, (int)L) & 0x7FFFFFFCu) | (uint32_t)K);         \
        key[K] = mk[K] != 0ull ? tb : 0xFFFFFFFFu;                                                         \
      }
      CRH_PCHILD(0) CRH_PCHILD(1) CRH_PCHILD(2) CRH_PCHILD(3)
#undef CRH_PCHILD
#undef CRH_PP
#undef CRH_QB
      { // four unique scalar keys, ascending: the children somebody hit come first (0xFFFFFFFF = nobody)
        uint32_t a0 = min(key[0], key[1]), a1 = max(key[0], key[1]), b0 = min(key[2], key[3]), b1 = max(key[2], key[3]);
        const uint32_t lo = min(a0, b0), hi = max(a1, b1), m0 = max(a0, b0), m1 = min(a1, b1);
        key[0] = lo; key[1] = min(m0, m1); key[2] = max(m0, m1); key[3] = hi;
      }
#define CRH_PREF(I) (((I) < ni ? base_inner : base_leaf) + (I))
      if (key[0] != 0xFFFFFFFFu) {
        if (sp > 61u) { ovf = true; break; }      // deeper than any tree of the builder (<= 60 pending entries): the whole packet takes the fall-back pass
#define CRH_PPUSH(KEY)                                                                                         \
        if ((KEY) != 0xFFFFFFFFu) {                                                                        \
          const uint32_t ci = (KEY) & 3u;                                                                  \
          const unsigned long long pm = pick_mask(mk[0], mk[1], mk[2], mk[3], ci);                                                 \
          stack_put(st_ref, st_mlo, st_mhi, CRH_PREF(ci), (uint32_t)pm, (uint32_t)(pm >> 32), sp);      /* lane `sp` of the three registers takes the entry */ \
          ++sp;                                                                                            \
        }
        CRH_PPUSH(key[3]) CRH_PPUSH(key[2]) CRH_PPUSH(key[1])                 // far .. near
#undef CRH_PPUSH
        const uint32_t c0 = key[0] & 3u;
        cur = CRH_PREF(c0); cm = pick_mask(mk[0], mk[1], mk[2], mk[3], c0);
        continue;
      }
#undef CRH_PREF
    } else {
      const uint32_t ti = cur & 0x0FFFFFFFu;                                  // one triangle per leaf; uniform address: scalar loads
      const float4* tp = tris + kTriStride * ti;
      const float4 a = tp[0], b = tp[1], c = tp[2];
      const v3 v0 = xyz(a), e0 = xyz(b), e1 = xyz(c), nrm = crh_mk3(a.w, b.w, c.w);
      const v3 to = crh_sub3(v0, o);                                          // trace_engine::tri_step, operation by operation
      const float inv = 1.0f / crh_dot3(nrm, d);
      const v3 vc = crh_cross3(d, to);
      const float tt = crh_dot3(nrm, to) * inv, uu = crh_dot3(vc, e1) * inv, vv = crh_dot3(vc, e0) * inv;
      const bool ok = in && tt >= 0.f && uu >= 0.f && vv >= 0.f && (uu + vv) <= 1.0f;
      amb = amb || (ok && found && tt == best);                               // two triangles at exactly this distance: the ray's own walk decides (fall-back pass)
      const bool acc = ok && tt < best;                                       // selects, not branches: every lane of the wavefront is here anyway
      best = acc ? tt : best; found = found || acc;
      hit.x = acc ? tt : hit.x; hit.y = acc ? uu : hit.y; hit.z = acc ? vv : hit.z; hit.w = acc ? __int_as_float((int)ti) : hit.w;
    }
    if (sp == 0u) break;
    --sp;
    cur = (uint32_t)__builtin_amdgcn_readlane((int)st_ref, (int)sp);
    cm = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)st_mhi, (int)sp) << 32) | (unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)st_mlo, (int)sp);
  }
  if (ovf) amb = true;
}

template <bool PN>
__global__ __launch_bounds__(kBlock, 8) void k_trace_packets(DScene S, DPaths P, const float4* __restrict__ nodes, const float4* __restrict__ pnodes, const float4* __restrict__ tris,      // = S.nodes, S.pnodes, S.tris: as restrict-qualified PARAMETERS the compiler may read them with scalar loads
                                                          const uint32_t* __restrict__ q, const uint32_t* __restrict__ count,
                                                          uint32_t* __restrict__ cursors, uint32_t* zero_a, uint32_t* zero_b, uint32_t* zero_c, uint32_t* zero_d,
                                                          uint32_t* __restrict__ fb_q, uint32_t* __restrict__ fb_count, DCounters* C)
{
  const uint32_t n = *count;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    *zero_a = 0u; *zero_b = 0u; *zero_c = 0u; *zero_d = 0u;      // as k_trace_nearest: the other queue's count, the shadow count, the second-pass counts
    cursors[1] = 0u; cursors[2] = 0u; cursors[5] = 0u;
    atomicAdd(&C->rays_nearest, (unsigned long long)n);
    atomicAdd(&C->packet_rays, (unsigned long long)n);
  }
  const uint32_t lane = lane_id();
  const float4* __restrict__ ray_o = P.ray_o[0]; const float4* __restrict__ ray_d = P.ray_d[0];
  const float4 gb = S.guard_box;
  for (;;) {
    uint32_t cbase = 0;
    if (lane == 0) cbase = atomicAdd(cursors, kPacketChunk);
    cbase = sgpr(__shfl(cbase, 0));
    if (cbase >= n) break;
    const uint32_t cend = min(cbase + kPacketChunk, n);
    for (uint32_t pb = cbase; pb < cend; pb += 64u) {
      const uint32_t idx = pb + lane;
      const bool act = idx < cend;
      uint32_t tag = 0; v3 o = crh_mk3(0.f, 0.f, 0.f), d = crh_mk3(1.f, 0.f, 0.f);
      if (act) { tag = q[idx]; const float4 o4 = ld_stream(&ray_o[tag]), d4 = ld_stream(&ray_d[tag]); o = xyz(o4); d = xyz(d4); }
      const float ix = inv_dir(d.x), iy = inv_dir(d.y), iz = inv_dir(d.z);
      const float R = CRH_FMA(gb.w, 3.0f, (crh_abs(o.x - gb.x) + crh_abs(o.y - gb.y)) + crh_abs(o.z - gb.z)) * kSlabGuard;      // trace_engine::set_guard
      const float gx = crh_abs(ix) * R, gy = crh_abs(iy) * R, gz = crh_abs(iz) * R;
      float4 hit = make_float4(CRH_MAXFLOAT, 0.f, 0.f, __int_as_float(-1)); bool amb = false;
      {
        // the packet's direction signs: uniform over the wavefront for all but the packets that straddle an axis of the view -- one specialised walk per octant
        const unsigned long long am_ = __ballot(act), bx_ = __ballot(act && ix < 0.f), by_ = __ballot(act && iy < 0.f), bz_ = __ballot(act && iz < 0.f);
        const bool uni = (bx_ == 0ull || bx_ == am_) && (by_ == 0ull || by_ == am_) && (bz_ == 0ull || bz_ == am_);
        const uint32_t oct = uni ? ((bx_ ? 1u : 0u) | (by_ ? 2u : 0u) | (bz_ ? 4u : 0u)) : 8u;
#define CRH_WALK(O) case O: packet_walk<O, PN>(PN ? pnodes : nodes, tris, S.root, lane, act, o, d, ix, iy, iz, gx, gy, gz, hit, amb); break;
        switch (oct) { CRH_WALK(0) CRH_WALK(1) CRH_WALK(2) CRH_WALK(3) CRH_WALK(4) CRH_WALK(5) CRH_WALK(6) CRH_WALK(7) default: packet_walk<8, false>(nodes, tris, S.root, lane, act, o, d, ix, iy, iz, gx, gy, gz, hit, amb); }
#undef CRH_WALK
      }
      if (act && !amb) st_stream(&P.hit[tag], hit);
      const unsigned long long am = __ballot(act && amb);
      if (am != 0ull) {
        uint32_t fb = 0;
        if (lane == 0) { fb = atomicAdd(fb_count, (uint32_t)__popcll(am)); atomicAdd(&C->packet_fallback, (unsigned long long)__popcll(am)); }
        fb = __shfl(fb, 0);
        if (act && amb) fb_q[fb + (uint32_t)__popcll(am & ((1ull << lane) - 1ull))] = tag;
      }
    }
  }
}

// API-level tracing of a caller ray buffer {o.xyz, tmax, d.xyz, -}; `cursor` must be zero at launch
template <bool ANY, bool COUNT, bool TWO>
__global__ CRH_TRACE_BOUNDS void k_trace_rays(DScene S, const float4* __restrict__ rays, uint32_t n, uint32_t* __restrict__ cursor,
                                               float4* __restrict__ out_hit, uint32_t* __restrict__ out_vis, DCounters* C)
{
  __shared__ uint32_t stk[kLdsStack * kBlock];
  uint32_t nn = 0, nt = 0;
  trace_engine<ANY, COUNT, TWO, false>(S.nodes, S.tris, S.inst_leaf, S.root, S.guard_box, top2_of(S, false), cursor, n, &stk[threadIdx.x],
    [&](uint32_t idx, v3& o, v3& d, float& tmax, uint32_t& tag) {
      tag = idx;
      const float4 o4 = rays[2u * idx], d4 = rays[2u * idx + 1u];
      o = xyz(o4); d = xyz(d4); tmax = o4.w;
    },
    [&](uint32_t tag, float4 h, bool f) {
      if (ANY) out_vis[tag] = f ? 0u : 1u;
      else {
        const int k = __float_as_int(h.w);
        if (k >= 0) h.w = S.tris[kTriStride * (uint32_t)k + 3u].x;   // leaf order -> caller's triangle index (fourth quarter of the 64-B record)
        out_hit[tag] = h;
      }
    }, nn, nt);
  if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(ANY ? &C->rays_any : &C->rays_nearest, (unsigned long long)n);
  if (COUNT) {
    nn = wave_sum(nn); nt = wave_sum(nt);
    if (lane_id() == 0) {
      atomicAdd(ANY ? &C->nodes_any : &C->nodes_nearest, (unsigned long long)nn);
      atomicAdd(ANY ? &C->tris_any : &C->tris_nearest, (unsigned long long)nt);
    }
  }
}

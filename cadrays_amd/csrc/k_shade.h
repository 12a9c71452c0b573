// k_shade.h -- part of kernels.hip (ONE translation unit: included there inside namespace crh::(anonymous), in this order: k_common, k_traversal, k_packets, k_bsdf,
// k_lights_env, k_raygen, k_shade, k_accumulate).  K_shade: emission / environment, next-event estimation, BSDF sampling, Russian roulette, survivor compaction.
// ================================================================== shade
constexpr int kLdsMats = 64;   // materials staged in LDS (8 KB); larger tables are read from HBM/L2

// workgroups per CU the register allocation must allow: 4 => <= 128 VGPRs (143 unconstrained, 3 waves/SIMD).  Measured, Mrays/s at
// 0 / 4 / 5: C3 3457 / 3474 / 3200, C2 4803 / 4856 / 4569, C5 2416 / 2423, C1 13390 / 13765 (5 = 96 VGPRs spills the BSDF code)
#ifndef CRH_SHADE_MINWAVES
#define CRH_SHADE_MINWAVES 4
#endif
#if CRH_SHADE_MINWAVES > 0
#define CRH_SHADE_BOUNDS __launch_bounds__(kBlock, CRH_SHADE_MINWAVES)
#else
#define CRH_SHADE_BOUNDS __launch_bounds__(kBlock)
#endif
// One path at one bounce: what k_shade does for a queue entry once its state is in registers (also the wave-level shading step of k_frame, k_frame.h).
// In: the ray that was traced (o4 = origin | rng state, d4 = direction | (slot << 1) | inside flag), throughput | pending implicit pdf, the hit record, the
// path's slot and bounce.  Out: the successor ray / the shadow ray to store once their positions are known.  The radiance record P.rad[pid] is
// updated here (implicit light / environment, emission).
template <bool SPLIT>
__device__ __forceinline__ void shade_path(const DScene& S, const DPaths& P, const float4* s_mats, const bool mats_in_lds, const uint32_t bounce, const bool first, const bool last,
                                           const float4 o4, const float4 d4, const float4 t4, const float4 h, const uint32_t pid,
                                           bool& cont, bool& shadow, float4& n_o, float4& n_d, float4& n_t, float4& s_o, float4& s_d, float4& s_c, uint32_t& n_shaded)
{
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
  const uint2 st = make_uint2(__float_as_uint(o4.w), __float_as_uint(d4.w) & 1u);   // rng state, flags
  const v3 o = xyz(o4), d = xyz(d4);
  v3 W = xyz(t4); float imp_pdf = t4.w;
  const int hk = __float_as_int(h.w);
  const bool found = hk >= 0;
  float exp_pdf;
  const v3 le = intersect_light(S, o, d, bounce, found ? h.x : CRH_MAXFLOAT, exp_pdf);
  if (le.x > 0.f || le.y > 0.f || le.z > 0.f || !found) {
    const float mis = (bounce == 0u || imp_pdf == CRH_MAXFLOAT) ? 1.0f : (imp_pdf * imp_pdf) / CRH_FMA(exp_pdf, exp_pdf, imp_pdf * imp_pdf);
    float4 r4 = first ? zero4 : P.rad[pid];              // the radiance record is touched only when something is added
    if (__float_as_uint(r4.w) != P.stamp) r4 = zero4;    // ... and holds this batch's stamp once it has been (DPaths::stamp): anything else reads as zero
    P.rad[pid] = mk4(crh_add3(xyz(r4), crh_scale3(crh_mul3(W, le), mis)), __uint_as_float(P.stamp));
  } else {
    ++n_shaded;
    // shading record: one 64-B sector {n0 | material, n1 | instance, n2, geometric normal}.  The geometric normal of a
    // single-level scene is precomputed on the host with the same inline arithmetic (crh_math.h) the kernel used to apply to
    // the three vertices -- same bits -- so shading no longer gathers the 48-B triangle record (a second 128-B line per hit);
    // a two-level scene still needs the vertices: the normal is taken from the TRANSFORMED corners
    const float4* sp = S.shade + 4u * (uint32_t)hk;
    const float4 s0 = sp[0], s1 = sp[1], s2 = sp[2];
    float M[12];
    v3 ng;
    // two-level scenes: s1.w = the object of a triangle that lives in an object tree (-1: a triangle of the static world-space tree)
    const bool in_object = S.two_level && __float_as_int(s1.w) >= 0;
    if (in_object) {                                     // object -> world through the instance's forward transform
      const float4* tp = S.verts + 3u * (uint32_t)hk;       // the three object-space vertices (the traversal record holds edges, not vertices)
      const float4 a = tp[0], b4 = tp[1], c4 = tp[2];
      const float4* ip = S.inst + 8u * (uint32_t)__float_as_int(s1.w);
      const float4 f0 = ip[3], f1 = ip[4], f2 = ip[5];
      M[0] = f0.x; M[1] = f0.y; M[2] = f0.z; M[3] = f0.w; M[4] = f1.x; M[5] = f1.y; M[6] = f1.z; M[7] = f1.w;
      M[8] = f2.x; M[9] = f2.y; M[10] = f2.z; M[11] = f2.w;
      const v3 p0 = crh_xform_point(M, xyz(a)), p1 = crh_xform_point(M, xyz(b4)), p2 = crh_xform_point(M, xyz(c4));
      ng = crh_norm3(crh_cross3(crh_sub3(p0, p2), crh_sub3(p1, p0)));
    } else ng = xyz(sp[3]);
    const float w0 = (1.0f - h.y) - h.z;
    v3 ns = crh_norm3(crh_mk3(CRH_FMA(s2.x, h.z, CRH_FMA(s1.x, h.y, s0.x * w0)),
                              CRH_FMA(s2.y, h.z, CRH_FMA(s1.y, h.y, s0.y * w0)),
                              CRH_FMA(s2.z, h.z, CRH_FMA(s1.z, h.y, s0.z * w0))));
    if (in_object) ns = crh_norm3(crh_xform_vector(M, ns));
    if (!(crh_dot3(ns, ns) > 0.f)) ns = ng;
    const v3 p = crh_madd3(o, d, h.x);
    int mat = __float_as_int(s0.w); if (mat < 0 || (uint32_t)mat >= S.n_mats) mat = 0;
    const float4* mp = mats_in_lds ? (s_mats + 8 * mat) : (S.mats + 8 * mat);
    Bsdf bs;
    { const float4 m0 = mp[0], m1 = mp[1], m2 = mp[2], m3 = mp[3], m4 = mp[4];
      bs.Kc = xyz(m0); bs.Rc = m0.w; bs.Kd = xyz(m1); bs.Ks = xyz(m2); bs.Rs = m2.w; bs.Kt = xyz(m3); bs.Le = xyz(m4);
      bs.ab = mp[5]; bs.fc = mp[6]; bs.fb = mp[7]; }
    if (S.n_tex != 0u) {                                   // wave-uniform: scenes without textures skip the call
      const int slot = (int)mp[1].w - 1;
      if (slot >= 0) {
        const float4 tx = sample_texture(S, (uint32_t)slot, (uint32_t)hk, h.y, h.z, w0, mp[3].w, mp[4].w);
        bs.Kd = crh_mul3(bs.Kd, xyz(tx));
        if (tx.w != 1.0f) {                                // alpha cut-out: the uncovered part transmits
          bs.Kd = crh_scale3(bs.Kd, tx.w);
          const float ia = 1.0f - tx.w;
          bs.Kt = crh_mk3(CRH_FMA(tx.w, bs.Kt.x, ia), CRH_FMA(tx.w, bs.Kt.y, ia), CRH_FMA(tx.w, bs.Kt.z, ia));
        }
      }
    }
    const Frame fr = make_frame(ns);
    const v3 wo = to_local(fr, crh_mk3(-d.x, -d.y, -d.z));
    bs.Fc = fresnel_media(wo.z, bs.fc);
    bool inside = (st.y & 1u) != 0u;
    if (inside) {
      const float k = -(h.x * bs.ab.w);
      W = crh_mul3(W, crh_mk3(crh_exp(k * (1.0f - bs.ab.x)), crh_exp(k * (1.0f - bs.ab.y)), crh_exp(k * (1.0f - bs.ab.z))));
    }
    if (bs.Le.x != 0.f || bs.Le.y != 0.f || bs.Le.z != 0.f) {      // emissive surfaces are rare: skip the read-modify-write otherwise
      float4 r4 = first ? zero4 : P.rad[pid];
      if (__float_as_uint(r4.w) != P.stamp) r4 = zero4;
      P.rad[pid] = mk4(crh_add3(xyz(r4), crh_mul3(W, bs.Le)), __uint_as_float(P.stamp));
    }                                                              // nobody initialises the record: an unstamped one reads as zero
    uint32_t rng = st.x;
    // ---- next event estimation
    {
      const v3 z3 = crh_mk3(0.f, 0.f, 0.f);
      const v3 nd = crh_add3(bs.Kd, crh_add3(bs.Rc > kBsdfEps ? bs.Kc : z3, bs.Rs > kBsdfEps ? bs.Ks : z3));
      if (S.n_lights > 0u && crh_dot3(nd, W) > kBsdfEps) {
        const float fl = crh_rng_next_mode(&rng, S.spec_u32) * (float)S.n_lights;
        uint32_t li = (uint32_t)fl; if (li > S.n_lights - 1u) li = S.n_lights - 1u;
        const float k1 = crh_rng_next_mode(&rng, S.spec_u32), k2 = crh_rng_next_mode(&rng, S.spec_u32);
        const float4 l0 = S.lights[2u * li], l1 = S.lights[2u * li + 1u];
        v3 axis; float dist, cm;
        if (l0.w != 0.f) { const v3 tl = crh_sub3(xyz(l0), p); dist = crh_len3(tl); axis = crh_scale3(tl, 1.0f / dist); cm = sphere_cosmax(l1.w, dist); }
        else { axis = xyz(l0); dist = CRH_MAXFLOAT; cm = l1.w; }
        const Frame lf = make_frame(axis);
        const float ct = CRH_FMA(-k2, 1.0f - cm, 1.0f);
        float sn, cs; crh_sincos2pi(k1, &sn, &cs);
        const float sq = crh_sqrt(crh_max(CRH_FMA(-ct, ct, 1.0f), 0.f));
        const v3 ld = crh_norm3(from_local(lf, crh_mk3(cs * sq, sn * sq, ct)));
        const float e_pdf = (cm < 1.0f) ? (1.0f / (float)S.n_lights) * cone_pdf(cm) : CRH_MAXFLOAT;
        const v3 wi = to_local(fr, ld);
        const float i_pdf = pdf_layered(bs, wo, wi, W, S.two_sided);
        const float mis = (e_pdf == CRH_MAXFLOAT) ? 1.0f : e_pdf / CRH_FMA(e_pdf, e_pdf, i_pdf * i_pdf);
        const v3 contrib = crh_scale3(crh_mul3(xyz(l1), eval_layered(bs, wi, wo, S.two_sided)), mis);
        const v3 wc = crh_mul3(W, contrib);
        if (contrib.x > S.spec_min_contrib || contrib.y > S.spec_min_contrib || contrib.z > S.spec_min_contrib) {      // crh_spec.h #11
          shadow = true;
          s_o = mk4(offset_origin(p, ld, ng, S.eps), dist);
          // split scenes: .w != 0 marks a shadow ray that touches a moved object (the first any-hit pass leaves its contribution to the second)
          s_d = mk4(ld, (SPLIT && ray_touches_instances(S, xyz(s_o), ld, dist)) ? 1.0f : 0.f);
          s_c = mk4(wc, __uint_as_float(pid));
        }
      }
    }
    // ---- BSDF sampling + Russian roulette (the last bounce has no successor ray)
    if (!last) {
      v3 wi; bool delta; const v3 Wsel = W; int lobe;
      const bool alive = sample_layered(bs, wo, wi, W, inside, delta, rng, S.two_sided, SpecB{S.spec_u32, S.spec_eta_nd}, lobe);
      if (alive) imp_pdf = delta ? CRH_MAXFLOAT : pdf_layered(bs, wo, wi, Wsel, S.two_sided, S.spec_mis1 ? lobe : -1);
      const bool roulette = S.rr && bounce >= S.spec_rr_start;      // crh_spec.h #9, #10, #12
      float survive = (W.x > S.spec_min_thr || W.y > S.spec_min_thr || W.z > S.spec_min_thr) ? 1.0f : 0.f;
      if (roulette)
        survive = crh_min(CRH_FMA(0.0722f, W.z, CRH_FMA(0.7152f, W.y, 0.2126f * W.x)), S.spec_rr_cap) * survive;
      const float kr = crh_rng_next_mode(&rng, S.spec_u32);
      if (alive && kr < survive) {
        if (roulette) W = crh_mk3(W.x / survive, W.y / survive, W.z / survive);
        const v3 nd2 = crh_norm3(from_local(fr, wi));
        n_o = mk4(offset_origin(p, nd2, ng, S.eps), __uint_as_float(rng));
        n_d = mk4(nd2, __uint_as_float((pid << 1) | (inside ? 1u : 0u)));
        n_t = mk4(W, imp_pdf);
        cont = true;
      }
    }
  }
}

#if CRH_COHERENCE_STATS
// Instrumented builds only (tools/ab_build.sh coh "-DCRH_COHERENCE_STATS=1"; tools/coherence_vote.py; round-5 verdict item 2): would the rays of bounce
// >= 1 form packets?  The survivors / shadow rays of a chunk are written in rank order, so 64 consecutive ranks ARE the wavefront that traces them next.
// Per bounce: [0] groups of 64 consecutive continuation rays with >= 48 entries, [1] those in which >= 48 entries leave the SAME triangle, [2] ... through a
// delta lobe, [3] continuation rays; [4] groups of shadow rays, [5] those with >= 48 entries from the same triangle, [6] shadow rays, [7] -.
__device__ unsigned long long g_coherence[32 * 8];
__device__ __forceinline__ void coherence_vote(const uint32_t* keys, uint32_t n, unsigned long long* out, bool with_delta)
{
  const uint32_t lane = lane_id(), wave = threadIdx.x >> 6;
  for (uint32_t g = wave; g * 64u < n; g += (uint32_t)kBlock / 64u) {
    const uint32_t i = g * 64u + lane; const bool valid = i < n;
    const uint32_t key = valid ? keys[i] : 0xFFFFFFFFu;
    const uint32_t nv = (uint32_t)__popcll(__ballot(valid));
    uint32_t best = 0, best_key = 0;
    for (int c = 0; c < 64; c += 21) {                      // three candidates: a key that >= 48 of 64 entries share is behind one of them with probability 0.98
      const uint32_t kk = (uint32_t)__shfl((int)key, c);
      const uint32_t m = (uint32_t)__popcll(__ballot(valid && key == kk));
      if (m > best) { best = m; best_key = kk; }
    }
    if (lane == 0 && nv >= 48u) {
      atomicAdd(&out[0], 1ull);
      if (best >= 48u) { atomicAdd(&out[1], 1ull); if (with_delta && (best_key >> 31)) atomicAdd(&out[2], 1ull); }
    }
  }
}
#endif

template <bool SPLIT>
__global__ CRH_SHADE_BOUNDS void k_shade(DScene S, DPaths P, int cur, uint32_t bounce,
                                                   const uint32_t* __restrict__ q_in, const uint32_t* __restrict__ count_in,
                                                   uint32_t* __restrict__ q_out, uint32_t* __restrict__ count_out,
                                                   uint32_t* __restrict__ q_sh, uint32_t* __restrict__ count_sh,
                                                   uint32_t* __restrict__ q2, uint32_t* __restrict__ count2,
                                                   uint32_t* __restrict__ q2_sh, uint32_t* __restrict__ count2_sh,
                                                   uint32_t* __restrict__ cursors, DCounters* C)
{
  __shared__ float4 s_mats[kLdsMats * 8];
  // survivors / shadow rays of kShadeIters x 256 paths are collected in LDS and appended with ONE global atomic each
  __shared__ uint32_t s_qc[kShadeIters * kBlock], s_qs[kShadeIters * kBlock];
  // split scenes: those of them that touch a moved object, for the second traversal pass
  __shared__ uint32_t s_q2c[SPLIT ? kShadeIters * kBlock : 1], s_q2s[SPLIT ? kShadeIters * kBlock : 1];
  __shared__ uint32_t s_base, s_nc, s_ns, s_gc, s_gs, s_n2c, s_n2s, s_g2c, s_g2s;
#if CRH_COHERENCE_STATS
  __shared__ uint32_t s_kc[kShadeIters * kBlock], s_ks[kShadeIters * kBlock];
#endif
  if (blockIdx.x == 0 && threadIdx.x == 0) { cursors[0] = 0u; cursors[4] = 0u; }     // nearest-hit cursors (both passes) of the next bounce
  const bool mats_in_lds = S.n_mats <= (uint32_t)kLdsMats;
  if (mats_in_lds) {
    for (uint32_t i = threadIdx.x; i < S.n_mats * 8u; i += kBlock) s_mats[i] = S.mats[i];
    __syncthreads();
  }
  const uint32_t n = *count_in;
  const bool last = bounce + 1u >= S.max_depth;
  const bool first = bounce == 0u;
  const float4* __restrict__ in_o = P.ray_o[cur]; const float4* __restrict__ in_d = P.ray_d[cur]; const float4* __restrict__ in_t = P.thr[cur];
  float4* __restrict__ out_o = P.ray_o[1 - cur]; float4* __restrict__ out_d = P.ray_d[1 - cur]; float4* __restrict__ out_t = P.thr[1 - cur];
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
  uint32_t n_shaded = 0;
  for (;;) {
    __syncthreads();
    if (threadIdx.x == 0) { s_base = atomicAdd(cursors + 1, kShadeIters * (uint32_t)kBlock); s_nc = 0u; s_ns = 0u; s_n2c = 0u; s_n2s = 0u; }
    __syncthreads();
    const uint32_t base = s_base;
    if (base >= n) break;
#pragma unroll 1
    for (uint32_t it = 0; it < kShadeIters; ++it) {
    const uint32_t i = base + it * kBlock + threadIdx.x;
    bool cont = false, shadow = false;
    float4 n_o = zero4, n_d = zero4, n_t = zero4, s_o = zero4, s_d = zero4, s_c = zero4;     // successor ray / shadow ray, stored after the ranks are known
#if CRH_COHERENCE_STATS
    uint32_t coh_tri = 0u;
#endif
    if (i < n) {
      const uint32_t pos = q_in[i];
      const float4 o4 = in_o[pos], d4 = in_d[pos], h = P.hit[pos];
#if CRH_COHERENCE_STATS
      coh_tri = __float_as_uint(h.w) & 0x7FFFFFFFu;
#endif
      const float4 t4 = first ? make_float4(1.0f, 1.0f, 1.0f, CRH_MAXFLOAT) : in_t[pos];     // k_raygen leaves thr / rad unwritten
      const uint32_t pid = __float_as_uint(d4.w) >> 1;                              // the path's slot (radiance record, pixel)
      shade_path<SPLIT>(S, P, s_mats, mats_in_lds, bounce, first, last, o4, d4, t4, h, pid, cont, shadow, n_o, n_d, n_t, s_o, s_d, s_c, n_shaded);
    }
    // The r-th shadow ray / survivor of this chunk takes the position of the chunk's r-th input entry (in the other ray buffer for
    // survivors): positions stay packed in runs, no two chunks ever share one, and no global atomic is needed to find them.
    if (S.n_lights > 0u) {
      const uint32_t r = lds_rank(shadow, &s_ns);
      uint32_t ps = 0u;
      if (shadow) { ps = q_in[base + r]; s_qs[r] = ps; P.sh_o[ps] = s_o; P.sh_d[ps] = s_d; P.sh_c[ps] = s_c; }
#if CRH_COHERENCE_STATS
      if (shadow) s_ks[r] = coh_tri;
#endif
      if (SPLIT) lds_append(shadow && s_d.w != 0.f, ps, s_q2s, &s_n2s);
    }
    {
      const uint32_t r = lds_rank(cont, &s_nc);
      uint32_t pn = 0u;
      if (cont) { pn = q_in[base + r]; s_qc[r] = pn; out_o[pn] = n_o; out_d[pn] = n_d; out_t[pn] = n_t; }
#if CRH_COHERENCE_STATS
      if (cont) s_kc[r] = coh_tri | (n_t.w == CRH_MAXFLOAT ? 0x80000000u : 0u);      // the successor left through a delta lobe
#endif
      if (SPLIT) lds_append(cont && ray_touches_instances(S, xyz(n_o), xyz(n_d), CRH_MAXFLOAT), pn, s_q2c, &s_n2c);
    }
    }
    __syncthreads();
#if CRH_COHERENCE_STATS
    if (bounce < 32u) {
      coherence_vote(s_kc, s_nc, &g_coherence[8u * bounce], true);
      coherence_vote(s_ks, s_ns, &g_coherence[8u * bounce + 4u], false);
      if (threadIdx.x == 0) { atomicAdd(&g_coherence[8u * bounce + 3u], (unsigned long long)s_nc); atomicAdd(&g_coherence[8u * bounce + 6u], (unsigned long long)s_ns); }
    }
#endif
    if (threadIdx.x == 0) {
      s_gc = s_nc ? atomicAdd(count_out, s_nc) : 0u; s_gs = s_ns ? atomicAdd(count_sh, s_ns) : 0u;
      if (SPLIT) { s_g2c = s_n2c ? atomicAdd(count2, s_n2c) : 0u; s_g2s = s_n2s ? atomicAdd(count2_sh, s_n2s) : 0u; }
    }
    __syncthreads();
    for (uint32_t j = threadIdx.x; j < s_nc; j += kBlock) q_out[s_gc + j] = s_qc[j];
    for (uint32_t j = threadIdx.x; j < s_ns; j += kBlock) q_sh[s_gs + j] = s_qs[j];
    if (SPLIT) {
      for (uint32_t j = threadIdx.x; j < s_n2c; j += kBlock) q2[s_g2c + j] = s_q2c[j];
      for (uint32_t j = threadIdx.x; j < s_n2s; j += kBlock) q2_sh[s_g2s + j] = s_q2s[j];
    }
  }
  n_shaded = wave_sum(n_shaded);
  if (lane_id() == 0 && n_shaded) atomicAdd(&C->shaded_hits, (unsigned long long)n_shaded);
}

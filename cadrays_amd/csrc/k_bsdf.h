// k_bsdf.h -- part of kernels.hip (ONE translation unit: included there inside namespace crh::(anonymous), in this order: k_common, k_traversal, k_packets, k_bsdf,
// k_lights_env, k_raygen, k_shade, k_accumulate).  The double-layer BSDF: Fresnel models, lobe sampling and evaluation.
// ================================================================== BSDF
struct Bsdf {
  v3 Kc, Kd, Ks, Kt, Le, Fc;
  float Rc, Rs;
  float4 fc, fb, ab;
};

__device__ v3 fresnel_media(float cosI, float4 f)
{
  if (f.x > -0.5f) {
    const float m = 1.0f - crh_abs(cosI); const float m2 = m * m; const float m5 = (m2 * m2) * m;
    return crh_mk3(CRH_FMA(1.0f - f.x, m5, f.x), CRH_FMA(1.0f - f.y, m5, f.y), CRH_FMA(1.0f - f.z, m5, f.z));
  }
  if (f.x > -1.5f) return crh_mk3(f.z, f.z, f.z);
  if (f.x > -2.5f) {
    const float ci = crh_abs(cosI), n = f.y, k = f.z;
    const float tmp = (2.0f * n) * ci;
    const float t1 = CRH_FMA(n, n, k * k);
    const float ci2 = ci * ci;
    const float sperp = ((t1 - tmp) + ci2) / ((t1 + tmp) + ci2);
    const float t2 = t1 * ci2;
    const float sparl = ((t2 - tmp) + 1.0f) / ((t2 + tmp) + 1.0f);
    const float r = (sperp + sparl) * 0.5f;
    return crh_mk3(r, r, r);
  }
  const float n = f.y;
  const float etaI = cosI > 0.f ? 1.0f : n, etaT = cosI > 0.f ? n : 1.0f;
  float r = 1.0f;
  const float ratio = etaI / etaT;
  const float sinT2 = (ratio * ratio) * CRH_FMA(-cosI, cosI, 1.0f);
  if (sinT2 < 1.0f) {
    const float ci = crh_abs(cosI), ct = crh_sqrt(1.0f - sinT2);
    const float p0 = etaT * ci, p1 = etaI * ct, q0 = etaI * ci, q1 = etaT * ct;
    const float parl = (p0 - p1) / (p0 + p1);
    const float perp = (q0 - q1) / (q0 + q1);
    const float pp = parl * parl, qq = perp * perp;
    r = (pp + qq) * 0.5f;
  }
  return crh_mk3(r, r, r);
}

__device__ float smith_g1(v3 w, v3 m, float rough)
{
  float r = 0.f;
  if (crh_dot3(w, m) * w.z > 0.f) {
    const float tanT = crh_sqrt(crh_max(CRH_FMA(-w.z, w.z, 1.0f), 0.f)) / w.z;
    if (tanT == 0.f) r = 1.0f;
    else {
      const float a = 1.0f / (rough * tanT);
      r = CRH_FMA(2.181f, a, 3.535f) / CRH_FMA(2.577f, a, 1.0f / a + 2.276f);
    }
  }
  return crh_min(r, 1.0f);
}

__device__ __forceinline__ float blinn_power(float rough) { return crh_max(2.0f / (rough * rough) - 2.0f, 0.f); }

__device__ v3 eval_blinn(v3 wi, v3 wo, float4 fr, float rough)
{
  if (wi.z <= 0.f || wo.z <= 0.f) return crh_mk3(0.f, 0.f, 0.f);
  const v3 h = crh_norm3(crh_add3(wi, wo));
  const float e = blinn_power(rough);
  const float D = ((e + 2.0f) * CRH_INV_TWOPI) * crh_pow(h.z, e);
  const float G = smith_g1(wo, h, rough) * smith_g1(wi, h, rough);
  const v3 F = fresnel_media(crh_dot3(wo, h), fr);
  const float s = (D * G) / (4.0f * wo.z);
  return crh_scale3(F, s);
}

__device__ v3 eval_layered(const Bsdf& b, v3 wi, v3 wo, int two_sided)
{
  if (two_sided) { const float sg = crh_sign(wo.z); wi.z *= sg; wo.z *= sg; }
  const float lam = (wi.z <= 0.f || wo.z <= 0.f) ? 0.f : wi.z * CRH_INV_PI;
  v3 r = crh_scale3(b.Kd, lam);
  if (b.Rs > kBsdfEps) r = crh_add3(r, crh_mul3(b.Ks, eval_blinn(wi, wo, b.fb, b.Rs)));
  r = crh_mul3(r, crh_mk3(1.0f - b.Fc.x, 1.0f - b.Fc.y, 1.0f - b.Fc.z));
  if (b.Rc > kBsdfEps) r = crh_add3(r, crh_mul3(b.Kc, eval_blinn(wi, wo, b.fc, b.Rc)));
  return r;
}

struct Lobes { float pc, pd, ps, pt, total; v3 Tc; };
__device__ __forceinline__ Lobes lobe_probs(const Bsdf& b, v3 W)
{
  Lobes L;
  L.Tc = crh_mk3(1.0f - b.Fc.x, 1.0f - b.Fc.y, 1.0f - b.Fc.z);
  L.pc = crh_dot3(crh_mul3(b.Kc, b.Fc), W);
  L.pd = crh_dot3(crh_mul3(b.Kd, L.Tc), W);
  L.ps = crh_dot3(crh_mul3(b.Ks, L.Tc), W);
  L.pt = crh_dot3(crh_mul3(b.Kt, L.Tc), W);
  L.total = ((L.pc + L.pd) + L.ps) + L.pt;
  return L;
}

__device__ float blinn_pdf(float hz, float dih, float rough)
{
  const float e = blinn_power(rough);
  return (((e + 2.0f) * CRH_INV_TWOPI) * crh_pow(crh_abs(hz), e + 1.0f)) / (4.0f * crh_abs(dih));
}

// lobe < 0: the mixture pdf over all non-delta lobes (the spec's MIS pdf); lobe = 0 coat / 1 diffuse / 2 glossy: that lobe's pdf times its
// selection probability only (crh_spec.h #3, mis_single_lobe)
__device__ float pdf_layered(const Bsdf& b, v3 wo, v3 wi, v3 W, int two_sided, int lobe = -1)
{
  const Lobes L = lobe_probs(b, W);
  if (!(L.total > kBsdfEps)) return 0.f;
  if (two_sided) { const float sg = crh_sign(wo.z); wi.z *= sg; wo.z *= sg; }
  float pdf = 0.f;
  if (wi.z > 0.f && wo.z > 0.f) {
    const v3 h = crh_norm3(crh_add3(wi, wo));
    const float dih = crh_dot3(wi, h);
    if (lobe < 0 || lobe == 1) pdf = L.pd * (wi.z * CRH_INV_PI);
    if (b.Rc > kBsdfEps && (lobe < 0 || lobe == 0)) pdf = CRH_FMA(L.pc, blinn_pdf(h.z, dih, b.Rc), pdf);
    if (b.Rs > kBsdfEps && (lobe < 0 || lobe == 2)) pdf = CRH_FMA(L.ps, blinn_pdf(h.z, dih, b.Rs), pdf);
  }
  return pdf / L.total;
}

__device__ v3 sample_blinn(v3 wo, v3& wi, float4 fr, float rough, uint32_t& rng, int two_sided, bool& ok, int u32)
{
  const float k1 = crh_rng_next_mode(&rng, u32), k2 = crh_rng_next_mode(&rng, u32);
  const float e = blinn_power(rough);
  const float cm = crh_pow(k1, 1.0f / (e + 2.0f));
  float s, c; crh_sincos2pi(k2, &s, &c);
  const float sm = crh_sqrt(crh_max(CRH_FMA(-cm, cm, 1.0f), 0.f));
  const v3 m = crh_mk3(c * sm, s * sm, cm);
  bool flip = false;
  if (two_sided && wo.z < 0.f) { flip = true; wo.z = -wo.z; }
  const float cd = crh_dot3(wo, m);
  const float cd2 = 2.0f * cd;
  wi = crh_mk3(CRH_FMA(cd2, m.x, -wo.x), CRH_FMA(cd2, m.y, -wo.y), CRH_FMA(cd2, m.z, -wo.z));
  if (wi.z <= 0.f || wo.z <= 0.f || !(cd > 0.f)) { ok = false; return crh_mk3(0.f, 0.f, 0.f); }
  const float G = smith_g1(wo, m, rough) * smith_g1(wi, m, rough);
  const v3 F = fresnel_media(cd, fr);
  const float w = (G * cd) / (wo.z * m.z);
  if (flip) wi.z = -wi.z;
  ok = true;
  return crh_scale3(F, w);
}

// the crh_spec.h switches the BSDF code sees (wave-uniform)
struct SpecB { int u32; float eta_nd; };

// returns alive; W multiplied by the lobe weight; inside toggled on transmission; lobe = 0 coat / 1 diffuse / 2 glossy / 3 transmission
__device__ bool sample_layered(const Bsdf& b, v3 wo, v3& wi, v3& W, bool& inside, bool& delta, uint32_t& rng, int two_sided, SpecB sp, int& lobe)
{
  const Lobes L = lobe_probs(b, W);
  const float ksi = L.total * crh_rng_next_mode(&rng, sp.u32);
  delta = false; lobe = 0;
  if (!(L.total > kBsdfEps)) { W = crh_mk3(0.f, 0.f, 0.f); return false; }
  const v3 mirror = crh_mk3(-wo.x, -wo.y, wo.z);
  bool ok = true; v3 k;
  if (ksi < L.pc) {
    k = crh_scale3(b.Kc, L.total / L.pc);
    if (b.Rc > kBsdfEps) k = crh_mul3(k, sample_blinn(wo, wi, b.fc, b.Rc, rng, two_sided, ok, sp.u32));
    else { k = crh_mul3(k, b.Fc); wi = mirror; delta = true; }
  } else if (ksi < L.pc + L.pd) {
    k = crh_scale3(crh_mul3(b.Kd, L.Tc), L.total / L.pd); lobe = 1;
    const float k1 = crh_rng_next_mode(&rng, sp.u32), k2 = crh_rng_next_mode(&rng, sp.u32);
    float s, c; crh_sincos2pi(k1, &s, &c);
    const float r = crh_sqrt(k2);
    wi = crh_mk3(c * r, s * r, crh_sqrt(1.0f - k2));
    if (two_sided) { if (wo.z < 0.f) wi.z = -wi.z; }
    else if (!(wo.z > 0.f)) ok = false;
  } else if (ksi < (L.pc + L.pd) + L.ps) {
    k = crh_scale3(crh_mul3(b.Ks, L.Tc), L.total / L.ps); lobe = 2;
    if (b.Rs > kBsdfEps) k = crh_mul3(k, sample_blinn(wo, wi, b.fb, b.Rs, rng, two_sided, ok, sp.u32));
    else { k = crh_mul3(k, fresnel_media(wo.z, b.fb)); wi = mirror; delta = true; }
  } else {
    k = crh_scale3(crh_mul3(b.Kt, L.Tc), L.total / L.pt); lobe = 3;
    const float ior = b.fc.x > -2.5f ? sp.eta_nd : b.fc.y;   // no dielectric coat: crh_spec.eta_no_dielectric (default 1 = index-matched, straight through)
    const float eta = wo.z > 0.f ? 1.0f / ior : ior;
    const float sinT2 = (eta * eta) * CRH_FMA(-wo.z, wo.z, 1.0f);
    if (!(sinT2 < 1.0f) || !(L.pt > 0.f)) ok = false;
    else {
      float ct = crh_sqrt(1.0f - sinT2); if (wo.z > 0.f) ct = -ct;
      wi = crh_norm3(crh_mk3(-(eta * wo.x), -(eta * wo.y), ct));
      inside = !inside; delta = true;
    }
  }
  if (!ok) { W = crh_mk3(0.f, 0.f, 0.f); return false; }
  W = crh_mul3(W, k);
  return true;
}

"""A THIRD implementation of the layered BSDF -- float64 numpy, written from the prose of DESIGN.md section 3 ("BSDF (a8-a10)",
"Lights (a11)") and the textbook forms of the ingredients (exact unpolarised Fresnel equations, Blinn normalisation, Smith
shadowing), NOT from kernels.hip / crh_oracle.c, which are transliterations of each other.  tests/test_bsdf_independent.py
compares the oracle (CPU) and the gfx950 code (crh_debug_bsdf) against it within float tolerance, so that a formula error
shared by the twins shows up."""
import numpy as np

BSDF_EPS = 1e-5


def fresnel(cos_i, f):
    """f = Fresnel.Serialize(): tag in f[0].  Returns rgb (float64)."""
    tag = f[0]
    c = abs(cos_i)
    if tag > -0.5:                                   # Schlick: F0 + (1 - F0)(1 - c)^5, F0 = f[0:3]
        f0 = np.asarray(f[:3], np.float64)
        return f0 + (1.0 - f0) * (1.0 - c) ** 5
    if tag > -1.5:                                   # constant
        return np.full(3, float(f[2]))
    if tag > -2.5:                                   # conductor (n, k): the usual approximation (n^2 + k^2 terms)
        n, k = float(f[1]), float(f[2])
        n2k2 = n * n + k * k
        r_perp = (n2k2 - 2 * n * c + c * c) / (n2k2 + 2 * n * c + c * c)
        r_parl = (n2k2 * c * c - 2 * n * c + 1) / (n2k2 * c * c + 2 * n * c + 1)
        return np.full(3, 0.5 * (r_perp + r_parl))
    n = float(f[1])                                  # dielectric, signed cosine: > 0 entering (1 -> n), < 0 leaving (n -> 1)
    eta_i, eta_t = (1.0, n) if cos_i > 0 else (n, 1.0)
    sin_t2 = (eta_i / eta_t) ** 2 * (1.0 - cos_i * cos_i)
    if sin_t2 >= 1.0:
        return np.ones(3)                            # total internal reflection
    ct = np.sqrt(1.0 - sin_t2)
    r_parl = (eta_t * c - eta_i * ct) / (eta_t * c + eta_i * ct)
    r_perp = (eta_i * c - eta_t * ct) / (eta_i * c + eta_t * ct)
    return np.full(3, 0.5 * (r_parl ** 2 + r_perp ** 2))


def smith_g1(w, m, rough):
    """rational approximation of the Smith shadowing term for a Beckmann-like lobe, a = 1 / (rough * tan(theta))"""
    if np.dot(w, m) * w[2] <= 0:
        return 0.0
    tan_t = np.sqrt(max(1.0 - w[2] * w[2], 0.0)) / w[2]
    if tan_t == 0.0:
        return 1.0
    a = 1.0 / (rough * tan_t)
    return min((3.535 * a + 2.181 * a * a) / (1.0 + 2.276 * a + 2.577 * a * a), 1.0)


def blinn_exponent(rough):
    return max(2.0 / (rough * rough) - 2.0, 0.0)


def blinn_lobe(wi, wo, fr, rough):
    """f * cos(theta_i) of the microfacet lobe: D G F / (4 cos_o)  [the cos_i of the rendering equation cancels the 1 / cos_i]"""
    if wi[2] <= 0 or wo[2] <= 0:
        return np.zeros(3)
    h = wi + wo; h = h / np.linalg.norm(h)
    e = blinn_exponent(rough)
    D = (e + 2.0) / (2.0 * np.pi) * h[2] ** e
    G = smith_g1(wo, h, rough) * smith_g1(wi, h, rough)
    return fresnel(float(np.dot(wo, h)), fr) * (D * G / (4.0 * wo[2]))


def _parts(b):
    Kc, Rc = np.asarray(b.Kc[:3], np.float64), float(b.Kc[3])
    Ks, Rs = np.asarray(b.Ks[:3], np.float64), float(b.Ks[3])
    return Kc, Rc, np.asarray(b.Kd[:3], np.float64), Ks, Rs, np.asarray(b.Kt[:3], np.float64), b.FresnelCoat.Serialize(), b.FresnelBase.Serialize()


def eval_fcos(b, wo, wi, two_sided=True):
    """layered f(wo, wi) * cos: [Lambert Kd / pi + Ks * Blinn(base Fresnel)] attenuated by (1 - F_coat(wo)), plus Kc * Blinn(coat Fresnel);
    delta lobes (roughness <= 1e-5) contribute nothing to an evaluation"""
    wo, wi = np.array(wo, np.float64), np.array(wi, np.float64)
    Kc, Rc, Kd, Ks, Rs, Kt, fc, fb = _parts(b)
    Fc = fresnel(float(wo[2]), fc)                   # coat Fresnel at the (unmirrored) outgoing direction
    if two_sided and wo[2] < 0:
        wo[2], wi[2] = -wo[2], -wi[2]
    lam = wi[2] / np.pi if (wi[2] > 0 and wo[2] > 0) else 0.0
    r = Kd * lam
    if Rs > BSDF_EPS:
        r = r + Ks * blinn_lobe(wi, wo, fb, Rs)
    r = r * (1.0 - Fc)
    if Rc > BSDF_EPS:
        r = r + Kc * blinn_lobe(wi, wo, fc, Rc)
    return r


def blinn_pdf(wi, wo, rough):
    h = wi + wo; h = h / np.linalg.norm(h)
    e = blinn_exponent(rough)
    return (e + 2.0) / (2.0 * np.pi) * abs(h[2]) ** (e + 1.0) / (4.0 * abs(np.dot(wi, h)))      # p(h) * dh/dwi


def pdf(b, wo, wi, weight=(1.0, 1.0, 1.0), two_sided=True):
    """lobe-mixture density of sampling wi: lobe probabilities proportional to <K * coat factor, path weight>"""
    wo, wi, W = np.array(wo, np.float64), np.array(wi, np.float64), np.asarray(weight, np.float64)
    Kc, Rc, Kd, Ks, Rs, Kt, fc, fb = _parts(b)
    Fc = fresnel(float(wo[2]), fc); Tc = 1.0 - Fc
    pc, pd, ps, pt = np.dot(Kc * Fc, W), np.dot(Kd * Tc, W), np.dot(Ks * Tc, W), np.dot(Kt * Tc, W)
    total = pc + pd + ps + pt
    if not total > BSDF_EPS:
        return 0.0
    if two_sided and wo[2] < 0:
        wo[2], wi[2] = -wo[2], -wi[2]
    p = 0.0
    if wi[2] > 0 and wo[2] > 0:
        p = pd * wi[2] / np.pi
        if Rc > BSDF_EPS:
            p += pc * blinn_pdf(wi, wo, Rc)
        if Rs > BSDF_EPS:
            p += ps * blinn_pdf(wi, wo, Rs)
    return p / total

"""Scene inputs for the path: the synthetic generator of BASELINE.json's configs and the two
fixtures the reference ships (data/scripts/CornellBox.tcl, data/scripts/Materials.tcl), restated
as plain arrays.  Inputs only -- no rendering arithmetic lives here.
"""
import math
from dataclasses import dataclass, field
from typing import List, Optional
import numpy as np

from .materials import BSDF, Fresnel


@dataclass
class Light:
    """V3d directional / positional light (src/Launcher/LightSourcesEditor.cxx:242-310)."""
    vec: tuple
    is_point: bool
    color: tuple = (1.0, 1.0, 1.0)
    intensity: float = 1.0
    smoothness: float = 0.0     # directional: cone half-angle [rad]; positional: sphere radius

    @staticmethod
    def directional(direction, smoothness=0.0, intensity=1.0, color=(1, 1, 1)):
        return Light(tuple(direction), False, tuple(color), intensity, smoothness)

    @staticmethod
    def positional(position, smoothness=0.0, intensity=1.0, color=(1, 1, 1)):
        return Light(tuple(position), True, tuple(color), intensity, smoothness)


@dataclass
class Camera:
    eye: tuple = (0.0, -3.6, 0.0)
    dir: tuple = (0.0, 1.0, 0.0)
    up: tuple = (0.0, 0.0, 1.0)
    fovy_deg: float = 45.0
    aspect: float = 0.0
    is_ortho: bool = False
    ortho_scale: float = 1.0
    aperture_radius: float = 0.0
    focal_dist: float = 1.0


@dataclass
class Params:
    """Graphic3d_RenderingParams subset; defaults = the reference's GI defaults
    (src/Launcher/SettingsWidget.cxx:65-90, data/scripts/CornellBox.tcl:76)."""
    width: int = 512
    height: int = 512
    max_depth: int = 5
    radiance_clamp: float = 0.0
    two_sided: bool = True
    coherent_rng: bool = False
    seed: int = 1
    tile_size: int = 32
    tonemap_mode: int = 0
    exposure: float = 0.0
    white_point: float = 1.0
    background: tuple = (0.0, 0.0, 0.0)
    env_as_background: bool = True
    scene_epsilon: float = 0.0
    russian_roulette: bool = True


@dataclass
class Scene:
    pos: np.ndarray
    nrm: np.ndarray
    tri: np.ndarray                      # (nT, 4) int32: i0, i1, i2, material
    materials: List[BSDF]
    lights: List[Light] = field(default_factory=list)
    env: Optional[np.ndarray] = None     # (H, W, 3) float32 linear lat-long, row 0 = zenith (+Z)
    camera: Camera = field(default_factory=Camera)
    params: Params = field(default_factory=Params)
    uv: Optional[np.ndarray] = None
    textures: List[np.ndarray] = field(default_factory=list)   # slot -> (H, W, 3) float32 linear RGB, row 0 = v 1
    tri_object: Optional[np.ndarray] = None   # (nT,) int32 object id per triangle  } two-level BVH: vertices in object space,
    obj_xform: Optional[np.ndarray] = None    # (nO, 12) row-major 3x4 transforms   } one tree per object + a top-level tree
    name: str = "scene"
    spec: Optional[dict] = None               # switches of include/crh_spec.h that differ from the defaults (None: the frozen spec)


# ---------------------------------------------------------------------------------------------
def splitmix64_uniform(seed, n, first=0):
    """n doubles in [0,1) from the splitmix64 stream of `seed`, starting at its element `first` (vectorised, in cache-sized chunks: at 90 M
    elements -- the 10 M-triangle scene -- whole-array temporaries made this 27 s instead of 1.3 s)."""
    n = int(n)
    out = np.empty(n, np.float64)
    chunk = 1 << 18
    with np.errstate(over="ignore"):
        for a in range(0, n, chunk):
            b = min(n, a + chunk)
            z = np.arange(first + a + 1, first + b + 1, dtype=np.uint64)
            z *= np.uint64(0x9E3779B97F4A7C15); z += np.uint64(seed)
            t = z >> np.uint64(30); z ^= t; z *= np.uint64(0xBF58476D1CE4E5B9)
            np.right_shift(z, np.uint64(27), out=t); z ^= t; z *= np.uint64(0x94D049BB133111EB)
            np.right_shift(z, np.uint64(31), out=t); z ^= t
            z >>= np.uint64(11)                                       # < 2^53: the conversion to double below is exact
            np.multiply(z, 1.0 / 9007199254740992.0, out=out[a:b])
    return out


def gen_scene(n_tris, seed=1, n_materials=1):
    """SURVEY.md section 8(d) generator: centre ~ U([-1,1]^3); two edges ~ U([-1,1]^3) * r with
    r = 1.5 * N^(-1/3); flat normals; material = i mod M.  Returns pos, nrm, tri."""
    n = int(n_tris)
    r = 1.5 * float(n) ** (-1.0 / 3.0)
    pos = np.empty((3 * n, 3), np.float32)
    nrm = np.empty((3 * n, 3), np.float32)
    chunk = 1 << 17                                    # triangles per pass: the float64 temporaries stay in the caches
    for a in range(0, n, chunk):
        b = min(n, a + chunk)
        u = splitmix64_uniform(seed, 9 * (b - a), 9 * a).reshape(b - a, 9) * 2.0 - 1.0
        c = u[:, 0:3]
        e1 = u[:, 3:6] * r
        e2 = u[:, 6:9] * r
        v = np.stack([c, c + e1, c + e2], axis=1).astype(np.float32)          # (m, 3, 3)
        fn = np.cross((v[:, 1] - v[:, 0]).astype(np.float64), (v[:, 2] - v[:, 0]).astype(np.float64))
        ln = np.linalg.norm(fn, axis=1, keepdims=True)
        fn = np.where(ln > 0, fn / np.maximum(ln, 1e-300), np.array([0.0, 0.0, 1.0]))
        pos[3 * a:3 * b] = v.reshape(3 * (b - a), 3)
        nrm[3 * a:3 * b] = np.repeat(fn.astype(np.float32), 3, axis=0)
    tri = np.empty((n, 4), np.int32)
    base = np.arange(n, dtype=np.int32) * 3
    tri[:, 0], tri[:, 1], tri[:, 2] = base, base + 1, base + 2
    tri[:, 3] = np.arange(n, dtype=np.int32) % max(int(n_materials), 1)
    return pos, nrm, tri


def procedural_sky(w=2048, h=1024, seed=1, sun=5.0e4):
    """Synthetic float lat-long HDR sky (the reference has no HDR loader):
    zenith-horizon gradient, ground, and a ~1 degree sun disc of radiance `sun`."""
    rng = splitmix64_uniform(seed, 3)
    az = 2.0 * np.pi * rng[0]
    el = np.deg2rad(35.0 + 30.0 * rng[1])
    sd = np.array([np.cos(el) * np.cos(az), np.cos(el) * np.sin(az), np.sin(el)])
    v = (np.arange(h) + 0.5) / h
    u = (np.arange(w) + 0.5) / w
    theta = v * np.pi                       # angle from +Z
    phi = u * 2.0 * np.pi - np.pi
    dz = np.cos(theta)[:, None] * np.ones((1, w))
    dx = np.sin(theta)[:, None] * np.cos(phi)[None, :]
    dy = np.sin(theta)[:, None] * np.sin(phi)[None, :]
    t = np.clip(dz, 0.0, 1.0)[..., None]
    zen = np.array([0.25, 0.45, 1.0]) * 1.4
    hor = np.array([0.9, 0.95, 1.0]) * 1.0
    sky = hor * (1.0 - t) + zen * t
    gnd = np.array([0.30, 0.28, 0.25]) * np.ones_like(sky)
    img = np.where(dz[..., None] >= 0.0, sky, gnd)
    cosang = dx * sd[0] + dy * sd[1] + dz * sd[2]
    img = np.where((cosang > np.cos(np.deg2rad(1.0)))[..., None], np.array([sun, sun * 0.95, sun * 0.85]), img)
    return np.ascontiguousarray(img.astype(np.float32))


# ---------------------------------------------------------------------------------------------
class _Mesh:
    def __init__(self):
        self.pos, self.nrm, self.tri = [], [], []
        self.nv = 0

    def add(self, pos, nrm, faces, mat):
        pos = np.asarray(pos, np.float32)
        nrm = np.asarray(nrm, np.float32)
        faces = np.asarray(faces, np.int32)
        t = np.empty((len(faces), 4), np.int32)
        t[:, :3] = faces + self.nv
        t[:, 3] = mat
        self.pos.append(pos)
        self.nrm.append(nrm)
        self.tri.append(t)
        self.nv += len(pos)

    def quad(self, p0, p1, p2, p3, normal, mat):
        self.add([p0, p1, p2, p3], [normal] * 4, [[0, 1, 2], [0, 2, 3]], mat)

    def box(self, size, mat, translate=(0, 0, 0), rot_z_deg=0.0):
        sx, sy, sz = size
        c, s = np.cos(np.deg2rad(rot_z_deg)), np.sin(np.deg2rad(rot_z_deg))
        R = np.array([[c, -s, 0], [s, c, 0], [0, 0, 1.0]])
        T = np.asarray(translate, np.float64)
        corners = np.array([[x, y, z] for z in (0, sz) for y in (0, sy) for x in (0, sx)], np.float64)
        faces = [((0, 2, 3, 1), (0, 0, -1)), ((4, 5, 7, 6), (0, 0, 1)), ((0, 1, 5, 4), (0, -1, 0)),
                 ((2, 6, 7, 3), (0, 1, 0)), ((0, 4, 6, 2), (-1, 0, 0)), ((1, 3, 7, 5), (1, 0, 0))]
        for idx, n in faces:
            p = [(R @ corners[i]) + T for i in idx]
            self.quad(p[0], p[1], p[2], p[3], R @ np.asarray(n, np.float64), mat)

    def sphere(self, center, radius, mat, n_lon=32, n_lat=16):
        pts, nrm = [], []
        for j in range(n_lat + 1):
            th = np.pi * j / n_lat
            for i in range(n_lon):
                ph = 2 * np.pi * i / n_lon
                n = np.array([np.sin(th) * np.cos(ph), np.sin(th) * np.sin(ph), np.cos(th)])
                pts.append(np.asarray(center) + radius * n)
                nrm.append(n)
        faces = []
        for j in range(n_lat):
            for i in range(n_lon):
                a = j * n_lon + i
                b = j * n_lon + (i + 1) % n_lon
                c = (j + 1) * n_lon + i
                d = (j + 1) * n_lon + (i + 1) % n_lon
                if j > 0:
                    faces.append([a, c, b])
                if j < n_lat - 1:
                    faces.append([b, c, d])
        self.add(pts, nrm, faces, mat)

    def arrays(self):
        return (np.ascontiguousarray(np.concatenate(self.pos)), np.ascontiguousarray(np.concatenate(self.nrm)),
                np.ascontiguousarray(np.concatenate(self.tri)))


def cornell_box(full=False, width=512, height=512):
    """data/scripts/CornellBox.tcl:10-76 restated.  Unit cube, front face (y = 0) open, camera on -Y
    looking +Y (vfront), sphere light at (.5,.5,.85) r=.06 intensity 25 (:12-14), depth 5 (:76).
    x=1 wall Kd (1,.3,.3) (:34), x=0 wall Kd (.3,.5,1) (:35), others Kd 1 (:36-38).
    full=False: BASELINE config C1 (walls + the two inner boxes, every BSDF diffuse-only, 34 triangles).
    full=True : adds the glass sphere (:44-49), glass box (:60-66) and the mirror-like sphere (:69-74)."""
    m = _Mesh()
    mats = [BSDF.CreateDiffuse((1.0, 0.3, 0.3)), BSDF.CreateDiffuse((0.3, 0.5, 1.0)), BSDF.CreateDiffuse(1.0)]
    RED, BLUE, WHITE = 0, 1, 2
    m.quad((1, 0, 0), (1, 1, 0), (1, 1, 1), (1, 0, 1), (-1, 0, 0), RED)
    m.quad((0, 0, 0), (0, 0, 1), (0, 1, 1), (0, 1, 0), (1, 0, 0), BLUE)
    m.quad((0, 1, 0), (0, 1, 1), (1, 1, 1), (1, 1, 0), (0, -1, 0), WHITE)
    m.quad((0, 0, 1), (1, 0, 1), (1, 1, 1), (0, 1, 1), (0, 0, -1), WHITE)
    m.quad((0, 0, 0), (0, 1, 0), (1, 1, 0), (1, 0, 0), (0, 0, 1), WHITE)
    # first inner box: 0.3 x 0.3 x 0.2 at (.55,.3,0), -30 deg about z (:52-57); -kd 1 .8 .2 -ks .3 -n
    c = BSDF.CreateDiffuse((1.0, 0.8, 0.2))
    if full:
        c.Ks = np.array([0.3, 0.3, 0.3, 0.1], np.float32)
        c.FresnelBase = Fresnel.CreateSchlick((0.8, 0.8, 0.8))
    else:
        c.Ks = np.array([0.3, 0.3, 0.3, 0.0], np.float32)
    c.Normalize()
    if not full:
        c.Ks[:] = 0
    mats.append(c)
    m.box((0.3, 0.3, 0.2), len(mats) - 1, (0.55, 0.3, 0.0), -30.0)
    # second inner box: 0.15 x 0.15 x 0.3 at (.7,.25,.2), +10 deg (:60-66); glass in the script
    if full:
        g = BSDF.CreateGlass(1.0, (0.8, 1.0, 0.8), 6.0, 1.5)
    else:
        g = BSDF.CreateDiffuse((0.8, 1.0, 0.8))
    mats.append(g)
    m.box((0.15, 0.15, 0.3), len(mats) - 1, (0.7, 0.25, 0.2), 10.0)
    if full:
        mats.append(BSDF.CreateGlass(1.0, (0.8, 0.8, 1.0), 6.0, 1.5))
        m.sphere((0.21, 0.3, 0.2), 0.2, len(mats) - 1)
        r = BSDF.CreateDiffuse((0.5, 0.9, 0.3))
        r.Ks = np.array([0.3, 0.3, 0.3, 0.0], np.float32)
        r.FresnelBase = Fresnel.CreateConstant(1.0)
        r.Normalize()
        mats.append(r)
        m.sphere((0.5, 0.65, 0.1), 0.1, len(mats) - 1)
    pos, nrm, tri = m.arrays()
    return Scene(pos, nrm, tri, mats,
                 lights=[Light.positional((0.5, 0.5, 0.85), smoothness=0.06, intensity=25.0)],
                 camera=Camera(eye=(0.5, -1.45, 0.5), dir=(0, 1, 0), up=(0, 0, 1), fovy_deg=45.0),
                 params=Params(width=width, height=height, max_depth=5, seed=1),
                 name="cornell_full" if full else "cornell")


def materials_scene(width=512, height=384, n_lon=48, n_lat=24):
    """data/scripts/Materials.tcl:9-203 restated: nine r=10 balls with the script's BSDF vectors on a
    12x12 checker floor (Kd .85/.45), camera (:193-199), directional light (:203)."""
    m = _Mesh()
    mats = [BSDF.CreateDiffuse(0.85), BSDF.CreateDiffuse(0.45)]
    for i in range(12):
        for j in range(1, 13):
            m.box((10, 10, 0.1), 0 if (i + j) % 2 == 0 else 1, (i * 10 - 90, j * 10 - 70, -0.15))

    def ball(loc, **kw):
        b = BSDF()
        b.Kc = np.array(list(kw.get("Kc", (0, 0, 0))) + [kw.get("coatRoughness", 0.0)], np.float32)
        b.Kd = np.array(kw.get("Kd", (0, 0, 0)), np.float32)
        b.Ks = np.array(list(kw.get("Ks", (0, 0, 0))) + [kw.get("baseRoughness", 0.0)], np.float32)
        b.Kt = np.array(kw.get("Kt", (0, 0, 0)), np.float32)
        b.Le = np.array(kw.get("Le", (0, 0, 0)), np.float32)
        b.Absorption = np.array(list(kw.get("absorpColor", (0, 0, 0))) + [kw.get("absorpCoeff", 0.0)], np.float32)
        b.FresnelCoat = kw.get("coatFresnel", Fresnel.CreateConstant(0.0))
        b.FresnelBase = kw.get("baseFresnel", Fresnel.CreateConstant(1.0))
        mats.append(b)
        m.sphere(loc, 10.0, len(mats) - 1, n_lon, n_lat)

    gold = Fresnel.CreateSchlick((0.58, 0.42, 0.2))
    ball((10, 0, 10), Kd=(0.272798, 0.746262, 0.104794), Ks=(0.253738,) * 3, baseRoughness=0.045, baseFresnel=gold)      # Ball1 :39-54
    ball((10, 40, 10), Kd=(0.8, 0.8, 0.8), Le=(2.02, 0.171915, 0.171915))                                                 # Ball2 :57-71
    ball((-30, -40, 10), Kc=(1, 1, 1), Kt=(1, 1, 1), absorpColor=(0.75, 0.95, 0.9), absorpCoeff=0.05,
         coatFresnel=Fresnel.CreateDielectric(1.62))                                                                       # Ball3 :74-88
    ball((-70, -40, 10), Ks=(0.985,) * 3, baseFresnel=gold)                                                               # Ball4 :91-105
    ball((-30, 0, 10), Kc=(1, 1, 1), Kt=(1, 1, 1), absorpColor=(0, 0.288061, 0.825532), absorpCoeff=0.3,
         coatFresnel=Fresnel.CreateDielectric(1.62))                                                                       # Ball5 :108-122
    ball((-30, 40, 10), Kc=(1, 1, 1), Kd=(0, 0.716033, 0.884507), Ks=(0.115493,) * 3, baseRoughness=0.045,
         coatFresnel=Fresnel.CreateDielectric(1.5), baseFresnel=gold)                                                      # Ball6 :125-139
    ball((-70, 0, 10), Kc=(1, 1, 1), Kd=(1e-06, 9.9999e-07, 9.9999e-07), Ks=(0.0479573, 0.804998, 0), baseRoughness=0.447,
         coatFresnel=Fresnel.CreateDielectric(1.5), baseFresnel=gold)                                                      # Ball7 :142-156
    ball((-70, 40, 10), Ks=(0.985,) * 3, baseRoughness=0.026,
         baseFresnel=Fresnel.CreateSchlick((0.913183, 0.921494, 0.924524)))                                                # Ball8 :159-173
    ball((10, -40, 10), Kd=(0.723404, 0.166229, 0.166229))                                                                 # Ball0 :176-190
    pos, nrm, tri = m.arrays()
    eye = np.array([139.412, -1.62643, 178.037])
    at = np.array([-22.3025, 0.0986351, 3.30327])
    return Scene(pos, nrm, tri, mats,
                 lights=[Light.directional((-0.303949, -0.434084, -0.848048), smoothness=0.3, intensity=12.0)],
                 camera=Camera(eye=tuple(eye), dir=tuple(at - eye), up=(-0.733931, -0.00311795, 0.679217), fovy_deg=25.0),
                 params=Params(width=width, height=height, max_depth=10, seed=1, background=(0.4225,) * 3),
                 name="materials")


DEFAULT_LIGHT = dict(direction=(-0.25, -1.0, -1.0), smoothness=0.3, intensity=10.0)   # AppGui.cxx:957


def gen_cad_like(n_tris=1_000_000, seed=1, grid=0, with_objects=False):
    """A synthetic scene shaped like what CADRays really renders (round-5 verdict, weak 8): TESSELLATED CAD SURFACES, as src/ImportExport/AisMesh.cxx:357-423
    hands them over -- indexed meshes with shared vertices and smooth per-vertex normals -- instead of a soup of unrelated triangles:

      * a grid^3 assembly of parts inside [-1, 1]^3 (grid = 0: about 580 triangles per part, 12^3 parts at a million triangles): tori, capped cylinders, boxes and thin plates, one per cell, chosen by a splitmix64 stream;
      * ANISOTROPIC tessellation, the way a mesher with an angular deflection refines curved directions only: a cylinder wall is one long strip per
        angular step (triangles ~50 - 100 times longer than wide), a torus is fine along the tube and coarse around it, box faces are cut into strips;
      * TOUCHING parts: a box or a plate fills its cell exactly along one or two axes, so neighbouring boxes meet in COINCIDENT faces (same plane, opposite
        normals, different tessellation) -- several per cent of the total area -- and cylinder caps lie in the plane of the box face above them;
      * three materials (diffuse, glossy, glass), picked per part.

    Returns (pos, nrm, tri) like gen_scene, + tri_object (the part of every triangle) when with_objects.  The triangle count is met within a few per cent
    (each part gets the same budget; the exact number is whatever the tessellation parameters give)."""
    G = int(grid) if grid else int(min(16, max(2, round((n_tris / 580.0) ** (1.0 / 3.0)))))
    u = splitmix64_uniform(seed, 8 * G ** 3)
    budget = max(64, n_tris // G ** 3)
    cell = 2.0 / G
    P, Nn, T, OB = [], [], [], []
    nv = 0

    def emit(pos, nrm, faces, mat, ob):
        nonlocal nv
        P.append(pos.astype(np.float32)); Nn.append(nrm.astype(np.float32))
        t = np.empty((len(faces), 4), np.int32); t[:, :3] = faces + nv; t[:, 3] = mat
        T.append(t); OB.append(np.full(len(faces), ob, np.int32)); nv += len(pos)

    def grid_faces(nu, nvv, wrap_u, wrap_v):
        """two triangles per cell of an nu x nvv parameter grid; vertex (i, j) has index i * cols + j"""
        cols = nvv if wrap_v else nvv + 1
        i, j = np.meshgrid(np.arange(nu), np.arange(nvv), indexing="ij")
        i1 = (i + 1) % nu if wrap_u else i + 1
        j1 = (j + 1) % nvv if wrap_v else j + 1
        a, b, c, d = i * cols + j, i1 * cols + j, i1 * cols + j1, i * cols + j1
        return np.concatenate([np.stack([a, b, c], -1).reshape(-1, 3), np.stack([a, c, d], -1).reshape(-1, 3)])

    def box_part(lo, hi, n_strips, mat, ob):
        # six faces, each cut into n_strips long strips across its first axis (shared vertices along the strip borders, flat normals)
        for ax in range(3):
            a1, a2 = (ax + 1) % 3, (ax + 2) % 3
            for side in (0, 1):
                s_ = np.linspace(lo[a1], hi[a1], n_strips + 1); t_ = np.array([lo[a2], hi[a2]])
                ss, tt = np.meshgrid(s_, t_, indexing="ij")
                pos = np.zeros((ss.size, 3)); pos[:, a1] = ss.ravel(); pos[:, a2] = tt.ravel(); pos[:, ax] = hi[ax] if side else lo[ax]
                nrm = np.zeros_like(pos); nrm[:, ax] = 1.0 if side else -1.0
                f = grid_faces(n_strips, 1, False, False)
                emit(pos, nrm, f if side else f[:, ::-1], mat, ob)

    k = 0
    for ix in range(G):
        for iy in range(G):
            for iz in range(G):
                r = u[8 * k:8 * k + 8]; ob = k; k += 1
                c = np.array([-1.0 + (ix + 0.5) * cell, -1.0 + (iy + 0.5) * cell, -1.0 + (iz + 0.5) * cell])
                kind = int(r[0] * 4.0)
                mat = int(r[1] * 3.0)
                h = 0.5 * cell
                if kind == 0:                                           # torus: fine along the tube (nu), coarse around it (nvv)
                    nvv = 12; nu = max(8, budget // (2 * nvv))
                    R, rr = 0.30 * cell, (0.08 + 0.08 * r[2]) * cell
                    th, ph = np.meshgrid(np.arange(nu) * (2 * math.pi / nu), np.arange(nvv) * (2 * math.pi / nvv), indexing="ij")
                    axis = int(r[3] * 3.0)
                    x = (R + rr * np.cos(ph)) * np.cos(th); y = (R + rr * np.cos(ph)) * np.sin(th); z = rr * np.sin(ph)
                    nx, ny, nz = np.cos(ph) * np.cos(th), np.cos(ph) * np.sin(th), np.sin(ph)
                    pos = np.stack([x, y, z], -1).reshape(-1, 3); nrm = np.stack([nx, ny, nz], -1).reshape(-1, 3)
                    perm = [(0, 1, 2), (2, 0, 1), (1, 2, 0)][axis]
                    emit(pos[:, perm] + c, nrm[:, perm], grid_faces(nu, nvv, True, True), mat, ob)
                elif kind == 1:                                         # capped cylinder: ONE strip along the axis per angular step, fan caps
                    nu = max(8, budget // 4)
                    rad = (0.25 + 0.15 * r[2]) * cell
                    axis = int(r[3] * 3.0)
                    th = np.arange(nu) * (2 * math.pi / nu)
                    ring = np.stack([rad * np.cos(th), rad * np.sin(th)], -1)
                    wall = np.zeros((nu, 2, 3)); wall[:, :, 0] = ring[:, None, 0]; wall[:, :, 1] = ring[:, None, 1]; wall[:, 0, 2] = -h; wall[:, 1, 2] = h      # caps in the cell's faces
                    wn = np.zeros((nu, 2, 3)); wn[:, :, 0] = np.cos(th)[:, None]; wn[:, :, 1] = np.sin(th)[:, None]
                    perm = [(0, 1, 2), (2, 0, 1), (1, 2, 0)][axis]
                    emit(wall.reshape(-1, 3)[:, perm] + c, wn.reshape(-1, 3)[:, perm], grid_faces(nu, 1, True, False), mat, ob)
                    for side, zc in ((0, -h), (1, h)):
                        pos = np.zeros((nu + 1, 3)); pos[:nu, 0] = ring[:, 0]; pos[:nu, 1] = ring[:, 1]; pos[:, 2] = zc
                        nrm = np.zeros_like(pos); nrm[:, 2] = 1.0 if side else -1.0
                        i = np.arange(nu); f = np.stack([np.full(nu, nu), i, (i + 1) % nu], -1)
                        emit(pos[:, perm] + c, nrm[:, perm], f if side else f[:, ::-1], mat, ob)
                elif kind == 2:                                         # box filling the cell along two axes: coincident faces with its neighbours
                    axis = int(r[3] * 3.0)
                    half = np.full(3, h); half[axis] = (0.2 + 0.25 * r[2]) * cell
                    box_part(c - half, c + half, max(2, budget // 12), mat, ob)
                else:                                                   # thin plate across the whole cell
                    axis = int(r[3] * 3.0)
                    half = np.full(3, h); half[axis] = 0.02 * cell
                    off = np.zeros(3); off[axis] = (r[2] - 0.5) * 0.6 * cell
                    box_part(c + off - half, c + off + half, max(2, budget // 12), mat, ob)
    pos, nrm, tri = np.concatenate(P), np.concatenate(Nn), np.concatenate(T)
    if with_objects:
        return pos, nrm, tri, np.concatenate(OB)
    return pos, nrm, tri


def baseline_config(which, width=None, height=None, n_tris=None):
    """BASELINE.json configs as Scene objects (SURVEY.md section 8(d) 'Synthetic inputs').
    which: 'C1' Cornell 512^2; 'C2' 100k diffuse, 1080p; 'C3' 1M glass+glossy + HDR sky, 1080p;
    'C5' 10M, 4K.  Sizes can be overridden for parity-test scale."""
    if which == "C1":
        return cornell_box(False, width or 512, height or 512)
    if which == "C2":
        n = n_tris or 100_000
        pos, nrm, tri = gen_scene(n, 1, 1)
        return Scene(pos, nrm, tri, [BSDF.CreateDiffuse(0.8)],
                     lights=[Light.directional(**DEFAULT_LIGHT)], env=None,
                     camera=Camera(),
                     params=Params(width=width or 1920, height=height or 1080, max_depth=5, radiance_clamp=30.0,
                                   seed=1, background=(0.65, 0.65, 0.65)),
                     name=f"C2_{n}")
    if which in ("C3", "C4", "C5"):
        n = n_tris or (10_000_000 if which == "C5" else 1_000_000)
        pos, nrm, tri = gen_scene(n, 1, 2)
        glass = BSDF.CreateGlass(1.0, (0.8, 0.8, 1.0), 6.0, 1.5)        # CornellBox.tcl:47-49, MaterialEditor.cxx:785-809
        glossy = BSDF.Glossy(0.5, 0.5, 0.1, 0.8)                         # MaterialEditor.cxx:733-757
        w, h = (3840, 2160) if which == "C5" else (1920, 1080)
        return Scene(pos, nrm, tri, [glass, glossy], lights=[], env=procedural_sky(2048, 1024, 1),
                     camera=Camera(),
                     params=Params(width=width or w, height=height or h, max_depth=10, radiance_clamp=30.0, seed=1),
                     name=f"{which}_{n}")
    if which == "CAD1M":
        # the CAD-like leg beside the soup (round-5 verdict, item 4): tessellated parts, diffuse + glossy + glass, the application's default light
        # (AppGui.cxx:957) + the sky of C3, 1080p, depth 10
        n = n_tris or 1_000_000
        pos, nrm, tri = gen_cad_like(n, 1)
        glass = BSDF.CreateGlass(1.0, (0.8, 0.8, 1.0), 6.0, 1.5)
        return Scene(pos, nrm, tri, [BSDF.CreateDiffuse(0.8), BSDF.Glossy(0.5, 0.5, 0.1, 0.8), glass],
                     lights=[Light.directional(**DEFAULT_LIGHT)], env=procedural_sky(2048, 1024, 1), camera=Camera(),
                     params=Params(width=width or 1920, height=height or 1080, max_depth=10, radiance_clamp=30.0, seed=1),
                     name=f"CAD1M_{len(tri)}")
    raise ValueError(which)


def spec_switch_scene(width=96, height=80):
    """A small scene in which EVERY switch of include/crh_spec.h changes the image (the reference's own CornellBox.tcl / Materials.tcl
    have no texture, no environment map and no transmitting material under a non-dielectric coat, so they cannot tell switches 2 and 7
    apart): the Cornell room of data/scripts/CornellBox.tcl with an 8-bit-range environment and a textured, coated, glossy wall material
    (texel_gamma2), a sphere light + a cone light over glossy and coated lobes (mis_single_lobe), a transmitting box under a Schlick
    coat (eta_no_dielectric), several bounces (eps_rule) -- and random numbers everywhere (uniform_32bit).  Exportable in the
    application's own scene format (cadrays_amd.scene_tcl.write_scene), so the real renderer can load it: tools/occt_pin."""
    import dataclasses
    sc = cornell_box(True, width, height)
    mats = list(sc.materials)
    thin = BSDF.CreateDiffuse((0.2, 0.2, 0.2))
    thin.Kt = np.array([0.7, 0.7, 0.7], np.float32)
    thin.Kc = np.array([0.3, 0.3, 0.3, 0.2], np.float32)
    thin.FresnelCoat = Fresnel.CreateSchlick((0.1, 0.1, 0.1))
    thin.Normalize()
    mats[4] = thin                                     # the second inner box: transmission under a NON-dielectric coat
    glossy = BSDF.CreateDiffuse((0.4, 0.4, 0.4))
    glossy.Ks = np.array([0.5, 0.5, 0.5, 0.15], np.float32); glossy.FresnelBase = Fresnel.CreateSchlick((0.8, 0.8, 0.8))
    glossy.Kc = np.array([0.6, 0.6, 0.6, 0.25], np.float32); glossy.FresnelCoat = Fresnel.CreateDielectric(1.5)
    glossy.texture = 0
    glossy.Normalize()
    mats[2] = glossy                                   # the white walls: diffuse + glossy + rough coat, textured
    r = np.random.default_rng(3)
    uv = r.random((len(sc.pos), 2)).astype(np.float32)
    tex = (np.round((0.25 + 0.75 * r.random((5, 7, 4))) * 255.0) / 255.0).astype(np.float32)      # exactly representable in an 8-bit PNG
    tex[..., :3] = tex[..., :3] * tex[..., :3]                                                      # as a reader hands it over: rgb squared
    env = np.clip(procedural_sky(32, 16, 2, sun=1.0), 0.0, 1.0)
    env = (np.round(np.sqrt(env) * 255.0) / 255.0).astype(np.float32) ** 2                          # an 8-bit image, linearised by squaring
    par = dataclasses.replace(sc.params, max_depth=6, env_as_background=True)
    cam = dataclasses.replace(sc.camera, eye=(0.5, -1.2, 0.5))
    return dataclasses.replace(sc, materials=mats, uv=uv, textures=[tex], env=env.astype(np.float32),
                               lights=[Light.positional((0.5, 0.5, 0.85), smoothness=0.12, intensity=12.0),
                                       Light.directional((-0.2, 0.4, -1.0), smoothness=0.2, intensity=2.0)],
                               params=par, camera=cam, name="switches")

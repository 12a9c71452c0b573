"""Host-side mirror of the material contract CADRays drives: Graphic3d_BSDF / Graphic3d_Fresnel.

Names and argument meaning follow the reference's call sites:
  Graphic3d_Fresnel::CreateConstant / CreateSchlick / CreateConductor / CreateDielectric
      (src/Launcher/MaterialEditor.cxx:177-201), Serialize() layout (:209-255,
      src/ImportExport/ImportExport.cxx:197-229)
  Graphic3d_BSDF::CreateDiffuse / CreateMetallic / CreateGlass, Normalize()
      (MaterialEditor.cxx:670, 692, 796; src/ImportExport/AisMesh.cxx:247, 319)
  MaterialEditor::setBSDF clamp + energy normalisation (MaterialEditor.cxx:281-338)
  getMaterialType class predicate (MaterialEditor.cxx:350-370)
  preset reset values (MaterialEditor.cxx:664-944)
"""
from dataclasses import dataclass, field
import numpy as np

from . import abi


@dataclass
class Fresnel:
    """Serialised as the vec4 the reference writes (MaterialEditor.cxx:209-255)."""
    kind: str = "constant"
    data: tuple = (0.0, 0.0, 0.0)

    @staticmethod
    def CreateConstant(f):
        return Fresnel("constant", (float(np.clip(f, 0.0, 1.0)),))

    @staticmethod
    def CreateSchlick(rgb):
        r, g, b = (float(np.clip(c, 0.0, 1.0)) for c in rgb)
        return Fresnel("schlick", (r, g, b))

    @staticmethod
    def CreateConductor(n, k):
        return Fresnel("conductor", (float(np.clip(n, 1e-2, 1e3)), float(np.clip(k, 1e-2, 1e3))))

    @staticmethod
    def CreateDielectric(n):
        return Fresnel("dielectric", (float(np.clip(n, 1.0, 1e3)),))

    def Serialize(self):
        if self.kind == "schlick":
            return (self.data[0], self.data[1], self.data[2], 0.0)
        if self.kind == "constant":
            return (abi.FRESNEL_CONSTANT, 0.0, self.data[0], 0.0)
        if self.kind == "conductor":
            return (abi.FRESNEL_CONDUCTOR, self.data[0], self.data[1], 0.0)
        if self.kind == "dielectric":
            return (abi.FRESNEL_DIELECTRIC, self.data[0], 0.0, 0.0)
        raise ValueError(self.kind)


def _v(x, n):
    a = np.atleast_1d(np.asarray(x, dtype=np.float32))
    if a.size == 1:
        a = np.repeat(a, n)
    assert a.size == n
    return a.astype(np.float32).copy()


@dataclass
class BSDF:
    Kc: np.ndarray = field(default_factory=lambda: np.zeros(4, np.float32))   # rgb + coat roughness
    Kd: np.ndarray = field(default_factory=lambda: np.zeros(3, np.float32))
    Ks: np.ndarray = field(default_factory=lambda: np.zeros(4, np.float32))   # rgb + base roughness
    Kt: np.ndarray = field(default_factory=lambda: np.zeros(3, np.float32))
    Le: np.ndarray = field(default_factory=lambda: np.zeros(3, np.float32))
    Absorption: np.ndarray = field(default_factory=lambda: np.zeros(4, np.float32))  # rgb + coeff
    FresnelCoat: Fresnel = field(default_factory=lambda: Fresnel.CreateConstant(0.0))
    FresnelBase: Fresnel = field(default_factory=lambda: Fresnel.CreateConstant(1.0))
    texture: int = -1                 # diffuse (Kd) texture slot; -1 = none   (aspect Kd map, AisMesh.cxx:321-346)
    texture_scale: tuple = (1.0, 1.0)  # rttexture -scale S T                    (ImportExportPlugin.cxx:679-727)

    # -- factories (reference call sites above) ------------------------------------------
    @staticmethod
    def CreateDiffuse(weight):
        b = BSDF()
        b.Kd = _v(weight, 3)
        return b

    @staticmethod
    def CreateMetallic(weight, fresnel, roughness):
        b = BSDF()
        b.Ks = np.concatenate([_v(weight, 3), [np.float32(roughness)]]).astype(np.float32)
        b.FresnelBase = fresnel
        return b

    @staticmethod
    def CreateGlass(weight, absorption_color, absorption_coeff, refraction_index):
        b = BSDF()
        b.FresnelCoat = Fresnel.CreateDielectric(refraction_index)
        b.Kt = _v(weight, 3)
        b.Kc = np.array([1, 1, 1, 0], np.float32)
        b.Absorption = np.concatenate([_v(absorption_color, 3), [np.float32(absorption_coeff)]]).astype(np.float32)
        return b

    # -- presets as MaterialEditor resets them -------------------------------------------
    @staticmethod
    def Matte(kd=0.8):                      # MaterialEditor.cxx:666-686
        return BSDF.CreateDiffuse(kd)

    @staticmethod
    def Metal(ks=1.0, roughness=0.1, f0=0.8):   # :688-721
        return BSDF.CreateMetallic(ks, Fresnel.CreateSchlick(_v(f0, 3)), roughness)

    @staticmethod
    def Glossy(kd=0.5, ks=0.5, roughness=0.1, f0=0.8):   # :723-783
        b = BSDF.CreateMetallic(ks, Fresnel.CreateSchlick(_v(f0, 3)), roughness)
        b.Kd = _v(kd, 3)
        return b

    @staticmethod
    def Glass(kt=1.0, absorption_color=(0, 0, 0), absorption_coeff=0.0, ior=1.5):   # :785-831
        return BSDF.CreateGlass(kt, absorption_color, absorption_coeff, ior)

    @staticmethod
    def Paint(kd=0.5, ks=0.5, roughness=0.1, f0=0.8, kc=1.0, coat_roughness=0.0, coat_ior=1.5):   # :833-944
        b = BSDF.Glossy(kd, ks, roughness, f0)
        b.Kc = np.concatenate([_v(kc, 3), [np.float32(coat_roughness)]]).astype(np.float32)
        b.FresnelCoat = Fresnel.CreateDielectric(coat_ior)
        return b

    # -- invariants the reference enforces before upload ---------------------------------
    def Normalize(self):
        """Energy normalisation: if max_c(Kd+Ks+Kt) > 1 divide Kd, Ks.rgb, Kt by it
        (MaterialEditor.cxx:311-329; Graphic3d_BSDF::Normalize at AisMesh.cxx:319)."""
        m = float(np.max(self.Kd[:3] + self.Ks[:3] + self.Kt[:3]))
        if m > 1.0:
            self.Kd = (self.Kd / np.float32(m)).astype(np.float32)
            self.Ks[:3] = self.Ks[:3] / np.float32(m)
            self.Kt = (self.Kt / np.float32(m)).astype(np.float32)
        return self

    def Sanitize(self):
        """MaterialEditor::setBSDF (MaterialEditor.cxx:281-338): clamp weights to [0,1], Le >= 0,
        absorption coeff >= 0, then normalise."""
        self.Kc = np.clip(self.Kc, 0, 1).astype(np.float32)
        self.Kd = np.clip(self.Kd, 0, 1).astype(np.float32)
        self.Ks = np.clip(self.Ks, 0, 1).astype(np.float32)
        self.Kt = np.clip(self.Kt, 0, 1).astype(np.float32)
        w = max(float(self.Absorption[3]), 0.0)
        self.Absorption = np.clip(self.Absorption, 0, 1).astype(np.float32)
        self.Absorption[3] = w
        self.Le = np.maximum(self.Le, 0).astype(np.float32)
        return self.Normalize()

    def MaterialType(self):
        """getMaterialType (MaterialEditor.cxx:350-370): 0 matte 1 metal 2 glossy 3 glass 4 paint 5 custom."""
        nz = lambda v: float(np.sum(v[:3])) > 1e-10
        hc, hd, hs, ht = nz(self.Kc), nz(self.Kd), nz(self.Ks), nz(self.Kt)
        if not hc:
            if not ht:
                return 0 if not hs else (1 if not hd else 2)
        else:
            return 4 if not ht else (5 if (hd or hs) else 3)
        return 5

    def to_abi(self):
        m = abi.crh_bsdf()
        m.Kc[:] = [float(x) for x in self.Kc]
        m.Kd[:] = [float(x) for x in self.Kd] + [float(self.texture + 1)]
        m.Ks[:] = [float(x) for x in self.Ks]
        m.Kt[:] = [float(x) for x in self.Kt] + [float(self.texture_scale[0]) if self.texture >= 0 else 0.0]
        m.Le[:] = [float(x) for x in self.Le] + [float(self.texture_scale[1]) if self.texture >= 0 else 0.0]
        m.Absorption[:] = [float(x) for x in self.Absorption]
        m.FresnelCoat[:] = self.FresnelCoat.Serialize()
        m.FresnelBase[:] = self.FresnelBase.Serialize()
        return m


def phong_to_roughness(shininess):
    """Ks.w = sqrt(2 / (shininess + 2)) (AisMesh.cxx:316)."""
    return float(np.sqrt(np.float32(2.0) / (np.float32(shininess) + np.float32(2.0))))


def pack_materials(bsdfs):
    arr = (abi.crh_bsdf * max(len(bsdfs), 1))()
    for i, b in enumerate(bsdfs):
        arr[i] = b.to_abi()
    return arr

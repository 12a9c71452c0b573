"""ctypes loader for the CPU oracle (oracle/libcrh_oracle.so).  TEST INFRASTRUCTURE ONLY:
imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg -- never by cadrays_amd/.
"""
import ctypes as C
import os
import sys
import subprocess

import numpy as np

from cadrays_amd import abi
from cadrays_amd.binding import Backend

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    so = os.path.join(_HERE, "libcrh_oracle.so")
    src = os.path.join(_HERE, "crh_oracle.c")
    inc = os.path.join(_HERE, "..", "include")
    hdrs = [os.path.join(inc, h) for h in sorted(os.listdir(inc)) if h.endswith(".h")]
    stale = (not os.path.exists(so)) or any(os.path.getmtime(p) > os.path.getmtime(so) for p in [src] + hdrs)
    if force or stale:
        subprocess.check_call(["make", "-s", "-C", _HERE, "-B", "libcrh_oracle.so"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        # CRH_ORACLE_LIB: another build of the same file (tests/hunts/run_sanitizers.sh loads the ASan / UBSan builds this way)
        _LIB = C.CDLL(os.environ.get("CRH_ORACLE_LIB") or build())
        _default_threads(_LIB)
    return _LIB


def _default_threads(l):
    """OpenMP would start one thread per hardware thread; a container with a CPU quota (16 of 256 on the pool's GPU boxes) is far slower
    that way than with as many threads as it may run"""
    try:
        sys.path.insert(0, os.path.join(_HERE, ".."))
        from cadrays_amd.hostinfo import usable_cpus
        if not os.environ.get("OMP_NUM_THREADS"):
            l.orc_set_threads(int(usable_cpus()))
    except Exception:
        pass


class Oracle(Backend):
    _libfn = staticmethod(lambda: lib())

    def __init__(self, threads=0):
        super().__init__(self._libfn(), "orc_")
        if threads:
            self._libfn().orc_set_threads(int(threads))

    @classmethod
    def set_threads(cls, n):
        return int(cls._libfn().orc_set_threads(int(n)))

    def read_accum(self):
        out = np.empty((self.height, self.width, 4), np.float32)
        self._call("read_accum", out.ctypes.data_as(C.POINTER(C.c_float)))
        return out


_FAST = None


def fast_lib():
    """libcrh_oracle_fast.so: bench.py's CPU-baseline build (-O3 -march=native, ORC_FAST).  -march=native ties the file to
    the CPU it was built on, so it is rebuilt when the sidecar names another CPU model (the GPU box's host is not this one)."""
    global _FAST
    if _FAST is None:
        so, tag = os.path.join(_HERE, "libcrh_oracle_fast.so"), os.path.join(_HERE, "libcrh_oracle_fast.cpu")
        here = ""
        try:
            here = next(l for l in open("/proc/cpuinfo") if l.startswith("model name"))
        except (OSError, StopIteration):
            pass
        src = os.path.join(_HERE, "crh_oracle.c")
        stale = (not os.path.exists(so)) or (not os.path.exists(tag)) or open(tag).read() != here or os.path.getmtime(src) > os.path.getmtime(so)
        if stale:
            subprocess.check_call(["make", "-s", "-C", _HERE, "-B", "libcrh_oracle_fast.so"])
        _FAST = C.CDLL(so)
        _default_threads(_FAST)
    return _FAST


class FastOracle(Oracle):
    """Same entry points on the fast build.  A baseline to time, never a checker (contraction changes the low bits)."""
    _libfn = staticmethod(lambda: fast_lib())


def oracle_class(kind="parity"):
    if kind == "parity":
        lib()
        return Oracle
    if kind == "fast":
        fast_lib()
        return FastOracle
    raise ValueError(kind)


# --- unit entry points ------------------------------------------------------------------------
_f32p = C.POINTER(C.c_float)


def fresnel(cos_i, f4):
    out = (C.c_float * 3)()
    lib().orc_fresnel(C.c_float(cos_i), (C.c_float * 4)(*f4), out)
    return np.array(out[:], np.float32)


def bsdf_eval(bsdf, wo, wi, two_sided=True):
    out = (C.c_float * 3)()
    m = bsdf.to_abi()
    lib().orc_bsdf_eval(C.byref(m), (C.c_float * 3)(*wo), (C.c_float * 3)(*wi), int(two_sided), out)
    return np.array(out[:], np.float32)


def bsdf_pdf(bsdf, wo, wi, weight=(1, 1, 1), two_sided=True):
    m = bsdf.to_abi()
    f = lib().orc_bsdf_pdf
    f.restype = C.c_float
    return float(f(C.byref(m), (C.c_float * 3)(*wo), (C.c_float * 3)(*wi), (C.c_float * 3)(*weight), int(two_sided)))


def bsdf_sample(bsdf, wo, rng_state, weight=(1, 1, 1), two_sided=True, inside=False):
    """returns alive, wi, weight, delta, inside, new rng state"""
    m = bsdf.to_abi()
    w = (C.c_float * 3)(*weight)
    wi = (C.c_float * 3)()
    st = C.c_uint32(rng_state)
    fl = C.c_int(0)
    alive = lib().orc_bsdf_sample(C.byref(m), (C.c_float * 3)(*wo), w, C.byref(st), int(two_sided), int(inside), wi, C.byref(fl))
    return bool(alive), np.array(wi[:], np.float32), np.array(w[:], np.float32), bool(fl.value & 1), bool(fl.value & 2), st.value


def math_fn(fn, a, b=None):
    a = np.ascontiguousarray(a, np.float32)
    b = np.ascontiguousarray(b if b is not None else np.zeros_like(a), np.float32)
    out, out2 = np.empty_like(a), np.empty_like(a)
    lib().orc_math(int(fn), a.ctypes.data_as(_f32p), b.ctypes.data_as(_f32p), out.ctypes.data_as(_f32p),
                   out2.ctypes.data_as(_f32p), C.c_uint32(a.size))
    return out, out2


def rng_stream(pixel, fseed, n):
    out = np.empty(n, np.float32)
    lib().orc_rng_stream(C.c_uint32(pixel), C.c_uint32(fseed), out.ctypes.data_as(_f32p), C.c_uint32(n))
    return out


def frame_seed(seed, n):
    f = lib().orc_frame_seed
    f.restype = C.c_uint32
    return int(f(C.c_uint32(seed), C.c_uint32(n)))

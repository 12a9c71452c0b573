#!/usr/bin/env python3
"""Script-driven host: the reference's test mode  `CADRays <script.tcl> <nFrames>`  on the MI355X backend.

Reference behaviour restated (src/Launcher/main.cxx:164-228, AppViewer.cxx:1045-1071, 1255-1264): evaluate the Tcl script,
render one Redraw() per frame until nFrames, dump the image as Output_<name>_<n>.png and the average frame rate as
Output_<name>_<n>.txt.  Scripts that drive the renderer themselves -- `vfps N` (render N frames) and `vdump file.png`, the
material-preview recipe of data/other/preview.tcl:59-64 -- are honoured live: every vfps renders on the GPU with the scene
state of that moment, every vdump writes the current image.

    python -m cadrays_amd.run_script script.tcl [nFrames] [--device 0] [--outdir DIR] [--max-vfps N] [--size WxH]
"""
import argparse
import json
import os
import time

import numpy as np

from .scene_tcl import MiniTcl, SceneBuilder


def write_png(path, rgb8):
    from PIL import Image
    Image.fromarray(np.ascontiguousarray(rgb8), "RGB").save(path)


class ScriptHost:
    def __init__(self, view_factory, outdir=".", size=None, max_vfps=None, lookahead=0, hdr=False):
        self.view_factory, self.outdir, self.size, self.max_vfps, self.lookahead, self.hdr = view_factory, outdir, size, max_vfps, lookahead, hdr
        self.view = None
        self.frames = 0
        self.seconds = 0.0
        self.dumps = []
        self.builder = None
        self._scene_key = None

    # -- the scene of this moment, loaded into the view (a changed scene restarts accumulation, like OCCT)
    def _sync_scene(self):
        w, h = self.size or self.builder.view_size or (512, 512)
        sc = self.builder.snapshot(w, h, "script")
        key = (sc.pos.tobytes(), sc.tri.tobytes(), b"".join(bytes(m.to_abi()) for m in sc.materials), repr(sc.lights), repr(sc.camera),
               repr(sc.params), None if sc.env is None else sc.env.tobytes(),
               sc.nrm.tobytes(), None if sc.uv is None else sc.uv.tobytes(), tuple(t.tobytes() for t in (sc.textures or [])),
               None if sc.tri_object is None else sc.tri_object.tobytes(), None if sc.obj_xform is None else sc.obj_xform.tobytes())
        if key != self._scene_key:
            if self.view is None:
                self.view = self.view_factory()
            self.view.load_scene(sc)
            # speculative look-ahead (crh_set_lookahead) keeps the 1-spp-per-Redraw loop wide: ~4 M paths per traced batch
            k = self.lookahead or min(1024, max(1, (4 << 20) // (w * h)))
            if k > 1 and hasattr(self.view, "set_lookahead"):
                self.view.set_lookahead(k)
            self._scene_key = key
        return sc

    def render(self, n):
        """n x Redraw() (AppViewer.cxx:1047)"""
        self._sync_scene()
        if self.max_vfps:
            n = min(n, self.max_vfps)
        t = time.perf_counter()
        for _ in range(n):
            self.view.Redraw()
        self.view.sync() if hasattr(self.view, "sync") else None
        self.seconds += time.perf_counter() - t
        self.frames += n
        return ""

    def dump(self, path):
        """BufferDump(Graphic3d_BT_RGB) -> image file (AppViewer.cxx:1259-1261); only the file name of the script's path is kept"""
        self._sync_scene()
        out = os.path.join(self.outdir, os.path.basename(path.replace("\\", "/")) or "dump.png")
        if not out.lower().endswith(".png"):
            out += ".png"
        write_png(out, self.view.read_ldr())
        self.dumps.append(out)
        return ""

    def run(self, script, n_frames=0):
        root = os.path.dirname(os.path.abspath(script))
        b = self.builder = SceneBuilder(root)
        b.on_vfps, b.on_vdump = self.render, self.dump
        interp = MiniTcl(b.commands, {"Root": root, "__script__": os.path.abspath(script), "env(APP_DATA)": root + "/"})
        interp.eval(open(script).read())
        name = os.path.splitext(os.path.basename(script))[0]
        if n_frames > 0:                                   # the reference's test mode: nFrames more Redraws, then Output_<name>_<n>.*
            self.render(n_frames)
            base = os.path.join(self.outdir, f"Output_{name}_{n_frames}")
            self.dump(base + ".png")
            if self.hdr and hasattr(self.view, "read_hdr"):    # beside the reference's two files: the linear HDR accumulator (tools/compare_runs.py)
                rgb = np.ascontiguousarray(self.view.read_hdr(), np.float32)
                with open(base + ".pfm", "wb") as f:
                    f.write(b"PF\n%d %d\n-1.0\n" % (rgb.shape[1], rgb.shape[0])); f.write(rgb[::-1].astype("<f4").tobytes())
            with open(base + ".txt", "w") as f:
                f.write("%g" % (self.frames / max(self.seconds, 1e-9)))
        return {"script": name, "frames": self.frames, "seconds": round(self.seconds, 4),
                "fps": round(self.frames / max(self.seconds, 1e-9), 2), "images": self.dumps, "unsupported": b.unsupported}


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("script")
    ap.add_argument("frames", nargs="?", type=int, default=0)
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--outdir", default=".")
    ap.add_argument("--max-vfps", type=int, default=0, help="cap the frames a single vfps renders")
    ap.add_argument("--size", default="", help="WxH render target (default: the script's vinit size, else 512x512)")
    ap.add_argument("--hdr", action="store_true", help="also write Output_<name>_<n>.pfm (linear HDR) for tools/compare_runs.py")
    a = ap.parse_args(argv)
    import torch  # noqa: F401  (runtime ordering: torch's HIP runtime first)
    from .view import View
    os.makedirs(a.outdir, exist_ok=True)
    size = tuple(int(x) for x in a.size.lower().split("x")) if a.size else None
    host = ScriptHost(lambda: View(a.device), a.outdir, size, a.max_vfps or None, hdr=a.hdr)
    print(json.dumps(host.run(a.script, a.frames)))


if __name__ == "__main__":
    main()

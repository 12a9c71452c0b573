import sys, os
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cadrays_amd import scenes, sharding
from cadrays_amd.view import View
bits = lambda a: a.view(np.uint32)
sc = scenes.cornell_box(True, 96, 80)
for trial in range(3):
    ref = View(0).load_scene(sc)
    v = View(0).load_scene(sc); v.set_lookahead(4)
    ref.Redraw(); v.Redraw()
    a, b = v.read_hdr(), ref.read_hdr()
    d = (bits(a) != bits(b)).any(2)
    print("lookahead trial", trial, "differing pixels", int(d.sum()), "max abs", float(np.abs(a - b).max()), np.argwhere(d)[:5].tolist())
    # same context type, no lookahead, twice
    r2 = View(0).load_scene(sc); r2.Redraw()
    print("   plain vs plain", int((bits(r2.read_hdr()) != bits(b)).any(2).sum()))
    # tiles subset
    t = View(0).load_scene(sc)
    tiles = sharding.tiles_for_rank(t.n_tiles(), 1, 3)
    t.render_tiles(tiles, 0, 1)
    c = t.read_hdr()
    ts = 32; tx = (96 + ts - 1) // ts
    m = np.zeros((80, 96), bool)
    for ti in tiles:
        m[(ti // tx) * ts:(ti // tx + 1) * ts, (ti % tx) * ts:(ti % tx + 1) * ts] = True
    dd = (bits(c) != bits(b)).any(2) & m
    print("   tile subset vs full: differing pixels inside the subset", int(dd.sum()), np.argwhere(dd)[:5].tolist())

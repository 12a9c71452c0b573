#!/usr/bin/env python3
"""Writes tests/golden/icon_features.json from the 25 material icons the reference ships (data/materials/*.png).

Those icons are OUTPUTS OF THE REAL RENDERER: OCCT's path tracer wrote them for data/other/preview.tcl:10-64 (128 x 128, -rayDepth 10, `vfps 8000`,
`vdump`), and the application loads them at start-up (src/Launcher/main.cxx:120-132).  They are the only reference-held evidence about the conventions
of rows a2 / a11 / a17 of SURVEY.md section 8.  This script runs in the BUILD container (where /root/reference is readable) and commits DATA derived
from them -- positions, masks as run lengths, block means -- never the images; tests/test_icon_features.py compares the oracle's and the HIP path's
render of tools/material_preview.tcl against this file with the same extraction function (tools/icon_features.py).

    python tests/golden/make_icon_features.py [--reference /root/reference]
"""
import argparse
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import icon_features as F  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", default="/root/reference")
    a = ap.parse_args()
    from PIL import Image
    d = os.path.join(a.reference, "data", "materials")
    icons = {}
    for stock, icon in F.STOCK_TO_ICON + [("", "custom")]:
        img = np.asarray(Image.open(os.path.join(d, icon + ".png")).convert("RGB"))
        assert img.shape == (F.N, F.N, 3), (icon, img.shape)
        f = F.extract(img, icon)
        f["stock"] = stock
        if icon == "custom":                                  # a composite picture (half chrome, half matte): geometry only
            for k in ("highlight", "shadow_left_over_right", "interior", "caustic", "ball_peak"):
                f.pop(k, None)
        icons[icon] = f

    # the geometry is the same in every icon: an edge is a TILE edge when (nearly) every icon has it -- the others (shadow terminators, caustic rims)
    # depend on the material and on the absent environment map and stay out of the comparison
    names = list(icons)
    consensus = {"rows": {}, "cols": {}}
    for kind in ("rows", "cols"):
        for key in icons[names[0]][kind]:
            pool = [e for n in names for e in icons[n][kind][key]]
            kept = []
            for x, s in sorted(icons["plastered"][kind][key]):
                near = [y for y, t in pool if t == s and abs(y - x) <= 0.5]
                if len(near) >= len(names) - 2:
                    kept.append([round(float(np.median(near)), 3), s, round(float(max(near) - min(near)), 3)])
            consensus[kind][key] = kept
    for n in names:
        for kind in ("rows", "cols"):
            for key, edges in icons[n][kind].items():
                tile = [e for e in edges if any(abs(e[0] - c[0]) <= 0.5 and e[1] == c[1] for c in consensus[kind][key])]
                icons[n].setdefault("other_edges", {})[kind[0] + key] = [e for e in edges if e not in tile]
                icons[n][kind][key] = tile
    matte = ("plastered", "stone", "plastified", "jade", "satined", "shiny_plastified")
    consensus["horizon"] = [round(float(np.median([icons[n]["horizon"][k] for n in names])), 3) for k in (0, 1)]
    consensus["cap_circle_matte"] = [round(float(np.median([icons[n]["cap_circle"][k] for n in matte])), 3) for k in (0, 1, 2)]
    gam = [icons[n]["display_gamma"] for n in names if icons[n]["display_gamma"]]
    consensus["display_gamma"] = {"median": round(float(np.median(gam)), 3), "min": min(gam), "max": max(gam), "n": len(gam)}
    out = {"source": "data/materials/*.png of the reference (64 x 64 RGB), rendered by OCCT from data/other/preview.tcl:10-64",
           "extractor": "tools/icon_features.py", "scan_rows": list(F.SCAN_ROWS), "scan_cols": list(F.SCAN_COLS), "grid": F.GRID,
           "consensus": consensus, "icons": icons}
    path = os.path.join(HERE, "icon_features.json")
    with open(path, "w") as fh:
        json.dump(out, fh, indent=None, separators=(",", ":"))
        fh.write("\n")
    print(path, os.path.getsize(path), "bytes;", json.dumps(consensus))


if __name__ == "__main__":
    main()

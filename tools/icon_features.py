#!/usr/bin/env python3
"""Geometric features of a material-preview image (ball on a two-tone tiled floor, black background).

The reference ships 25 outputs of the real OCCT path tracer: data/materials/*.png, 64 x 64 RGB, the images data/other/preview.tcl:10-64
wrote (128 x 128, -rayDepth 10, 8000 frames per stock material, vdump) scaled down to icon size and loaded by main.cxx:120-132.  The
environment map of that recipe (preview.tcl:56, a file on the author's disk) and OCCT's stock BSDFs are not in the reference, so the icons
cannot pin radiance -- but everything in them that is GEOMETRY or a CONVENTION is independent of both, and this module measures exactly
that, with one function for both sides (tests/golden/make_icon_features.py reads the icons, tests/test_icon_features.py renders
tools/material_preview.tcl with the oracle and with the HIP path):

  horizon        sub-pixel row of the floor's far edge at the left and right image border (camera pitch, fovy, V orientation)
  cap            circle through the ball's upper silhouette against the black background (centre, radius) + that cap's mask as run lengths
  rows / cols    sub-pixel positions of the tile edges along three scan rows of the foreground and two scan columns (perspective)
  highlight      centroid and size of the saturated spot on the ball (sign of the light direction; cone angle on a mirror)
  shadow         which side of the ball the floor is darker on, and the shadow's relative depth profile along one scan row
  interior       the ball interior's luminance on a coarse polar grid (refraction: the floor seen upside down through glass / water / diamond)
  caustic        presence and centroid of the spot on the floor that is brighter than any lit tile (light through two delta interfaces)
  tile_ratio     display value of a lit bright tile over its dark neighbour: (0.85 / 0.45) ** (1 / gamma) pins the display gamma
  background     largest value in the rows above everything (the environment is not shown behind the scene)

Pixel coordinates: x to the right, y down, pixel (i, j) covers [i, i + 1) x [j, j + 1); a sub-pixel position is an edge of a coverage-weighted
box, so a box filter over a finer image leaves it where it was.
"""
import json
import math
import sys

import numpy as np

# OCCT's Graphic3d_NameOfMaterial order = the order of preview.tcl:3; the icon file is Graphic3d_MaterialAspect::MaterialName() of the same
# index (main.cxx:120-126).  "custom" (main.cxx:130) is a composite picture: geometry features only.
STOCK_TO_ICON = [("brass", "brass"), ("bronze", "bronze"), ("copper", "copper"), ("gold", "gold"), ("pewter", "pewter"), ("plaster", "plastered"),
                 ("plastic", "plastified"), ("silver", "silver"), ("steel", "steel"), ("stone", "stone"), ("shiny_plastic", "shiny_plastified"),
                 ("satin", "satined"), ("metalized", "metalized"), ("neon_gnc", "ionized"), ("chrome", "chrome"), ("aluminium", "aluminium"),
                 ("obsidian", "obsidian"), ("neon_phc", "neon"), ("jade", "jade"), ("charcoal", "charcoal"), ("water", "water"), ("glass", "glass"),
                 ("diamond", "diamond"), ("transparent", "transparent")]
REFRACTIVE = ("water", "glass", "diamond")
MIRROR_LIKE = ("chrome", "silver", "steel", "gold", "brass", "bronze", "copper", "aluminium", "pewter")
SCAN_ROWS = (58, 60, 62, 63)
SCAN_COLS = (1, 62)
N = 64
GRID = 14


def luminance(rgb8):
    a = np.asarray(rgb8, np.float64)[..., :3] / 255.0
    return a @ np.array([0.2126, 0.7152, 0.0722])


def box_down(rgb8, n=N):
    """average k x k blocks of display values, like scaling the dumped PNG down"""
    a = np.asarray(rgb8, np.float64)
    k = a.shape[0] // n
    assert a.shape[0] == a.shape[1] == n * k, a.shape
    return a.reshape(n, k, n, k, -1).mean(axis=(1, 3))


def _edges_1d(v, lo, hi, min_step=0.07):
    """sub-pixel positions of the steps of v[lo:hi]: an edge is where |v[i+1] - v[i-1]| peaks; its position is the one that splits the
    three pixels around it between the plateau values two pixels to either side (area-preserving: exact for a box-filtered step)"""
    out = []
    g = np.zeros_like(v)
    g[1:-1] = v[2:] - v[:-2]
    i = max(lo, 2)
    while i < min(hi, len(v) - 2):
        if abs(g[i]) >= min_step and abs(g[i]) >= abs(g[i - 1]) and abs(g[i]) > abs(g[i + 1]):
            a, b = v[i - 2], v[i + 2]
            if abs(a - b) >= min_step:
                cover = sum((v[j] - b) / (a - b) for j in (i - 1, i, i + 1))
                out.append([round(float(i - 1 + cover), 3), 1 if b > a else -1])
                i += 2
                continue
        i += 1
    return out


def _fit_circle(xs, ys):
    A = np.stack([xs, ys, np.ones_like(xs)], 1)
    sol, *_ = np.linalg.lstsq(A, -(xs ** 2 + ys ** 2), rcond=None)
    cx, cy = -sol[0] / 2, -sol[1] / 2
    return float(cx), float(cy), float(math.sqrt(max(cx * cx + cy * cy - sol[2], 0.0)))


def _runs(mask):
    """run-length code of a boolean image, row-major, starting with a run of False"""
    flat = np.asarray(mask, bool).ravel()
    change = np.flatnonzero(np.diff(flat.astype(np.int8))) + 1
    bounds = np.concatenate([[0], change, [flat.size]])
    runs = np.diff(bounds).tolist()
    return runs if not flat[0] else [0] + runs


def mask_from_runs(runs, shape):
    flat = np.zeros(shape[0] * shape[1], bool)
    pos, val = 0, False
    for r in runs:
        flat[pos:pos + r] = val
        pos += r
        val = not val
    return flat.reshape(shape)


def iou(a, b):
    a, b = np.asarray(a, bool), np.asarray(b, bool)
    u = np.logical_or(a, b).sum()
    return float(np.logical_and(a, b).sum() / u) if u else 1.0


def extract(img64, name=""):
    """img64: (64, 64, 3) display values 0 .. 255 (float or u8)"""
    rgb = np.asarray(img64, np.float64)
    assert rgb.shape[:2] == (N, N), rgb.shape
    L = luminance(rgb)
    f = {"name": name}

    # -- background: rows that neither floor nor ball reach
    f["background_max"] = round(float(L[:5].max()), 4)

    # -- horizon at the borders: the row where the border columns climb to half of the floor's level just below
    hz = []
    for cols in ((0, 1, 2), (61, 62, 63)):
        v = L[:, cols].mean(1)
        r = int(np.argmax(v > 0.05))
        level = v[r + 2:r + 8].mean()
        hz.append(r + 1 - min(v[r] / level, 1.0) if v[r] < 0.8 * level else float(r))
    f["horizon"] = [round(float(h), 3) for h in hz]
    hrow = int(math.floor(min(hz)))

    # -- cap: the ball's silhouette above the horizon, against black: every pixel the ball touches (the background is exactly 0 in the icons, so a
    #    threshold of 3 / 255 is "any coverage worth a grey level" for a charcoal ball and a chrome one alike)
    top = L[:hrow]
    cap = top > 0.012
    f["cap_runs"] = _runs(cap)
    f["cap_rows"] = hrow
    xs, ys = [], []
    for x in range(N):                                   # the circle goes through the top edge of the first touched pixel of each column: a rim estimate from
        nz = np.flatnonzero(cap[:, x])                   # coverage would depend on how bright the rim is (dark on chrome, bright on plaster)
        if nz.size == 0 or nz[0] + 2 >= hrow:
            continue
        xs.append(x + 0.5)
        ys.append(float(nz[0]))
    if len(xs) >= 8:
        cx, cy, rad = _fit_circle(np.array(xs), np.array(ys))
        f["cap_circle"] = [round(cx, 3), round(cy, 3), round(rad, 3)]
        f["cap_columns"] = [int(xs[0] - 0.5), int(xs[-1] - 0.5)]
    else:
        f["cap_circle"] = None

    # -- tile edges along scan rows (whole width) and scan columns (below the horizon's fine pattern)
    f["rows"] = {str(r): _edges_1d(L[r], 0, N) for r in SCAN_ROWS}
    f["cols"] = {str(c): _edges_1d(L[:, c], 26, N) for c in SCAN_COLS}

    # -- tile ratio: a lit bright tile over its dark neighbour AT THEIR COMMON EDGE (a straight line through each plateau, both evaluated at the edge:
    #    the irradiance is continuous there, so the ratio of display values is (0.85 / 0.45) ** (1 / gamma) whatever lights the floor), on the scan rows'
    #    right half (the shadow falls to the left)
    ratios = []
    for r in SCAN_ROWS:
        e = [x for x, _ in f["rows"][str(r)] if x >= 30]
        bounds = e + [float(N)]
        plate = []
        for a, b in zip(bounds[:-1], bounds[1:]):
            i0, i1 = int(math.ceil(a + 1.5)), int(math.floor(b - 1.5))
            if i1 - i0 >= 5:
                xs_ = np.arange(i0, i1) + 0.5
                plate.append((a, b, np.polyfit(xs_, L[r, i0:i1], 1)))
        for (a0, b0, p0), (a1, b1, p1) in zip(plate[:-1], plate[1:]):
            if abs(b0 - a1) < 1e-9:
                u, w = float(np.polyval(p0, b0)), float(np.polyval(p1, b0))
                hi_, lo_ = max(u, w), min(u, w)
                if lo_ > 0.2 and hi_ / lo_ > 1.15:
                    ratios.append(hi_ / lo_)
    f["tile_ratio"] = round(float(np.median(ratios)), 4) if ratios else None
    f["display_gamma"] = round(math.log(0.85 / 0.45) / math.log(f["tile_ratio"]), 3) if ratios else None

    if f["cap_circle"]:
        cx, cy, rad = f["cap_circle"]
        yy, xx = np.mgrid[0:N, 0:N] + 0.5
        rr = np.hypot(xx - cx, yy - cy)
        ball = rr <= rad - 1.0

        # -- highlight: the saturated spot in the ball's upper right quarter (where the mirror direction of `vlight ... direction -0.25 -1 -1` lies)
        peak = float(L[ball].max())
        f["ball_peak"] = round(peak, 4)
        spot = ball & (L >= 0.97) & (xx > cx) & (yy < cy + 0.1 * rad)
        if spot.sum() >= 3:
            f["highlight"] = [round(float(xx[spot].mean()), 3), round(float(yy[spot].mean()), 3), int(spot.sum())]
        else:
            f["highlight"] = None

        # -- shadow: floor to the left and to the right of the ball, rows just above its contact point, outside the disc
        band = (yy > cy + 0.55 * rad) & (yy < cy + 1.05 * rad) & (rr > rad + 1.5)
        left, right = band & (xx < cx - 0.2 * rad), band & (xx > cx + 0.2 * rad)
        f["shadow_left_over_right"] = round(float(L[left].mean() / L[right].mean()), 4)

        # -- interior: block means of the luminance on a GRID x GRID raster over the ball's bounding square (cells whose centre lies within 0.9 radius)
        grid = []
        for gy in range(GRID):
            for gx in range(GRID):
                x0, x1 = cx - rad + 2 * rad * gx / GRID, cx - rad + 2 * rad * (gx + 1) / GRID
                y0, y1 = cy - rad + 2 * rad * gy / GRID, cy - rad + 2 * rad * (gy + 1) / GRID
                if math.hypot((x0 + x1) / 2 - cx, (y0 + y1) / 2 - cy) > 0.9 * rad:
                    grid.append(None)
                    continue
                m = (xx >= x0) & (xx < x1) & (yy >= y0) & (yy < y1)
                grid.append(round(float(L[m].mean()), 4) if m.any() else None)
        f["interior"] = grid

        # -- caustic: floor pixels outside the ball that are brighter than every plain lit tile
        floor = (rr > rad + 1.0) & (yy > max(hz) + 2)
        lit = float(np.percentile(L[floor & (xx > cx + 0.5 * rad)], 98))
        hot = floor & (L > lit + 0.04) & (xx < cx + 0.3 * rad) & (yy > cy)
        if hot.sum() >= 4:
            w = L[hot] - lit
            f["caustic"] = [round(float((xx[hot] * w).sum() / w.sum()), 3), round(float((yy[hot] * w).sum() / w.sum()), 3), int(hot.sum())]
        else:
            f["caustic"] = None
    return f


def match_edges(ref, got, tol=1.0):
    """every reference edge must have an edge of the same polarity within tol; returns (worst distance, unmatched reference edges)"""
    worst, missing = 0.0, []
    for x, s in ref:
        d = [abs(x - y) for y, t in got if t == s]
        if not d or min(d) > tol:
            missing.append(x)
        else:
            worst = max(worst, min(d))
    return worst, missing


def correlation(a, b):
    keep = [i for i, (x, y) in enumerate(zip(a, b)) if x is not None and y is not None]
    a, b = np.asarray([a[i] for i in keep], float), np.asarray([b[i] for i in keep], float)
    a, b = a - a.mean(), b - b.mean()
    d = math.sqrt(float((a * a).sum() * (b * b).sum()))
    return float((a * b).sum() / d) if d > 0 else 0.0


if __name__ == "__main__":
    from PIL import Image
    for p in sys.argv[1:]:
        im = np.asarray(Image.open(p).convert("RGB"), np.float64)
        if im.shape[0] != N:
            im = box_down(im)
        print(json.dumps(extract(im, p)))

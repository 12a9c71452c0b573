"""Property test of the node quantiser / packer (include/crh_bvh_format.h), which the oracle and the product SHARE -- so the
"GPU == oracle" tests cannot see an error in it.  Every packed node of both builders is decoded here with
independent numpy arithmetic (float64, exact for these operands) and every child's TRUE box -- recomputed from the triangles
underneath it -- must lie inside the decoded box; the grid must also be tight (no face further than one step from the truth)
and implicit child references must tile the node and leaf arrays exactly once."""
import numpy as np
import pytest

from cadrays_amd import scenes
from cadrays_amd.materials import BSDF


def decode(w):
    """packed node (16 dwords) -> origin[3], step[3], n_inner, n_children, qlo[4, 3], qhi[4, 3], child_base, leaf_base"""
    org = w[0:3].view(np.float32).astype(np.float64)
    k = np.array([w[3] & 0xff, (w[3] >> 8) & 0xff, (w[3] >> 16) & 0xff], np.int64)
    k = np.where(k >= 128, k - 256, k)                   # signed bytes: step = 2^k (include/crh_bvh_format.h)
    e = k + 127
    step = np.ldexp(1.0, k.astype(int))
    ni, nc = int((w[3] >> 24) & 7), int((w[3] >> 28) & 7)
    qlo = np.array([[(int(w[4 + a]) >> (8 * k)) & 0xff for a in range(3)] for k in range(4)], np.float64)
    qhi = np.array([[(int(w[7 + a]) >> (8 * k)) & 0xff for a in range(3)] for k in range(4)], np.float64)
    return org, step, ni, nc, qlo, qhi, int(w[10]), int(w[11])


def check_tree(nodes, tris, root=0, n_expected_leaves=None):
    W = nodes.view(np.uint32)
    tri_lo = tris[:, [0, 4, 8, 1, 5, 9, 2, 6, 10]].reshape(-1, 3, 3).min(2)       # per leaf-order triangle: min / max corner
    tri_hi = tris[:, [0, 4, 8, 1, 5, 9, 2, 6, 10]].reshape(-1, 3, 3).max(2)
    seen_nodes, seen_leaves = set(), set()
    worst_slack = 0.0

    def visit(i):
        nonlocal worst_slack
        assert i not in seen_nodes, f"node {i} referenced twice"
        seen_nodes.add(i)
        org, step, ni, nc, qlo, qhi, cb, lb = decode(W[i])
        assert 0 <= ni <= nc <= 4
        lo_all, hi_all = np.full(3, np.inf), np.full(3, -np.inf)
        for k in range(nc):
            if k < ni:
                tlo, thi = visit(cb + k)
            else:
                ref = lb + (k - ni)
                assert ref & 0x80000000 and (ref & 0xF0000000) != 0xF0000000
                p = ref & 0x0FFFFFFF
                assert p not in seen_leaves, f"leaf {p} referenced twice"
                seen_leaves.add(p)
                tlo, thi = tri_lo[p].astype(np.float64), tri_hi[p].astype(np.float64)
            dlo, dhi = org + qlo[k] * step, org + qhi[k] * step                    # exact in float64
            assert (dlo <= tlo).all() and (dhi >= thi).all(), f"node {i} child {k}: true box sticks out of the decoded box"
            # tightness: the grid is never more than one step away (a clamped face excepted, which cannot happen: 255 steps >= extent)
            assert ((tlo - dlo) < step * (1 + 1e-12)).all() and ((dhi - thi) < step * (1 + 1e-12)).all(), f"node {i} child {k}: loose box"
            worst_slack = max(worst_slack, float(((tlo - dlo) / step).max()), float(((dhi - thi) / step).max()))
            lo_all, hi_all = np.minimum(lo_all, tlo), np.maximum(hi_all, thi)
        if nc:
            assert (org == lo_all).all(), f"node {i}: origin is not the min corner of the union"      # origin is copied, not rounded
            assert (255.0 * step >= hi_all - lo_all).all() and ((127.0 * step < hi_all - lo_all) | (hi_all == lo_all) | (step <= 2.0 ** -126)).all(), \
                f"node {i}: exponent is not the smallest power of two that spans the extent"
        return lo_all, hi_all

    visit(root)
    if n_expected_leaves is not None:
        assert seen_leaves == set(range(n_expected_leaves))
    return len(seen_nodes), worst_slack


CASES = [(1, 3, 1.0), (2, 4, 1.0), (5, 5, 1.0), (37, 6, 1e-3), (1000, 7, 1.0), (6000, 8, 50.0), (6000, 9, 1e-2)]


@pytest.mark.parametrize("n,seed,scale", CASES)
def test_every_packed_child_box_contains_the_truth_oracle_builder(oracle_lib, n, seed, scale):
    pos, nrm, tri = scenes.gen_scene(n, seed, 1)
    pos = (pos * np.float32(scale)).astype(np.float32)
    o = oracle_lib.Oracle().load_scene(scenes.Scene(pos, nrm, tri, [BSDF.CreateDiffuse(0.5)]))
    nodes, tris = o.get_bvh()
    n_nodes, slack = check_tree(nodes, tris, 0, n)
    assert slack < 1.0 + 1e-9


@pytest.mark.parametrize("n,seed,scale", CASES)
def test_every_packed_child_box_contains_the_truth_product_builder(hip_lib, n, seed, scale):
    from cadrays_amd.view import build_bvh_host
    pos, nrm, tri = scenes.gen_scene(n, seed, 1)
    pos = (pos * np.float32(scale)).astype(np.float32)
    nodes, order = build_bvh_host(pos, tri, threads=3)
    v = pos[tri[order][:, :3]]                                                   # leaf-order triangle corners
    tris = np.zeros((n, 12), np.float32); tris[:, 0:3] = v[:, 0]; tris[:, 4:7] = v[:, 1]; tris[:, 8:11] = v[:, 2]
    check_tree(nodes, tris, 0, n)


def test_degenerate_and_offset_boxes(oracle_lib):
    """flat (zero-extent) axes, coincident triangles, a scene far from the origin (large coordinates, tiny extents)"""
    r = np.random.default_rng(4)
    n = 300
    c = r.random((n, 3)).astype(np.float32) * np.float32(1e-3) + np.float32(4096.0)
    c[:, 2] = np.float32(4096.5)                                                  # everything in one plane
    v = np.stack([c, c + np.float32([1e-4, 0, 0]), c + np.float32([0, 1e-4, 0])], 1)
    v[10:20] = v[10]                                                              # ten identical triangles
    pos = v.reshape(-1, 3); nrm = np.tile(np.float32([0, 0, 1]), (3 * n, 1))
    tri = np.concatenate([np.arange(3 * n, dtype=np.int32).reshape(-1, 3), np.zeros((n, 1), np.int32)], 1)
    o = oracle_lib.Oracle().load_scene(scenes.Scene(pos, nrm, tri, [BSDF.CreateDiffuse(0.5)]))
    nodes, tris = o.get_bvh()
    check_tree(nodes, tris, 0, n)

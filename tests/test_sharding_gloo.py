"""N > 1 path on CPU: two gloo ranks shard the tiles (cadrays_amd.sharding), each renders its shard with
the CPU oracle standing in for the GPU backend, the float4 framebuffers are reduced to rank 0, and the
result must be bit-identical to a single-process render (tile sharding keeps the
1-GPU RNG, so there is no sum-order issue)."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, spp, out_path):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    from cadrays_amd import scenes, sharding
    from oracle.pyoracle import Oracle
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        Oracle.set_threads(2)
        o = Oracle().load_scene(scenes.cornell_box(True, 96, 72))
        # two passes like bench.py's steps: samples [0, spp) then [spp, 2*spp)
        for step in range(2):
            tiles = sharding.render_shard(o, rank, world, step * spp, spp)
        assert len(tiles) > 0 and np.array_equal(tiles, sharding.tiles_for_rank(o.n_tiles(), rank, world, 3))      # 96 x 72 in 32 x 32 tiles: 3 columns, Morton-interleaved
        acc = torch.from_numpy(o.read_accum().copy())
        total = sharding.reduce_framebuffer(acc, 0)
        assert torch.equal(acc, torch.from_numpy(o.read_accum()))      # the rank's own buffer is left untouched
        # the cheaper exchange step: gather of owned tiles -- 1 / world of the bytes, the same frame on the root
        g = sharding.TileGather(o.width, o.height, o.tile_size, world, acc.device)
        assert g.bytes_per_rank * world < 16 * o.width * o.height * 1.6      # 12 tiles (96 x 72, the last row 8 pixels high) over 2 / 3 ranks, padded to the longest list
        frame = g.assemble(acc, rank, 0)
        assert torch.equal(acc, torch.from_numpy(o.read_accum()))
        if rank == 0:
            assert torch.equal(frame, total), "tile gather and full-frame reduce assemble different frames"
            frame2 = g.assemble(acc, rank, 0)                            # persistent output buffer, second call
            assert torch.equal(frame2, total)
            np.save(out_path, total.numpy())
        else:
            assert frame is None
            g.assemble(acc, rank, 0)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_tile_sharded_render_equals_single_process(tmp_path, oracle_lib, world):
    import torch.multiprocessing as mp
    from cadrays_amd import scenes
    spp = 2
    out = str(tmp_path / "acc.npy")
    mp.spawn(_worker, args=(world, _free_port(), spp, out), nprocs=world, join=True)
    got = np.load(out)
    o = oracle_lib.Oracle().load_scene(scenes.cornell_box(True, 96, 72))
    o.render(2 * spp)
    ref = o.read_accum()
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))
    assert (got[..., 3] == 2 * spp).all()


def test_tile_assignment_partitions_all_tiles():
    from cadrays_amd import sharding
    for n, w, tx in [(2040, 8, 0), (2040, 3, 0), (7, 8, 0), (1, 1, 0), (2040, 8, 60), (2040, 4, 60), (2040, 2, 60), (2040, 3, 60), (8160, 8, 120), (12, 5, 3), (6, 4, 6)]:
        parts = [sharding.tiles_for_rank(n, r, w, tx) for r in range(w)]
        assert sorted(np.concatenate(parts).tolist()) == list(range(n))
        assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1
        assert all((np.diff(p.astype(np.int64)) > 0).all() for p in parts if len(p) > 1)            # ascending: the frame is still walked row by row


def test_morton_interleave_spreads_every_small_block_over_the_ranks():
    """the point of the Z-order interleave: neighbouring tiles cost about the same, so the four tiles of every aligned 2 x 2
    block (and the eight of every aligned 4 x 2 block) must go to different ranks -- whatever the grid width; `t mod N` only does that by luck"""
    from cadrays_amd import sharding
    for tx, ty in [(60, 34), (120, 68), (16, 16)]:
        for world in (4, 8):
            owner = np.empty(tx * ty, np.int32)
            for r in range(world):
                owner[sharding.tiles_for_rank(tx * ty, r, world, tx)] = r
            owner = owner.reshape(ty, tx)
            bw, bh = (2, 2) if world == 4 else (4, 2)
            full = 0
            for y in range(0, ty - bh + 1, bh):
                for x in range(0, tx - bw + 1, bw):
                    if (x // bw) * bw == x and (y // bh) * bh == y:
                        blk = owner[y:y + bh, x:x + bw].ravel()
                        full += len(set(blk.tolist())) == world
            # the curve is cut where the image is not a power of two wide, so a few blocks straddle a seam: nearly all are perfect
            assert full >= 0.9 * ((tx // bw) * (ty // bh)), (tx, ty, world, full)

"""Multi-GPU sharding of the path: screen tiles are the unit (the reference's RT tile concept,
src/Launcher/SettingsWidget.cxx:451-476).  Every GPU holds the whole scene (replicated; it is tiny next
to 288 GB), renders ALL samples of its own tiles with the single-GPU RNG (seed = f(pixel, frame)), so
the union of the shards is bit-identical to a 1-GPU render; one reduce of the float4 framebuffer over
RCCL/xGMI (disjoint support, zero elsewhere -> the sum is exact) assembles the image on rank 0.

Backend-agnostic: works on anything with render_tiles()/n_tiles(); the CPU tests drive it with the
oracle over gloo, bench.py with the HIP backend over nccl (= RCCL).
"""
import numpy as np


def morton_order(tiles_x, tiles_y):
    """tile ids (row-major numbering, crh_render_tiles) sorted along the Z-order curve of their (column, row)"""
    ids = np.arange(tiles_x * tiles_y, dtype=np.uint32)
    x, y = (ids % tiles_x).astype(np.uint64), (ids // tiles_x).astype(np.uint64)

    def spread(v):                       # bit k of v -> bit 2k
        v = (v | (v << 16)) & np.uint64(0x0000FFFF0000FFFF); v = (v | (v << 8)) & np.uint64(0x00FF00FF00FF00FF)
        v = (v | (v << 4)) & np.uint64(0x0F0F0F0F0F0F0F0F); v = (v | (v << 2)) & np.uint64(0x3333333333333333)
        return (v | (v << 1)) & np.uint64(0x5555555555555555)
    return ids[np.argsort(spread(x) | (spread(y) << np.uint64(1)), kind="stable")]


def tiles_for_rank(n_tiles, rank, world, tiles_x=0):
    """Interleaved assignment of the screen tiles to `world` GPUs.  With the grid width `tiles_x`: the k-th tile along the Z-order
    (Morton) curve goes to rank k mod world -- the tiles of any 2 x 2 / 4 x 2 block of the image, which cost
    about the same, land on different GPUs whatever the grid width; measured shard imbalance on the 1 M-triangle benchmark frame
    in DESIGN.md section 5 (tools/tile_cost.py).  Without it: tile t -> rank t mod world.  The list is returned in ascending tile order."""
    if tiles_x and n_tiles % tiles_x == 0:
        return np.sort(morton_order(tiles_x, n_tiles // tiles_x)[rank::world]).astype(np.uint32)
    return np.arange(rank, n_tiles, world, dtype=np.uint32)


def tiles_x_of(backend):
    ts = getattr(backend, "tile_size", 0) or 32
    return (backend.width + ts - 1) // ts if getattr(backend, "width", 0) else 0


def render_shard(backend, rank, world, first_sample, n_samples):
    tiles = tiles_for_rank(backend.n_tiles(), rank, world, tiles_x_of(backend))
    if len(tiles):
        backend.render_tiles(tiles, first_sample, n_samples)
    return tiles


_staging = {}


def reduce_framebuffer(accum, dst=0):
    """Sum the per-rank float4 accumulators into rank `dst` (torch.distributed reduce; nccl == RCCL on ROCm).
    The reduce runs on a staging copy: a rank's own accumulator must keep holding only its own tiles, because
    rendering continues into it (progressive display reduces every few iterations).  The staging buffer is PERSISTENT (one per
    shape / device, reused by every step: no allocation in the timed region, and RCCL sees the same registered address every
    time).  Returns the staging tensor (the assembled frame on rank `dst`, valid until the next call); the caller's stream is
    synchronised before returning."""
    import torch
    import torch.distributed as dist
    key = (tuple(accum.shape), accum.dtype, str(accum.device))
    out = _staging.get(key)
    if out is None:
        out = _staging[key] = torch.empty_like(accum)
    out.copy_(accum)
    if dist.is_available() and dist.is_initialized():
        if out.is_cuda and dist.get_backend() == "gloo":       # gloo has no GPU reduce; only used to rehearse the flow on one GPU
            dist.all_reduce(out, op=dist.ReduceOp.SUM)
        else:
            dist.reduce(out, dst=dst, op=dist.ReduceOp.SUM)
    if out.is_cuda:
        torch.cuda.current_stream(out.device).synchronize()
    return out


def load_scene_shared(view, scene, dist=None, device=None, src=0):
    """One BVH build per job instead of one per rank (the scene is replicated): rank `src` builds (all of the host's build threads are
    its own -- the other ranks wait), exports its tree (crh_get_bvh) and broadcasts node array + leaf order; every other rank hands them to
    crh_build_prebuilt.  At 10 M triangles: 350 MB of nodes + 40 MB of order over xGMI / shared memory against 8 builds on 2 threads each.  Two-level
    scenes and single-rank jobs build locally.  Returns {"built_here": bool, "tree_bytes": n, "seconds": {...}}."""
    import time
    t0 = time.perf_counter()
    world = dist.get_world_size() if dist is not None and dist.is_initialized() else 1
    if world == 1 or getattr(scene, "tri_object", None) is not None:
        view.load_scene(scene)
        return {"built_here": True, "tree_bytes": 0, "seconds": {"load_scene": round(time.perf_counter() - t0, 3)}}
    import torch
    rank = dist.get_rank()
    dev = device if device is not None else torch.device("cpu")
    hdr = torch.zeros(2, dtype=torch.int64, device=dev)
    if rank == src:
        view.load_scene(scene)
        t1 = time.perf_counter()
        nodes, order = view.export_tree()
        hdr[0], hdr[1] = nodes.shape[0], order.shape[0]
        dist.broadcast(hdr, src=src)
        tn = torch.from_numpy(nodes.view(np.uint8).reshape(-1)).to(dev)
        to = torch.from_numpy(order.view(np.uint8).reshape(-1)).to(dev)
        dist.broadcast(tn, src=src); dist.broadcast(to, src=src)
        if tn.is_cuda:
            torch.cuda.synchronize(tn.device)
        return {"built_here": True, "tree_bytes": int(tn.numel() + to.numel()),
                "seconds": {"load_scene": round(t1 - t0, 3), "export_and_broadcast": round(time.perf_counter() - t1, 3)}}
    dist.broadcast(hdr, src=src)
    n_nodes, n_tris = int(hdr[0].item()), int(hdr[1].item())
    t1 = time.perf_counter()
    from . import abi
    tn = torch.empty(n_nodes * abi.NODE_DWORDS * 4, dtype=torch.uint8, device=dev)
    to = torch.empty(n_tris * 4, dtype=torch.uint8, device=dev)
    dist.broadcast(tn, src=src); dist.broadcast(to, src=src)
    nodes = tn.cpu().numpy().view(np.float32).reshape(n_nodes, abi.NODE_DWORDS)
    order = to.cpu().numpy().view(np.uint32)
    t2 = time.perf_counter()
    view.load_scene(scene, prebuilt=(nodes, order))
    return {"built_here": False, "tree_bytes": int(tn.numel() + to.numel()),
            "seconds": {"wait_for_builder": round(t1 - t0, 3), "receive": round(t2 - t1, 3), "load_prebuilt": round(time.perf_counter() - t2, 3)}}


class TileGather:
    """The cheaper exchange step of SURVEY.md 8e: every rank owns a disjoint set of tiles, so instead of summing whole frames (33 MB at 1080p, 133 MB at
    4K per rank) each rank sends ONLY its tiles' pixels -- W*H*16/N bytes -- and the root scatters them into the frame.  Pixel lists are built once per
    (frame shape, world size): rank r's pixels in tile order, rows inside a tile, clipped to the image, padded to the longest list with repeats of the
    rank's last pixel (a repeated pixel is written twice with the same value).  Values are copied, not summed: bit-exact by construction."""

    def __init__(self, width, height, tile_size, world, device):
        import torch
        self.world, self.shape = world, (height, width, 4)
        tx, ty = (width + tile_size - 1) // tile_size, (height + tile_size - 1) // tile_size
        lists = []
        for r in range(world):
            tiles = tiles_for_rank(tx * ty, r, world, tx)
            px = []
            yy, xx = np.mgrid[0:tile_size, 0:tile_size]
            for t in tiles:
                x = (int(t) % tx) * tile_size + xx; y = (int(t) // tx) * tile_size + yy
                ok = (x < width) & (y < height)
                px.append((y[ok].astype(np.int64) * width + x[ok]).reshape(-1))
            lists.append(np.concatenate(px) if px else np.zeros(0, np.int64))
        self.n_max = max(1, max(len(l) for l in lists))
        pad = [np.concatenate([l, np.full(self.n_max - len(l), l[-1] if len(l) else 0, np.int64)]) for l in lists]
        self.idx = [torch.from_numpy(p).to(device) for p in pad]
        self.bytes_per_rank = self.n_max * 16
        self._out = None

    def assemble(self, accum, rank, dst=0):
        """gather the owned tiles of every rank's accumulator into a persistent frame on rank `dst` (returned there; None elsewhere)"""
        import torch
        import torch.distributed as dist
        send = accum.view(-1, 4).index_select(0, self.idx[rank])
        via_cpu = send.is_cuda and dist.get_backend() == "gloo"        # rehearsal on one GPU: gloo has no device-side gather
        if via_cpu:
            send = send.cpu()
        if rank == dst:
            recv = [torch.empty_like(send) for _ in range(self.world)]
            dist.gather(send, recv, dst=dst)
            if self._out is None:
                self._out = torch.zeros(self.shape, dtype=accum.dtype, device=accum.device)
            flat = self._out.view(-1, 4)
            for r in range(self.world):
                flat.index_copy_(0, self.idx[r], recv[r].to(accum.device))
            if accum.is_cuda:
                torch.cuda.current_stream(accum.device).synchronize()
            return self._out
        dist.gather(send, None, dst=dst)
        if accum.is_cuda:
            torch.cuda.current_stream(accum.device).synchronize()
        return None


class DeviceFramebuffer:
    """Zero-copy torch view of a backend's device accumulator (crh_accum_device_ptr)."""

    def __init__(self, view):
        import torch
        if not torch.cuda.is_available():
            raise RuntimeError("torch sees no GPU: `import torch` must come before the first cadrays_amd.View in the process "
                               "(torch bundles its own HIP runtime and cannot initialise after the system one)")
        ptr, nbytes = view.accum_device_ptr()
        self.__cuda_array_interface__ = {"shape": (view.height, view.width, 4), "typestr": "<f4",
                                         "data": (ptr, False), "version": 2}
        self.tensor = torch.as_tensor(self, device=f"cuda:{view.device}")

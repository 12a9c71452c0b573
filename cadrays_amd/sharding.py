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

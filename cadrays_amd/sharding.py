"""Multi-GPU sharding of the path: screen tiles are the unit (the reference's RT tile concept,
src/Launcher/SettingsWidget.cxx:451-476).  Every GPU holds the whole scene (replicated; it is tiny next
to 288 GB), renders ALL samples of its own tiles with the single-GPU RNG (seed = f(pixel, frame)), so
the union of the shards is bit-identical to a 1-GPU render; one reduce of the float4 framebuffer over
RCCL/xGMI (disjoint support, zero elsewhere -> the sum is exact) assembles the image on rank 0.

Backend-agnostic: works on anything with render_tiles()/n_tiles(); the CPU tests drive it with the
oracle over gloo, bench.py with the HIP backend over nccl (= RCCL).
"""
import numpy as np


def tiles_for_rank(n_tiles, rank, world):
    """Interleaved assignment: tile t -> rank t mod world (load balance across image regions)."""
    return np.arange(rank, n_tiles, world, dtype=np.uint32)


def render_shard(backend, rank, world, first_sample, n_samples):
    tiles = tiles_for_rank(backend.n_tiles(), rank, world)
    if len(tiles):
        backend.render_tiles(tiles, first_sample, n_samples)
    return tiles


def reduce_framebuffer(accum, dst=0):
    """Sum the per-rank float4 accumulators into rank `dst` (torch.distributed reduce; nccl == RCCL on ROCm)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.reduce(accum, dst=dst, op=dist.ReduceOp.SUM)
    return accum


class DeviceFramebuffer:
    """Zero-copy torch view of a backend's device accumulator (crh_accum_device_ptr)."""

    def __init__(self, view):
        import torch
        ptr, nbytes = view.accum_device_ptr()
        self.__cuda_array_interface__ = {"shape": (view.height, view.width, 4), "typestr": "<f4",
                                         "data": (ptr, False), "version": 2}
        self.tensor = torch.as_tensor(self, device=f"cuda:{view.device}")

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
    """Sum the per-rank float4 accumulators into rank `dst` (torch.distributed reduce; nccl == RCCL on ROCm).
    The reduce runs on a staging copy: a rank's own accumulator must keep holding only its own tiles, because
    rendering continues into it (progressive display reduces every few iterations).  Returns the staging tensor
    (the assembled frame on rank `dst`); the caller's stream is synchronised before returning."""
    import torch
    import torch.distributed as dist
    out = accum.clone()
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

"""The RCCL legs of the N > 1 flow on the ONE GPU this pool has: a process group of world size 1 over the `nccl` backend (= RCCL on ROCm) runs the very
collectives bench.py --gpus N issues -- broadcast (tree hand-over), reduce (full-frame exchange), gather (owned tiles), all_reduce / all_gather (the
line's bookkeeping) -- on device tensors.  A backend that lacks one of them refuses it whatever the world size, so this is what can be known here about
the driver's 8-GPU run besides the gloo rehearsals (tests/test_sharding_gloo.py, tests/test_bench_cli.py).  Reference: SURVEY.md 8e."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CODE = r'''
import os, sys
sys.path.insert(0, %(root)r)
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=%(port)r, RANK="0", WORLD_SIZE="1", GPU_MAX_HW_QUEUES="16")
import numpy as np, torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
from cadrays_amd import scenes, sharding
from cadrays_amd.view import View
sc = scenes.baseline_config("C2", 256, 160, n_tris=20000)
v = View(0)
info = sharding.load_scene_shared(v, sc, dist, torch.device("cuda:0"))          # world 1: builds here
assert info["built_here"]
tiles = sharding.tiles_for_rank(v.n_tiles(), 0, 1, sharding.tiles_x_of(v))
v.render_tiles(tiles, 0, 4); v.sync()
fb = sharding.DeviceFramebuffer(v)
total = sharding.reduce_framebuffer(fb.tensor, 0)                               # dist.reduce over RCCL
g = sharding.TileGather(v.width, v.height, v.tile_size, 1, fb.tensor.device)
frame = g.assemble(fb.tensor, 0, 0)                                             # dist.gather over RCCL
assert torch.equal(frame, total) and torch.equal(total, fb.tensor)
# the tree hand-over's broadcasts, with a second context taking the tree
nodes, order = v.export_tree()
tn = torch.from_numpy(nodes.view(np.uint8).reshape(-1)).cuda(); dist.broadcast(tn, src=0)
w = View(0).load_scene(sc, prebuilt=(tn.cpu().numpy().view(np.float32).reshape(nodes.shape), order))
w.render_tiles(tiles, 0, 4)
assert np.array_equal(w.read_hdr().view(np.uint32), v.read_hdr().view(np.uint32))
t = torch.tensor([1.5], dtype=torch.float64, device="cuda:0"); dist.all_reduce(t, op=dist.ReduceOp.MAX)
out = [torch.zeros_like(t)]; dist.all_gather(out, t)
assert float(out[0].item()) == 1.5
print("rccl", ".".join(str(x) for x in torch.cuda.nccl.version()), "ok")
dist.barrier(); dist.destroy_process_group()
'''


@pytest.mark.gpu
def test_rccl_collectives_of_the_sharded_flow_on_one_gpu(hip_lib):
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = str(s.getsockname()[1]); s.close()
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    p = subprocess.run([sys.executable, "-c", CODE % {"root": ROOT, "port": port}], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    ok = [l for l in p.stdout.splitlines() if l.startswith("rccl ") and l.endswith(" ok")]      # RCCL prints its own banner after it
    assert p.returncode == 0 and ok, (p.stdout + p.stderr)[-3000:]

"""The two schedule x config cells the round-3 verdict found untested inside `-m gpu` (item 4b), through bench.py with its parity gates on:

  * BASELINE.json configs[3], C4 AS WRITTEN: C3's scene, 4096 spp per step (32 wide batches of 128 spp), fixed job = strong scaling, on the one GPU
    this pool has (`--gpus 1`; the driver's 8-GPU node runs the sharded thing);
  * BASELINE.json configs[4]'s scene, C5: the WIDE-batch schedule (one 265 M-path batch: 4K x 32 spp) on the 10 M-triangle tree -- the other C5 tests
    render 1 spp, i.e. the small-batch, work-donating kernels.
The reference's harness also runs every script it is given, not one (testing/CADRays_Testing.py:177-185)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(args, timeout):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    assert p.returncode == 0, (p.stdout + p.stderr)[-3000:]
    return json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])


@pytest.mark.gpu
def test_c4_as_written_on_one_gpu(hip_lib):
    out = _bench(["--config", "C4", "--spp", "4096", "--gpus", "1", "--steps", "1", "--warmup", "0", "--no-cpu", "--no-interactive", "--parity-seconds", "4", "--step0-seconds", "4"], 1500)
    assert out["scaling"] == "strong" and out["n_gpus"] == 1 and out["value"] > 500
    assert out["config"]["workload"].startswith("C4: 1000000") and out["config"]["spp_per_step_per_rank"] == 4096
    assert out["parity"]["bit_exact"] is True and out["parity"]["spp"] == 4096 and out["parity"]["pixels"] > 0
    assert out["parity_step0"]["bit_exact"] is True and out["parity_step0"]["tiles"] >= 32
    assert out["roofline"]["launches"] == 16 * 10                    # four tile groups of 512 tiles x four batches of 1024 samples (crh_schedule.cpp), ten bounces each


@pytest.mark.gpu
def test_c5_wide_batch_on_the_10m_triangle_tree(hip_lib):
    out = _bench(["--config", "C5", "--steps", "1", "--warmup", "0", "--no-cpu", "--no-interactive"], 1500)
    assert out["config"]["workload"].startswith("C5: 10000000") and out["config"]["spp_per_step_per_rank"] == 256
    assert "3840x2160" in out["config"]["workload"] and out["value"] > 500
    assert out["parity"]["bit_exact"] is True and "wide" in out["parity"]["schedule"] and out["parity"]["pixels"] > 0
    assert out["parity_step0"]["bit_exact"] is True and out["parity_step0"]["tiles"] >= 32
    assert out["roofline"]["scene_bytes"] > (256 << 20)              # not cache-resident: this is the config where HBM is the roof
    assert out["roofline"]["nodes_per_ray"] > 30

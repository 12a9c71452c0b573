"""BASELINE.json configs[3] (C4: C3's scene, 4096 spp, tiles sharded across the GPUs, fixed job = strong scaling) must at least execute once on the
one GPU this pool has: `bench.py --config C4 --spp 64 --gpus 1` (the driver's 8-GPU node runs the real thing)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_c4_code_path_runs_on_one_gpu(hip_lib):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", "C4", "--spp", "64", "--gpus", "1", "--steps", "1", "--warmup", "1",
                        "--no-cpu", "--no-interactive", "--parity-seconds", "3"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, (p.stdout + p.stderr)[-3000:]
    out = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert out["scaling"] == "strong" and out["n_gpus"] == 1 and out["value"] > 500
    assert out["config"]["workload"].startswith("C4: 1000000") and out["config"]["spp_per_step_per_rank"] == 64
    assert out["parity"]["bit_exact"] is True and out["parity"]["spp"] == 64

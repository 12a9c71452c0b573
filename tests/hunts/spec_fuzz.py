#!/usr/bin/env python3
"""One-off hunt: the random scenes of tests/test_gpu_fuzz.py under random combinations of ALL thirteen run-time switches of include/crh_spec.h (the suite's
tests/test_spec_switches.py runs 24 seeds of this), HIP path vs oracle, image + every counter; also with the counters off (the timed instantiations).
  python tests/hunts/spec_fuzz.py [first] [last]"""
import dataclasses, sys, importlib.util, numpy as np
sys.path.insert(0, '.')
import torch
spec = importlib.util.spec_from_file_location("fz", "tests/test_gpu_fuzz.py"); fz = importlib.util.module_from_spec(spec); spec.loader.exec_module(fz)
from cadrays_amd.view import View
from oracle.pyoracle import Oracle
a, b = (int(sys.argv[1]) if len(sys.argv) > 1 else 0), (int(sys.argv[2]) if len(sys.argv) > 2 else 200)
bad = []
for seed in range(a, b):
    r = np.random.default_rng(91000 + seed)
    kw = dict(uniform_32bit=int(r.integers(0, 2)), texel_gamma2=int(r.integers(0, 2)), mis_single_lobe=int(r.integers(0, 2)),
              eps_rule=int(r.integers(0, 2)), eta_no_dielectric=float(r.choice([1.0, 1.33, 1.5, 0.8])),
              rr_start_bounce=int(r.choice([3, 0, 1, 2, 5, 32])), rr_survival_cap=float(r.choice([0.95, 1.0, 0.5, 0.25, 0.01])),
              min_contribution=float(r.choice([1e-2, 0.0, 0.1, 0.5, 10.0])), min_throughput=float(r.choice([1e-3, 0.0, 0.05, 0.2, 2.0])),
              raygen_bilinear=int(r.integers(0, 3)), env_orientation=int(r.integers(0, 2)))
    sc = dataclasses.replace(fz.random_scene(5000 + seed), spec=kw)
    v = View(0).load_scene(sc); v.enable_counters(True); v.reset(); o = Oracle().load_scene(sc)
    v.render(2); o.render(2)
    gs, cs = v.stats(), o.stats()
    ok = np.array_equal(fz.bits(v.read_hdr()), fz.bits(o.read_hdr())) and all(gs[k] == cs[k] for k in gs if k != "seconds")
    w = View(0).load_scene(sc); w.render(1); w.render(1)
    ok = ok and np.array_equal(fz.bits(w.read_hdr()), fz.bits(o.read_hdr()))
    if not ok: bad.append((seed, kw))
    v.close(); w.close(); o.close()
print(f"{b - a} random scenes under random switches, mismatches:", bad)

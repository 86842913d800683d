"""Spread of |gpu hogwild hold-out loss - sequential oracle's| for every scenario of tests/test_zz_gpu_hogwild_quality.py, over repeated
runs and for the routes of the hot LR entry (kernels.hip hot_lr_flush): 0 = plain read-modify-writes, 1 = atomics per example (default),
32 = weight deltas pending 32 examples of a workgroup (HOT_ROUTES=0,1,32; default: the shipped route only).  The tests' TOL table is 1.3 x the
`max gap` column of this script's output on the shipped build (profiles/r04_holdout_spread.txt).
usage: python scripts/holdout_spread.py [reps=8] [scenario ...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import test_zz_gpu_hogwild_quality as Q  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 8
names = sys.argv[2:] or list(Q.SCENARIOS)
print(f"{'scenario':20s} {'hot_lr':>6s} {'oracle':>8s} {'gpu min':>8s} {'gpu max':>8s} {'max gap':>8s} {'gap/3':>8s}")
for name in names:
    for hot in [int(x) for x in os.environ.get("HOT_ROUTES", "1").split(",")]:
        vals, ref = [], None
        for _ in range(reps):
            g, ref = Q.SCENARIOS[name](hot_lr=hot)
            vals.append(g)
        v = np.array(vals)
        print(f"{name:20s} {hot:6d} {ref:8.4f} {v.min():8.4f} {v.max():8.4f} {np.abs(v - ref).max():8.4f} {(Q.LN2 - ref) / 3:8.4f}", flush=True)

"""The metric's own protocol as a test (main.rs:184-185, 238-241; benchmark/calc_loss.py:5-25): bench.py's driver shape at config C's FULL size
(28-bit tables, 65 536-example launches, 1.64 M training examples, the 262 144-example hold-out) against the committed curves of the CPU oracle on the same
stream -- the reference's single thread AND its 16-thread hogwild mode (tests/golden/bench_oracle_curve_*.json, scripts/make_bench_oracle_curve.py).
The concurrent GPU mode must not be behind the better of the two by more than 1.3 x its own measured run-to-run spread (three passes of `bench.py --long`
on one box: 0.0020 at 16.8 M examples, profiles/r05_long_protocol*.json; at 1.64 M the passes differ by less)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SPREAD = 0.0020


def test_driver_shape_holdout_loss_is_not_behind_the_reference_modes():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "20", "--warmup", "5", "--no-cpu-baseline", "--no-traffic", "--no-config-e",
                        "--no-config-b"], env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    d = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["config"]["holdout_examples"] == 262144 and d["steps"] == 20 and d["warmup"] == 5
    seq, hog = d["oracle_final_logloss"], d["oracle_hogwild16_final_logloss"]
    assert seq is not None and hog and len(hog) >= 3, "the committed oracle curves do not cover this run's stream"
    ref = min([seq] + hog)
    assert d["final_logloss"] <= ref + 1.3 * SPREAD, (d["final_logloss"], seq, hog)
    assert d["final_logloss"] < d["holdout_prior_logloss"] - 0.03  # ... and it has learned
    # every checkpoint of the run has its reference value beside it
    assert all(v is not None for v in d["oracle_logloss_after_examples"].values())
    assert d["saturated_fraction_last_step"] == 0.0
    r = d["roofline"]
    assert r["bound"] == "hbm" and 0.3 < r["frac"] < 1.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9

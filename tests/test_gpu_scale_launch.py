"""The driver's multi-GPU launch of bench.py (scripts/scale_run.sh = `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`)
with N = 8 PROCESSES on the one GPU of this box: torch.distributed over gloo as the side channel, the library's own communicator over the
shared-memory stand-in of tests/fake_rccl.  What it pins before a real node runs it: the launcher starts before any GPU call (the script
execs the launcher from bash), eight ranks come up at once on one device (eight regressors' placement searches, eight communicators),
the timed replica exchange, the sparse and sharded legs and the peer-sharded mode with its IPC-mapped tables (incl. the LR-shard
workaround for hipIpcOpenMemHandle) all complete, and the line reports the communicator's own rank count.  Functional, not a rate."""
import json
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _launch(n, *flags, timeout=900):
    env = dict(os.environ, FAKE="1", MASTER_PORT=str(29600 + n + len(flags)))
    cmd = ["bash", os.path.join(ROOT, "scripts", "scale_run.sh"), str(n), "--steps", "4", "--warmup", "1", "--batch", "2048", "--bits", "20", "--ffm-bits", "20",
           "--holdout", "1024", "--no-cpu-baseline", "--no-traffic", "--other-modes-timeout", "300", *flags]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout)
    assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-4000:])
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert lines, p.stdout[-2000:]
    return json.loads(lines[-1])


def test_eight_ranks_replica_mode_with_the_sparse_and_sharded_legs():
    d = _launch(8)
    assert d["n_gpus"] == 8 and d["rccl_ranks"] == 8, (d["n_gpus"], d.get("rccl_ranks"))
    assert d["value"] > 0 and 0.3 < d["final_logloss"] < 0.70
    assert "error" not in d.get("dp_modes", {}), d.get("dp_modes")
    assert d["dp_modes"]["sparse"]["value"] > 0 and d["dp_modes"]["sharded"]["value"] > 0
    lb = d["link_bytes_per_example"]
    assert lb["owner_apply"] < lb["peer"] < lb["sparse_upper_bound"]


def test_eight_ranks_peer_sharded_mode():
    d = _launch(8, "--dp-mode", "peer", "--no-other-modes")
    assert d["n_gpus"] == 8 and d["rccl_ranks"] == 8
    assert d["value"] > 0 and 0.3 < d["final_logloss"] < 0.70

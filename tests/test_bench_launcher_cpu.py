"""bench.py as its own launcher (`python3 bench.py --gpus N` with no WORLD_SIZE): what can be checked without a GPU -- the parent imports neither
torch nor the library, gives every rank the torchrun environment, relays rank 0's stdout only, and comes back non-zero when a rank fails
(here every rank does: there is no GPU and no CPU path)."""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_self_launch_without_gpu_fails_loudly_and_quickly():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    t0 = time.time()
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode != 0
    assert "[bench launcher] rank" in p.stderr and "needs a GPU" in p.stderr, p.stderr[-2000:]
    assert "[rank 0]" in p.stderr and "[rank 1]" in p.stderr
    assert not p.stdout.strip()
    assert time.time() - t0 < 240


def test_parent_makes_no_gpu_call():
    # the launcher path must return before torch / the library are imported (a parent holding a GPU context must never be the one that starts ranks)
    code = ("import sys, types; sys.argv=['bench.py','--gpus','2'];\n"
            "import bench\n"
            "bench.self_launch = lambda a: (print('torch' in sys.modules, 'fwumious_wabbit_amd' in sys.modules), 0)[1]\n"
            "try:\n    bench.main()\nexcept SystemExit as e:\n    print('rc', e.code)\n")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, capture_output=True, text=True, timeout=120)
    assert p.stdout.split() == ["False", "False", "rc", "0"], (p.stdout, p.stderr[-1000:])

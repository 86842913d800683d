"""Builds tests/golden/ffm_example/*.vw.gz: the reference's own synthetic FFM example data (BASELINE config A), produced by
running the reference's generator script examples/ffm/generate.py with the parameters its CI-style check uses
(examples/ffm/run_fw_with_prediction_tests.sh:45-49: 300 animals, 200 foods, 30 000 training examples), random_seed 1 (its
default).  Only the generated DATA is committed (gzip-compressed); the generator itself stays in the reference tree.
Run from the repo root, in the container that has /root/reference:  python tests/golden/make_ffm_example_data.py"""
import gzip
import os
import shutil
import subprocess
import sys
import tempfile

REF = "/root/reference/examples/ffm/generate.py"
out_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ffm_example")
os.makedirs(out_dir, exist_ok=True)
with tempfile.TemporaryDirectory() as d:
    subprocess.check_call([sys.executable, REF, "--num_animals", "300", "--num_foods", "200", "--num_train_examples", "30000",
                           "--num_eval_examples", "3000"], cwd=d)
    for name in ("train.vw", "test-easy.vw", "test-hard.vw", "vw_namespace_map.csv"):
        src = os.path.join(d, "datasets", name)
        with open(src, "rb") as f, gzip.GzipFile(os.path.join(out_dir, name + ".gz"), "wb", mtime=0) as g:
            g.write(f.read())
        print(name, os.path.getsize(src), "bytes")

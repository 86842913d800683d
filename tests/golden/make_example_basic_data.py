"""Copies the DATA files of the reference's examples/basic (== examples/vw-compatibility) into tests/golden/example_basic:
train.vw (100 production-like lines: 58 namespaces, multi-feature namespaces, namespace weights, empty features) and
vw_namespace_map.csv.  They are data the reference ships for its own example runs (examples/basic/run.sh,
examples/vw-compatibility/run.sh); nothing else is taken.  Run from the repo root in the container with /root/reference."""
import gzip
import os
import shutil

SRC = "/root/reference/examples/basic/datasets"
dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "example_basic")
os.makedirs(dst, exist_ok=True)
for name in ("vw_namespace_map.csv", "train.vw"):
    with open(os.path.join(SRC, name), "rb") as f, gzip.GzipFile(os.path.join(dst, name + ".gz"), "wb", mtime=0) as g:
        g.write(f.read())
print(os.listdir(dst))

"""GPU: end-to-end rate of the HogwildTrainer replacement fed from HOST memory (records -> PCIe -> device translate+learn)."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import fwumious_wabbit_amd as fw
import bench
class A: pass
args = A(); args.fields, args.k, args.bits, args.ffm_bits = 30, 8, 28, 28
args.nn_layers, args.nn_width = 0, 256
args.mean_extra, args.zipf, args.ids, args.p_weighted, args.seed = 5.67, 1.05, 10_000_000, 0.1, 20240612
n = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
mi = bench.build_model_instance(fw, args, 0)
re = fw.Regressor(mi)
recs, off = bench.gen_records(fw, args, 0, n)
for mb in (4096, 16384, 65536):
    tr = fw.HogwildTrainer(re, mi, micro_batch=mb)
    tr.digest_records(recs[: int(off[mb])], off[: mb + 1]); tr.block_until_workers_finished()   # warm-up
    t0 = time.perf_counter()
    tr.digest_records(recs, off)
    tr.block_until_workers_finished()
    dt = time.perf_counter() - t0
    print(f"micro_batch={mb}: {n} records ({recs.nbytes/1e6:.0f} MB) in {dt*1e3:.1f} ms = {n/dt/1e6:.2f} M examples/s from host memory", flush=True)
    tr.close()

# ---- the same stream through a .fwcache file: file -> RecordCache.next_records -> digest_records -> device
import tempfile
from fwumious_wabbit_amd.feed import RecordCache, VwNamespaceMap
vw = VwNamespaceMap("".join(f"A{i},ns{i}\n" for i in range(args.fields)))
d = tempfile.mkdtemp(dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
inp = os.path.join(d, "stream.vw")
c = RecordCache(inp, True, vw); c.push_records(recs); c.write_finish(); c.close()
for rep in range(2):
    tr = fw.HogwildTrainer(re, mi, micro_batch=16384)
    c = RecordCache(inp, True, vw)
    t0 = time.perf_counter(); seen = 0
    while True:
        w, o = c.next_records(words_cap=1 << 25, max_records=1 << 17)
        if len(o) <= 1:
            break
        tr.digest_records(w, o); seen += len(o) - 1
    tr.block_until_workers_finished()
    dt = time.perf_counter() - t0
    print(f"cache file -> trainer (pass {rep}): {seen} records in {dt*1e3:.1f} ms = {seen/dt/1e6:.2f} M examples/s", flush=True)
    c.close(); tr.close()
for rep in range(2):
    tr = fw.HogwildTrainer(re, mi, micro_batch=16384)
    c = RecordCache(inp, True, vw)
    tr.digest_cache(c, max_records=min(65536, n // 4))  # warm-up: the trainer pins its staging memory and allocates its device buffers on first use
    tr.block_until_workers_finished()
    t0 = time.perf_counter()
    seen = tr.digest_cache(c)
    tr.block_until_workers_finished()
    dt = time.perf_counter() - t0
    print(f"cache file -> trainer, native loop (pass {rep}, after a 65 536-record warm-up): {seen} records in {dt*1e3:.1f} ms = {seen/dt/1e6:.2f} M examples/s", flush=True)
    c.close(); tr.close()
os.remove(inp + ".fwcache")

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

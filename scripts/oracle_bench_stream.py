"""CPU: what hold-out log-loss does the sequential reference algorithm reach on exactly bench.py's stream?"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import fwumious_wabbit_amd as fw
import bench
from oracle import fwo
class A: pass
args = A(); args.fields, args.k, args.bits, args.ffm_bits = 30, 8, 28, 28
args.nn_layers, args.nn_width = 0, 256
args.mean_extra, args.zipf, args.ids, args.p_weighted, args.seed = 5.67, 1.05, 10_000_000, 0.1, 20240612
n_train = int(sys.argv[1]) if len(sys.argv) > 1 else 14 * 16384
threads = int(sys.argv[2]) if len(sys.argv) > 2 else 1
F = 30
ocfg = fwo.make_config(optimizer=fwo.OPT_ADAGRAD_LUT, learning_rate=bench.LR, ffm_learning_rate=bench.LR, power_t=bench.POWER_T,
                       ffm_power_t=bench.POWER_T, init_acc_gradient=bench.INIT_ACC, ffm_init_acc_gradient=bench.INIT_ACC,
                       bit_precision=28, num_combos=F + 1, ffm_k=8, ffm_bit_precision=28, ffm_num_fields=F)
ots = fwo.TranslatorSpec([([(i, False)], 1.0) for i in range(F)], [[(i, False)] for i in range(F)], True, 28, 8, 28)
om = fwo.Model(ocfg, native=True)
recs, off = bench.gen_records(fw, args, 0, n_train)
t = time.time(); om.run_stream(ots, recs, off, nthreads=threads, want_preds=False); dt = time.time() - t
hrecs, hoff = bench.gen_records(fw, args, 1_000_000_000, 8192)
hy = hrecs[hoff[:-1].astype(np.int64) + 1].astype(np.float32)
_, p = om.run_stream(ots, hrecs, hoff, holdout_after=1, nthreads=1)
print(f"oracle threads={threads}: trained {n_train} examples in {dt:.1f}s ({n_train/dt:.0f} ex/s); hold-out log-loss {bench.logloss(p, hy):.4f}")

"""Two tenants on one GPU: two regressors learn at the same time on two HIP streams (two hardware queues), each in the in-order mode, and each
must come out exactly as the sequential oracle says -- predictions per example and final tables.  Round 3 traced an irreproducibility of the
PHASE kernels of two queues overlapping to scalar registers spilled to VGPR lanes; this holds the fused kernels -- the
config-C kernel with its 20 kept rows among them -- to exactness under the same overlap, with a third stream running hogwild launches of a
third model beside them to keep every CU busy."""
import numpy as np
import pytest
import torch

import fwumious_wabbit_amd as fw
from fwumious_wabbit_amd import _capi as capi
from helpers import logloss, make_pair, record_labels
from oracle import fwo

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("shape", ["config_c_kernel", "generic_kernel", "deep_head_kernel"])
def test_two_learners_on_two_streams_are_each_the_sequential_reference(shape):
    nn_layers = None
    if shape == "deep_head_kernel":  # k = 16 rows (two chunks) + a 2 x 64 ReLU head: fw_example_kernel<4, AdagradLUT, true, 0, true, false>, the instantiation with the most lane-spilled scalars
        geo = dict(n_ns=12, k=16, bits=16, ffm_bits=18, mean_extra=2.0, p_weighted=0.1, ids=5000, lr=0.025, power_t=0.38)
        n = 600
        nn_layers = [(64, "relu", "hu"), (64, "relu", "hu")]
    elif shape == "config_c_kernel":  # 30 fields, k = 8, the large-table update path forced onto small tables: fw_example_kernel_r<.., 20, true, 1, POL>
        geo = dict(n_ns=30, k=8, bits=18, ffm_bits=20, mean_extra=5.67, p_weighted=0.1, ids=20000, lr=0.025, power_t=0.38)
        n = 700
    else:                            # k = 10: the generic kernel
        geo = dict(n_ns=6, k=10, bits=16, ffm_bits=16, mean_extra=1.0, p_weighted=0.1, ids=3000, lr=0.1, power_t=0.5)
        n = 3000
    streams = [torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()]
    tenants = []
    for t, seed in enumerate((1231, 4567)):
        mi, ocfg, ots = make_pair(geo["n_ns"], geo["k"], geo["bits"], geo["ffm_bits"], fw.Optimizer.AdagradLUT, lr=geo["lr"], ffm_lr=geo["lr"],
                                  power_t=geo["power_t"], ffm_power_t=geo["power_t"])
        recs, off = fw.synth_records(geo["n_ns"], geo["mean_extra"], 1.1, geo["ids"], geo["p_weighted"], seed, 0, n)
        nn = None
        if nn_layers:
            mi.nn_layers = [dict(width=w, activation=a, init=i) for w, a, i in nn_layers]
            mi.nn_topology = "one"
            mi.nn_learning_rate, mi.nn_power_t, mi.nn_init_acc_gradient = 0.025, 0.38, 1.0
            nn = fwo.make_nn_config(nn_layers, "one", 0.025, 0.38, 1.0)
        re = fw.Regressor(mi)
        if nn_layers:  # (Hu draws are implementation-defined: the oracle model below starts from the same ones)
            om0 = fwo.Model(ocfg, nn=nn)
            w0 = np.concatenate([om0.nn_weights(l).copy() for l in range(len(nn_layers) + 1)])
            om0.close()
            re.table_write(capi.TABLE_NN_W, w0)
            ocfg = (ocfg, nn)
        if shape == "config_c_kernel":
            re.set_whole_line_updates(3)
        fbt = fw.FeatureBufferTranslator(mi)
        # three consecutive launches per tenant, so that launches of the two tenants start and end at different moments
        cuts = [0, n // 3, 2 * n // 3, n]
        batches = [re.record_batch(fbt, recs[int(off[a]):int(off[b])], off[a:b + 1] - off[a]) for a, b in zip(cuts[:-1], cuts[1:])]
        tenants.append((mi, ocfg, ots, recs, off, re, batches))
    # the noise maker: a hogwild learner of its own
    mi3, _, _ = make_pair(10, 4, 18, 18, fw.Optimizer.AdagradLUT)
    recs3, off3 = fw.synth_records(10, 0.0, 1.1, 100000, 0.0, 99, 0, 60000)
    re3 = fw.Regressor(mi3)
    b3 = re3.record_batch(fw.FeatureBufferTranslator(mi3), recs3, off3)
    for rep in range(3):
        for t, (_, _, _, _, _, re, batches) in enumerate(tenants):
            re.learn_batch(batches[rep], capi.MODE_SEQUENTIAL, True, streams[t].cuda_stream)
        re3.learn_batch(b3, capi.MODE_HOGWILD, True, streams[2].cuda_stream)
    torch.cuda.synchronize()
    for mi, ocfg, ots, recs, off, re, batches in tenants:
        y = record_labels(recs, off)
        om = fwo.Model(ocfg[0], nn=ocfg[1]) if isinstance(ocfg, tuple) else fwo.Model(ocfg)
        _, p_ref = om.run_stream(ots, recs, off, holdout_after=0, nthreads=1)
        p_gpu = np.concatenate([b.predictions() for b in batches])
        assert np.abs(logloss(p_gpu, y) - logloss(p_ref, y)).max() < 1e-4
        assert np.abs(p_gpu - p_ref).max() < 5e-5
        w_gpu, w_ref = re.table_read(capi.TABLE_FFM_W), np.asarray(om.ffm_weights)
        bad = np.abs(w_gpu - w_ref) > 2e-5 + 1e-5 * np.abs(w_ref)
        assert int(bad.sum()) <= max(3, w_gpu.size // 10000), int(bad.sum())  # (AdagradLUT bucket edges: a last-bit difference in acc may pick the neighbouring LUT entry)
        for b in batches:
            b.close()
        re.close()
    b3.close()
    re3.close()

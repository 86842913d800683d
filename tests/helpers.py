"""Shared helpers for the parity tests: build matching oracle / GPU models and synthetic streams."""
import numpy as np

import fwumious_wabbit_amd as fw
from oracle import fwo


def mi_from_cfg(d, wiring="regressor", n_fields=None):
    """ModelInstance from a golden-scenario config dict."""
    F = d.get("ffm_num_fields", 0) if n_fields is None else n_fields
    nc = d.get("num_combos", 1)
    mi = fw.ModelInstance(
        learning_rate=d["learning_rate"], ffm_learning_rate=d["ffm_learning_rate"], bit_precision=d["bit_precision"],
        power_t=d["power_t"], ffm_power_t=d["ffm_power_t"], add_constant_feature=True,
        feature_combo_descs=[fw.FeatureComboDesc([fw.NamespaceDescriptor(i)]) for i in range(nc - 1)],
        ffm_fields=[[fw.NamespaceDescriptor(i)] for i in range(F)], ffm_k=d.get("ffm_k", 0),
        ffm_bit_precision=d.get("ffm_bit_precision", 18), ffm_init_acc_gradient=d.get("ffm_init_acc_gradient", 0.0),
        init_acc_gradient=d["init_acc_gradient"], optimizer=d["optimizer"],
        wiring=fw.capi.WIRING_FFM_ONLY if wiring == "ffm_only" else fw.capi.WIRING_REGRESSOR)
    return mi


def make_pair(n_ns, k, bits, ffm_bits, optimizer, lr=0.1, ffm_lr=0.1, power_t=0.5, ffm_power_t=0.5, init_acc=1.0,
              ffm_init_acc=None, interactions=()):
    """(ModelInstance, fw translator, oracle config, oracle translator) for n_ns namespaces == fields, LR --keep for
    every namespace (+ the given namespace-pair interactions) + constant.

    ffm_init_acc defaults to init_acc, as the reference's command line does (model_instance.rs:423:
    ffm_init_acc_gradient defaults to init_acc_gradient).  NOTE: ffm_init_acc=0 with AdaGrad makes every first
    step on a weight +-learning_rate whatever the gradient size; with lr=0.1 on ~10^4 weights per example the
    training dynamics are chaotic and f32 summation-order noise is amplified to O(1) within tens of examples, for
    ANY two implementations that do not add in the same order (the reference's own SSE and scalar paths included).
    Stream-parity tests therefore run in the default regime; the exactness of each mechanism at
    ffm_init_acc=0 is pinned by the KAT scenarios and the crafted single-example cases."""
    if ffm_init_acc is None:
        ffm_init_acc = init_acc
    combos = [fw.FeatureComboDesc([fw.NamespaceDescriptor(i)]) for i in range(n_ns)]
    combos += [fw.FeatureComboDesc([fw.NamespaceDescriptor(a), fw.NamespaceDescriptor(b)]) for a, b in interactions]
    mi = fw.ModelInstance(learning_rate=lr, ffm_learning_rate=ffm_lr, bit_precision=bits, power_t=power_t,
                          ffm_power_t=ffm_power_t, add_constant_feature=True, feature_combo_descs=combos,
                          ffm_fields=[[fw.NamespaceDescriptor(i)] for i in range(n_ns)] if k else [], ffm_k=k,
                          ffm_bit_precision=ffm_bits, init_acc_gradient=init_acc, ffm_init_acc_gradient=ffm_init_acc,
                          optimizer=optimizer)
    ocfg = fwo.make_config(optimizer=optimizer, learning_rate=lr, power_t=power_t, init_acc_gradient=init_acc,
                           bit_precision=bits, num_combos=mi.num_combos, ffm_k=k, ffm_bit_precision=ffm_bits,
                           ffm_num_fields=n_ns if k else 0, ffm_learning_rate=ffm_lr, ffm_power_t=ffm_power_t,
                           ffm_init_acc_gradient=ffm_init_acc)
    ots = fwo.TranslatorSpec([([(i, False)], 1.0) for i in range(n_ns)] + [([(a, False), (b, False)], 1.0) for a, b in interactions],
                             [[(i, False)] for i in range(n_ns)] if k else [], True, bits, k, ffm_bits)
    return mi, ocfg, ots


def logloss(p, y):
    """benchmark/calc_loss.py:5-25"""
    p = np.clip(np.asarray(p, dtype=np.float64), 1e-15, 1 - 1e-15)
    y = np.asarray(y)
    return -np.where(y == 1, np.log(p), np.log(1 - p))


def record_labels(records, rec_off):
    return np.array([records[int(o) + 1] for o in rec_off[:-1]], dtype=np.float32)

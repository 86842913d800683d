"""Host-side mirror of the reference interface for the LR+FFM path, over the C ABI (include/fwgpu.h).

Names, argument meaning and error behaviour follow the reference so that the parity tests read like
the reference's own tests:

  ModelInstance            model_instance.rs:47-97 (only the fields this path reads)
  FeatureBuffer            feature_buffer.rs:24-31
  FeatureBufferTranslator  feature_buffer.rs:33-44, 138-338
  Regressor                regressor.rs:142-147, 356-395, 426-469
  HogwildTrainer           hogwild.rs:13-61

All compute happens in libfwgpu.so on the GPU; nothing here falls back to the CPU.
"""
import ctypes as C
from dataclasses import dataclass, field
from typing import List, Optional, Tuple

import numpy as np

from . import _capi as capi
from ._capi import FFM_ENTRY, LR_ENTRY, check, ptr


class Optimizer:
    """model_instance.rs:24-28"""
    SGD = capi.OPT_SGD
    AdagradFlex = capi.OPT_ADAGRAD_FLEX
    AdagradLUT = capi.OPT_ADAGRAD_LUT


@dataclass
class NamespaceDescriptor:
    """vwmap.rs:23-27 (primitive namespaces only)."""
    namespace_index: int
    namespace_format_f32: bool = False


@dataclass
class FeatureComboDesc:
    """model_instance.rs FeatureComboDesc {namespace_descriptors, weight}."""
    namespace_descriptors: List[NamespaceDescriptor]
    weight: float = 1.0


@dataclass
class ModelInstance:
    """Defaults = ModelInstance::new_empty() (model_instance.rs:120-150)."""
    learning_rate: float = 0.5
    ffm_learning_rate: float = 0.5
    bit_precision: int = 18
    power_t: float = 0.5
    ffm_power_t: float = 0.5
    add_constant_feature: bool = True
    feature_combo_descs: List[FeatureComboDesc] = field(default_factory=list)
    ffm_fields: List[List[NamespaceDescriptor]] = field(default_factory=list)
    ffm_k: int = 0
    ffm_bit_precision: int = 18
    ffm_init_center: float = 0.0
    ffm_init_width: float = 0.0
    ffm_init_zero_band: float = 0.0
    ffm_init_acc_gradient: float = 0.0
    init_acc_gradient: float = 1.0
    optimizer: int = Optimizer.SGD
    # deep head (nn_config.layers / topology, model_instance.rs:32-44, 139-142): one dict per hidden layer with the
    # reference's keys: width (default 20), activation ("relu" | "none"), init ("hu" default | "xavier" | "one" | "zero")
    nn_layers: List[dict] = field(default_factory=list)
    nn_topology: str = "one"
    nn_learning_rate: float = 0.02
    nn_power_t: float = 0.45
    nn_init_acc_gradient: float = 0.0
    # not in the reference: which graph the regressor is wired as (tests of the bare FFM block) and the device
    wiring: int = capi.WIRING_REGRESSOR
    device: int = 0

    @property
    def num_combos(self):  # block_lr.rs:53-56
        return len(self.feature_combo_descs) + (1 if self.add_constant_feature else 0)

    def to_config(self):
        return capi.Config(self.optimizer, self.learning_rate, self.power_t, self.init_acc_gradient, self.bit_precision,
                           self.num_combos, self.ffm_k, self.ffm_bit_precision, len(self.ffm_fields),
                           self.ffm_learning_rate, self.ffm_power_t, self.ffm_init_acc_gradient, self.ffm_init_center,
                           self.ffm_init_width, self.ffm_init_zero_band, self.wiring, self.device)


@dataclass
class FeatureBuffer:
    """feature_buffer.rs:24-31.  lr_buffer rows (hash, value, combo_index); ffm_buffer rows
    (hash, value, contra_field_index)."""
    label: float = 0.0
    example_importance: float = 1.0
    example_number: int = 0
    lr_buffer: np.ndarray = field(default_factory=lambda: np.zeros(0, dtype=LR_ENTRY))
    ffm_buffer: np.ndarray = field(default_factory=lambda: np.zeros(0, dtype=FFM_ENTRY))


def lr_vec(rows, label=0.0, importance=1.0):
    """regressor.rs:544-552 test helper."""
    return FeatureBuffer(label, importance, 0, np.array([tuple(r) for r in rows], dtype=LR_ENTRY).reshape(-1))


def ffm_vec(rows, label=0.0, importance=1.0):
    """block_ffm.rs:1219-1227 test helper."""
    return FeatureBuffer(label, importance, 0, np.zeros(0, dtype=LR_ENTRY),
                         np.array([tuple(r) for r in rows], dtype=FFM_ENTRY).reshape(-1))


def lr_and_ffm_vec(lr_rows, ffm_rows, label=0.0, importance=1.0):
    """persistence.rs:420-433 test helper."""
    return FeatureBuffer(label, importance, 0, np.array([tuple(r) for r in lr_rows], dtype=LR_ENTRY).reshape(-1),
                         np.array([tuple(r) for r in ffm_rows], dtype=FFM_ENTRY).reshape(-1))


class FeatureBufferTranslator:
    """feature_buffer.rs:33-44: record (&[u32]) -> FeatureBuffer.  Pure host code inside libfwgpu."""

    def __init__(self, mi: ModelInstance):
        self.mi = mi
        co, cn, cf, cw = [0], [], [], []
        for cd in mi.feature_combo_descs:
            for nd in cd.namespace_descriptors:
                cn.append(nd.namespace_index)
                cf.append(int(nd.namespace_format_f32))
            co.append(len(cn))
            cw.append(cd.weight)
        fo, fn, ff = [0], [], []
        for fld in mi.ffm_fields:
            for nd in fld:
                fn.append(nd.namespace_index)
                ff.append(int(nd.namespace_format_f32))
            fo.append(len(fn))
        self._keep = [np.array(co, np.uint32), np.array(cn, np.uint32), np.array(cf, np.uint8),
                      np.array(cw, np.float32), np.array(fo, np.uint32), np.array(fn, np.uint32),
                      np.array(ff, np.uint8)]
        k = self._keep
        self.c = capi.TranslatorConfig(len(mi.feature_combo_descs), ptr(k[0]), ptr(k[1]), ptr(k[2]), ptr(k[3]),
                                       int(mi.add_constant_feature), len(mi.ffm_fields), ptr(k[4]), ptr(k[5]),
                                       ptr(k[6]), mi.bit_precision, mi.ffm_k, mi.ffm_bit_precision)
        L = capi.lib()
        self.lr_hash_mask = L.fwgpu_lr_hash_mask(mi.bit_precision)
        self.ffm_hash_mask = L.fwgpu_ffm_hash_mask(mi.ffm_bit_precision, mi.ffm_k)
        self.feature_buffer = FeatureBuffer()

    def translate(self, record_buffer, example_number=0, cap=8192):
        """feature_buffer.rs:174-176"""
        L = capi.lib()
        rec = np.ascontiguousarray(record_buffer, dtype=np.uint32)
        lr = np.zeros(cap, dtype=LR_ENTRY)
        ffm = np.zeros(cap, dtype=FFM_ENTRY)
        n_lr, n_ffm = C.c_uint32(0), C.c_uint32(0)
        label, imp = C.c_float(0), C.c_float(0)
        check(L.fwgpu_translate(C.byref(self.c), ptr(rec), len(rec), ptr(lr), cap, C.byref(n_lr), ptr(ffm), cap,
                                C.byref(n_ffm), C.byref(label), C.byref(imp)))
        self.feature_buffer = FeatureBuffer(label.value, imp.value, example_number, lr[: n_lr.value].copy(),
                                            ffm[: n_ffm.value].copy())
        return self.feature_buffer


class Batch:
    """A device-resident micro-batch of FeatureBuffers (CSR)."""

    def __init__(self, regressor, handle):
        self.regressor, self.h = regressor, handle
        n, n_lr, n_ffm = C.c_uint32(0), C.c_uint64(0), C.c_uint64(0)
        check(capi.lib().fwgpu_batch_size(self.h, C.byref(n), C.byref(n_lr), C.byref(n_ffm)))
        self.n, self.n_lr, self.n_ffm = n.value, n_lr.value, n_ffm.value

    def set_cache(self, cache):
        """predict-only launches of this entry batch start every example from the context cache (None detaches it)"""
        check(capi.lib().fwgpu_batch_set_cache(self.h, cache.h if cache is not None else None))

    def predictions(self, stream=None):
        out = np.zeros(self.n, dtype=np.float32)
        check(capi.lib().fwgpu_batch_predictions(self.h, ptr(out), self.n, stream))
        return out

    def close(self):
        if self.h:
            capi.lib().fwgpu_batch_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class SplitBuffers:
    """device buffers of one synchronous micro-batch (fwgpu_split)"""

    def __init__(self, regressor, n_examples, max_ffm_per_example):
        self.h = C.c_void_p()
        check(capi.lib().fwgpu_split_create(regressor.h, n_examples, max_ffm_per_example, C.byref(self.h)))

    def close(self):
        if self.h:
            capi.lib().fwgpu_split_free(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class BlockCache:
    """Vec<BlockCache> (regressor.rs:40-50): what setup_cache leaves on the device for predict_with_cache"""

    def __init__(self, regressor):
        self.h = C.c_void_p()
        self.regressor = regressor

    def filter(self, ffm_entries) -> np.ndarray:
        """the FFM entries forward_with_cache still gathers (block_ffm.rs:548, 600)"""
        ffm = np.ascontiguousarray(ffm_entries, dtype=FFM_ENTRY)
        out = np.zeros(len(ffm), dtype=FFM_ENTRY)
        n = C.c_uint32(0)
        check(capi.lib().fwgpu_block_cache_filter(self.h, ptr(ffm), len(ffm), ptr(out), C.byref(n)))
        return out[: n.value]

    def close(self):
        if self.h:
            capi.lib().fwgpu_block_cache_free(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Regressor:
    """regressor.rs:142-147.  `Regressor(mi)` == Regressor::new(&mi): builds the graph and initialises weights."""

    def __init__(self, mi: ModelInstance, init_weights: bool = True):
        self.mi = mi
        self.L = capi.lib()
        self.h = C.c_void_p()
        cfg = mi.to_config()
        check(self.L.fwgpu_create(C.byref(cfg), C.byref(self.h)))
        if mi.nn_layers:  # regressor.rs:191-320
            nn = capi.NNConfig()
            nn.n_layers = len(mi.nn_layers)
            for i, layer in enumerate(mi.nn_layers):
                nn.width[i] = int(layer.get("width", 20))
                nn.relu[i] = int(layer.get("activation", "none") == "relu")
                nn.init[i] = capi.NN_INIT[layer.get("init", "hu")]
            if mi.nn_topology not in ("one", "two"):
                raise ValueError(f'unknown nn topology: "{mi.nn_topology}"')
            nn.topology = {"one": 1, "two": 2}[mi.nn_topology]
            nn.nn_learning_rate, nn.nn_power_t, nn.nn_init_acc_gradient = (mi.nn_learning_rate, mi.nn_power_t,
                                                                            mi.nn_init_acc_gradient)
            check(self.L.fwgpu_set_nn(self.h, C.byref(nn)))
        if init_weights:
            self.allocate_and_init_weights()

    @classmethod
    def from_handle(cls, mi, handle):
        """wrap a regressor the library created itself (fwgpu_model_load)"""
        self = cls.__new__(cls)
        self.L = capi.lib()
        self.mi = mi
        self.h = handle
        return self

    @classmethod
    def new_without_weights(cls, mi):  # regressor.rs:173
        return cls(mi, init_weights=False)

    def allocate_and_init_weights(self, mi=None):  # regressor.rs:352-354
        check(self.L.fwgpu_init_weights(self.h))

    def get_name(self):  # regressor.rs:177
        names = {Optimizer.SGD: "SGD", Optimizer.AdagradFlex: "AdagradFlex", Optimizer.AdagradLUT: "AdagradLUT"}
        return f'Regressor with optimizer "{names[self.mi.optimizer]}"'

    def new_portbuffer(self):  # regressor.rs:348-350: the tape lives in LDS on the device
        return None

    def close(self):
        if self.h:
            self.L.fwgpu_free(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- regressor.rs:356-395
    def learn(self, fb: FeatureBuffer, pb=None, update: bool = True) -> float:
        lr = np.ascontiguousarray(fb.lr_buffer, dtype=LR_ENTRY)
        ffm = np.ascontiguousarray(fb.ffm_buffer, dtype=FFM_ENTRY)
        out = C.c_float(0)
        check(self.L.fwgpu_learn(self.h, ptr(lr), len(lr), ptr(ffm), len(ffm), fb.label, fb.example_importance,
                                 int(update), C.byref(out)))
        return out.value

    def predict(self, fb: FeatureBuffer, pb=None) -> float:
        lr = np.ascontiguousarray(fb.lr_buffer, dtype=LR_ENTRY)
        ffm = np.ascontiguousarray(fb.ffm_buffer, dtype=FFM_ENTRY)
        out = C.c_float(0)
        check(self.L.fwgpu_predict(self.h, ptr(lr), len(lr), ptr(ffm), len(ffm), C.byref(out)))
        return out.value

    # ---- serving context cache (regressor.rs:397-423)
    def setup_cache(self, fb: FeatureBuffer, cache=None):
        """Regressor::setup_cache: the context features' field sums are computed once on the device; returns the cache
        (pass an existing one to refill it, as the reference does with should_create == false)"""
        lr = np.ascontiguousarray(fb.lr_buffer, dtype=LR_ENTRY)
        ffm = np.ascontiguousarray(fb.ffm_buffer, dtype=FFM_ENTRY)
        cache = cache if cache is not None else BlockCache(self)
        check(self.L.fwgpu_setup_cache(self.h, ptr(lr), len(lr), ptr(ffm), len(ffm), C.byref(cache.h)))
        return cache

    def predict_with_cache(self, fb: FeatureBuffer, pb, cache) -> float:
        """Regressor::predict_with_cache: fb is the FeatureBuffer of context + candidate"""
        lr = np.ascontiguousarray(fb.lr_buffer, dtype=LR_ENTRY)
        ffm = np.ascontiguousarray(fb.ffm_buffer, dtype=FFM_ENTRY)
        out = C.c_float(0)
        check(self.L.fwgpu_predict_with_cache(self.h, cache.h, ptr(lr), len(lr), ptr(ffm), len(ffm), C.byref(out)))
        return out.value

    # ---- micro-batches
    def batch(self, fbs: List[FeatureBuffer]) -> Batch:
        lr = np.concatenate([np.ascontiguousarray(f.lr_buffer, dtype=LR_ENTRY) for f in fbs]) if fbs else np.zeros(0, LR_ENTRY)
        ffm = np.concatenate([np.ascontiguousarray(f.ffm_buffer, dtype=FFM_ENTRY) for f in fbs]) if fbs else np.zeros(0, FFM_ENTRY)
        lr_off = np.zeros(len(fbs) + 1, dtype=np.uint32)
        ffm_off = np.zeros(len(fbs) + 1, dtype=np.uint32)
        lr_off[1:] = np.cumsum([len(f.lr_buffer) for f in fbs])
        ffm_off[1:] = np.cumsum([len(f.ffm_buffer) for f in fbs])
        label = np.array([f.label for f in fbs], dtype=np.float32)
        imp = np.array([f.example_importance for f in fbs], dtype=np.float32)
        return self.batch_from_arrays(lr, lr_off, ffm, ffm_off, label, imp)

    def batch_from_arrays(self, lr, lr_off, ffm, ffm_off, label, importance) -> Batch:
        h = C.c_void_p()
        lr = np.ascontiguousarray(lr, dtype=LR_ENTRY)
        ffm = np.ascontiguousarray(ffm, dtype=FFM_ENTRY)
        lr_off = np.ascontiguousarray(lr_off, dtype=np.uint32)
        ffm_off = np.ascontiguousarray(ffm_off, dtype=np.uint32)
        label = np.ascontiguousarray(label, dtype=np.float32)
        importance = np.ascontiguousarray(importance, dtype=np.float32)
        # zero-length arrays still need a valid offsets pointer
        check(self.L.fwgpu_batch_create(self.h, ptr(lr), lr_off.ctypes.data_as(C.c_void_p), ptr(ffm),
                                        ffm_off.ctypes.data_as(C.c_void_p), ptr(label), ptr(importance), len(label),
                                        C.byref(h)))
        return Batch(self, h)

    def batch_from_records(self, translator: FeatureBufferTranslator, records, rec_off) -> Batch:
        records = np.ascontiguousarray(records, dtype=np.uint32)
        rec_off = np.ascontiguousarray(rec_off, dtype=np.uint64)
        h = C.c_void_p()
        check(self.L.fwgpu_batch_from_records(self.h, C.byref(translator.c), ptr(records),
                                              rec_off.ctypes.data_as(C.c_void_p), len(rec_off) - 1, C.byref(h)))
        return Batch(self, h)

    def record_batch(self, translator: FeatureBufferTranslator, records, rec_off) -> Batch:
        """Raw records to HBM; translation happens on the device inside the example kernel."""
        records = np.ascontiguousarray(records, dtype=np.uint32)
        rec_off = np.ascontiguousarray(rec_off, dtype=np.uint64)
        h = C.c_void_p()
        check(self.L.fwgpu_record_batch_create(self.h, C.byref(translator.c), ptr(records),
                                               rec_off.ctypes.data_as(C.c_void_p), len(rec_off) - 1, C.byref(h)))
        return Batch(self, h)

    def learn_batch(self, batch: Batch, mode=capi.MODE_SEQUENTIAL, update=True, stream=None):
        check(self.L.fwgpu_learn_batch(self.h, batch.h, mode, int(update), stream))

    # ---- synchronous micro-batches (the step of the sharded multi-GPU mode and of the mini-batched deep head)
    def split_buffers(self, n_examples, max_ffm_per_example=512):
        return SplitBuffers(self, n_examples, max_ffm_per_example)

    def learn_batch_sync(self, batch, split, mode=capi.MODE_HOGWILD, stream=None):
        """every example of the batch is scored with the weights of the batch start, then all updates are applied"""
        check(self.L.fwgpu_learn_batch_sync(self.h, batch.h, split.h, mode, stream))

    def set_launch(self, threads=0, workgroups_per_cu=0):
        check(self.L.fwgpu_set_launch(self.h, threads, workgroups_per_cu))

    def set_whole_line_updates(self, mode):
        """update path of large models (fwgpu_debug_set_option 2): 0 = round-1 path (float-granular, repeated rows serialised);
        1 = auto, the default (tables beyond the Infinity Cache: duplicate-row chains, whole 128 B lines only when w and acc
        contend for one memory region); 2 = chains + whole-line accesses always (kernels.hip update_rows_win); 3 = chains with
        float-granular accesses always (the path of large, well-placed tables on a table of any size)"""
        check(self.L.fwgpu_debug_set_option(self.h, 2, int(mode)))

    def set_store_policy(self, policy, flush_every=-1):
        """HOGWILD launches of the large-table update path (fwgpu_debug_set_option 5 / 6): how FFM row stores reach memory -- 0 both
        tables device-scope write-through, 1 weight rows write-back through the XCD's L2, 2 both tables write-back, 3 (the default) = 1 with thinned accumulator stores on hot kept rows,
        4 = 3 with the thinned store replaced by a thinned atomic add of m g^2 on every hot row (lossless in expectation),
        -1 the build's default; with 1 / 2 a workgroup writes its XCD's dirty L2 lines back every `flush_every` of its examples (0 = only when the
        launch ends, -1 = the build's default).  kernels.hip "store policy", tests/test_gpu_conservation.py"""
        check(self.L.fwgpu_debug_set_option(self.h, 5, int(policy)))
        check(self.L.fwgpu_debug_set_option(self.h, 6, int(flush_every)))

    def set_hot_row_sampling(self, theta=-1.0, sample_log2=-1):
        """store policies 3 / 4 (fwgpu_debug_set_option 9 / 10): a row is hot once its accumulators have grown by more than `theta` (default 0.5;
        resolution 1/1024); one example in 2^sample_log2 (default 3) then touches its accumulator row, with that many times its g^2"""
        check(self.L.fwgpu_debug_set_option(self.h, 9, -1 if theta < 0 else int(round(theta * 1024))))
        check(self.L.fwgpu_debug_set_option(self.h, 10, int(sample_log2)))

    def set_kept_rows(self, on):
        """HOGWILD launches of the large-table kernel keep 20 + 3 rows per wave from the gather and write them back as w_gather - step (1, default: the faster learner, damped on the rows
        many examples hold) or re-read every row in the update (0: 14 % slower; on the bench's stream the reference's own curve) -- fwgpu_debug_set_option 13"""
        check(self.L.fwgpu_debug_set_option(self.h, 13, int(on)))

    def set_head_kernel(self, v2):
        """deep head of HOGWILD launches with two-chunk rows (config E): as a phase of the large-table kernel (1; default -1 = where two workgroups fit a CU)
        or on the generic kernel (0) -- fwgpu_debug_set_option 11"""
        check(self.L.fwgpu_debug_set_option(self.h, 11, int(v2)))

    def set_prefetch(self, on):
        """updating launches copy the next example's record to LDS during the current example (fwgpu_debug_set_option 7; default on)"""
        check(self.L.fwgpu_debug_set_option(self.h, 7, int(bool(on))))

    def set_lds_keep(self, rows):
        """rows per wave beyond the register-kept ones whose gather-time weights stay in LDS for the update (fwgpu_debug_set_option 8;
        -1 = automatic, the default)"""
        check(self.L.fwgpu_debug_set_option(self.h, 8, int(rows)))

    def set_hot_lr_entry(self, every):
        """HOGWILD launches (fwgpu_debug_set_option 4): the constant feature's LR entry is stepped with atomics -- the step size
        always from the GLOBAL accumulator (returning atomic add of g^2), the weight delta by a fire-and-forget atomic, sent at
        once (`every` = 1, the default) or kept pending in LDS for `every` examples of the workgroup; 0 = plain
        read-modify-writes per example, which serialise on that entry and lose most of the concurrent ones"""
        check(self.L.fwgpu_debug_set_option(self.h, 4, int(every)))

    # ---- tables
    def set_max_in_flight(self, n):
        """cap on the examples a HOGWILD launch processes concurrently (16 = hogwild.rs's default thread count; 0 = no cap)"""
        check(self.L.fwgpu_set_max_in_flight(self.h, n))

    def table_len(self, which):
        n = C.c_uint64(0)
        check(self.L.fwgpu_table_len(self.h, which, C.byref(n)))
        return n.value

    def table_read(self, which, offset=0, count=None):
        count = self.table_len(which) - offset if count is None else count
        out = np.zeros(count, dtype=np.float32)
        check(self.L.fwgpu_table_read(self.h, which, offset, count, ptr(out)))
        return out

    def table_write(self, which, values, offset=0):
        v = np.ascontiguousarray(values, dtype=np.float32)
        check(self.L.fwgpu_table_write(self.h, which, offset, len(v), ptr(v)))

    def table_fill(self, which, value):
        check(self.L.fwgpu_table_fill(self.h, which, value))

    def table_checksum(self, which):
        c = C.c_uint64(0)
        check(self.L.fwgpu_table_checksum(self.h, which, C.byref(c)))
        return c.value

    def table_device_ptr(self, which):
        p = C.c_void_p()
        check(self.L.fwgpu_table_device_ptr(self.h, which, C.byref(p)))
        return p.value

    def placement(self):
        """(candidate allocations tried for the FFM accumulators, fastest, slowest pair-probe ms) -- fwgpu_debug_placement"""
        n, lo, hi = C.c_int32(), C.c_float(), C.c_float()
        check(self.L.fwgpu_debug_placement(self.h, C.byref(n), C.byref(lo), C.byref(hi)))
        return n.value, lo.value, hi.value

    def table_as_torch(self, which):
        """Zero-copy torch view of a table (device memory stays owned by the library)."""
        import torch

        n = self.table_len(which)

        class _View:
            pass

        v = _View()
        v.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f4", "data": (self.table_device_ptr(which), False),
                                      "version": 2}
        t = torch.as_tensor(v, device=f"cuda:{self.mi.device}")
        t._fwgpu_owner = self  # keep the regressor alive as long as the view
        return t

    def ffm_fill(self, w):
        """The reference tests' ffm_init (block_ffm.rs:1228-1235): weights = w, accumulators = initial_data()."""
        self.table_fill(capi.TABLE_FFM_W, w)
        a0 = self.mi.ffm_init_acc_gradient if self.mi.optimizer == Optimizer.AdagradFlex else 0.0
        self.table_fill(capi.TABLE_FFM_ACC, a0)

    # ---- regressor.rs:426-469
    def write_weights_to_buf(self) -> bytes:
        n = C.c_uint64(0)
        check(self.L.fwgpu_serialized_len(self.h, C.byref(n)))
        buf = np.zeros(n.value, dtype=np.uint8)
        w = C.c_uint64(0)
        check(self.L.fwgpu_write_weights(self.h, ptr(buf), n.value, C.byref(w)))
        return buf[: w.value].tobytes()

    def overwrite_weights_from_buf(self, blob: bytes):
        b = np.frombuffer(blob, dtype=np.uint8)
        check(self.L.fwgpu_read_weights(self.h, ptr(b), len(b)))


class HogwildTrainer:
    """hogwild.rs:13-61.  num_workers has no meaning on the device; micro_batch replaces it."""

    def __init__(self, regressor: Regressor, model_instance: ModelInstance, num_workers: int = 16,
                 micro_batch: int = 4096):
        self.regressor = regressor
        self.translator = FeatureBufferTranslator(model_instance)
        self.h = C.c_void_p()
        check(capi.lib().fwgpu_trainer_create(regressor.h, C.byref(self.translator.c), micro_batch, C.byref(self.h)))

    def digest_example(self, feature_buffer):
        """hogwild.rs:51-53: one record (Vec<u32>)."""
        rec = np.ascontiguousarray(feature_buffer, dtype=np.uint32)
        off = np.array([0, len(rec)], dtype=np.uint64)
        check(capi.lib().fwgpu_digest_records(self.h, ptr(rec), ptr(off), 1))

    def digest_records(self, records, rec_off):
        records = np.ascontiguousarray(records, dtype=np.uint32)
        rec_off = np.ascontiguousarray(rec_off, dtype=np.uint64)
        check(capi.lib().fwgpu_digest_records(self.h, ptr(records), ptr(rec_off), len(rec_off) - 1))

    def set_holdout(self, holdout_after=0, testonly=False):
        """main.rs:184-185, 238-241: examples numbered >= holdout_after (from 1), or all with testonly, are predicted, not learned"""
        check(capi.lib().fwgpu_trainer_set_holdout(self.h, holdout_after, int(testonly)))

    def predictions(self) -> np.ndarray:
        """predictions of the examples that were not learned, in stream order (call after block_until_workers_finished)"""
        n = C.c_uint64()
        check(capi.lib().fwgpu_trainer_predictions(self.h, None, 0, C.byref(n)))
        out = np.zeros(n.value, dtype=np.float32)
        check(capi.lib().fwgpu_trainer_predictions(self.h, ptr(out), out.size, C.byref(n)))
        return out

    def digest_cache(self, cache, max_records=0) -> int:
        """the example loop over a RecordCache opened for reading (main.rs:213-270), in native code"""
        n = C.c_uint64()
        check(capi.lib().fwgpu_trainer_digest_cache(self.h, cache.h, max_records, C.byref(n)))
        return n.value

    def digest_text(self, parser, text: bytes, cache=None, threads=0):
        """VW text -> device in native code; returns (examples learned, bytes consumed, status): status is OK, PARSE_FLUSH,
        PARSE_HOGWILD_LOAD; parse errors raise"""
        n, used = C.c_uint64(), C.c_uint64()
        rc = capi.lib().fwgpu_trainer_digest_text(self.h, parser.h, cache.h if cache is not None else None, text, len(text),
                                                  threads, C.byref(n), C.byref(used))
        if rc not in (capi.OK, capi.PARSE_FLUSH, capi.PARSE_HOGWILD_LOAD):
            check(rc)
        return n.value, used.value, rc

    def digest_file(self, parser, filename: str, cache=None, threads=0):
        """the example loop over a .vw / .gz / .zst file (main.rs:213-270), in native code -> (examples learned, status)"""
        n = C.c_uint64()
        rc = capi.lib().fwgpu_trainer_digest_file(self.h, parser.h, cache.h if cache is not None else None, filename.encode(),
                                                  threads, C.byref(n))
        if rc not in (capi.OK, capi.PARSE_FLUSH, capi.PARSE_HOGWILD_LOAD):
            check(rc)
        return n.value, rc

    def block_until_workers_finished(self):
        """hogwild.rs:55-60"""
        check(capi.lib().fwgpu_finish(self.h))

    def examples_seen(self):
        n = C.c_uint64(0)
        check(capi.lib().fwgpu_trainer_examples_seen(self.h, C.byref(n)))
        return n.value

    def close(self):
        if self.h:
            capi.lib().fwgpu_trainer_free(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def synth_records(n_namespaces, mean_extra, zipf_s, ids_per_ns, p_weighted, seed, first_example, n):
    """Synthetic record stream of the BASELINE.json configs (see fwgpu_synth_records in include/fwgpu.h).
    Returns (records u32[], rec_off u64[n+1])."""
    L = capi.lib()
    cfg = capi.SynthConfig(n_namespaces, mean_extra, zipf_s, ids_per_ns, p_weighted, seed)
    nw = C.c_uint64(0)
    rec_off = np.zeros(n + 1, dtype=np.uint64)
    check(L.fwgpu_synth_records(C.byref(cfg), first_example, n, None, 0, ptr(rec_off), C.byref(nw)))
    records = np.zeros(nw.value, dtype=np.uint32)
    check(L.fwgpu_synth_records(C.byref(cfg), first_example, n, ptr(records), nw.value, ptr(rec_off), C.byref(nw)))
    return records, rec_off

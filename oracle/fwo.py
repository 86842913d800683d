"""ctypes binding of the CPU parity oracle (oracle/fw_oracle.c).

TEST INFRASTRUCTURE ONLY.  Import this from tests/, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` -- never from ``fwumious_wabbit_amd``.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

OPT_SGD, OPT_ADAGRAD_FLEX, OPT_ADAGRAD_LUT = 100, 200, 300
WIRING_REGRESSOR, WIRING_FFM_ONLY = 0, 1

LR_ENTRY = np.dtype([("hash", "<u4"), ("value", "<f4"), ("combo_index", "<u4")])
FFM_ENTRY = np.dtype([("hash", "<u4"), ("value", "<f4"), ("contra_field_index", "<u4")])


class Config(C.Structure):
    _fields_ = [
        ("optimizer", C.c_int32),
        ("learning_rate", C.c_float),
        ("power_t", C.c_float),
        ("init_acc_gradient", C.c_float),
        ("bit_precision", C.c_uint32),
        ("num_combos", C.c_uint32),
        ("ffm_k", C.c_uint32),
        ("ffm_bit_precision", C.c_uint32),
        ("ffm_num_fields", C.c_uint32),
        ("ffm_learning_rate", C.c_float),
        ("ffm_power_t", C.c_float),
        ("ffm_init_acc_gradient", C.c_float),
        ("ffm_init_center", C.c_float),
        ("ffm_init_width", C.c_float),
        ("ffm_init_zero_band", C.c_float),
        ("wiring", C.c_int32),
    ]


NN_INIT = {"xavier": 0, "hu": 1, "one": 2, "zero": 3}


class NNConfig(C.Structure):
    _fields_ = [
        ("n_layers", C.c_uint32),
        ("width", C.c_uint32 * 8),
        ("relu", C.c_uint32 * 8),
        ("init", C.c_uint32 * 8),
        ("topology", C.c_uint32),
        ("nn_learning_rate", C.c_float),
        ("nn_power_t", C.c_float),
        ("nn_init_acc_gradient", C.c_float),
    ]


def make_nn_config(layers, topology="one", nn_learning_rate=0.02, nn_power_t=0.45, nn_init_acc_gradient=0.0):
    """layers: list of (width, activation 'relu'|'none', init 'hu'|'xavier'|'one'|'zero')."""
    c = NNConfig()
    c.n_layers = len(layers)
    for i, (w, act, init) in enumerate(layers):
        c.width[i], c.relu[i], c.init[i] = w, int(act == "relu"), NN_INIT[init]
    c.topology = {"one": 1, "two": 2}[topology]
    c.nn_learning_rate, c.nn_power_t, c.nn_init_acc_gradient = nn_learning_rate, nn_power_t, nn_init_acc_gradient
    return c


class Translator(C.Structure):
    _fields_ = [
        ("n_combos", C.c_uint32),
        ("combo_off", C.c_void_p),
        ("combo_ns", C.c_void_p),
        ("combo_ns_f32", C.c_void_p),
        ("combo_weight", C.c_void_p),
        ("add_constant_feature", C.c_int32),
        ("bit_precision", C.c_uint32),
        ("ffm_k", C.c_uint32),
        ("ffm_bit_precision", C.c_uint32),
        ("n_fields", C.c_uint32),
        ("field_off", C.c_void_p),
        ("field_ns", C.c_void_p),
        ("field_ns_f32", C.c_void_p),
    ]


def build(native=False):
    target = "native" if native else "all"
    subprocess.run(["make", "-C", _HERE, target], check=True, stdout=subprocess.DEVNULL)


_lib_cache = {}


def lib(native=False):
    """Load (building on demand) the oracle shared library."""
    if native in _lib_cache:
        return _lib_cache[native]
    name = "libfworacle_native.so" if native else "libfworacle.so"
    path = os.path.join(_HERE, "_build", name)
    src_m = max(os.path.getmtime(os.path.join(_HERE, f)) for f in ("fw_oracle.c", "fw_oracle.h"))
    if native or not os.path.exists(path) or os.path.getmtime(path) < src_m:
        build(native)
    L = C.CDLL(path)
    f32p, u32p, u64p, vp = C.POINTER(C.c_float), C.POINTER(C.c_uint32), C.POINTER(C.c_uint64), C.c_void_p
    L.fwo_lut_init.argtypes = [f32p, C.c_float, C.c_float, C.c_float]
    L.fwo_step_sgd.restype = C.c_float
    L.fwo_step_sgd.argtypes = [C.c_float, C.c_float]
    L.fwo_step_flex.restype = C.c_float
    L.fwo_step_flex.argtypes = [C.c_float, C.c_float, C.c_float, f32p]
    L.fwo_step_lut.restype = C.c_float
    L.fwo_step_lut.argtypes = [f32p, C.c_float, f32p]
    L.fwo_murmur3_32.restype = C.c_uint32
    L.fwo_murmur3_32.argtypes = [C.c_char_p, C.c_size_t, C.c_uint32]
    L.fwo_merand48.restype = C.c_float
    L.fwo_merand48.argtypes = [C.c_uint64]
    L.fwo_create.restype = vp
    L.fwo_create.argtypes = [C.POINTER(Config)]
    L.fwo_free.argtypes = [vp]
    L.fwo_init_weights.argtypes = [vp]
    L.fwo_ffm_fill.argtypes = [vp, C.c_float]
    for fn in ("fwo_lr_table", "fwo_ffm_weights", "fwo_ffm_acc"):
        getattr(L, fn).restype = f32p
        getattr(L, fn).argtypes = [vp, u64p]
    for fn in ("fwo_lut_lr", "fwo_lut_ffm"):
        getattr(L, fn).restype = f32p
        getattr(L, fn).argtypes = [vp]
    ex = [vp, vp, C.c_uint32, vp, C.c_uint32]
    L.fwo_learn.restype = C.c_float
    L.fwo_learn.argtypes = ex + [C.c_float, C.c_float, C.c_int]
    L.fwo_forward_backward.restype = C.c_float
    L.fwo_forward_backward.argtypes = ex + [C.c_float, C.c_float, C.c_int]
    L.fwo_predict.restype = C.c_float
    L.fwo_predict.argtypes = ex
    L.fwo_set_nn.restype = C.c_int
    L.fwo_set_nn.argtypes = [vp, C.POINTER(NNConfig)]
    for fn in ("fwo_nn_weights", "fwo_nn_acc"):
        getattr(L, fn).restype = f32p
        getattr(L, fn).argtypes = [vp, C.c_uint32, u64p]
    L.fwo_neuron_layer_fb.argtypes = [C.c_int, C.c_float, C.c_float, C.c_float, f32p, f32p, C.c_uint32, C.c_uint32, f32p,
                                      f32p, f32p, C.c_int]
    L.fwo_triangle_forward.argtypes = [f32p, C.c_uint32, f32p]
    L.fwo_triangle_backward.argtypes = [f32p, C.c_uint32, f32p]
    L.fwo_lr_hash_mask.restype = C.c_uint32
    L.fwo_lr_hash_mask.argtypes = [C.c_uint32]
    L.fwo_ffm_hash_mask.restype = C.c_uint32
    L.fwo_ffm_hash_mask.argtypes = [C.c_uint32, C.c_uint32]
    L.fwo_translate.restype = C.c_int
    L.fwo_translate.argtypes = [C.POINTER(Translator), vp, vp, C.c_uint32, u32p, vp, C.c_uint32, u32p, f32p, f32p]
    L.fwo_learn_window_emulation.restype = None
    L.fwo_learn_window_emulation.argtypes = [vp, C.POINTER(Translator), vp, vp, C.c_uint64, C.c_uint32, C.c_uint32, vp]
    L.fwo_learn_minibatch.restype = None
    L.fwo_learn_minibatch.argtypes = [vp, C.POINTER(Translator), vp, vp, C.c_uint64, vp]
    L.fwo_learn_sparse.restype = None
    L.fwo_learn_sparse.argtypes = [vp, C.POINTER(Translator), vp, vp, C.c_uint64, vp, C.c_uint32, vp]
    L.fwo_run_stream.restype = C.c_double
    L.fwo_run_stream.argtypes = [vp, C.POINTER(Translator), vp, vp, C.c_uint64, C.c_uint64, C.c_int, vp]
    L.fwo_predict_stream.restype = None
    L.fwo_predict_stream.argtypes = [vp, C.POINTER(Translator), vp, vp, C.c_uint64, C.c_int, vp]
    _lib_cache[native] = L
    return L


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None and a.size else None


def lr_entries(rows):
    """rows: iterable of (hash, value, combo_index)."""
    return np.array([tuple(r) for r in rows], dtype=LR_ENTRY).reshape(-1)


def ffm_entries(rows):
    """rows: iterable of (hash, value, contra_field_index)."""
    return np.array([tuple(r) for r in rows], dtype=FFM_ENTRY).reshape(-1)


def make_config(optimizer=OPT_SGD, learning_rate=0.5, power_t=0.5, init_acc_gradient=1.0, bit_precision=18,
                num_combos=1, ffm_k=0, ffm_bit_precision=18, ffm_num_fields=0, ffm_learning_rate=0.5,
                ffm_power_t=0.5, ffm_init_acc_gradient=0.0, ffm_init_center=0.0, ffm_init_width=0.0,
                ffm_init_zero_band=0.0, wiring=WIRING_REGRESSOR):
    """Defaults = ModelInstance::new_empty (model_instance.rs:120-150); num_combos=1 is the
    constant-feature slot those tests get from add_constant_feature=true with no combos."""
    return Config(optimizer, learning_rate, power_t, init_acc_gradient, bit_precision, num_combos, ffm_k,
                  ffm_bit_precision, ffm_num_fields, ffm_learning_rate, ffm_power_t, ffm_init_acc_gradient,
                  ffm_init_center, ffm_init_width, ffm_init_zero_band, wiring)


class TranslatorSpec:
    """Host-side description of combos/fields -> fwo_translator (arrays kept alive here)."""

    def __init__(self, combos, fields, add_constant_feature, bit_precision, ffm_k, ffm_bit_precision):
        """combos: list of (list of (ns_index, is_f32), weight); fields: list of list of (ns_index, is_f32)."""
        self.combos, self.fields = combos, fields
        self.add_constant_feature = int(bool(add_constant_feature))
        self.bit_precision, self.ffm_k, self.ffm_bit_precision = bit_precision, ffm_k, ffm_bit_precision
        co, cn, cf, cw = [0], [], [], []
        for members, w in combos:
            for ns, is_f32 in members:
                cn.append(ns)
                cf.append(int(is_f32))
            co.append(len(cn))
            cw.append(w)
        fo, fn, ff = [0], [], []
        for members in fields:
            for ns, is_f32 in members:
                fn.append(ns)
                ff.append(int(is_f32))
            fo.append(len(fn))
        self._a = dict(
            co=np.array(co, dtype=np.uint32), cn=np.array(cn, dtype=np.uint32), cf=np.array(cf, dtype=np.uint8),
            cw=np.array(cw, dtype=np.float32), fo=np.array(fo, dtype=np.uint32), fn=np.array(fn, dtype=np.uint32),
            ff=np.array(ff, dtype=np.uint8))
        a = self._a
        self.c = Translator(len(combos), _ptr(a["co"]), _ptr(a["cn"]), _ptr(a["cf"]), _ptr(a["cw"]),
                            self.add_constant_feature, bit_precision, ffm_k, ffm_bit_precision, len(fields),
                            _ptr(a["fo"]), _ptr(a["fn"]), _ptr(a["ff"]))

    @property
    def num_combos(self):
        return len(self.combos) + self.add_constant_feature

    def translate(self, record, cap=8192, native=False):
        L = lib(native)
        rec = np.ascontiguousarray(record, dtype=np.uint32)
        lr = np.zeros(cap, dtype=LR_ENTRY)
        ffm = np.zeros(cap, dtype=FFM_ENTRY)
        n_lr, n_ffm = C.c_uint32(0), C.c_uint32(0)
        label, imp = C.c_float(0), C.c_float(0)
        rc = L.fwo_translate(C.byref(self.c), _ptr(rec), _ptr(lr), cap, C.byref(n_lr), _ptr(ffm), cap,
                             C.byref(n_ffm), C.byref(label), C.byref(imp))
        if rc != 0:
            raise RuntimeError("fwo_translate overflow")
        return lr[: n_lr.value].copy(), ffm[: n_ffm.value].copy(), label.value, imp.value


class Model:
    def __init__(self, cfg, init=True, native=False, nn=None):
        self.L = lib(native)
        self.cfg = cfg
        self.h = self.L.fwo_create(C.byref(cfg))
        if not self.h:
            raise ValueError("fwo_create failed (k*F^2 > 41472?)")
        self.nn = nn
        if nn is not None and self.L.fwo_set_nn(self.h, C.byref(nn)) != 0:
            raise ValueError("fwo_set_nn failed")
        if init:
            self.L.fwo_init_weights(self.h)

    def close(self):
        if self.h:
            self.L.fwo_free(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def ffm_fill(self, w):
        self.L.fwo_ffm_fill(self.h, w)

    def _view(self, fn, mult=1):
        n = C.c_uint64(0)
        p = getattr(self.L, fn)(self.h, C.byref(n))
        if not n.value:
            return np.zeros(0, dtype=np.float32)
        return np.ctypeslib.as_array(p, shape=(n.value * mult,))

    @property
    def lr_table(self):
        """Interleaved {w, acc}: shape (2*2^b,)."""
        return self._view("fwo_lr_table", 2)

    @property
    def ffm_weights(self):
        return self._view("fwo_ffm_weights")

    @property
    def ffm_acc(self):
        return self._view("fwo_ffm_acc")

    def nn_weights(self, layer):
        n = C.c_uint64(0)
        p = self.L.fwo_nn_weights(self.h, layer, C.byref(n))
        return np.ctypeslib.as_array(p, shape=(n.value,))

    def nn_acc(self, layer):
        n = C.c_uint64(0)
        p = self.L.fwo_nn_acc(self.h, layer, C.byref(n))
        return np.ctypeslib.as_array(p, shape=(n.value,))

    def lut(self, which="lr"):
        p = (self.L.fwo_lut_lr if which == "lr" else self.L.fwo_lut_ffm)(self.h)
        return np.ctypeslib.as_array(p, shape=(2048,)).copy()

    def _ex(self, lr, ffm):
        lr = np.zeros(0, dtype=LR_ENTRY) if lr is None else np.ascontiguousarray(lr, dtype=LR_ENTRY)
        ffm = np.zeros(0, dtype=FFM_ENTRY) if ffm is None else np.ascontiguousarray(ffm, dtype=FFM_ENTRY)
        return lr, ffm

    def learn(self, lr=None, ffm=None, label=0.0, importance=1.0, update=True):
        lr, ffm = self._ex(lr, ffm)
        return self.L.fwo_learn(self.h, _ptr(lr), len(lr), _ptr(ffm), len(ffm), label, importance, int(update))

    def forward_backward(self, lr=None, ffm=None, label=0.0, importance=1.0, update=True):
        lr, ffm = self._ex(lr, ffm)
        return self.L.fwo_forward_backward(self.h, _ptr(lr), len(lr), _ptr(ffm), len(ffm), label, importance,
                                           int(update))

    def predict(self, lr=None, ffm=None):
        lr, ffm = self._ex(lr, ffm)
        return self.L.fwo_predict(self.h, _ptr(lr), len(lr), _ptr(ffm), len(ffm))

    def learn_minibatch(self, tspec, records, rec_off):
        """synchronous micro-batch (fw_oracle.h): all examples scored against the current weights, then all updates"""
        records = np.ascontiguousarray(records, dtype=np.uint32)
        rec_off = np.ascontiguousarray(rec_off, dtype=np.uint64)
        n = len(rec_off) - 1
        preds = np.zeros(n, dtype=np.float32)
        self.L.fwo_learn_minibatch(self.h, C.byref(tspec.c), _ptr(records), _ptr(rec_off), n, _ptr(preds))
        return preds

    def learn_window_emulation(self, tspec, records, rec_off, window, flags, want_preds=False):
        """emulation of the device's concurrent mode (fw_oracle.c fwo_learn_window_emulation): analysis only, not a reference code path"""
        records = np.ascontiguousarray(records, dtype=np.uint32)
        rec_off = np.ascontiguousarray(rec_off, dtype=np.uint64)
        n = len(rec_off) - 1
        preds = np.zeros(n, dtype=np.float32) if want_preds else None
        self.L.fwo_learn_window_emulation(self.h, C.byref(tspec.c), _ptr(records), _ptr(rec_off), n, int(window), int(flags), _ptr(preds) if want_preds else None)
        return preds

    def learn_sparse(self, tspec, records, rec_off, part_end=None):
        """row-sparse gradient buckets (fw_oracle.h): one optimizer step per row with the gradient summed over the batch;
        part_end = cumulative example counts of the ranks' micro-batches (default: one part)"""
        records = np.ascontiguousarray(records, dtype=np.uint32)
        rec_off = np.ascontiguousarray(rec_off, dtype=np.uint64)
        n = len(rec_off) - 1
        pe = np.ascontiguousarray([n] if part_end is None else part_end, dtype=np.uint64)
        assert int(pe[-1]) == n
        preds = np.zeros(n, dtype=np.float32)
        self.L.fwo_learn_sparse(self.h, C.byref(tspec.c), _ptr(records), _ptr(rec_off), n, _ptr(pe), len(pe), _ptr(preds))
        return preds

    def run_stream(self, tspec, records, rec_off, holdout_after=0, nthreads=1, want_preds=True):
        records = np.ascontiguousarray(records, dtype=np.uint32)
        rec_off = np.ascontiguousarray(rec_off, dtype=np.uint64)
        n = len(rec_off) - 1
        preds = np.zeros(n, dtype=np.float32) if want_preds else None
        dt = self.L.fwo_run_stream(self.h, C.byref(tspec.c), _ptr(records), _ptr(rec_off), n, holdout_after,
                                   nthreads, _ptr(preds) if preds is not None else None)
        return dt, preds

    def predict_stream(self, tspec, records, rec_off, nthreads=1):
        """update=false pass over every record (main.rs:238-241) on nthreads threads; order-independent."""
        records = np.ascontiguousarray(records, dtype=np.uint32)
        rec_off = np.ascontiguousarray(rec_off, dtype=np.uint64)
        n = len(rec_off) - 1
        preds = np.zeros(n, dtype=np.float32)
        self.L.fwo_predict_stream(self.h, C.byref(tspec.c), _ptr(records), _ptr(rec_off), n, nthreads, _ptr(preds))
        return preds

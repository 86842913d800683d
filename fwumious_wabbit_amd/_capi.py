"""ctypes loader for libfwgpu.so (the C ABI declared in include/fwgpu.h).

The library is built in-tree by ``__graft_entry__.build()`` / ``make -C fwumious_wabbit_amd/csrc``.
There is no fallback of any kind: a missing library is an ImportError-grade failure, and the library
itself refuses to create a regressor without a HIP device.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# FWGPU_LIBRARY: an alternative build of the same C ABI (kernel tuning experiments); default = the in-tree build
LIB_PATH = os.environ.get("FWGPU_LIBRARY") or os.path.join(_HERE, "lib", "libfwgpu.so")

OK = 0
ERR_PARSE, ERR_IO, PARSE_FLUSH, PARSE_HOGWILD_LOAD = 6, 7, 100, 101
OPT_SGD, OPT_ADAGRAD_FLEX, OPT_ADAGRAD_LUT = 100, 200, 300
WIRING_REGRESSOR, WIRING_FFM_ONLY = 0, 1
MODE_SEQUENTIAL, MODE_HOGWILD = 0, 1
TABLE_LR, TABLE_FFM_W, TABLE_FFM_ACC, TABLE_NN_W, TABLE_NN_ACC = 0, 1, 2, 3, 4
NN_INIT = {"xavier": 0, "hu": 1, "one": 2, "zero": 3}

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
        ("device", C.c_int32),
    ]


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


class TranslatorConfig(C.Structure):
    _fields_ = [
        ("n_combos", C.c_uint32),
        ("combo_off", C.c_void_p),
        ("combo_ns", C.c_void_p),
        ("combo_ns_f32", C.c_void_p),
        ("combo_weight", C.c_void_p),
        ("add_constant_feature", C.c_int32),
        ("n_fields", C.c_uint32),
        ("field_off", C.c_void_p),
        ("field_ns", C.c_void_p),
        ("field_ns_f32", C.c_void_p),
        ("bit_precision", C.c_uint32),
        ("ffm_k", C.c_uint32),
        ("ffm_bit_precision", C.c_uint32),
    ]


class SynthConfig(C.Structure):
    _fields_ = [
        ("n_namespaces", C.c_uint32),
        ("mean_extra", C.c_float),
        ("zipf_s", C.c_double),
        ("ids_per_ns", C.c_uint32),
        ("p_weighted", C.c_float),
        ("seed", C.c_uint64),
    ]


class FwgpuError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"fwgpu error {code}: {msg}")
        self.code = code
        self.message = msg


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(there is no CPU fallback)")
    L = C.CDLL(LIB_PATH)
    vp, u32, u64, i32, f32 = C.c_void_p, C.c_uint32, C.c_uint64, C.c_int, C.c_float
    P = C.POINTER
    L.fwgpu_last_error.restype = C.c_char_p
    L.fwgpu_abi_version.restype = i32
    sig = {
        "fwgpu_create": [P(Config), P(vp)],
        "fwgpu_free": [vp],
        "fwgpu_init_weights": [vp],
        "fwgpu_set_nn": [vp, P(NNConfig)],
        "fwgpu_learn": [vp, vp, u32, vp, u32, f32, f32, i32, P(f32)],
        "fwgpu_predict": [vp, vp, u32, vp, u32, P(f32)],
        "fwgpu_setup_cache": [vp, vp, u32, vp, u32, P(vp)],
        "fwgpu_predict_with_cache": [vp, vp, vp, u32, vp, u32, P(f32)],
        "fwgpu_block_cache_filter": [vp, vp, u32, vp, P(u32)],
        "fwgpu_block_cache_free": [vp],
        "fwgpu_batch_set_cache": [vp, vp],
        "fwgpu_dist_unique_id": [vp, u64],
        "fwgpu_dist_init": [vp, vp, i32, i32, P(vp)],
        "fwgpu_dist_free": [vp],
        "fwgpu_dist_set_mode": [vp, i32],
        "fwgpu_dist_group_set_mode": [vp, i32],
        "fwgpu_dist_rank": [vp, P(i32), P(i32)],
        "fwgpu_dist_comm_count": [vp, P(i32)],
        "fwgpu_dist_ranges": [vp, P(u32), P(u32), P(u32), P(u32)],
        "fwgpu_dist_learn_sharded": [vp, P(TranslatorConfig), vp, vp, u32, vp],
        "fwgpu_dist_learn_sharded_batch": [vp, P(TranslatorConfig), vp],
        "fwgpu_dist_gather_tables": [vp],
        "fwgpu_dist_all_reduce_sum": [vp, vp, u64, vp],
        "fwgpu_dist_group_create": [vp, i32, P(vp)],
        "fwgpu_dist_group_free": [vp],
        "fwgpu_dist_group_learn_sharded": [vp, P(TranslatorConfig), vp, vp, u32, vp],
        "fwgpu_dist_group_gather_tables": [vp],
        "fwgpu_dist_learn_sparse": [vp, P(TranslatorConfig), vp, vp, u32, vp],
        "fwgpu_dist_learn_sparse_batch": [vp, P(TranslatorConfig), vp],
        "fwgpu_dist_sparse_last_rows": [vp, P(u32), P(u32)],
        "fwgpu_dist_group_learn_sparse": [vp, P(TranslatorConfig), vp, vp, vp, vp],
        "fwgpu_dist_group_learn_peer": [vp, P(TranslatorConfig), vp, vp, vp, vp, i32],
        "fwgpu_dist_peer_attach": [vp],
        "fwgpu_dist_owner_attach": [vp, u32, u32],
        "fwgpu_dist_learn_owner": [vp, P(TranslatorConfig), vp, vp, u32, vp, i32],
        "fwgpu_dist_group_learn_owner": [vp, P(TranslatorConfig), vp, vp, vp, vp, i32],
        "fwgpu_dist_group_learn_owner_stream": [vp, P(TranslatorConfig), vp, vp, vp, vp, vp, i32, u32, u32, u32],
        "fwgpu_dist_owner_stream_attach": [vp, u32, u32],
        "fwgpu_dist_learn_owner_stream": [vp, P(TranslatorConfig), vp, vp, u32, vp, i32, u32],
        "fwgpu_dist_learn_peer": [vp, P(TranslatorConfig), vp, vp, u32, vp, i32],
        "fwgpu_dist_learn_peer_batch": [vp, P(TranslatorConfig), vp, i32, vp],
        "fwgpu_dist_barrier": [vp],
        "fwgpu_split_create": [vp, u32, u32, P(vp)],
        "fwgpu_split_free": [vp],
        "fwgpu_learn_batch_sync": [vp, vp, vp, i32, vp],
        "fwgpu_serialized_len": [vp, P(u64)],
        "fwgpu_write_weights": [vp, vp, u64, P(u64)],
        "fwgpu_read_weights": [vp, vp, u64],
        "fwgpu_table_len": [vp, i32, P(u64)],
        "fwgpu_table_read": [vp, i32, u64, u64, vp],
        "fwgpu_table_write": [vp, i32, u64, u64, vp],
        "fwgpu_table_fill": [vp, i32, f32],
        "fwgpu_table_checksum": [vp, i32, P(u64)],
        "fwgpu_table_device_ptr": [vp, i32, P(vp)],
        "fwgpu_batch_create": [vp, vp, vp, vp, vp, vp, vp, u32, P(vp)],
        "fwgpu_batch_free": [vp],
        "fwgpu_batch_size": [vp, P(u32), P(u64), P(u64)],
        "fwgpu_learn_batch": [vp, vp, i32, i32, vp],
        "fwgpu_batch_predictions": [vp, vp, u32, vp],
        "fwgpu_batch_predictions_device": [vp, P(vp)],
        "fwgpu_translate": [P(TranslatorConfig), vp, u32, vp, u32, P(u32), vp, u32, P(u32), P(f32), P(f32)],
        "fwgpu_batch_from_records": [vp, P(TranslatorConfig), vp, vp, u32, P(vp)],
        "fwgpu_record_batch_create": [vp, P(TranslatorConfig), vp, vp, u32, P(vp)],
        "fwgpu_trainer_create": [vp, P(TranslatorConfig), u32, P(vp)],
        "fwgpu_digest_records": [vp, vp, vp, u32],
        "fwgpu_finish": [vp],
        "fwgpu_trainer_free": [vp],
        "fwgpu_trainer_examples_seen": [vp, P(u64)],
        "fwgpu_trainer_set_holdout": [vp, u64, i32],
        "fwgpu_trainer_predictions": [vp, vp, u64, P(u64)],
        "fwgpu_set_launch": [vp, u32, u32],
        "fwgpu_set_max_in_flight": [vp, u32],
        "fwgpu_delta_start": [vp, vp, vp, vp, u64, f32, vp],
        "fwgpu_delta_finish": [vp, vp, vp, vp, u64, vp],
        "fwgpu_debug_phase_ticks": [vp, i32, vp],
        "fwgpu_debug_placement": [vp, P(i32), P(C.c_float), P(C.c_float)],
        "fwgpu_debug_set_kernel_version": [vp, i32],
        "fwgpu_debug_set_option": [vp, i32, i32],
        "fwgpu_debug_coherence_probe": [i32, i32, u32, P(u32), P(u32)],
        "fwgpu_debug_head_gemm": [vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp, vp, i32, i32, vp],
        "fwgpu_synth_records": [P(SynthConfig), u64, u32, vp, u64, vp, P(u64)],
        "fwgpu_debug_format_f32": [f32, C.c_char_p, u32],
        "fwgpu_vwmap_from_csv": [C.c_char_p, u64, P(vp)],
        "fwgpu_vwmap_from_json": [C.c_char_p, u64, P(vp)],
        "fwgpu_vwmap_to_json": [vp, vp, u64, P(u64)],
        "fwgpu_vwmap_lookup": [vp, C.c_char_p, u64, i32, P(u32), P(u32)],
        "fwgpu_parser_create": [vp, P(vp)],
        "fwgpu_parser_parse_line": [vp, C.c_char_p, u64, vp, u32, P(u32)],
        "fwgpu_parser_parse_with_prefix": [vp, C.c_char_p, u64, C.c_char_p, u64, vp, u32, P(u32)],
        "fwgpu_parse_prefix_create": [vp, C.c_char_p, u64, P(vp)],
        "fwgpu_parse_prefix_resumable": [vp],
        "fwgpu_parser_parse_after_prefix": [vp, vp, C.c_char_p, u64, vp, u32, P(u32)],
        "fwgpu_parse_prefix_is_record": [vp, vp, u32],
        "fwgpu_parser_parse_candidate": [vp, vp, C.c_char_p, u64, vp, u32, P(u32), P(i32)],
        "fwgpu_parser_parse_buffer": [vp, C.c_char_p, u64, vp, u64, vp, u64, P(u64), P(u64), P(u64)],
        "fwgpu_mi_from_json": [C.c_char_p, u64, P(vp)],
        "fwgpu_mi_to_json": [vp, vp, u64, P(u64)],
        "fwgpu_mi_configs": [vp, i32, P(Config), P(TranslatorConfig), P(NNConfig)],
        "fwgpu_mi_set_inference": [vp, i32],
        "fwgpu_model_save": [C.c_char_p, vp, vp, vp, i32],
        "fwgpu_model_read_header": [C.c_char_p, P(vp), P(vp)],
        "fwgpu_model_load": [C.c_char_p, i32, i32, P(vp), P(vp), P(vp)],
        "fwgpu_model_convert_inference": [C.c_char_p, C.c_char_p, i32],
        "fwgpu_quantize_ffm_weights": [vp, u64, vp, u64],
        "fwgpu_dequantize_ffm_weights": [vp, u64, vp],
        "fwgpu_trainer_digest_cache": [vp, vp, u64, P(u64)],
        "fwgpu_trainer_digest_text": [vp, vp, vp, C.c_char_p, u64, u32, P(u64), P(u64)],
        "fwgpu_parser_clone": [vp, P(vp)],
        "fwgpu_input_open": [C.c_char_p, P(vp)],
        "fwgpu_input_read": [vp, vp, u64, P(u64)],
        "fwgpu_trainer_digest_file": [vp, vp, vp, C.c_char_p, u32, P(u64)],
        "fwgpu_cache_open": [C.c_char_p, vp, P(vp)],
        "fwgpu_cache_push_records": [vp, vp, u64],
        "fwgpu_cache_write_finish": [vp],
        "fwgpu_cache_next_records": [vp, vp, u64, vp, u64, P(u64), P(u64)],
    }
    for name, args in sig.items():
        fn = getattr(L, name)
        fn.argtypes = args
        fn.restype = i32
    for name in ("fwgpu_vwmap_free", "fwgpu_parser_free", "fwgpu_parse_prefix_free", "fwgpu_cache_free", "fwgpu_mi_free", "fwgpu_input_close"):
        getattr(L, name).argtypes = [vp]
        getattr(L, name).restype = None
    for name in ("fwgpu_vwmap_num_namespaces", "fwgpu_vwmap_num_entries"):
        getattr(L, name).argtypes = [vp]
        getattr(L, name).restype = u32
    for name in ("fwgpu_cache_is_reading", "fwgpu_cache_is_writing"):
        getattr(L, name).argtypes = [vp]
        getattr(L, name).restype = i32
    L.fwgpu_parser_command_argument.argtypes = [vp]
    L.fwgpu_parser_command_argument.restype = C.c_char_p
    L.fwgpu_murmur3_32.argtypes = [C.c_char_p, C.c_size_t, u32]
    L.fwgpu_murmur3_32.restype = u32
    L.fwgpu_lr_hash_mask.argtypes = [u32]
    L.fwgpu_lr_hash_mask.restype = u32
    L.fwgpu_ffm_hash_mask.argtypes = [u32, u32]
    L.fwgpu_ffm_hash_mask.restype = u32
    _lib = L
    return L


def check(rc):
    if rc != OK:
        raise FwgpuError(rc, lib().fwgpu_last_error().decode(errors="replace"))


def ptr(a):
    if a is None:
        return None
    if isinstance(a, np.ndarray):
        return a.ctypes.data_as(C.c_void_p) if a.size else None
    return a

"""ctypes binding of the reference's serving FFI as libfwgpu.so exports it (include/fw_ffi.h, lib.rs:150-236) --
the same calls a JNI / ctypes client of the reference's libfw makes."""
import ctypes as C

import numpy as np

from . import _capi as capi


def _lib():
    L = capi.lib()
    if not getattr(L, "_fw_ffi_ready", False):
        L.new_fw_predictor_prototype.argtypes = [C.c_char_p]
        L.new_fw_predictor_prototype.restype = C.c_void_p
        L.clone_lite.argtypes = [C.c_void_p]
        L.clone_lite.restype = C.c_void_p
        for name in ("fw_predict", "fw_predict_with_cache", "fw_setup_cache"):
            getattr(L, name).argtypes = [C.c_void_p, C.c_char_p]
            getattr(L, name).restype = C.c_float
        L.free_predictor.argtypes = [C.c_void_p]
        L.free_predictor.restype = None
        L.fwgpu_predictor_predict_batch.argtypes = [C.c_void_p, C.POINTER(C.c_char_p), C.c_uint32, C.c_int, C.c_void_p]
        L.fwgpu_predictor_predict_batch.restype = C.c_int
        L._fw_ffi_ready = True
    return L


class Predictor:
    def __init__(self, command: str = None, _ptr=None):
        self.L = _lib()
        self.p = _ptr if _ptr is not None else self.L.new_fw_predictor_prototype(command.encode())
        if not self.p:
            raise capi.FwgpuError(-1, self.L.fwgpu_last_error().decode(errors="replace"))

    def clone_lite(self):
        return Predictor(_ptr=self.L.clone_lite(self.p))

    def predict(self, text: str) -> float:
        return float(self.L.fw_predict(self.p, text.encode()))

    def setup_cache(self, text: str) -> float:
        return float(self.L.fw_setup_cache(self.p, text.encode()))

    def predict_with_cache(self, text: str) -> float:
        return float(self.L.fw_predict_with_cache(self.p, text.encode()))

    @staticmethod
    def encode_batch(texts):
        """the char** a C / Rust caller already holds: build it once, outside any timed region"""
        return (C.c_char_p * len(texts))(*[t.encode() for t in texts])

    def predict_batch(self, texts, with_cache=False) -> np.ndarray:
        """texts: a list of str, or the array Predictor.encode_batch made of one"""
        arr = texts if isinstance(texts, C.Array) else self.encode_batch(texts)
        out = np.zeros(len(arr), dtype=np.float32)
        capi.check(self.L.fwgpu_predictor_predict_batch(self.p, arr, len(arr), int(with_cache), capi.ptr(out)))
        return out

    def close(self):
        if self.p:
            self.L.free_predictor(self.p)
            self.p = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

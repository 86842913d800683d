"""Host-side mirror of persistence.rs over the C ABI: ModelInstance <-> JSON, model files, inference conversion."""
import ctypes as C
import json

import numpy as np

from . import _capi as capi
from .feed import VwNamespaceMap
from .regressor import FeatureComboDesc, ModelInstance, NamespaceDescriptor, Optimizer, Regressor

_OPT_NAME = {Optimizer.SGD: "SGD", Optimizer.AdagradFlex: "AdagradFlex", Optimizer.AdagradLUT: "AdagradLUT"}
_OPT_ID = {v: k for k, v in _OPT_NAME.items()}


def _nd(d: NamespaceDescriptor):
    return {"namespace_index": d.namespace_index, "namespace_type": "Primitive",
            "namespace_format": "F32" if d.namespace_format_f32 else "Categorical"}


class ModelInstanceHandle:
    """fwgpu_model_instance: the ModelInstance as the library holds it (made from / rendered to the reference's JSON)"""

    def __init__(self, h):
        self.L = capi.lib()
        self.h = h

    @classmethod
    def from_json(cls, text: bytes):
        L = capi.lib()
        h = C.c_void_p()
        capi.check(L.fwgpu_mi_from_json(text, len(text), C.byref(h)))
        return cls(h)

    @classmethod
    def from_model_instance(cls, mi: ModelInstance, extra=None):
        """the Python mirror's fields -> the reference's JSON document (fields this path does not read get new_empty()'s
        defaults, model_instance.rs:120-150)"""
        d = {
            "learning_rate": mi.learning_rate, "minimum_learning_rate": 0.0, "power_t": mi.power_t,
            "bit_precision": mi.bit_precision, "add_constant_feature": mi.add_constant_feature,
            "feature_combo_descs": [{"namespace_descriptors": [_nd(x) for x in c.namespace_descriptors], "weight": c.weight}
                                    for c in mi.feature_combo_descs],
            "ffm_fields": [[_nd(x) for x in f] for f in mi.ffm_fields],
            "ffm_k": mi.ffm_k, "ffm_bit_precision": mi.ffm_bit_precision, "fastmath": True,
            "ffm_initialization_type": "default", "ffm_k_threshold": 0.0, "ffm_init_center": mi.ffm_init_center,
            "ffm_init_width": mi.ffm_init_width, "ffm_init_zero_band": mi.ffm_init_zero_band,
            "ffm_init_acc_gradient": mi.ffm_init_acc_gradient, "init_acc_gradient": mi.init_acc_gradient,
            "ffm_learning_rate": mi.ffm_learning_rate, "ffm_power_t": mi.ffm_power_t,
            "nn_init_acc_gradient": mi.nn_init_acc_gradient, "nn_learning_rate": mi.nn_learning_rate,
            "nn_power_t": mi.nn_power_t,
            "nn_config": {"layers": [{k: str(v) for k, v in layer.items()} for layer in mi.nn_layers],
                          "topology": mi.nn_topology},
            "optimizer": _OPT_NAME[mi.optimizer], "transform_namespaces": {"v": []}, "dequantize_weights": False,
        }
        d.update(extra or {})
        return cls.from_json(json.dumps(d).encode())

    def to_json(self) -> bytes:  # serde_json::to_vec_pretty(&ModelInstance)
        n = C.c_uint64()
        capi.check(self.L.fwgpu_mi_to_json(self.h, None, 0, C.byref(n)))
        buf = C.create_string_buffer(n.value)
        capi.check(self.L.fwgpu_mi_to_json(self.h, buf, n.value, C.byref(n)))
        return buf.raw[: n.value]

    def to_model_instance(self, device=0) -> ModelInstance:
        d = json.loads(self.to_json())
        # going through fwgpu_mi_configs makes the library reject what this path does not implement
        cfg, tr, nn = capi.Config(), capi.TranslatorConfig(), capi.NNConfig()
        capi.check(self.L.fwgpu_mi_configs(self.h, device, C.byref(cfg), C.byref(tr), C.byref(nn)))

        def nd(x):
            return NamespaceDescriptor(x["namespace_index"], x["namespace_format"] == "F32")

        return ModelInstance(
            learning_rate=d["learning_rate"], ffm_learning_rate=d["ffm_learning_rate"], bit_precision=d["bit_precision"],
            power_t=d["power_t"], ffm_power_t=d["ffm_power_t"], add_constant_feature=d["add_constant_feature"],
            feature_combo_descs=[FeatureComboDesc([nd(x) for x in c["namespace_descriptors"]], c["weight"])
                                 for c in d["feature_combo_descs"]],
            ffm_fields=[[nd(x) for x in f] for f in d["ffm_fields"]], ffm_k=d["ffm_k"],
            ffm_bit_precision=d["ffm_bit_precision"], ffm_init_center=d["ffm_init_center"],
            ffm_init_width=d["ffm_init_width"], ffm_init_zero_band=d["ffm_init_zero_band"],
            ffm_init_acc_gradient=d["ffm_init_acc_gradient"], init_acc_gradient=d["init_acc_gradient"],
            optimizer=_OPT_ID[d["optimizer"]],
            nn_layers=[dict(layer) for layer in d["nn_config"]["layers"]], nn_topology=d["nn_config"]["topology"],
            nn_learning_rate=d["nn_learning_rate"], nn_power_t=d["nn_power_t"],
            nn_init_acc_gradient=d["nn_init_acc_gradient"], device=device)

    def close(self):
        if self.h:
            self.L.fwgpu_mi_free(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def save_regressor_to_filename(filename: str, mi: ModelInstance, vwmap: VwNamespaceMap, re: Regressor,
                               quantize_weights: bool = False, extra=None):
    """persistence.rs:73-89"""
    if quantize_weights:  # the header has to announce the f16-bucket FFM blob (main.rs:141-145 sets it before saving)
        extra = dict(extra or {}, dequantize_weights=True)
    h = ModelInstanceHandle.from_model_instance(mi, extra)
    capi.check(capi.lib().fwgpu_model_save(filename.encode(), vwmap.h, h.h, re.h, int(quantize_weights)))
    h.close()


def load_regressor_without_weights(filename: str):
    """persistence.rs:91-125 (header only) -> (ModelInstanceHandle, VwNamespaceMap)"""
    L = capi.lib()
    vw, mi = C.c_void_p(), C.c_void_p()
    capi.check(L.fwgpu_model_read_header(filename.encode(), C.byref(vw), C.byref(mi)))
    return ModelInstanceHandle(mi), VwNamespaceMap(_handle=vw)


def new_regressor_from_filename(filename: str, immutable: bool, device: int = 0):
    """persistence.rs:127-174 -> (ModelInstance, VwNamespaceMap, Regressor)"""
    L = capi.lib()
    vw, mih, r = C.c_void_p(), C.c_void_p(), C.c_void_p()
    capi.check(L.fwgpu_model_load(filename.encode(), device, int(immutable), C.byref(vw), C.byref(mih), C.byref(r)))
    h = ModelInstanceHandle(mih)
    mi = h.to_model_instance(device)
    h.close()
    return mi, VwNamespaceMap(_handle=vw), Regressor.from_handle(mi, r)


def hogwild_load(re: Regressor, filename: str):
    """persistence.rs:176-187: overwrite the weights of a live regressor from a model file"""
    r = C.c_void_p(re.h.value)
    capi.check(capi.lib().fwgpu_model_load(filename.encode(), re.mi.device, 0, None, None, C.byref(r)))


def convert_inference_regressor(in_filename: str, out_filename: str, quantize_weights: bool = False):
    """main.rs:136-148, host only"""
    capi.check(capi.lib().fwgpu_model_convert_inference(in_filename.encode(), out_filename.encode(), int(quantize_weights)))


def quantize_ffm_weights(weights: np.ndarray) -> bytes:  # quantization.rs:42-80
    w = np.ascontiguousarray(weights, dtype=np.float32)
    out = np.zeros(8 + 2 * w.size, dtype=np.uint8)
    capi.check(capi.lib().fwgpu_quantize_ffm_weights(capi.ptr(w), w.size, capi.ptr(out), out.size))
    return out.tobytes()


def dequantize_ffm_weights(blob: bytes, n: int) -> np.ndarray:  # quantization.rs:82-98
    src = np.frombuffer(blob, dtype=np.uint8).copy()
    out = np.zeros(n, dtype=np.float32)
    capi.check(capi.lib().fwgpu_dequantize_ffm_weights(capi.ptr(src), n, capi.ptr(out)))
    return out

"""MI355X-native LR + field-aware FM online learner behind fwumious_wabbit's Regressor/HogwildTrainer boundary.

The product is libfwgpu.so (HIP kernels + C ABI, include/fwgpu.h); this package is its thin host-side mirror
of the reference interface.  Importing the package does not load the library; the first use does, and it
fails loudly if the library has not been built or no GPU is present."""
from . import _capi as capi  # noqa: F401
from .regressor import (Batch, BlockCache, SplitBuffers, FeatureBuffer, FeatureBufferTranslator, FeatureComboDesc, HogwildTrainer,  # noqa: F401
                        ModelInstance, NamespaceDescriptor, Optimizer, Regressor, ffm_vec, lr_and_ffm_vec, lr_vec,
                        synth_records)

"""pytorch_object_detection_amd — MI355X-native FCOS / HISFCOS detection hot path.

Host side mirrors the reference's Python model API (bulider.Builder, config YAML, model.od.*, model.modules.head,
model.loss, utill.utills.load_config); the hot path itself is libfcosdet_hip.so (hand-written HIP for gfx950)
behind the C-ABI declared in include/fcosdet.h.  There is no CPU fallback: every op raises if the HIP library
is missing or a tensor is not on the GPU.
"""
__version__ = "0.1.0"

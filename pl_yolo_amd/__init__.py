"""pl_yolo_amd -- MI355X-native (gfx950) YOLOX detection path behind the pl_YOLO
plugin API.  Host logic in Python on PyTorch-ROCm tensors, compute in
libplyolo_hip.so (hand-written HIP) through the C ABI of include/plyolo.h."""
from .build_detection import build_model, OneStageD  # noqa: F401
from ._lib import PlyoloError  # noqa: F401

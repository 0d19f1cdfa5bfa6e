"""pl_yolo_amd -- MI355X-native (gfx950) YOLOX detection path behind the pl_YOLO
plugin API.  Host logic in Python on PyTorch-ROCm tensors, compute in
libplyolo_hip.so (hand-written HIP) through the C ABI of include/plyolo.h."""
import os as _os

# The launch plans use four HIP streams (main lane, weight-gradient lane, two head-level lanes).  ROCclr maps a process's
# streams onto GPU_MAX_HW_QUEUES hardware queues (default 4).  Measured on MI355X / ROCm 7.2 (tools/ab_hwq.sh, YOLOX-s step):
# 1 queue 11.8 ms, 2 queues 10.0, 3 queues 9.64, 4 queues 9.99, 5 queues 23.3 (!), 6 queues 20.0 -- three is the optimum,
# five or more halve the throughput (the same cliff a prioritised or a sixth stream runs into).  The variable is read when the
# HIP runtime initialises, so it must be in the environment BEFORE the first HIP call of the process: importing this package
# first (or exporting it in the shell) is enough; an explicit setting of the user is respected.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "3")
from .build_detection import build_model, OneStageD  # noqa: F401
from ._lib import PlyoloError  # noqa: F401

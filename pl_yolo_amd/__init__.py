"""pl_yolo_amd -- MI355X-native (gfx950) YOLOX detection path behind the pl_YOLO
plugin API.  Host logic in Python on PyTorch-ROCm tensors, compute in
libplyolo_hip.so (hand-written HIP) through the C ABI of include/plyolo.h."""
import os as _os

# ROCclr maps a process's HIP streams onto GPU_MAX_HW_QUEUES hardware queues (default 4), and on MI355X / ROCm 7.2 the step time
# depends on that number like this (tools/ab_hwq.sh, YOLOX-s step): 1 queue 11.8 ms, 2 queues 10.0, 3 queues 9.64, 4 queues 9.99,
# 5 queues 23.3 (!), 6 queues 20.0.  The optimum is ONE QUEUE PER BUSY STREAM, never more than four:
#   * single process: the launch plans use three streams (the default stream as the main lane, the weight-gradient lane, one
#     lane for the two smaller head levels) -> 3 queues;
#   * under torch.distributed (RANK / WORLD_SIZE in the environment) the process group owns a stream of its own, created
#     before the lanes: with 3 queues the weight-gradient lane then shares a queue with the main lane and the step takes
#     11.8 ms instead of 9.5 (tools/ab_ddp.sh) -> 4 queues, and a lane per head level again (pl_yolo_amd/heads.py).
# The variable is read when the HIP runtime initialises, so it must be in the environment BEFORE the first HIP call of the
# process: importing this package first (or exporting it in the shell) is enough; an explicit setting of the user is respected.
DISTRIBUTED_LAUNCH = "RANK" in _os.environ or int(_os.environ.get("WORLD_SIZE", "1") or 1) > 1
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "4" if DISTRIBUTED_LAUNCH else "3")
from .build_detection import build_model, OneStageD  # noqa: F401
from ._lib import PlyoloError  # noqa: F401

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
#     before the lanes, and which stream shares a queue with which is decided by creation order (round-robin): with 3 queues
#     the weight-gradient lane then lands on the main lane's queue and the step takes 11.9 ms instead of 9.6 (tools/ab_ddp.sh).
#     There every plan runs its main lane on a stream of its OWN, created back to back with its side streams
#     (PLYOLO_OWN_MAIN=1, csrc/api.hip): consecutive streams sit on distinct queues whatever was created before them.  With 4
#     queues that is as fast as the single-process setup (9.60 vs 9.62 ms, one-rank group, tools/ab_own.sh) and does not depend
#     on what else the application created; in a single process the caller's stream as main lane is 1 % faster still.
# The variables are read when the HIP runtime / the library initialise, so they must be in the environment BEFORE the first HIP
# call of the process: importing this package first (or exporting them in the shell) is enough; explicit settings are respected.
DISTRIBUTED_LAUNCH = "RANK" in _os.environ or int(_os.environ.get("WORLD_SIZE", "1") or 1) > 1
_os.environ.setdefault("PLYOLO_OWN_MAIN", "1" if DISTRIBUTED_LAUNCH else "0")
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "4" if DISTRIBUTED_LAUNCH else "3")
from .build_detection import build_model, OneStageD  # noqa: F401
from ._lib import PlyoloError  # noqa: F401

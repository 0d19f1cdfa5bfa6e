"""pl_yolo_amd -- MI355X-native (gfx950) YOLOX detection path behind the pl_YOLO
plugin API.  Host logic in Python on PyTorch-ROCm tensors, compute in
libplyolo_hip.so (hand-written HIP) through the C ABI of include/plyolo.h."""
# Streams and hardware queues (MI355X / ROCm 7.2; tools/ab/ab_r3h.sh, profiles/r03_stream_layout.txt).  A plan replays on three
# streams of its OWN -- main lane, weight-gradient lane, side lane (csrc/api.hip: issue_lanes; PLYOLO_OWN_MAIN=0 puts lane 0 on the
# caller's stream instead) -- created back to back, so they sit on distinct hardware queues whatever the application created before;
# the caller's stream only forks into / joins from the plan.  ROCclr maps a process's streams onto GPU_MAX_HW_QUEUES queues, default 4,
# which is what this layout wants: YOLOX-s step 9.26 ms at 4 queues, 9.59 at 3, 11.3 at 2, 16.9 at 5 (five or more queues halve the
# throughput); issued from a user stream with 1-4 other user streams around 9.31-9.59 ms, in a one-rank RCCL process group 9.37 ms.
# The package therefore sets NOTHING in the environment (rounds 1-2 exported GPU_MAX_HW_QUEUES / PLYOLO_OWN_MAIN at import): the
# runtime's default is the measured optimum, and an application that changes GPU_MAX_HW_QUEUES gets the table above.
from .build_detection import build_model, OneStageD  # noqa: F401
from ._lib import PlyoloError  # noqa: F401

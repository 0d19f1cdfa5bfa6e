"""pl_yolo_amd -- MI355X-native (gfx950) YOLOX detection path behind the pl_YOLO
plugin API.  Host logic in Python on PyTorch-ROCm tensors, compute in
libplyolo_hip.so (hand-written HIP) through the C ABI of include/plyolo.h."""
import os as _os

# ROCclr maps a process's HIP streams onto GPU_MAX_HW_QUEUES hardware queues (default 4), round-robin in creation order, and on
# MI355X / ROCm 7.2 the step time depends on that mapping more than on any kernel (YOLOX-s step, tools/ab_hwq.sh, ab_dummy.sh,
# ab_own.sh, user_stream_check.py):
#   * number of queues, caller's stream as the main lane: 1 queue 11.8 ms, 2: 10.0, 3: 9.64, 4: 9.99, 5: 23.3 (!), 6: 20.0 --
#     five or more queues halve the throughput (the cliff a prioritised stream or a sixth lane runs into);
#   * WHICH lanes share a queue: with the caller's stream as lane 0 the answer depends on every stream the application created
#     before the plans' side streams -- the tuned single-process setup (3 queues) runs 9.53 ms, and 11.7 ms as soon as the step is
#     issued from a user-created stream; under torch.distributed (the process group owns a stream) 11.9 ms;
#   * PLYOLO_OWN_MAIN=1 (csrc/api.hip: issue_lanes): every plan runs its main lane on a stream of its OWN, created back to back
#     with its side streams -- consecutive streams sit on distinct queues whatever was created before them; the caller's stream
#     only forks into / joins from the plan.  With 4 queues: 9.58 ms on the default stream, 9.58-9.76 ms with one to three user
#     streams around, 9.57 ms in a one-rank process group -- 0.5 % behind the tuned setup in its best case, never near its worst.
# Hence the defaults: own main stream, four queues (three lanes per plan: main, weight gradients, the two smaller head levels).
# GPU_MAX_HW_QUEUES is read when the HIP runtime initialises, PLYOLO_OWN_MAIN at the first plan replay: both must be in the
# environment BEFORE the first HIP call of the process -- importing this package first (or exporting them in the shell) is enough;
# explicit settings of the user are respected (the tuned single-process setup: GPU_MAX_HW_QUEUES=3 PLYOLO_OWN_MAIN=0).
_os.environ.setdefault("PLYOLO_OWN_MAIN", "1")
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "4")
from .build_detection import build_model, OneStageD  # noqa: F401
from ._lib import PlyoloError  # noqa: F401

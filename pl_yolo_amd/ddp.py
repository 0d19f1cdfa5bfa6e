"""Data-parallel training over the GPUs of one node: one process per GPU, identical
weights, rank-local BatchNorm statistics and rank-local num_fg normalisation (the
reference has no SyncBN and no cross-rank num_fg reduction, yolox_loss.py:148-154), and
exactly one exchange step per iteration: the mean all-reduce of the flat fp32 gradient
buffer with RCCL over xGMI (torch.distributed backend "nccl" is RCCL on ROCm; "gloo" is
used by the CPU tests)."""
import torch
import torch.distributed as dist


FORCE_COLLECTIVE = False   # self-test: issue the collective even in a one-rank group (bench.py PLYOLO_BENCH_FORCE_DDP)


class GradAllReduce:
    """Averages a flat gradient buffer across ranks.  The whole model is ONE bucket
    (YOLOX-s: 36 MB): xGMI is point-to-point, so few large collectives beat many small
    ones; the call is issued on the stream the backward plan ran on."""

    def __init__(self, group=None):
        self.group = group
        self.world = dist.get_world_size(group)

    def all_reduce_(self, flat):
        if self.world == 1 and not FORCE_COLLECTIVE:
            return flat
        if dist.get_backend(self.group) == "nccl":   # RCCL averages inside the collective: no second pass over the buffer
            dist.all_reduce(flat, op=dist.ReduceOp.AVG, group=self.group)
        else:                                        # gloo (CPU tests) has no AVG
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
            flat.mul_(1.0 / self.world)
        return flat


def attach(model, group=None):
    """Make `model` (a pl_yolo_amd OneStageD) average its gradients across ranks at the
    end of every backward, and start from rank 0's weights."""
    if not dist.is_initialized():
        raise RuntimeError("torch.distributed is not initialised")
    r = model.runner()
    r.ddp = GradAllReduce(group)
    for t in list(model.parameters()) + list(model.buffers()):
        dist.broadcast(t.data, src=0, group=group)
    return model

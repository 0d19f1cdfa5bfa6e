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
    dev = next(model.parameters()).device
    if not r._adopted_ok(dev):
        r.adopt(dev)            # parameters / buffers become views of three flat buffers ...
    for key in ("w", "fbuf", "ibuf"):
        dist.broadcast(r.flat[key], src=0, group=group)   # ... so the start state is three collectives, not one per tensor
    model.__dict__['_ddp'] = GradAllReduce(group)         # kept on the model: a runner rebuilt later (compute_dtype change) re-attaches it
    r.ddp = model.__dict__['_ddp']
    return model

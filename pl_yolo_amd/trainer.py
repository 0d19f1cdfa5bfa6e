"""The training-step arithmetic of the reference's Lightning module, without Lightning
(reference PL_Modules/pl_detection.py:51-64,107-111; models/layers/lr_scheduler.py:5-19;
models/utils/ema.py:22-60):

    losses = model(imgs, labels); optimizer.zero_grad(); backward; optimizer.step()
    ema.update(model); lr_scheduler.step()

with SGD(lr, momentum), no weight decay, no nesterov; per-step cosine x linear-warmup LR;
EMA decay 0.9998 * (1 - exp(-n / 2000)) over every float state tensor.

Because the runner keeps parameters, gradients and float buffers in flat device
buffers, the optimizer and the EMA are ONE launch each (plyolo_sgd_momentum /
plyolo_ema_update) instead of ~260 + ~870 tiny kernels.
"""
import copy
import math

import numpy as np
import torch

from ._lib import call, PlyoloError


def lr_factor(step, warmup, max_iters):
    """CosineWarmupScheduler.get_lr_factor (lr_scheduler.py:15-19)."""
    f = 0.5 * (1 + np.cos(np.pi * step / max_iters))
    if step <= warmup:
        f *= (step * 1.0 + 0.00001) / warmup
    return float(f)


class Trainer:
    def __init__(self, model, learning_rate=0.01, momentum=0.9, warmup=0.1, total_steps=1000, ema=True, ema_decay=0.9998):
        self.model = model
        self.base_lr, self.momentum = float(learning_rate), float(momentum)
        self.total_steps = int(total_steps)
        self.warmup_steps = warmup * total_steps  # pl_detection.py:110
        self.step_idx = 0
        self.use_ema = bool(ema)
        self.ema_decay = float(ema_decay)
        self.ema_updates = 0
        self.ema_model = None
        self._mom = None
        self._ema_flat = None

    def _stream(self):
        return torch.cuda.current_stream().cuda_stream

    def current_lr(self):
        return self.base_lr * lr_factor(self.step_idx, self.warmup_steps, self.total_steps)

    def _ensure_state(self):
        r = self.model.runner()
        f = r.flat
        if f is None:
            raise PlyoloError("run one forward before the first optimizer step")
        if self._mom is None or self._mom.numel() != f["n"] or self._mom.device != f["w"].device:
            self._mom = torch.zeros_like(f["w"])
            self._first = True
        if self.use_ema and self.ema_model is None:
            # ModelEMA.__init__: deep copy in eval mode (ema.py:41); its tensors are re-pointed at
            # flat buffers mirroring the training model's so one launch updates all of them
            self.ema_model = copy.deepcopy(self.model).eval()
            for p in self.ema_model.parameters():
                p.requires_grad_(False)
            er = self.ema_model.runner()
            er.adopt(f["device"])
            ef = er.flat
            ef["w"].copy_(f["w"])
            ef["fbuf"].copy_(f["fbuf"])
            ef["ibuf"].copy_(f["ibuf"])
        return f

    def train_step(self, imgs, labels):
        model = self.model
        model.train()
        losses = model(imgs, labels)
        model.zero_grad(set_to_none=True)
        losses["loss"].backward()
        f = self._ensure_state()
        lr = self.current_lr()
        # parameters without a gradient (the dead Bottleneck.bn affine pairs) have zeros in the
        # flat gradient buffer, so one launch over the whole buffer equals torch.optim.SGD
        call("plyolo_sgd_momentum", f["w"].data_ptr(), f["g"].data_ptr(), self._mom.data_ptr(), f["n"], None, lr, self.momentum,
             int(self._first), self._stream())
        self._first = False
        if self.use_ema:
            self.ema_updates += 1
            d = self.ema_decay * (1 - math.exp(-self.ema_updates / 2000))
            ef = self.ema_model.runner().flat
            call("plyolo_ema_update", ef["w"].data_ptr(), f["w"].data_ptr(), f["n"], d, self._stream())
            call("plyolo_ema_update", ef["fbuf"].data_ptr(), f["fbuf"].data_ptr(), f["fbuf"].numel(), d, self._stream())
        self.step_idx += 1
        return losses

    def eval_model(self):
        """validation_step uses the EMA weights when present (pl_detection.py:68-71)."""
        return self.ema_model if self.ema_model is not None else self.model

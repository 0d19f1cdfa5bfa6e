"""Harness arithmetic the runner must reproduce (TEST ORACLE).

  * cosine x linear-warmup LR factor   models/layers/lr_scheduler.py:5-19
  * EMA over every float state tensor  models/utils/ema.py:22-60 (decay ramp
    d = 0.9998*(1-exp(-n/2000)), ctor arg at PL_Modules/pl_detection.py:48)
  * torch.optim.SGD(lr, momentum), no weight decay, no nesterov, dampening 0
                                        PL_Modules/pl_detection.py:107-111
"""
import math

import numpy as np
import torch


def lr_factor(step, warmup, max_iters):
    f = 0.5 * (1 + np.cos(np.pi * step / max_iters))
    if step <= warmup:
        f *= (step * 1.0 + 0.00001) / warmup
    return f


def ema_decay(updates, decay=0.9998):
    return decay * (1 - math.exp(-updates / 2000))


def ema_update(ema_state, model_state, updates, decay=0.9998):
    """One ModelEMA.update; returns the new `updates` counter."""
    updates += 1
    d = ema_decay(updates, decay)
    with torch.no_grad():
        for k, v in ema_state.items():
            if v.dtype.is_floating_point:
                v *= d
                v += (1.0 - d) * model_state[k].detach()
    return updates


def sgd_step(params, grads, bufs, lr, momentum):
    """torch.optim.SGD step (first step initialises the buffer with the grad)."""
    with torch.no_grad():
        for k, p in params.items():
            g = grads.get(k)
            if g is None:
                continue
            if momentum != 0:
                if k not in bufs:
                    bufs[k] = g.clone()
                else:
                    bufs[k].mul_(momentum).add_(g)
                g = bufs[k]
            p.add_(g, alpha=-lr)

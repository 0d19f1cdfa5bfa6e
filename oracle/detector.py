"""`OneStageD.forward(x, labels)` restated as one function (TEST ORACLE).

PL_Modules/build_detection.py:23-53: backbone -> neck -> head -> (loss if
labels is not None).  The train/eval switch of the loss follows the module's
`training` flag (yolox_loss.py:25)."""
import torch

from . import net, yolox_loss


def forward(state, cfg, num_classes, x, labels=None, training=True, return_assign=False, use_l1=False):
    if cfg["backbone"]["name"] not in ("cspdarknet", "ecmnet") or cfg["head"]["name"] != "decoupled_head":
        raise NameError("oracle.detector covers the cspdarknet/csppafpn and ecmnet/al_pafpn + decoupled_head/yolox paths")
    if cfg["backbone"]["name"] == "ecmnet":
        from . import net_e
        maps = net_e.eyolox_network(state, cfg, x, training)
    else:
        maps = net.yolox_network(state, cfg, x, training)
    if labels is None:
        return maps
    strides = cfg["loss"]["stride"]
    if not training:
        return yolox_loss.eval_decode(maps, strides, num_classes)
    return yolox_loss.yolox_loss(maps, labels, strides, num_classes, return_assign=return_assign, use_l1=use_l1)


def train_step_grads(state, cfg, num_classes, x, labels, use_l1=False):
    """fwd + loss + bwd; returns (loss dict, {param name: grad})."""
    names = net.param_names(state)
    for k in names:
        state[k].requires_grad_(True)
        state[k].grad = None
    out = forward(state, cfg, num_classes, x, labels, training=True, return_assign=True, use_l1=use_l1)
    out["loss"].backward()
    grads = {k: state[k].grad for k in names if state[k].grad is not None}
    return out, grads


def synthetic_batch(batch, size, num_classes, num_gt=30, max_gt=100, seed=1234):
    """SURVEY.md section 8(d) synthetic inputs: images rand*255, targets with
    G valid rows (cls, cx, cy, w, h) in pixels, zero padded to `max_gt`."""
    g = torch.Generator().manual_seed(seed)
    imgs = torch.rand(batch, 3, size, size, generator=g) * 255
    labels = torch.zeros(batch, max_gt, 5)
    labels[:, :num_gt, 0] = torch.randint(0, num_classes, (batch, num_gt), generator=g).float()
    labels[:, :num_gt, 1:3] = (0.15 + 0.7 * torch.rand(batch, num_gt, 2, generator=g)) * size
    labels[:, :num_gt, 3:5] = 8 + torch.rand(batch, num_gt, 2, generator=g) * 0.3 * size
    return imgs, labels

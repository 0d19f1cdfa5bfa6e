"""Functional fp32 CPU restatement of the YOLOv7 network graph (TEST ORACLE).

Same conventions as oracle/net.py (flat `state` dict with the reference's key layout).
Reference semantics restated (file:line into /root/reference):
  * EELAN backbone, its ELAN `CSPLayer` and `Transition`   models/backbones/eelan.py:15-141
  * SPPCSPC                                                models/layers/network_blocks.py:158-175
  * YOLOv7NECK, its ELAN-W `CSPLayer` and `Transition`     models/necks/yolov7_neck.py:7-164
  * ImplicitHead / ImplicitA / ImplicitM                   models/heads/implicit_head.py:5-62
  * eval decode of YOLOv7Loss                              models/losses/yolov7/yolov7_loss.py:43-78
"""
import torch
import torch.nn.functional as F

from .net import conv_unit, spp_bottleneck, _q, _qw


def _chain(state, prefix, x, n, training, norm, act):
    for i in range(n):
        x = conv_unit(state, "%s.%d" % (prefix, i), x, 1, training, norm, act)
    return x


def elan(state, prefix, x, num_bottle, training, norm, act):
    n = num_bottle // 2 if num_bottle > 2 else 1
    x1 = conv_unit(state, prefix + ".conv1", x, 1, training, norm, act)
    x2 = conv_unit(state, prefix + ".conv2", x, 1, training, norm, act)
    x3 = _chain(state, prefix + ".conv3", x2, n, training, norm, act)
    x4 = _chain(state, prefix + ".conv4", x3, n, training, norm, act)
    return conv_unit(state, prefix + ".conv5", torch.cat([x1, x2, x3, x4], 1), 1, training, norm, act)


def transition(state, prefix, x, training, norm, act):
    # conv1/conv2 always use bn/silu (eelan.py:130-131, yolov7_neck.py:153-154)
    x1 = conv_unit(state, prefix + ".conv1", F.max_pool2d(x, 2, 2), 1, training, "bn", "silu")
    x2 = conv_unit(state, prefix + ".conv2", x, 1, training, "bn", "silu")
    x2 = conv_unit(state, prefix + ".conv3", x2, 2, training, norm, act)
    return torch.cat([x2, x1], 1)


def eelan(state, cfg, x, training, prefix="backbone"):
    d, outs, norm, act = cfg["depths"], cfg["outputs"], cfg["norm"], cfg["act"]
    feats = {}
    x = conv_unit(state, prefix + ".stem.0", _q(x), 1, training, norm, act)
    x = conv_unit(state, prefix + ".stem.1", x, 2, training, norm, act)
    x = conv_unit(state, prefix + ".stem.2", x, 1, training, norm, act)
    feats["stem"] = x
    x = conv_unit(state, prefix + ".stage1.0", x, 2, training, norm, act)
    x = elan(state, prefix + ".stage1.1", x, d[0], training, norm, act)
    feats["stage1"] = x
    for s in (2, 3):
        x = transition(state, "%s.stage%d.0" % (prefix, s), x, training, norm, act)
        x = elan(state, "%s.stage%d.1" % (prefix, s), x, d[s - 1], training, norm, act)
        feats["stage%d" % s] = x
    x = transition(state, prefix + ".stage4.0", x, training, norm, act)
    x = spp_bottleneck(state, prefix + ".stage4.1", x, training, norm, act)
    x = elan(state, prefix + ".stage4.2", x, d[3], training, norm, act)
    feats["stage4"] = x
    if len(outs) <= 1:
        return x
    return [v for k, v in feats.items() if k in outs]


def sppcspc(state, prefix, x, training):
    u = lambda name, t, s=1: conv_unit(state, prefix + "." + name, t, s, training, "bn", "silu")
    x1 = u("cv4", u("cv3", u("cv1", x)))
    y1 = u("cv6", u("cv5", torch.cat([x1] + [F.max_pool2d(x1, k, 1, k // 2) for k in (5, 9, 13)], 1)))
    y2 = u("cv2", x)
    return u("cv7", torch.cat((y1, y2), 1))


def elan_w(state, prefix, x, num_bottle, training, norm, act):
    x1 = conv_unit(state, prefix + ".conv1", x, 1, training, norm, act)
    x2 = conv_unit(state, prefix + ".conv2", x, 1, training, norm, act)
    x3 = conv_unit(state, prefix + ".conv3", x2, 1, training, norm, act)
    xs = [x1, x2, x3]
    for i in range(num_bottle):
        x3 = conv_unit(state, "%s.conv4.%d" % (prefix, i), x3, 1, training, norm, act)
        xs.append(x3)
    return conv_unit(state, prefix + ".conv5", torch.cat(xs, 1), 1, training, norm, act)


def yolov7neck(state, cfg, inputs, training, prefix="neck"):
    n, norm, act = cfg["depths"][0], cfg["norm"], cfg["act"]
    c3, c4, c5 = inputs
    u = lambda name, t, s=1: conv_unit(state, prefix + "." + name, t, s, training, norm, act)
    up = lambda t: F.interpolate(t, scale_factor=2, mode="nearest")
    p5 = sppcspc(state, prefix + ".spp", c5, training)
    p4 = elan_w(state, prefix + ".p5_p4", torch.cat([up(u("conv_for_P5", p5)), u("conv_for_C4", c4)], 1), n, training, norm, act)
    p3 = elan_w(state, prefix + ".p4_p3", torch.cat([up(u("conv_for_P4", p4)), u("conv_for_C3", c3)], 1), n, training, norm, act)
    n3 = p3
    n4 = elan_w(state, prefix + ".n3_n4", torch.cat([transition(state, prefix + ".downsample_conv1", n3, training, norm, act), p4], 1), n, training, norm, act)
    n5 = elan_w(state, prefix + ".n4_n5", torch.cat([transition(state, prefix + ".downsample_conv2", n4, training, norm, act), p5], 1), n, training, norm, act)
    return (u("n3", n3), u("n4", n4), u("n5", n5))


def implicit_head(state, inputs, prefix="head"):
    outs = []
    for k, x in enumerate(inputs):
        x = state["%s.ia.%d.implicit" % (prefix, k)] + x
        x = F.conv2d(x, _qw(state["%s.conv.%d.weight" % (prefix, k)]), state["%s.conv.%d.bias" % (prefix, k)])
        outs.append(state["%s.im.%d.implicit" % (prefix, k)] * x)
    return outs


def yolov7_network(state, cfg, x, training):
    f = eelan(state, cfg["backbone"], x, training)
    f = yolov7neck(state, cfg["neck"], f, training)
    return implicit_head(state, f)


def eval_decode(maps, strides, anchors, num_classes):
    """yolov7_loss.py:43-78 -> [B, na*sum(hw), 5+C] = (x1,y1,x2,y2, sig(obj), sig(cls))."""
    ch = 5 + num_classes
    preds = []
    for m, s, anc in zip(maps, strides, anchors):
        B, _, h, w = m.shape
        na = len(anc)
        p = m.view(B, na, ch, h, w).permute(0, 1, 3, 4, 2).contiguous().sigmoid()
        yv, xv = torch.meshgrid([torch.arange(h), torch.arange(w)], indexing="ij")
        grid = torch.stack((xv, yv), 2).view(1, 1, h, w, 2).to(p.dtype)
        ag = torch.tensor(anc, dtype=p.dtype).view(1, na, 1, 1, 2)
        xy = (p[..., :2] * 2.0 - 0.5 + grid) * s
        wh = (p[..., 2:4] * 2) ** 2 * ag
        preds.append(torch.cat([xy, wh, p[..., 4:]], -1).reshape(B, -1, ch))
    P = torch.cat(preds, 1)
    box = torch.stack([P[..., 0] - P[..., 2] / 2, P[..., 1] - P[..., 3] / 2, P[..., 0] + P[..., 2] / 2, P[..., 1] + P[..., 3] / 2], -1)
    return torch.cat([box, P[..., 4:]], -1)


def repconv(state, prefix, x, training, act="silu"):
    """RepConv, train-time form (models/necks/yolov7_neck.py:167-211): act(bn(conv3x3(x)) + bn(conv1x1(x))
    [+ bn_id(x) when c1 == c2 and stride 1]).  Plain nn.BatchNorm2d defaults (eps 1e-5, momentum 0.1).
    state keys: <prefix>.rbr_dense.{0.weight,1.*}, <prefix>.rbr_1x1.{0.weight,1.*}, <prefix>.rbr_identity.*"""
    def bn(p, t):
        return F.batch_norm(t, state[p + ".running_mean"], state[p + ".running_var"], state[p + ".weight"], state[p + ".bias"],
                            training, 0.1, 1e-5)
    out = bn(prefix + ".rbr_dense.1", F.conv2d(x, state[prefix + ".rbr_dense.0.weight"], None, 1, 1))
    out = out + bn(prefix + ".rbr_1x1.1", F.conv2d(x, state[prefix + ".rbr_1x1.0.weight"], None, 1, 0))
    if prefix + ".rbr_identity.weight" in state:
        out = out + bn(prefix + ".rbr_identity", x)
    return F.silu(out) if act == "silu" else out

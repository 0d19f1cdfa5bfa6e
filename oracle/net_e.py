"""Functional fp32 CPU restatement of the e-yolox network graph (TEST ORACLE; never imported by pl_yolo_amd).

Reference semantics restated here (file:line into /root/reference):
  * Bottleneck  dw3x3+BN -> 1x1 (no BN)+act -> 1x1+BN -> dw3x3+BN (+x) -> act      models/backbones/ecmnet.py:142-177
                                                                                 (== models/necks/pafpn_al.py:147-182)
  * CSPLayer    conv1 || conv2 -> conv3 chain on x_2 -> conv4 chain on x_3 -> cat(x_1, x_2, x_3, x_4)   ecmnet.py:91-139
                hidden width in_channels // 4, num_conv = num_bottle // 2 if num_bottle > 2 else 1
  * ECMNet      Focus stem + four stages (3x3 s2 conv, CSPLayer; SPP in stage 4)   ecmnet.py:17-88
  * AL_PAFPN    1x1 shrink convs, bicubic x2 upsampling, CSPLayers, 3x3 s2 convs   pafpn_al.py:7-93
The head and the loss are the YOLOX ones (oracle/net.py, oracle/yolox_loss.py).
Pinned to tests/golden/network_eyolox_test.npz (tools/gen_golden.py: gen_network_e runs the reference itself)."""
import torch
import torch.nn.functional as F

from .net import conv_unit, activation, focus, spp_bottleneck, decoupled_head, _q


def bottleneck(state, prefix, x, shortcut, training, norm, act):
    c = x.shape[1]
    y = conv_unit(state, prefix + ".conv0", x, 1, training, norm, None, groups=c)
    y = conv_unit(state, prefix + ".conv1", y, 1, training, None, act)
    y = conv_unit(state, prefix + ".conv2", y, 1, training, norm, None)
    co = y.shape[1]
    y = conv_unit(state, prefix + ".conv3", y, 1, training, norm, None, residual=x if (shortcut and co == c) else None, groups=co)
    return _q(activation(y, act))


def csp_layer(state, prefix, x, num_bottle, training, norm, act):
    num_conv = num_bottle // 2 if num_bottle > 2 else 1
    x_1 = conv_unit(state, prefix + ".conv1", x, 1, training, norm, act)
    x_2 = conv_unit(state, prefix + ".conv2", x, 1, training, norm, act)
    x_3 = x_2
    for i in range(num_conv):
        x_3 = bottleneck(state, "%s.conv3.%d" % (prefix, i), x_3, True, training, norm, act)
    x_4 = x_3
    for i in range(num_conv):
        x_4 = bottleneck(state, "%s.conv4.%d" % (prefix, i), x_4, True, training, norm, act)
    return torch.cat([x_1, x_2, x_3, x_4], 1)


def ecmnet(state, cfg, x, training, prefix="backbone"):
    norm, act, d = cfg["norm"], cfg["act"], cfg["depths"]
    x = conv_unit(state, prefix + ".stem.conv", _q(focus(x)), 1, training, norm, act)
    outs = {"stem": x}
    for s in (1, 2, 3):
        x = conv_unit(state, "%s.stage%d.0" % (prefix, s), x, 2, training, norm, act)
        x = csp_layer(state, "%s.stage%d.1" % (prefix, s), x, d[s - 1], training, norm, act)
        outs["stage%d" % s] = x
    x = conv_unit(state, prefix + ".stage4.0", x, 2, training, norm, act)
    x = spp_bottleneck(state, prefix + ".stage4.1", x, training, norm, act)
    x = csp_layer(state, prefix + ".stage4.2", x, d[3], training, norm, act)
    outs["stage4"] = x
    return [v for k, v in outs.items() if k in cfg["outputs"]]


def _up(x):
    return _q(F.interpolate(x, scale_factor=2, mode="bicubic"))


def al_pafpn(state, cfg, inputs, training, prefix="neck"):
    n, norm, act = cfg["depths"][0], cfg["norm"], cfg["act"]
    c3, c4, c5 = inputs
    p5_expand = conv_unit(state, prefix + ".shrink_conv1", c5, 1, training, norm, act)
    p4 = conv_unit(state, prefix + ".shrink_conv2", torch.cat([_up(p5_expand), c4], 1), 1, training, norm, act)
    p4 = csp_layer(state, prefix + ".p5_p4", p4, n, training, norm, act)
    p4_expand = conv_unit(state, prefix + ".shrink_conv3", p4, 1, training, norm, act)
    p3 = conv_unit(state, prefix + ".shrink_conv4", torch.cat([_up(p4_expand), c3], 1), 1, training, norm, act)
    p3 = csp_layer(state, prefix + ".p4_p3", p3, n, training, norm, act)
    n3 = p3
    n4 = torch.cat([conv_unit(state, prefix + ".downsample_conv1", n3, 2, training, norm, act), p4_expand], 1)
    n4 = csp_layer(state, prefix + ".n3_n4", n4, n, training, norm, act)
    n5 = torch.cat([conv_unit(state, prefix + ".downsample_conv2", n4, 2, training, norm, act), p5_expand], 1)
    n5 = csp_layer(state, prefix + ".n4_n5", n5, n, training, norm, act)
    return (n3, n4, n5)


def eyolox_network(state, cfg, x, training):
    """backbone -> neck -> head; returns the list of 3 raw NCHW head maps."""
    f = ecmnet(state, cfg["backbone"], x, training)
    f = al_pafpn(state, cfg["neck"], f, training)
    return decoupled_head(state, cfg["head"], f, training)

"""Functional fp32 CPU restatement of the YOLOX network graph (TEST ORACLE).

Everything is driven by a flat ``state`` dict that uses the reference's
state_dict key layout (e.g. ``backbone.stem.conv.conv.weight``), so a
state_dict captured from the imported reference plugs in directly.

Reference semantics restated here (file:line into /root/reference):
  * conv unit  = act(bn(conv(x)))          models/layers/network_blocks.py:7-40
      BN eps 1e-3, momentum 0.03           models/layers/normalization.py:8
  * Focus      = space-to-depth TL,BL,TR,BR models/layers/network_blocks.py:43-65
  * Bottleneck = 1x1 -> 3x3 (+x)            models/layers/network_blocks.py:68-91
  * CSPLayer                                models/layers/network_blocks.py:94-131
  * SPPBottleneck (pools 5/9/13)            models/layers/network_blocks.py:134-155
  * CSPDarkNet                              models/backbones/darknet_csp.py:15-75
  * CSPPAFPN (all CSP depths = depths[0])   models/necks/pafpn_csp.py:7-86
  * DecoupledHead                           models/heads/decoupled_head.py:7-95
"""
import math

import torch
import torch.nn.functional as F

BN_EPS = 1e-3
BN_MOMENTUM = 0.03

# Optional emulation of the HIP bf16 pipeline's storage roundings (used to check the bf16
# MFMA path tightly): bf16 weights, bf16 raw conv output `z` (the batch statistics are taken
# from the fp32 accumulators first), bf16 activations, bf16 image after the Focus gather;
# BatchNorm / SiLU arithmetic stays fp32, prediction convs write fp32.  Under autograd the
# same points round the GRADIENT to bf16 on the way back (`_q`: the cast pair's backward is a
# cast pair) -- the HIP backward stores d(activation) and dz as bf16 at exactly these places --
# while weight gradients stay fp32 like the MFMA weight-gradient slabs (`_qw`).
# Enabled with `with emulate_bf16(): ...`.
_EMU = [False]


class emulate_bf16:
    def __enter__(self):
        self.prev = _EMU[0]
        _EMU[0] = True

    def __exit__(self, *exc):
        _EMU[0] = self.prev
        return False


def _q(t):
    return t.to(torch.bfloat16).to(torch.float32) if _EMU[0] else t


def _qw(w):
    """bf16-rounded weights in the forward, UNROUNDED (fp32) gradient in the backward."""
    return w + (w.to(torch.bfloat16).to(torch.float32) - w).detach() if _EMU[0] else w


def activation(x, name):
    """models/layers/activation.py:5-20 (module semantics, functional form)."""
    if name is None:
        return x
    if name == "silu":
        return F.silu(x)
    if name == "relu":
        return F.relu(x)
    if name == "lrelu":
        return F.leaky_relu(x, 0.1)
    if name == "hswish":
        return x * F.relu6(x + 3) / 6
    if name == "gelu":
        return F.gelu(x)
    raise AttributeError("Unsupported activation function type: {}".format(name))


def conv_unit(state, prefix, x, stride, training, norm="bn", act="silu", residual=None, groups=1):
    """One BaseConv: conv (no bias, same padding) -> BN (batch stats in
    training, running stats in eval; running buffers updated in place) -> act."""
    w = state[prefix + ".conv.weight"]
    k = w.shape[-1]
    z = F.conv2d(x, _qw(w), state.get(prefix + ".conv.bias"), stride, (k - 1) // 2, 1, groups)
    if norm is not None and _EMU[0]:
        if norm != "bn":
            raise AttributeError("Unsupported normalization function type: {}".format(norm))
        g, b = state[prefix + ".norm.weight"], state[prefix + ".norm.bias"]
        rm, rv = state[prefix + ".norm.running_mean"], state[prefix + ".norm.running_var"]
        if training:
            mean = z.mean((0, 2, 3))
            var = z.var((0, 2, 3), unbiased=False)
            n = z.numel() / z.shape[1]
            with torch.no_grad():
                rm.mul_(1 - BN_MOMENTUM).add_(BN_MOMENTUM * mean)
                rv.mul_(1 - BN_MOMENTUM).add_(BN_MOMENTUM * var * n / max(n - 1, 1))
        else:
            mean, var = rm, rv
        scale = g / torch.sqrt(var + BN_EPS)
        shift = b - mean * scale
        z = _q(z) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)
    elif norm is not None:
        if norm != "bn":
            raise AttributeError("Unsupported normalization function type: {}".format(norm))
        z = F.batch_norm(
            z,
            state[prefix + ".norm.running_mean"],
            state[prefix + ".norm.running_var"],
            state[prefix + ".norm.weight"],
            state[prefix + ".norm.bias"],
            training,
            BN_MOMENTUM,
            BN_EPS,
        )
        if training:
            nbt = prefix + ".norm.num_batches_tracked"
            if nbt in state:
                state[nbt] += 1
    out = activation(z, act)
    if residual is not None:
        out = out + residual  # Bottleneck shortcut (network_blocks.py:89-90)
    return _q(out)


def focus(x):
    """Space-to-depth x2, channel blocks ordered TL, BL, TR, BR."""
    return torch.cat(
        (x[..., ::2, ::2], x[..., 1::2, ::2], x[..., ::2, 1::2], x[..., 1::2, 1::2]), dim=1
    )


def bottleneck(state, prefix, x, shortcut, training, norm, act):
    y = conv_unit(state, prefix + ".conv1", x, 1, training, norm, act)
    use_add = shortcut and (x.shape[1] == state[prefix + ".conv2.conv.weight"].shape[0])
    return conv_unit(state, prefix + ".conv2", y, 1, training, norm, act, residual=x if use_add else None)


def csp_layer(state, prefix, x, n, shortcut, training, norm, act):
    x1 = conv_unit(state, prefix + ".conv1", x, 1, training, norm, act)
    x2 = conv_unit(state, prefix + ".conv2", x, 1, training, norm, act)
    for i in range(n):
        x1 = bottleneck(state, "%s.m.%d" % (prefix, i), x1, shortcut, training, norm, act)
    return conv_unit(state, prefix + ".conv3", torch.cat((x1, x2), 1), 1, training, norm, act)


def spp_bottleneck(state, prefix, x, training, norm, act):
    x = conv_unit(state, prefix + ".conv1", x, 1, training, norm, act)
    pools = [F.max_pool2d(x, k, 1, k // 2) for k in (5, 9, 13)]
    # conv2 always carries a BN: the reference never forwards `norm` to it
    # (network_blocks.py:149).
    return conv_unit(state, prefix + ".conv2", torch.cat([x] + pools, 1), 1, training, "bn", act)


def cspdarknet(state, cfg, x, training, prefix="backbone"):
    depths, outs = cfg["depths"], cfg["outputs"]
    norm, act = cfg["norm"], cfg["act"]
    assert outs, "please provide output features of Darknet!"
    feats = {}
    x = conv_unit(state, prefix + ".stem.conv", _q(focus(x)), 1, training, norm, act)
    feats["stem"] = x
    for s in (1, 2, 3):
        x = conv_unit(state, "%s.stage%d.0" % (prefix, s), x, 2, training, norm, act)
        x = csp_layer(state, "%s.stage%d.1" % (prefix, s), x, depths[s - 1], True, training, norm, act)
        feats["stage%d" % s] = x
    x = conv_unit(state, prefix + ".stage4.0", x, 2, training, norm, act)
    x = spp_bottleneck(state, prefix + ".stage4.1", x, training, norm, act)
    x = csp_layer(state, prefix + ".stage4.2", x, depths[3], False, training, norm, act)
    feats["stage4"] = x
    if len(outs) <= 1:
        return x
    return [v for k, v in feats.items() if k in outs]


def csppafpn(state, cfg, inputs, training, prefix="neck"):
    n = cfg["depths"][0]
    norm, act = cfg["norm"], cfg["act"]
    c3, c4, c5 = inputs
    p5_expand = conv_unit(state, prefix + ".shrink_conv1", c5, 1, training, norm, act)
    p4 = torch.cat([F.interpolate(p5_expand, scale_factor=2, mode="nearest"), c4], 1)
    p4 = csp_layer(state, prefix + ".p5_p4", p4, n, False, training, norm, act)
    p4_expand = conv_unit(state, prefix + ".shrink_conv2", p4, 1, training, norm, act)
    p3 = torch.cat([F.interpolate(p4_expand, scale_factor=2, mode="nearest"), c3], 1)
    p3 = csp_layer(state, prefix + ".p4_p3", p3, n, False, training, norm, act)
    n3 = p3
    n4 = torch.cat([conv_unit(state, prefix + ".downsample_conv1", n3, 2, training, norm, act), p4_expand], 1)
    n4 = csp_layer(state, prefix + ".n3_n4", n4, n, False, training, norm, act)
    n5 = torch.cat([conv_unit(state, prefix + ".downsample_conv2", n4, 2, training, norm, act), p5_expand], 1)
    n5 = csp_layer(state, prefix + ".n4_n5", n5, n, False, training, norm, act)
    return (n3, n4, n5)


def decoupled_head(state, cfg, inputs, training, prefix="head"):
    norm, act = cfg["norm"], cfg["act"]
    outs = []
    for k, x in enumerate(inputs):
        # stems ignore the `norm` argument (decoupled_head.py:29-31) -> always bn
        x = conv_unit(state, "%s.stems.%d" % (prefix, k), x, 1, training, "bn", act)
        c = x
        r = x
        for j in (0, 1):
            c = conv_unit(state, "%s.cls_convs.%d.%d" % (prefix, k, j), c, 1, training, norm, act)
            r = conv_unit(state, "%s.reg_convs.%d.%d" % (prefix, k, j), r, 1, training, norm, act)

        def pred(name, t):
            return F.conv2d(t, _qw(state["%s.%s.%d.weight" % (prefix, name, k)]), state["%s.%s.%d.bias" % (prefix, name, k)])

        outs.append(torch.cat([pred("reg_preds", r), pred("obj_preds", r), pred("cls_preds", c)], 1))
    return outs


def yolox_network(state, cfg, x, training):
    """backbone -> neck -> head; returns the list of 3 raw NCHW head maps."""
    f = cspdarknet(state, cfg["backbone"], x, training)
    if cfg["neck"]["name"] != "none":
        f = csppafpn(state, cfg["neck"], f, training)
    return decoupled_head(state, cfg["head"], f, training)


# ----------------------------------------------------------------------------
# State construction (PyTorch default init + head prior bias), key layout and
# creation ORDER identical to the reference module tree so that
# `torch.manual_seed(s)` followed by init reproduces the reference weights.
# ----------------------------------------------------------------------------

def _conv_init(cout, cin, k, bias=False):
    """nn.Conv2d default init: kaiming_uniform(a=sqrt(5)) on the weight, then
    (if bias) uniform(-1/sqrt(fan_in), 1/sqrt(fan_in)) -- same RNG call order."""
    w = torch.empty(cout, cin, k, k)
    torch.nn.init.kaiming_uniform_(w, a=math.sqrt(5))
    b = None
    if bias:
        bound = 1 / math.sqrt(cin * k * k)
        b = torch.empty(cout).uniform_(-bound, bound)
    return w, b


class StateBuilder:
    def __init__(self):
        self.state = {}

    def bn(self, prefix, c):
        s = self.state
        s[prefix + ".weight"] = torch.ones(c)
        s[prefix + ".bias"] = torch.zeros(c)
        s[prefix + ".running_mean"] = torch.zeros(c)
        s[prefix + ".running_var"] = torch.ones(c)
        s[prefix + ".num_batches_tracked"] = torch.tensor(0, dtype=torch.long)

    def unit(self, prefix, cin, cout, k):
        w, _ = _conv_init(cout, cin, k)
        self.state[prefix + ".conv.weight"] = w
        self.bn(prefix + ".norm", cout)

    def bottleneck(self, prefix, c):
        self.bn(prefix + ".bn", c)  # dead BN kept for key compatibility (network_blocks.py:81)
        self.unit(prefix + ".conv1", c, c, 1)
        self.unit(prefix + ".conv2", c, c, 3)

    def csp(self, prefix, cin, cout, n):
        h = int(cout * 0.5)
        self.unit(prefix + ".conv1", cin, h, 1)
        self.unit(prefix + ".conv2", cin, h, 1)
        self.unit(prefix + ".conv3", 2 * h, cout, 1)
        for i in range(n):
            self.bottleneck("%s.m.%d" % (prefix, i), h)


def build_state(cfg, num_classes):
    """Fresh state dict for a YOLOX-family config (cspdarknet + csppafpn +
    decoupled_head).  Consumes the torch global RNG in the reference's order."""
    b = StateBuilder()
    cb, cn, ch = cfg["backbone"], cfg["neck"], cfg["head"]
    c, d = cb["channels"], cb["depths"]
    b.unit("backbone.stem.conv", 12, c[0], 3)
    for s in (1, 2, 3):
        b.unit("backbone.stage%d.0" % s, c[s - 1], c[s], 3)
        b.csp("backbone.stage%d.1" % s, c[s], c[s], d[s - 1])
    b.unit("backbone.stage4.0", c[3], c[4], 3)
    b.unit("backbone.stage4.1.conv1", c[4], c[4] // 2, 1)
    b.unit("backbone.stage4.1.conv2", (c[4] // 2) * 4, c[4], 1)
    b.csp("backbone.stage4.2", c[4], c[4], d[3])
    if cn["name"] != "none":
        ic, n = cn["channels"], cn["depths"][0]
        b.unit("neck.shrink_conv1", ic[2], ic[1], 1)
        b.unit("neck.shrink_conv2", ic[1], ic[0], 1)
        b.csp("neck.p5_p4", 2 * ic[1], ic[1], n)
        b.csp("neck.p4_p3", 2 * ic[0], ic[0], n)
        b.unit("neck.downsample_conv1", ic[0], ic[0], 3)
        b.unit("neck.downsample_conv2", ic[1], ic[1], 3)
        b.csp("neck.n3_n4", 2 * ic[0], ic[1], n)
        b.csp("neck.n4_n5", 2 * ic[1], ic[2], n)
    hc = ch["channels"]
    na = ch["num_anchor"]
    for k in range(len(hc)):
        b.unit("head.stems.%d" % k, hc[k], hc[0], 1)
        for j in (0, 1):
            b.unit("head.cls_convs.%d.%d" % (k, j), hc[0], hc[0], 3)
        w, bias = _conv_init(na * num_classes, hc[0], 1, True)
        b.state["head.cls_preds.%d.weight" % k], b.state["head.cls_preds.%d.bias" % k] = w, bias
        for j in (0, 1):
            b.unit("head.reg_convs.%d.%d" % (k, j), hc[0], hc[0], 3)
        w, bias = _conv_init(na * 4, hc[0], 1, True)
        b.state["head.reg_preds.%d.weight" % k], b.state["head.reg_preds.%d.bias" % k] = w, bias
        w, bias = _conv_init(na * 1, hc[0], 1, True)
        b.state["head.obj_preds.%d.weight" % k], b.state["head.obj_preds.%d.bias" % k] = w, bias
    prior = -math.log((1 - 1e-2) / 1e-2)  # decoupled_head.py:64-75
    for k in range(len(hc)):
        b.state["head.cls_preds.%d.bias" % k].fill_(prior)
        b.state["head.obj_preds.%d.bias" % k].fill_(prior)
    return b.state


def param_names(state):
    """Keys that are trainable parameters (weights/biases, not BN buffers)."""
    return [k for k in state if not (k.endswith("running_mean") or k.endswith("running_var") or k.endswith("num_batches_tracked"))]

"""Deploy-time folding restated on the CPU (TEST ORACLE; never imported by pl_yolo_amd).

Reference semantics (file:line into /root/reference):
  * fold a BatchNorm into the convolution it follows      models/necks/yolov7_neck.py:222-249 (_fuse_bn_tensor),
                                                          :265-286 (fuse_conv_bn)
        std = sqrt(running_var + eps);  t = gamma / std
        kernel' = kernel * t[:, None, None, None];  bias' = beta - running_mean * gamma / std
  * RepConv -> one 3x3 conv + bias                        :213-220 (get_equivalent_kernel_bias), :288-348 (fuse_repvgg_block)
        kernel = k3x3' + pad(k1x1', [1,1,1,1]) + k_id'   (k_id = identity kernel folded with rbr_identity's BatchNorm)
        bias   = b3x3' + b1x1' + b_id'
  * deploy forward                                        :203-204  act(rbr_reparam(x));   network_blocks.py:39-40 fuseforward
Pinned to tests/golden/deploy_fold.npz (tools/gen_golden.py: gen_deploy runs the reference's own methods)."""
import torch
import torch.nn.functional as F

from .net import activation


def fold_bn(kernel, gamma, beta, running_mean, running_var, eps):
    std = (running_var + eps).sqrt()
    t = (gamma / std).reshape(-1, 1, 1, 1)
    return kernel * t, beta - running_mean * gamma / std


def _branch(state, prefix, kernel, eps):
    return fold_bn(kernel, state[prefix + ".weight"], state[prefix + ".bias"], state[prefix + ".running_mean"],
                   state[prefix + ".running_var"], eps)


def repconv_equivalent(state, eps=1e-5):
    """state: a RepConv state_dict (rbr_dense.0.weight, rbr_dense.1.*, rbr_1x1.*, optional rbr_identity.*)."""
    k3, b3 = _branch(state, "rbr_dense.1", state["rbr_dense.0.weight"], eps)
    k1, b1 = _branch(state, "rbr_1x1.1", state["rbr_1x1.0.weight"], eps)
    kernel, bias = k3 + F.pad(k1, [1, 1, 1, 1]), b3 + b1
    if "rbr_identity.weight" in state:
        c = state["rbr_identity.weight"].shape[0]
        kid = torch.zeros(c, c, 3, 3)
        kid[torch.arange(c), torch.arange(c), 1, 1] = 1.0
        ki, bi = _branch(state, "rbr_identity", kid, eps)
        kernel, bias = kernel + ki, bias + bi
    return kernel, bias


def repconv_deploy_forward(x, kernel, bias):
    return F.silu(F.conv2d(x, kernel, bias, 1, 1))


def baseconv_fold(state, eps=1e-3):
    """state: a BaseConv state_dict (conv.weight, norm.*) -> (weight, bias) of the fused convolution."""
    return fold_bn(state["conv.weight"], state["norm.weight"], state["norm.bias"], state["norm.running_mean"],
                   state["norm.running_var"], eps)


def baseconv_fuseforward(x, weight, bias, stride, act="silu"):
    k = weight.shape[-1]
    return activation(F.conv2d(x, weight, bias, stride, (k - 1) // 2), act)

"""CPU: the deploy-time folding oracle (oracle/deploy.py) against the fixture written by the reference's own
RepConv.get_equivalent_kernel_bias / fuse_repvgg_block / fuse_conv_bn and BaseConv.fuseforward."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import deploy as od


def _state(g, prefix):
    return {k[len(prefix):]: torch.from_numpy(v.copy()) for k, v in g.items() if k.startswith(prefix)}


@pytest.mark.parametrize("tag", ["ne", "id"])
def test_repconv_equivalent_kernel(tag):
    g = load_golden("deploy_fold")
    st = _state(g, "rep_%s/state/" % tag)
    kernel, bias = od.repconv_equivalent(st)
    np.testing.assert_allclose(kernel.numpy(), g["rep_%s/kernel" % tag], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(bias.numpy(), g["rep_%s/bias" % tag], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(kernel.numpy(), g["rep_%s/reparam_weight" % tag], rtol=1e-5, atol=1e-6)   # the second code path (:288-348)
    np.testing.assert_allclose(bias.numpy(), g["rep_%s/reparam_bias" % tag], rtol=1e-5, atol=1e-6)
    y = od.repconv_deploy_forward(torch.from_numpy(g["rep_%s/x" % tag]), kernel, bias)
    np.testing.assert_allclose(y.numpy(), g["rep_%s/y_fused" % tag], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(y.numpy(), g["rep_%s/y_eval" % tag], rtol=1e-4, atol=1e-5)     # == the unfused eval forward


@pytest.mark.parametrize("tag", ["k3", "k1", "k3s2"])
def test_baseconv_fold(tag):
    g = load_golden("deploy_fold")
    st = _state(g, "base_%s/state/" % tag)
    w, b = od.baseconv_fold(st)
    np.testing.assert_allclose(w.numpy(), g["base_%s/fused_weight" % tag], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(b.numpy(), g["base_%s/fused_bias" % tag], rtol=1e-6, atol=1e-7)
    y = od.baseconv_fuseforward(torch.from_numpy(g["base_%s/x" % tag]), w, b, int(g["base_%s/stride" % tag]))
    np.testing.assert_allclose(y.numpy(), g["base_%s/y_fused" % tag], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(y.numpy(), g["base_%s/y_eval" % tag], rtol=1e-4, atol=1e-5)

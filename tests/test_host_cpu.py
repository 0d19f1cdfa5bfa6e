"""CPU checks of the host side: C-ABI completeness, plugin API surface, launch-plan
recording (no kernel runs without a GPU), loud failure without a device."""
import ctypes
import os
import re

import pytest
import torch
import yaml

import pl_yolo_amd
from pl_yolo_amd import _lib, graph as G, trainer
from conftest import ROOT, load_golden


def _cfg(name):
    with open(os.path.join(ROOT, "configs", "model", "yolox", name + ".yaml")) as f:
        return yaml.safe_load(f)


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "plyolo.h")).read()
    declared = set(re.findall(r"\b(plyolo_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"plyolo_plan", "plyolo_conv_desc", "plyolo_pack_entry", "plyolo_yolox_desc", "plyolo_nms_desc"}
    lib = ctypes.CDLL(_lib.LIB_PATH)
    missing = [s for s in sorted(declared) if not hasattr(lib, s)]
    assert not missing, missing
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    l = _lib.lib()
    assert l.plyolo_version() == _lib.ABI_VERSION == 6 and l.plyolo_arch() == b"gfx950"   # include/plyolo.h: PLYOLO_ABI_VERSION


def test_error_reporting_across_the_abi():
    d = _lib.ConvDesc()
    d.dtype, d.N, d.H, d.W, d.Cin, d.Cout, d.ksize, d.stride, d.x_ld, d.y_ld = 0, 1, 8, 8, 8, 8, 5, 1, 8, 8
    with pytest.raises(_lib.PlyoloError, match="ksize"):
        _lib.call("plyolo_conv2d_fwd", ctypes.byref(d), None, None, None, None, None, None)


def test_plugin_surface_and_state_dict_layout():
    g = load_golden("network_yolox_test")
    model = pl_yolo_amd.build_model(_cfg("yolox_test"), int(g["num_classes"]))
    ref_keys = {k[6:] for k in g if k.startswith("state/")}
    assert set(model.state_dict()) == ref_keys
    for k, v in model.state_dict().items():
        assert tuple(v.shape) == tuple(g["state/" + k].shape), k
    assert all(hasattr(model, a) for a in ("backbone", "neck", "head", "loss"))
    # YOLOX-s: the reference's parameter count and state_dict size
    m = pl_yolo_amd.build_model(_cfg("yolox_s"), 80)
    assert sum(p.numel() for p in m.parameters()) == 8971135
    assert len(m.state_dict()) == 522
    # prior-probability bias init of the head (decoupled_head.py:64-75)
    import math
    assert torch.allclose(m.head.cls_preds[0].bias, torch.full((80,), -math.log(99.0)))
    # compute_dtype: constructor argument / optional YAML key (default bf16, PLYOLO_DTYPE overrides the default)
    assert pl_yolo_amd.build_model(dict(_cfg("yolox_test"), compute_dtype="fp32"), 3).compute_dtype == "fp32"
    assert pl_yolo_amd.OneStageD(compute_dtype="bf16").compute_dtype == "bf16"
    with pytest.raises(pl_yolo_amd.PlyoloError):
        pl_yolo_amd.OneStageD(compute_dtype="fp16")
    with pytest.raises(NameError):
        pl_yolo_amd.build_model(dict(_cfg("yolox_test"), backbone=dict(_cfg("yolox_test")["backbone"], name="nosuch")), 3)
    with pytest.raises(KeyError):
        pl_yolo_amd.build_model({"backbone": {"name": "cspdarknet"}, "neck": {"name": "none"}, "head": {}, "loss": {}}, 3)
    with pytest.raises(AttributeError):
        pl_yolo_amd.build_model(dict(_cfg("yolox_test"), backbone=dict(_cfg("yolox_test")["backbone"], act="swishh")), 3)


def test_no_cpu_fallback():
    model = pl_yolo_amd.build_model(_cfg("yolox_test"), 3)
    with pytest.raises(pl_yolo_amd.PlyoloError, match="no CPU path"):
        model(torch.zeros(1, 3, 64, 64), torch.zeros(1, 4, 5))
    with pytest.raises(pl_yolo_amd.PlyoloError):
        model.backbone(torch.zeros(1, 3, 64, 64))
    from pl_yolo_amd import postprocess
    with pytest.raises(pl_yolo_amd.PlyoloError):
        postprocess.postprocess(torch.zeros(1, 10, 8))


@pytest.mark.parametrize("dtype", ["bf16", "fp32"])
def test_plan_recording_dry_run(dtype):
    """Trace + buffer planning + launch recording work without a GPU (record mode never
    touches the device): zero-copy concats, every used parameter gets a gradient slot,
    no gradient view is left partially initialised."""
    model = pl_yolo_amd.build_model(_cfg("yolox_test"), 3).train()
    model.compute_dtype = dtype
    r = model.runner()
    dev = torch.device("cpu")
    r.adopt(dev)
    s = r._build(2, 64, 64, 8, "train", dev)
    assert s.fwd.size() > 100 and s.bwd.size() > 200
    assert not [op for op in s.g.ops if isinstance(op, G.CopyOp)], "a concat fell back to a copy"
    n_params = len(list(model.parameters()))
    assert len(s.used_params) == n_params - 16  # all but the dead Bottleneck.bn pairs (network_blocks.py:81)
    assert all(all(st.ginit) for st in s.g.storages if st.grad is not None)
    # parameters are views into one flat buffer, gradients into another
    p = s.used_params[0]
    assert r.flat["w"].data_ptr() <= p.data_ptr() < r.flat["w"].data_ptr() + r.flat["n"] * 4
    ev = r._build(2, 64, 64, 1, "eval", dev)
    assert ev.bwd is None and ev.fwd.size() > 100


def test_lr_schedule_matches_reference_vectors():
    g = load_golden("lr_schedule")
    for i in range(3):
        warm, T = float(g["sched%d_warm" % i]), int(g["sched%d_T" % i])
        for t in (0, 1, int(warm), int(warm) + 1, T // 2, T):
            assert abs(trainer.lr_factor(t, warm, T) - float(g["sched%d_factor" % i][t])) < 1e-12


def test_repconv_module_keys_match_reference_fixture():
    """RepConv (row a13) keeps the reference's sub-module names, so its checkpoints load unchanged."""
    from conftest import load_golden
    from pl_yolo_amd.necks import RepConv
    g = load_golden("repconv_blocks")
    for tag, c1, c2 in (("ne", 16, 32), ("id", 24, 24)):
        want = {k[len(tag) + 7:]: v.shape for k, v in g.items() if k.startswith(tag + "/state/")}
        got = {k: tuple(v.shape) for k, v in RepConv(c1, c2, 3, 1).state_dict().items()}
        assert got == {k: tuple(v) for k, v in want.items()}


def test_model_summary_counts_match_the_reference_tables(capsys):
    """utils/flops.py:5-9 (thop) counts 0 operations on a model whose leaf modules never run; pl_yolo_amd.utils.model_summary prints the
    reference's line from a CPU dry trace of the launch graph.  Numbers: BASELINE.md section 2 (counted from the reference's own modules;
    the upstream YOLOX tables: 26.8 / 155.6 GFLOPs)."""
    from pl_yolo_amd.utils import conv_macs, model_summary
    want = {("yolox", "yolox_s", 640): (26.69, 8.971), ("yolox", "yolox_l", 640): (155.29, 54.226), ("yolov7", "yolov7", 640): (112.48, 47.698),
            ("yolox", "yolox_nano", 416): (2.91, 2.258)}
    for (family, name, size), (gf, mp) in want.items():
        with open(os.path.join(ROOT, "configs", "model", family, name + ".yaml")) as f:
            model = pl_yolo_amd.build_model(yaml.safe_load(f), 80)
        assert abs(2.0 * conv_macs(model, size, size) / 1e9 - gf) < 0.006, name
        assert abs(sum(p.numel() for p in model.parameters()) / 1e6 - mp) < 0.0006, name
    assert model_summary(model, (416, 416), None) is None
    assert capsys.readouterr().out.strip() == "------- params: 2.258M ------- flops: 2.913G"
    with pytest.raises(_lib.PlyoloError, match="multiple of 32"):
        conv_macs(model, 400, 416)

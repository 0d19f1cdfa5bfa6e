#!/usr/bin/env python3
"""Generate golden vectors by IMPORTING the reference (build container only).

Run:  python tools/gen_golden.py            (needs /root/reference mounted)

The reference publishes no tests or golden vectors, so
every fixture under tests/golden/ is produced here by running the reference's
own modules on seeded inputs.  Fixtures are DATA (inputs + expected outputs);
no reference source text is stored.  /root/reference does not exist on the GPU
box, so nothing at test time imports this script.
"""
import os
import sys

import numpy as np
import torch
import yaml

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, REF)

from PL_Modules.build_detection import build_model  # noqa: E402
from models.layers.network_blocks import BaseConv, CSPLayer, SPPBottleneck, Focus, Bottleneck  # noqa: E402
from models.losses.yolox import yolox_loss as ref_loss_mod  # noqa: E402
from models.losses.yolox.yolox_loss import YOLOXLoss  # noqa: E402
from models.layers.lr_scheduler import CosineWarmupScheduler  # noqa: E402
from models.utils.ema import ModelEMA  # noqa: E402

torch.set_num_threads(4)


def npy(d):
    return {k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in d.items()}


def save(name, d):
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **npy(d))
    print("wrote", path, "%.1f KB" % (os.path.getsize(path) / 1024))


def load_cfg(name):
    with open(os.path.join(ROOT, "configs", "model", "yolox", name + ".yaml")) as f:
        return yaml.safe_load(f)


# --------------------------------------------------------------------------
# capture the SimOTA assignment from inside the reference
# --------------------------------------------------------------------------
_calls = []
_orig_dk = ref_loss_mod.dynamic_k_matching


def _dk_spy(fg_mask, cost, pair_wise_ious, gt_classes, num_gt):
    r = _orig_dk(fg_mask, cost, pair_wise_ious, gt_classes, num_gt)
    # boundary gap diagnostic on the reference's own cost matrix
    n_k = min(10, pair_wise_ious.size(1))
    ksum = pair_wise_ious.sort(descending=True)[0][:, :n_k].sum(1)
    ks = torch.clamp(ksum.int(), min=1)
    # distance of every dynamic-k sum from the integer where int() would change k (sums below 1 clamp to k = 1: safe below 1)
    kmargin = float(torch.where(ksum < 1.0, 1.0 - ksum, torch.minimum(ksum - ksum.floor(), ksum.floor() + 1.0 - ksum)).min()) if num_gt > 0 else float("inf")
    gap = float("inf")
    for g in range(num_gt):
        sc = cost[g].sort()[0]
        k = int(ks[g])
        if k < sc.numel() - 1:
            gap = min(gap, abs(float(sc[k]) - float(sc[k - 1])) / max(abs(float(sc[k - 1])), 1.0))
    _calls.append(dict(fg=r[0].clone(), num_fg=int(r[1]), mg=r[2].clone(), iou=r[4].clone(), gap=gap, ks=ks.clone(), kmargin=kmargin))
    return r


ref_loss_mod.dynamic_k_matching = _dk_spy


def run_ref_loss(maps, labels, strides, num_classes, use_l1=False):
    """maps: list of leaf-less tensors; returns dict of outputs incl. dense assignment."""
    _calls.clear()
    loss_fn = YOLOXLoss(num_classes, strides, use_l1=use_l1)
    loss_fn.train()
    leafs = [m.clone().requires_grad_(True) for m in maps]
    # the loss writes through views of its inputs: hand it non-leaf copies
    out = loss_fn([l * 1.0 for l in leafs], labels)
    out["loss"].backward()
    B = maps[0].shape[0]
    A = sum(m.shape[2] * m.shape[3] for m in maps)
    fg = torch.zeros(B, A, dtype=torch.bool)
    mg = torch.full((B, A), -1, dtype=torch.int64)
    mi = torch.zeros(B, A)
    nlabel = (labels.sum(dim=2) > 0).sum(dim=1)
    ci = 0
    gap = float("inf")
    for b in range(B):
        if int(nlabel[b]) == 0:
            continue
        c = _calls[ci]
        ci += 1
        fg[b] = c["fg"]
        mg[b, c["fg"]] = c["mg"]
        mi[b, c["fg"]] = c["iou"]
        gap = min(gap, c["gap"])
    assert ci == len(_calls)
    res = dict(
        loss=out["loss"], loss_iou=out["loss_iou"], loss_obj=out["loss_obj"], loss_cls=out["loss_cls"],
        proportion=float(out["proportion"]), fg=fg, matched_gt=mg, matched_iou=mi, boundary_gap=gap,
    )
    if use_l1:
        res["loss_l1"] = out["loss_l1"]
        res["use_l1"] = np.asarray(1)
    for i, l in enumerate(leafs):
        res["grad%d" % i] = l.grad
    loss_fn.eval()
    res["eval_decode"] = loss_fn([m.clone() for m in maps], labels)
    return res


def rand_maps(gen, B, C, sizes, reg_std=0.5):
    maps = []
    for h, w in sizes:
        m = torch.randn(B, 5 + C, h, w, generator=gen)
        m[:, :4] *= reg_std
        m[:, 4] = m[:, 4] * 2.0 - 2.0
        m[:, 5:] = m[:, 5:] * 2.0 - 2.0
        maps.append(m)
    return maps


def rand_labels(gen, counts, C, size, max_gt, min_wh=8.0):
    B = len(counts)
    lab = torch.zeros(B, max_gt, 5)
    for b, g in enumerate(counts):
        lab[b, :g, 0] = torch.randint(0, C, (g,), generator=gen).float()
        lab[b, :g, 1:3] = (0.15 + 0.7 * torch.rand(g, 2, generator=gen)) * size
        lab[b, :g, 3:5] = min_wh + torch.rand(g, 2, generator=gen) * 0.3 * size
    return lab


def gen_loss_cases(only=None):
    cases = {}
    # A: realistic multi-level case incl. an image with zero GTs
    gen = torch.Generator().manual_seed(101)
    maps = rand_maps(gen, 4, 80, [(20, 20), (10, 10), (5, 5)])
    labels = rand_labels(gen, [12, 0, 30, 5], 80, 160, 40)
    cases["A"] = (maps, labels, [8, 16, 32], 80)
    # B: few classes, overlapping GTs (conflict resolution) + a tiny edge GT
    gen = torch.Generator().manual_seed(202)
    maps = rand_maps(gen, 2, 3, [(16, 16), (8, 8), (4, 4)])
    labels = torch.zeros(2, 10, 5)
    labels[0, 0] = torch.tensor([0, 60.0, 60.0, 50.0, 40.0])
    labels[0, 1] = torch.tensor([1, 62.0, 58.0, 48.0, 44.0])
    labels[0, 2] = torch.tensor([2, 64.0, 64.0, 90.0, 90.0])
    labels[0, 3] = torch.tensor([1, 20.0, 100.0, 30.0, 30.0])
    labels[1, 0] = torch.tensor([2, 3.0, 3.0, 2.5, 2.5])     # tiny GT hugging the corner
    labels[1, 1] = torch.tensor([0, 100.0, 30.0, 40.0, 20.0])
    cases["B"] = (maps, labels, [8, 16, 32], 3)
    # C: single 2x2 level, well-fitting predictions -> k >= N_c - 1 branch (take ALL)
    gen = torch.Generator().manual_seed(303)
    m = torch.zeros(2, 5 + 4, 2, 2)
    m[:, 4:] = torch.randn(2, 5, 2, 2, generator=gen)
    # decoded box = ((t+g)*32, exp(t)*32): aim every anchor at the GT (32,32,40,40)
    for gy in range(2):
        for gx in range(2):
            m[:, 0, gy, gx] = 1.0 - gx + 0.01 * (gx + 2 * gy)
            m[:, 1, gy, gx] = 1.0 - gy - 0.01 * (gx + 2 * gy)
            m[:, 2, gy, gx] = float(np.log(40.0 / 32)) + 0.02 * gx
            m[:, 3, gy, gx] = float(np.log(40.0 / 32)) - 0.02 * gy
    labels = torch.zeros(2, 4, 5)
    labels[0, 0] = torch.tensor([1, 32.0, 32.0, 40.0, 40.0])
    labels[1, 0] = torch.tensor([3, 32.0, 32.0, 40.0, 40.0])
    labels[1, 1] = torch.tensor([0, 30.0, 34.0, 36.0, 44.0])
    cases["C"] = ([m], labels, [32], 4)
    # D: non-square map -> documents the grid quirk (yolox_loss.py:198-200)
    gen = torch.Generator().manual_seed(404)
    maps = rand_maps(gen, 2, 2, [(2, 3)])
    labels = torch.zeros(2, 3, 5)
    labels[0, 0] = torch.tensor([1, 20.0, 12.0, 18.0, 14.0])
    labels[1, 0] = torch.tensor([0, 30.0, 20.0, 25.0, 22.0])
    labels[1, 1] = torch.tensor([1, 10.0, 10.0, 12.0, 12.0])
    cases["D"] = (maps, labels, [16], 2)
    # E: larger candidate sets, 60 GTs in one image, COCO-like class count
    gen = torch.Generator().manual_seed(505)
    maps = rand_maps(gen, 2, 80, [(32, 32), (16, 16), (8, 8)], reg_std=0.3)
    labels = rand_labels(gen, [60, 17], 80, 256, 64, min_wh=6.0)
    cases["E"] = (maps, labels, [8, 16, 32], 80)

    # F, G: YOLOXLoss(use_l1=True) (yolox_loss.py:128-135,157-158) on the inputs of A and E
    cases["F"] = cases["A"] + (True,)
    cases["G"] = cases["E"] + (True,)
    for name, case in cases.items():
        if only is not None and name not in only:
            continue
        maps, labels, strides, C = case[:4]
        r = run_ref_loss(maps, labels, strides, C, use_l1=len(case) > 4)
        d = dict(labels=labels, strides=np.asarray(strides), num_classes=C, nmaps=len(maps))
        for i, m in enumerate(maps):
            d["map%d" % i] = m
        d.update(r)
        print("loss case", name, "loss=%.6f" % float(r["loss"]), "num_fg=%d" % int(r["fg"].sum()), "gap=%.3g" % r["boundary_gap"])
        save("loss_case_" + name, d)


def gen_loss_side_effects():
    """What YOLOXLoss.__call__ leaves in the CALLER's head maps (yolox_loss.py:204-219): `permute(...).reshape(...)` of an NCHW map with one
    anchor is a view, so the decode writes the boxes (cx, cy, w, h in pixels) through into channels 0..3 -- in training and in eval mode.
    Inputs: the maps of the committed loss cases A (three square levels) and D (one NON-square level: the grid quirk).  Stored: the box
    channels after the call; the generator asserts that the other channels are untouched and that train and eval leave the same."""
    out = {}
    for case in ("A", "D"):
        g = dict(np.load(os.path.join(OUT, "loss_case_%s.npz" % case)))
        maps = [torch.from_numpy(g["map%d" % i]) for i in range(int(g["nmaps"]))]
        labels, strides, nc = torch.from_numpy(g["labels"]), [int(v) for v in g["strides"]], int(g["num_classes"])
        fn = YOLOXLoss(nc, strides)
        fn.train()
        tr = [m.clone().requires_grad_(True) * 1.0 for m in maps]
        fn(tr, labels)
        fn.eval()
        ev = [m.clone() for m in maps]
        fn(ev, labels)
        for i, (m, a, b) in enumerate(zip(maps, tr, ev)):
            a = a.detach()
            assert torch.equal(a, b) and torch.equal(a[:, 4:], m[:, 4:]) and not torch.equal(a[:, :4], m[:, :4])
            out["%s/boxes_after%d" % (case, i)] = a[:, :4].contiguous()
    save("loss_side_effects", out)


def gen_blocks():
    torch.manual_seed(7)
    d = {}

    def run(tag, mod, x):
        mod.train()
        # non-trivial BN affine so gamma/beta gradients are exercised
        for n, p in mod.named_parameters():
            if n.endswith("norm.weight") or n.endswith("bn.weight"):
                p.data.uniform_(0.5, 1.5)
            if n.endswith("norm.bias") or n.endswith("bn.bias"):
                p.data.uniform_(-0.5, 0.5)
        for k, v in mod.state_dict().items():
            d["%s/state/%s" % (tag, k)] = v.clone()
        xl = x.clone().requires_grad_(True)
        y = mod(xl)
        r = torch.randn(y.shape)
        (y * r).sum().backward()
        d[tag + "/x"] = x
        d[tag + "/y"] = y
        d[tag + "/r"] = r
        d[tag + "/dx"] = xl.grad
        for n, p in mod.named_parameters():
            if p.grad is not None:
                d["%s/grad/%s" % (tag, n)] = p.grad
        for k, v in mod.state_dict().items():
            if "running" in k:
                d["%s/state_after/%s" % (tag, k)] = v.clone()

    run("conv3s2", BaseConv(8, 16, 3, 2), torch.randn(2, 8, 16, 16))
    run("conv3s1", BaseConv(8, 16, 3, 1), torch.randn(2, 8, 12, 12))
    run("conv1", BaseConv(16, 8, 1, 1), torch.randn(2, 16, 8, 8))
    run("focus", Focus(3, 8, ksize=3), torch.rand(2, 3, 16, 16) * 255)
    run("bottleneck", Bottleneck(8, 8, True, 1.0), torch.randn(2, 8, 8, 8))
    run("csp", CSPLayer(16, 16, num_bottle=2), torch.randn(2, 16, 8, 8))
    run("csp_noshort", CSPLayer(32, 16, num_bottle=1, shortcut=False), torch.randn(2, 32, 8, 8))
    run("spp", SPPBottleneck(32, 32), torch.randn(2, 32, 8, 8))
    save("blocks", d)


def gen_network():
    cfg = load_cfg("yolox_test")
    C = 3
    torch.manual_seed(96)
    model = build_model(cfg, C)
    # perturb BN affine + running stats so eval mode is non-trivial too
    g = torch.Generator().manual_seed(5)
    for n, p in model.named_parameters():
        if n.endswith(".norm.weight"):
            p.data = 0.5 + torch.rand(p.shape, generator=g)
        if n.endswith(".norm.bias"):
            p.data = torch.rand(p.shape, generator=g) - 0.5
    gen = torch.Generator().manual_seed(1234)
    x = torch.rand(2, 3, 64, 64, generator=gen) * 255
    labels = torch.zeros(2, 8, 5)
    labels[0, :3] = torch.tensor([[0, 20.0, 24.0, 18.0, 22.0], [2, 40.0, 40.0, 30.0, 26.0], [1, 50.0, 14.0, 16.0, 12.0]])
    labels[1, :2] = torch.tensor([[1, 30.0, 30.0, 40.0, 36.0], [0, 12.0, 50.0, 14.0, 18.0]])
    d = dict(x=x, labels=labels, num_classes=C)
    for k, v in model.state_dict().items():
        d["state/" + k] = v.clone()
    model.train()
    _calls.clear()
    maps = model(x)  # labels=None -> raw head maps (this also updates BN running stats once)
    for i, m in enumerate(maps):
        d["maps_train%d" % i] = m.detach().clone()
    # restore buffers, then the real training step
    model.load_state_dict({k[len("state/"):]: torch.as_tensor(v) for k, v in d.items() if k.startswith("state/")})
    model.zero_grad()
    out = model(x, labels)
    out["loss"].backward()
    for k in ("loss", "loss_iou", "loss_obj", "loss_cls"):
        d["out/" + k] = out[k].detach()
    d["out/proportion"] = float(out["proportion"])
    d["boundary_gap"] = min([c["gap"] for c in _calls] + [float("inf")])
    nograd = []
    for n, p in model.named_parameters():
        if p.grad is None:
            nograd.append(n)
        else:
            d["grad/" + n] = p.grad.clone()
    d["nograd_names"] = np.asarray(nograd)
    for k, v in model.state_dict().items():
        if "running" in k or "num_batches" in k:
            d["state_after/" + k] = v.clone()
    # eval branch (uses the running stats after that one step)
    model.eval()
    with torch.no_grad():
        d["eval_out"] = model(x, labels).clone()
        for i, m in enumerate(model(x)):
            d["maps_eval%d" % i] = m.clone()
    print("network fixture: loss=%.6f gap=%.3g nograd=%d" % (float(out["loss"]), d["boundary_gap"], len(nograd)))
    save("network_yolox_test", d)

    # one full SGD(momentum) x2 + EMA + LR schedule trajectory on the same model (a25)
    model.load_state_dict({k[len("state/"):]: torch.as_tensor(v) for k, v in d.items() if k.startswith("state/")})
    model.train()
    opt = torch.optim.SGD(model.parameters(), lr=0.01, momentum=0.9)
    sched = CosineWarmupScheduler(opt, warmup=0.1 * 20, max_iters=20)
    ema = ModelEMA(model, 0.9998)
    h = dict(x=x, labels=labels)
    lrs = []
    for step in range(3):
        lrs.append(opt.param_groups[0]["lr"])
        out = model(x, labels)
        opt.zero_grad()
        out["loss"].backward()
        opt.step()
        ema.update(model)
        sched.step()
        h["loss%d" % step] = out["loss"].detach()
    h["lrs"] = np.asarray(lrs)
    for k, v in model.state_dict().items():
        h["final/" + k] = v.clone()
    for k, v in ema.ema.state_dict().items():
        h["ema/" + k] = v.clone()
    save("harness_trajectory", h)


def gen_network_rect():
    """The toy YOLOX on a RECTANGULAR image (64 rows x 96 columns: head maps 8x12 / 4x6 / 2x3), i.e. with the reference's grid quirk
    (yolox_loss.py:198-200: meshgrid(arange(h), arange(w), indexing='xy') re-viewed as (h, w)) inside a whole training step: raw maps,
    losses, every gradient, running statistics, eval output.  Same weights and perturbation as gen_network()."""
    cfg = load_cfg("yolox_test")
    C = 3
    torch.manual_seed(96)
    model = build_model(cfg, C)
    g = torch.Generator().manual_seed(5)
    for n, p in model.named_parameters():
        if n.endswith(".norm.weight"):
            p.data = 0.5 + torch.rand(p.shape, generator=g)
        if n.endswith(".norm.bias"):
            p.data = torch.rand(p.shape, generator=g) - 0.5
    gen = torch.Generator().manual_seed(4321)
    x = torch.rand(2, 3, 64, 96, generator=gen) * 255
    labels = torch.zeros(2, 8, 5)
    labels[0, :3] = torch.tensor([[0, 20.0, 24.0, 18.0, 22.0], [2, 70.0, 40.0, 30.0, 26.0], [1, 84.0, 14.0, 16.0, 12.0]])
    labels[1, :2] = torch.tensor([[1, 48.0, 30.0, 40.0, 36.0], [0, 12.0, 50.0, 14.0, 18.0]])
    d = dict(x=x, labels=labels, num_classes=C)
    state = {k: v.clone() for k, v in model.state_dict().items()}
    # (the state is NOT stored: it is the one of network_yolox_test.npz, asserted here)
    ref = dict(np.load(os.path.join(OUT, "network_yolox_test.npz")))
    assert all(np.array_equal(ref["state/" + k], v.numpy()) for k, v in state.items()) and len(state) == sum(1 for k in ref if k.startswith("state/"))
    model.train()
    _calls.clear()
    for i, m in enumerate(model(x)):
        d["maps_train%d" % i] = m.detach().clone()
    model.load_state_dict(state)
    model.zero_grad()
    out = model(x, labels)
    out["loss"].backward()
    for k in ("loss", "loss_iou", "loss_obj", "loss_cls"):
        d["out/" + k] = out[k].detach()
    d["out/proportion"] = float(out["proportion"])
    d["boundary_gap"] = min([c["gap"] for c in _calls] + [float("inf")])
    for n, p in model.named_parameters():
        if p.grad is not None:
            d["grad/" + n] = p.grad.clone()
    for k, v in model.state_dict().items():
        if "running" in k:
            d["state_after/" + k] = v.clone()
    model.eval()
    with torch.no_grad():
        d["eval_out"] = model(x, labels).clone()
    print("rectangular network fixture: loss=%.6f gap=%.3g" % (float(out["loss"]), d["boundary_gap"]))
    save("network_yolox_rect", d)


def gen_schedule():
    d = {}
    for i, (warm, T) in enumerate([(0.1 * 100, 100), (0.0 + 5, 37), (300, 3000)]):
        p = torch.nn.Parameter(torch.zeros(1))
        opt = torch.optim.SGD([p], lr=0.01, momentum=0.9)
        s = CosineWarmupScheduler(opt, warmup=warm, max_iters=T)
        fac = [s.get_lr_factor(t) for t in range(T + 1)]
        d["sched%d_warm" % i] = warm
        d["sched%d_T" % i] = T
        d["sched%d_factor" % i] = np.asarray(fac, dtype=np.float64)
    save("lr_schedule", d)


def gen_network_v7():
    """YOLOv7 family (eelan + yolov7neck + implicit_head + yolov7 loss): tiny config."""
    with open(os.path.join(ROOT, "configs", "model", "yolov7", "yolov7_test.yaml")) as f:
        cfg = yaml.safe_load(f)
    C = 3
    torch.manual_seed(97)
    model = build_model(cfg, C)
    g = torch.Generator().manual_seed(6)
    for n, p in model.named_parameters():
        if n.endswith(".norm.weight"):
            p.data = 0.5 + torch.rand(p.shape, generator=g)
        if n.endswith(".norm.bias"):
            p.data = torch.rand(p.shape, generator=g) - 0.5
        if ".im." in n:   # ImplicitM is N(0, .02) upstream: make the head outputs non-trivial
            p.data = 1.0 + 0.2 * torch.randn(p.shape, generator=g)
    gen = torch.Generator().manual_seed(4321)
    x = torch.rand(2, 3, 64, 64, generator=gen) * 255
    labels = torch.zeros(2, 8, 5)
    labels[0, :3] = torch.tensor([[0, 20.0, 24.0, 18.0, 22.0], [2, 40.0, 40.0, 30.0, 26.0], [1, 50.0, 14.0, 16.0, 12.0]])
    labels[1, :2] = torch.tensor([[1, 30.0, 30.0, 40.0, 36.0], [0, 12.0, 50.0, 14.0, 18.0]])
    d = dict(x=x, labels=labels, num_classes=C)
    for k, v in model.state_dict().items():
        d["state/" + k] = v.clone()
    model.train()
    maps = model(x)
    rs = [torch.randn(m.shape, generator=gen) for m in maps]
    model.zero_grad()
    sum((m * r).sum() for m, r in zip(maps, rs)).backward()
    for i, (m, r) in enumerate(zip(maps, rs)):
        d["maps_train%d" % i] = m.detach().clone()
        d["r%d" % i] = r
    for n, p in model.named_parameters():
        assert p.grad is not None, n
        d["grad/" + n] = p.grad.clone()
    for k, v in model.state_dict().items():
        if "running" in k or "num_batches" in k:
            d["state_after/" + k] = v.clone()
    # full training loss of the reference (CPU-only as written), for the loss kernels to come
    model.zero_grad()
    out = model(x, labels.clone())
    out["loss"].backward()
    d["out/loss"] = out["loss"].detach()
    for n, p in model.named_parameters():
        if p.grad is not None:
            d["lossgrad/" + n] = p.grad.clone()
    model.eval()
    with torch.no_grad():
        d["eval_out"] = model(x, labels).clone()
    print("yolov7 fixture: loss=%.6f eval %s" % (float(out["loss"]), tuple(d["eval_out"].shape)))
    save("network_yolov7_test", d)


V7_ANCHORS = [[[12, 16], [19, 36], [40, 28]], [[36, 75], [76, 55], [72, 146]], [[142, 110], [192, 243], [459, 401]]]


def _v7_case(name, B, nc, H, W, labels, seed, scale=1.5):
    """Run the reference YOLOv7Loss (train branch) on seeded head maps; record what build_targets
    returned, the loss and d loss / d head maps."""
    from models.losses.yolov7.yolov7_loss import YOLOv7Loss
    strides = [8, 16, 32]
    loss = YOLOv7Loss(nc, strides, V7_ANCHORS)
    loss.train()
    gen = torch.Generator().manual_seed(seed)
    maps = [(torch.randn(B, 3 * (5 + nc), H // s, W // s, generator=gen) * scale).requires_grad_(True) for s in strides]
    rec = {}
    orig = loss.build_targets

    def spy(predictions, targets):
        r = orig(predictions, targets)
        rec["r"] = r
        return r

    loss.build_targets = spy
    out = loss([m for m in maps], labels.clone())
    out["loss"].backward()
    d = dict(labels=labels, num_classes=nc, strides=np.asarray(strides), anchors=np.asarray(V7_ANCHORS), loss=out["loss"].detach())
    for i, m in enumerate(maps):
        d["map%d" % i] = m.detach()
        d["dmap%d" % i] = m.grad
    bs, as_, gjs, gis, tg, an = rec["r"]
    for i in range(3):
        d["m%d_b" % i] = bs[i].long(); d["m%d_a" % i] = as_[i].long(); d["m%d_gj" % i] = gjs[i].long(); d["m%d_gi" % i] = gis[i].long()
        d["m%d_t" % i] = tg[i].reshape(-1, 6) if tg[i].numel() else torch.zeros(0, 6)
        d["m%d_anch" % i] = an[i].reshape(-1, 2) if an[i].numel() else torch.zeros(0, 2)
    print(name, "loss %.6f matched per level" % float(out["loss"]), [int(b.shape[0]) for b in bs])
    save(name, d)


def gen_v7loss_cases():
    # A: hand-placed GTs: overlapping pair in one cell, border boxes (index clamps), a 2x2 px box that
    #    no anchor accepts, an image without GTs
    lab = torch.zeros(3, 10, 5)
    lab[0, :6] = torch.tensor([[0, 40.0, 44.0, 30.0, 36.0], [2, 42.0, 46.0, 26.0, 40.0], [1, 120.0, 30.0, 60.0, 50.0],
                               [4, 4.0, 150.0, 14.0, 18.0], [3, 156.0, 156.0, 20.0, 16.0], [0, 80.0, 80.0, 150.0, 140.0]])
    lab[2, :3] = torch.tensor([[1, 60.0, 100.0, 90.0, 40.0], [2, 100.0, 20.0, 2.0, 2.0], [0, 30.0, 30.0, 12.0, 16.0]])
    _v7_case("v7loss_case_A", 3, 5, 160, 160, lab, 11)
    # B: random, 80 classes, up to 20 GTs
    gen = torch.Generator().manual_seed(5)
    Bn, M, S = 3, 24, 192
    lab = torch.zeros(Bn, M, 5)
    for b, n in enumerate([20, 7, 13]):
        lab[b, :n, 0] = torch.randint(0, 20, (n,), generator=gen).float()
        lab[b, :n, 1:3] = torch.rand(n, 2, generator=gen) * S * 0.9 + S * 0.05
        lab[b, :n, 3:5] = torch.exp(torch.rand(n, 2, generator=gen) * 3.2 + 1.8)
    _v7_case("v7loss_case_B", Bn, 20, S, S, lab, 12)
    # E: near-zero logits (decoded boxes = anchors at cell centres) so IoUs are large and dynamic k > 1
    _v7_case("v7loss_case_E", Bn, 20, S, S, lab, 15, scale=0.3)
    # C: no GT anywhere (objectness negatives only)
    _v7_case("v7loss_case_C", 2, 3, 64, 64, torch.zeros(2, 4, 5), 13)
    # D: non-square input, confident (large-magnitude) logits
    lab = torch.zeros(2, 6, 5)
    lab[0, :4] = torch.tensor([[1, 30.0, 60.0, 40.0, 30.0], [0, 170.0, 100.0, 36.0, 80.0], [2, 96.0, 64.0, 100.0, 90.0], [1, 185.0, 10.0, 20.0, 16.0]])
    lab[1, :2] = torch.tensor([[0, 96.0, 64.0, 180.0, 120.0], [2, 10.0, 120.0, 18.0, 14.0]])
    _v7_case("v7loss_case_D", 2, 3, 128, 192, lab, 14, scale=3.0)


def gen_repconv():
    """RepConv blocks (train-time form) straight from the reference class: with and without the identity branch."""
    from models.necks.yolov7_neck import RepConv
    d = {}
    gen = torch.Generator().manual_seed(77)
    for tag, c1, c2 in (("ne", 16, 32), ("id", 24, 24)):
        torch.manual_seed(5)
        m = RepConv(c1, c2, 3, 1)
        for n, p in m.named_parameters():
            if n.endswith("1.weight") or n == "rbr_identity.weight":
                p.data = 0.5 + torch.rand(p.shape, generator=gen)
            if n.endswith("1.bias") or n == "rbr_identity.bias":
                p.data = torch.rand(p.shape, generator=gen) - 0.5
        x = torch.randn(2, c1, 12, 10, generator=gen).requires_grad_(True)
        for k, v in m.state_dict().items():
            d["%s/state/%s" % (tag, k)] = v.clone()
        m.train()
        y = m(x)
        r = torch.randn(y.shape, generator=gen)
        (y * r).sum().backward()
        d["%s/x" % tag], d["%s/y" % tag], d["%s/r" % tag], d["%s/dx" % tag] = x.detach(), y.detach(), r, x.grad
        for n, p in m.named_parameters():
            d["%s/grad/%s" % (tag, n)] = p.grad.clone()
        for k, v in m.state_dict().items():
            if "running" in k or "num_batches" in k:
                d["%s/state_after/%s" % (tag, k)] = v.clone()
    save("repconv_blocks", d)


def gen_deploy():
    """Deploy-time folding: the reference's RepConv re-parameterisation (get_equivalent_kernel_bias,
    fuse_repvgg_block) and BaseConv BN folding (RepConv.fuse_conv_bn + BaseConv.fuseforward) on seeded modules in eval mode
    with non-trivial running statistics: inputs, unfused eval outputs, fused kernels / biases, fused outputs."""
    from models.necks.yolov7_neck import RepConv
    d = {}
    gen = torch.Generator().manual_seed(321)

    def perturb_bn(bn):
        bn.weight.data = 0.5 + torch.rand(bn.weight.shape, generator=gen)
        bn.bias.data = torch.rand(bn.bias.shape, generator=gen) - 0.5
        bn.running_mean.data = torch.randn(bn.running_mean.shape, generator=gen) * 0.3
        bn.running_var.data = 0.3 + torch.rand(bn.running_var.shape, generator=gen)

    for tag, c1, c2 in (("ne", 16, 32), ("id", 24, 24)):
        torch.manual_seed(7)
        m = RepConv(c1, c2, 3, 1)
        for bn in [m.rbr_dense[1], m.rbr_1x1[1]] + ([m.rbr_identity] if m.rbr_identity is not None else []):
            perturb_bn(bn)
        m.eval()
        x = torch.randn(2, c1, 12, 10, generator=gen)
        for k, v in m.state_dict().items():
            d["rep_%s/state/%s" % (tag, k)] = v.clone()
        with torch.no_grad():
            d["rep_%s/x" % tag] = x
            d["rep_%s/y_eval" % tag] = m(x).clone()
            k, b = m.get_equivalent_kernel_bias()
            d["rep_%s/kernel" % tag], d["rep_%s/bias" % tag] = k.clone(), b.clone()
            m.fuse_repvgg_block()
            d["rep_%s/reparam_weight" % tag] = m.rbr_reparam.weight.detach().clone()
            d["rep_%s/reparam_bias" % tag] = m.rbr_reparam.bias.detach().clone()
            d["rep_%s/y_fused" % tag] = m(x).clone()
    helper = RepConv(8, 8, 3, 1)
    for tag, cin, cout, k, s in (("k3", 8, 16, 3, 1), ("k1", 16, 24, 1, 1), ("k3s2", 8, 16, 3, 2)):
        torch.manual_seed(11)
        m = BaseConv(cin, cout, k, s)
        perturb_bn(m.norm)
        m.eval()
        x = torch.randn(2, cin, 12, 10, generator=gen)
        for kk, v in m.state_dict().items():
            d["base_%s/state/%s" % (tag, kk)] = v.clone()
        with torch.no_grad():
            d["base_%s/x" % tag] = x
            d["base_%s/y_eval" % tag] = m(x).clone()
            m.conv = helper.fuse_conv_bn(m.conv, m.norm)
            d["base_%s/fused_weight" % tag] = m.conv.weight.detach().clone()
            d["base_%s/fused_bias" % tag] = m.conv.bias.detach().clone()
            d["base_%s/y_fused" % tag] = m.fuseforward(x).clone()
        d["base_%s/ksize" % tag], d["base_%s/stride" % tag] = k, s
    for tag in ("ne", "id"):
        print("repconv", tag, "fused vs unfused max diff %.3g" % float((d["rep_%s/y_eval" % tag] - d["rep_%s/y_fused" % tag]).abs().max()))
    save("deploy_fold", d)


def gen_network_e():
    """The e-yolox family (ecmnet + al_pafpn: depthwise 3x3, BatchNorm-free 1x1, bicubic upsampling) through the reference:
    toy width, one training step (losses + every gradient + running statistics), raw maps, eval output."""
    with open(os.path.join(ROOT, "configs", "model", "e-yolox", "e-yolox_test.yaml")) as f:
        cfg = yaml.safe_load(f)
    C = 3
    torch.manual_seed(96)
    model = build_model(cfg, C)
    g = torch.Generator().manual_seed(5)
    for n, p in model.named_parameters():
        if n.endswith(".norm.weight"):
            p.data = 0.5 + torch.rand(p.shape, generator=g)
        if n.endswith(".norm.bias"):
            p.data = torch.rand(p.shape, generator=g) - 0.5
    gen = torch.Generator().manual_seed(1234)
    x = torch.rand(2, 3, 96, 96, generator=gen) * 255
    labels = torch.zeros(2, 8, 5)
    labels[0, :3] = torch.tensor([[0, 30.0, 36.0, 28.0, 32.0], [2, 60.0, 60.0, 44.0, 40.0], [1, 76.0, 20.0, 24.0, 18.0]])
    labels[1, :2] = torch.tensor([[1, 44.0, 44.0, 60.0, 54.0], [0, 18.0, 76.0, 20.0, 28.0]])
    d = dict(x=x, labels=labels, num_classes=C)
    for k, v in model.state_dict().items():
        d["state/" + k] = v.clone()
    model.train()
    maps = model(x)
    for i, m in enumerate(maps):
        d["maps_train%d" % i] = m.detach().clone()
    model.load_state_dict({k[len("state/"):]: torch.as_tensor(v) for k, v in d.items() if k.startswith("state/")})
    model.zero_grad()
    _calls.clear()
    out = model(x, labels)
    out["loss"].backward()
    for k in ("loss", "loss_iou", "loss_obj", "loss_cls"):
        d["out/" + k] = out[k].detach()
    d["out/proportion"] = float(out["proportion"])
    d["boundary_gap"] = min([c["gap"] for c in _calls] + [float("inf")])
    for n, p in model.named_parameters():
        if p.grad is not None:
            d["grad/" + n] = p.grad.clone()
    for k, v in model.state_dict().items():
        if "running" in k or "num_batches" in k:
            d["state_after/" + k] = v.clone()
    model.eval()
    with torch.no_grad():
        d["eval_out"] = model(x, labels).clone()
    print("e-yolox fixture: loss=%.6f gap=%.3g params=%d" % (float(out["loss"]), d["boundary_gap"], len(list(model.parameters()))))
    save("network_eyolox_test", d)


def gen_format_outputs():
    """`format_outputs` of the reference (models/evaluators/postprocess.py:95-138) on seeded detections.  The module's first
    line imports torchvision (not installable here), so only the FUNCTION is taken from the reference file: its source is
    parsed and executed with numpy / torch / the reference's own xyxy2xywh in scope -- the reference's code, no stand-in library."""
    import ast
    from models.utils.bbox import xyxy2xywh
    path = os.path.join(REF, "models", "evaluators", "postprocess.py")
    tree = ast.parse(open(path).read())
    fn = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "format_outputs"][0]
    scope = {"np": np, "torch": torch, "xyxy2xywh": xyxy2xywh}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), path, "exec"), scope)
    ref_format = scope["format_outputs"]
    gen = torch.Generator().manual_seed(77)
    class_ids = [1, 2, 3, 5, 8, 13, 21]
    outs, d = [], {}
    for i, n in enumerate((5, 0, 17, 1)):
        if n == 0:
            outs.append(None)
            continue
        xy = torch.rand(n, 2, generator=gen) * 300
        wh = 5 + torch.rand(n, 2, generator=gen) * 200
        o = torch.cat([xy, xy + wh, torch.rand(n, 1, generator=gen), torch.randint(0, len(class_ids), (n, 1), generator=gen).float()], 1)
        outs.append(o)
        d["in%d" % i] = o.clone()
    ids = [42, 7, 99, 1000]
    hws = ([480, 375, 427, 640], [640, 500, 640, 480])
    val_size = (416, 416)
    d["ids"], d["hs"], d["ws"], d["val_size"], d["class_ids"] = np.asarray(ids), np.asarray(hws[0]), np.asarray(hws[1]), np.asarray(val_size), np.asarray(class_ids)
    json_list, det_list = ref_format(outs, ids, hws, val_size, class_ids, None)
    d["json_image_id"] = np.asarray([j["image_id"] for j in json_list])
    d["json_category_id"] = np.asarray([j["category_id"] for j in json_list])
    d["json_bbox"] = np.asarray([j["bbox"] for j in json_list], dtype=np.float64)
    d["json_score"] = np.asarray([j["score"] for j in json_list], dtype=np.float64)
    for i, o in enumerate(outs):
        if o is not None:
            d["after%d" % i] = o.clone()                  # boxes rescaled in place
        for c in range(len(class_ids)):
            d["det_%d_%d" % (i, c)] = np.asarray(det_list[i][c], dtype=np.float64).reshape(-1, 5)
    print("format_outputs fixture: %d records" % len(json_list))
    save("format_outputs", d)


def gen_network_warm():
    """"Warm weights": the toy YOLOX after 50 SGD steps of the REFERENCE on a small fixed data set, then one
    recorded training step at that state (losses, every gradient, head maps).  A trained-for-a-while BatchNorm net
    is far less chaotic than a random-initialised one, so the bf16 path can be held to tight end-to-end bounds on it."""
    cfg = load_cfg("yolox_test")
    C = 3
    torch.manual_seed(96)
    model = build_model(cfg, C)
    model.train()
    gen = torch.Generator().manual_seed(4242)
    S, B = 128, 4
    data = []
    for i in range(4):
        x = torch.rand(B, 3, S, S, generator=gen) * 255
        labels = rand_labels(gen, [3, 4, 2, 3], C, S, 8, min_wh=12.0)
        data.append((x, labels))
    opt = torch.optim.SGD(model.parameters(), lr=0.01, momentum=0.9)
    losses = []
    for step in range(50):
        x, labels = data[step % 4]
        out = model(x, labels)
        opt.zero_grad()
        out["loss"].backward()
        opt.step()
        losses.append(float(out["loss"]))
    x, labels = data[0]
    d = dict(x=x, labels=labels, num_classes=C, warm_losses=np.asarray(losses))
    for k, v in model.state_dict().items():
        d["state/" + k] = v.clone()
    maps = model(x)
    for i, m in enumerate(maps):
        d["maps_train%d" % i] = m.detach().clone()
    model.load_state_dict({k[len("state/"):]: torch.as_tensor(v) for k, v in d.items() if k.startswith("state/")})
    model.zero_grad()
    _calls.clear()
    out = model(x, labels)
    out["loss"].backward()
    for k in ("loss", "loss_iou", "loss_obj", "loss_cls"):
        d["out/" + k] = out[k].detach()
    d["out/proportion"] = float(out["proportion"])
    d["boundary_gap"] = min([c["gap"] for c in _calls] + [float("inf")])
    for n, p in model.named_parameters():
        if p.grad is not None:
            d["grad/" + n] = p.grad.clone()
    print("warm fixture: loss %.4f -> %.4f over 50 steps; recorded step loss=%.6f gap=%.3g" % (losses[0], losses[-1], float(out["loss"]), d["boundary_gap"]))
    save("network_yolox_warm", d)


# gradients of the warm yolox_s fixture stored in full besides every tensor up to 40k elements: the 128 -> 128 3x3 layers the
# benchmark spends its time in (backbone / neck / head, 20x20 and 10x10 maps at 160x160), a stride-2 layer, a wide 1x1
_WARM_S_FULL = ("backbone.stage3.1.m.0.conv2.conv.weight", "backbone.stage2.0.conv.weight", "backbone.stage3.0.conv.weight",
                "neck.p5_p4.m.0.conv2.conv.weight", "neck.n3_n4.m.0.conv2.conv.weight", "head.cls_convs.0.0.conv.weight",
                "head.cls_convs.0.1.conv.weight", "head.reg_convs.0.1.conv.weight", "head.reg_convs.1.0.conv.weight",
                "backbone.stage3.1.conv3.conv.weight")


def gen_network_warm_s():
    """The benchmarked network itself, warm: yolox_s.yaml (80 classes) after 50 SGD steps of the REFERENCE at 160x160, batch 2, then
    one recorded training step.  The state is stored as bf16 bit patterns (the 50-step weights are rounded to bf16 FIRST and the
    recorded step runs FROM the rounded state, so the stored inputs are exact); gradients: every tensor up to 40k elements and the
    _WARM_S_FULL layers in full, the L2 norm of all of them.  Covers the tile shapes the benchmark runs (128-channel blocks, 32-channel chunks, 8-row
    tiles, 64x64 weight-gradient slabs) with a tight end-to-end bf16 check, which the 8-channel toy net cannot."""
    cfg = load_cfg("yolox_s")
    C = 80
    torch.manual_seed(96)
    model = build_model(cfg, C)
    model.train()
    gen = torch.Generator().manual_seed(777)
    S, B = 160, 2
    data = []
    for i in range(4):
        x = torch.rand(B, 3, S, S, generator=gen) * 255
        labels = rand_labels(gen, [3, 4], C, S, 8, min_wh=12.0)
        data.append((x, labels))
    opt = torch.optim.SGD(model.parameters(), lr=0.01, momentum=0.9)
    losses = []
    for step in range(50):
        x, labels = data[step % 4]
        out = model(x, labels)
        opt.zero_grad()
        out["loss"].backward()
        opt.step()
        losses.append(float(out["loss"]))
    # round the warm state to bf16 and continue FROM the rounded state
    sd = {}
    for k, v in model.state_dict().items():
        sd[k] = v.to(torch.bfloat16).float() if v.dtype.is_floating_point else v.clone()
    model.load_state_dict(sd)
    # the recorded batch: SimOTA is discrete, and the bf16 path moves the head logits by ~1e-3 -- pick a batch whose assignment
    # is as far from a knife's edge as 300 candidate batches offer (relative cost gap at every k-th boundary, distance of every
    # dynamic-k sum from an integer), so that
    # the end-to-end bf16 comparison measures the kernels and not a flipped anchor
    x = labels = None
    best = (-1.0, None)
    for seed in range(1000, 1300):
        g2 = torch.Generator().manual_seed(seed)
        xc = torch.rand(B, 3, S, S, generator=g2) * 255
        lc = rand_labels(g2, [2, 3], C, S, 8, min_wh=12.0)
        model.load_state_dict(sd)
        _calls.clear()
        with torch.no_grad():
            model(xc, lc)
        gap = min([c["gap"] for c in _calls] + [float("inf")])
        km = min([c["kmargin"] for c in _calls] + [float("inf")])
        score = min(gap / 5e-3, km / 0.05)
        if score > best[0]:
            best = (score, (seed, gap, km, xc, lc))
        if score >= 2.0:
            break
    seed, gap, km, x, labels = best[1]
    print("recorded batch: seed %d, boundary gap %.3g, dynamic-k margin %.3g" % (seed, gap, km))
    assert x is not None
    model.load_state_dict(sd)
    d = dict(x=x, labels=labels, num_classes=C, warm_losses=np.asarray(losses))
    for k, v in sd.items():
        if v.dtype.is_floating_point:
            d["state16/" + k] = v.to(torch.bfloat16).view(torch.int16).numpy().copy()
        else:
            d["state/" + k] = v.clone()
    maps = model(x)
    for i, m in enumerate(maps):
        d["maps_train%d" % i] = m.detach().clone()
    model.load_state_dict(sd)          # the labels=None forward above moved the running statistics
    model.zero_grad()
    _calls.clear()
    out = model(x, labels)
    out["loss"].backward()
    for k in ("loss", "loss_iou", "loss_obj", "loss_cls"):
        d["out/" + k] = out[k].detach()
    d["out/proportion"] = float(out["proportion"])
    d["boundary_gap"] = min([c["gap"] for c in _calls] + [float("inf")])
    d["k_margin"] = min([c["kmargin"] for c in _calls] + [float("inf")])
    full = 0
    for n, p in model.named_parameters():
        if p.grad is not None:
            d["gnorm/" + n] = float(p.grad.double().norm())
            if p.numel() <= 40000 or n in _WARM_S_FULL:
                d["grad/" + n] = p.grad.clone()
                full += p.numel()
    print("warm yolox_s fixture: loss %.4f -> %.4f over 50 steps; recorded step loss=%.6f gap=%.3g; %d gradient elements in full"
          % (losses[0], losses[-1], float(out["loss"]), d["boundary_gap"], full))
    save("network_yolox_s_warm", d)


def gen_cfg1():
    """BASELINE.json configs[0]: "YOLOX-nano" (yolox_s.yaml at width 0.25) 416x416 batch 4 through the
    REFERENCE on the benchmark's synthetic batch: loss scalars + a handful of gradients + weight checksums."""
    cfg = load_cfg("yolox_nano")
    C = 80
    torch.manual_seed(96)
    model = build_model(cfg, C)
    model.train()
    g = torch.Generator().manual_seed(1234)
    B, S, num_gt, max_gt = 4, 416, 30, 100
    imgs = torch.rand(B, 3, S, S, generator=g) * 255
    labels = torch.zeros(B, max_gt, 5)
    labels[:, :num_gt, 0] = torch.randint(0, C, (B, num_gt), generator=g).float()
    labels[:, :num_gt, 1:3] = (0.15 + 0.7 * torch.rand(B, num_gt, 2, generator=g)) * S
    labels[:, :num_gt, 3:5] = 8 + torch.rand(B, num_gt, 2, generator=g) * 0.3 * S
    d = dict(batch=B, size=S, num_classes=C, seed_weights=96, seed_data=1234)
    d["param_sum"] = float(sum(p.double().sum() for p in model.parameters()))
    d["param_abs_sum"] = float(sum(p.double().abs().sum() for p in model.parameters()))
    d["stem_weight"] = model.backbone.stem.conv.conv.weight.detach().clone()
    _calls.clear()
    out = model(imgs, labels)
    out["loss"].backward()
    for k in ("loss", "loss_iou", "loss_obj", "loss_cls"):
        d["out/" + k] = out[k].detach()
    d["out/proportion"] = float(out["proportion"])
    d["boundary_gap"] = min([c["gap"] for c in _calls] + [float("inf")])
    d["num_fg"] = sum(c["num_fg"] for c in _calls)
    for n in ("backbone.stem.conv.conv.weight", "backbone.stage4.1.conv1.norm.weight", "neck.p4_p3.conv3.conv.weight",
              "head.cls_preds.0.bias", "head.obj_preds.2.bias", "head.reg_preds.1.weight"):
        d["grad/" + n] = dict(model.named_parameters())[n].grad.clone()
    d["grad_sq_sum"] = float(sum((p.grad.double() ** 2).sum() for p in model.parameters() if p.grad is not None))
    print("cfg1 fixture: loss=%.6f num_fg=%d gap=%.3g" % (float(out["loss"]), d["num_fg"], d["boundary_gap"]))
    save("cfg1_nano416", d)


def gen_wide(which=("yolox_l", "yolox_x", "yolov7")):
    """BASELINE.json configs[2..4] at their FULL width (yolox_l.yaml, yolox_x.yaml, yolov7.yaml unchanged) through the REFERENCE on
    a small synthetic batch (128x128, batch 2): seeded initialisation (manual_seed(96), checked by the first convolution's weights
    and a parameter checksum), the raw head maps of the labels=None path, the loss scalars, ~20 whole gradients (every prediction
    bias, one convolution / BatchNorm per stage, small enough to store) and the L2 norm of EVERY parameter's gradient."""
    for name in which:
        fam = "yolov7" if name.startswith("yolov7") else "yolox"
        with open(os.path.join(ROOT, "configs", "model", fam, name + ".yaml")) as f:
            cfg = yaml.safe_load(f)
        C = 80
        torch.manual_seed(96)
        model = build_model(cfg, C)
        model.train()
        B, S, num_gt, max_gt = 2, 128, 6, 20

        def batch(seed):
            g = torch.Generator().manual_seed(seed)
            imgs = torch.rand(B, 3, S, S, generator=g) * 255
            labels = torch.zeros(B, max_gt, 5)
            labels[:, :num_gt, 0] = torch.randint(0, C, (B, num_gt), generator=g).float()
            labels[:, :num_gt, 1:3] = (0.15 + 0.7 * torch.rand(B, num_gt, 2, generator=g)) * S
            labels[:, :num_gt, 3:5] = 8 + torch.rand(B, num_gt, 2, generator=g) * 0.3 * S
            return imgs, labels

        seed = 4321
        if fam == "yolox":
            # SimOTA is discrete: among a few candidate batches keep the one whose k-th cost boundary is widest (as network_yolox_s_warm does),
            # so that an fp32 re-implementation cannot legitimately flip an assignment
            sd_init = {k: v.clone() for k, v in model.state_dict().items()}
            best = (-1.0, seed)
            for cand in range(4321, 4333):
                model.load_state_dict(sd_init)
                _calls.clear()
                with torch.no_grad():
                    model(*batch(cand))
                gap = min([c["gap"] for c in _calls] + [float("inf")])
                best = max(best, (gap, cand))
            model.load_state_dict(sd_init)
            seed = best[1]
            print("%s: data seed %d (k-th cost boundary gap %.4g)" % (name, seed, best[0]))
        imgs, labels = batch(seed)
        d = dict(batch=B, size=S, num_classes=C, seed_weights=96, seed_data=seed, num_gt=num_gt, max_gt=max_gt)
        params = list(model.named_parameters())
        d["param_sum"] = float(sum(p.double().sum() for _, p in params))
        d["param_abs_sum"] = float(sum(p.double().abs().sum() for _, p in params))
        d["first_weight"] = params[0][1].detach().clone()
        d["n_params"] = len(params)
        sd0 = {k: v.clone() for k, v in model.state_dict().items()}
        with torch.no_grad():
            maps = model(imgs, None)          # raw NCHW head maps (train-mode BatchNorm: batch statistics)
        for i, m in enumerate(maps):
            d["maps/%d" % i] = m.detach().clone()
        model.load_state_dict(sd0)             # (the running statistics moved; the trained step starts from the seeded state)
        _calls.clear()
        out = model(imgs, labels)
        loss = out["loss"].sum()
        loss.backward()
        for k, v in out.items():
            d["out/" + k] = v.detach().clone().reshape(-1) if torch.is_tensor(v) else float(v)
        if fam == "yolox":
            d["boundary_gap"] = min([c["gap"] for c in _calls] + [float("inf")])
            d["num_fg"] = sum(c["num_fg"] for c in _calls)
        names = [n for n, _ in params]
        pick = [n for n in names if n.startswith("head") and (n.endswith(".bias") and "norm" not in n or n.endswith(".implicit"))]
        small = [n for n, p in params if p.dim() == 4 and p.numel() <= 120000 and n not in pick]
        pick += [small[i] for i in sorted({int(round(j * (len(small) - 1) / 5.0)) for j in range(6)})]
        norms = [n for n, p in params if p.dim() == 1 and n.endswith("norm.weight")]
        pick += [norms[i] for i in sorted({0, len(norms) // 3, 2 * len(norms) // 3, len(norms) - 1})]
        for n in pick:
            d["grad/" + n] = dict(params)[n].grad.clone()
        d["grad_names"] = np.array(names)
        d["grad_norms"] = np.array([float(p.grad.double().norm()) if p.grad is not None else -1.0 for _, p in params])
        d["grad_sq_sum"] = float(sum((p.grad.double() ** 2).sum() for _, p in params if p.grad is not None))
        print("%s fixture: loss=%.6f  %d stored gradients, %d parameters with a gradient" %
              (name, float(loss), len(pick), int((d["grad_norms"] >= 0).sum())), {k: v for k, v in d.items() if k in ("num_fg", "boundary_gap")})
        save("wide_" + name, d)



def toy_detection_dataset(seed=5, n=7, size=(48, 64)):
    """A tiny in-memory data set in the shape MosaicDetection reads (cocoDataset: annotations / imgs / img_size): images of
    assorted sizes and aspect ratios, one of them without labels."""
    rng = np.random.RandomState(seed)

    class Toy:
        pass
    ds = Toy()
    ds.img_size = size
    ds.imgs, ds.annotations = [], []
    for i in range(n):
        h, w = int(rng.randint(24, 80)), int(rng.randint(24, 96))
        ds.imgs.append(rng.randint(0, 256, (h, w, 3)).astype(np.uint8))
        k = 0 if i == 3 else int(rng.randint(1, 5))
        x1 = rng.uniform(0, w * 0.6, k); y1 = rng.uniform(0, h * 0.6, k)
        bw = rng.uniform(4, w * 0.4, k); bh = rng.uniform(4, h * 0.4, k)
        lab = np.stack([x1, y1, np.minimum(x1 + bw, w), np.minimum(y1 + bh, h), rng.randint(0, 80, k).astype(np.float64)], 1) if k else np.zeros((0, 5))
        ds.annotations.append((lab.astype(np.float64), (h, w), (h, w), "img%d" % i))
    ds.__class__.__len__ = lambda self: len(self.imgs)
    ds.object_cls, ds.back_cls = None, None
    return ds


def _stub_cv2():
    """`import cv2` inside the reference's data modules, served by the oracle's restatements of the OpenCV calls they make."""
    import types
    sys.path.insert(0, ROOT)
    from oracle import augment as oa, mosaic as om
    cv2 = types.ModuleType("cv2")
    cv2.INTER_LINEAR, cv2.COLOR_BGR2HSV, cv2.COLOR_HSV2BGR = 1, 40, 54
    cv2.resize = lambda img, dsize, interpolation=1: om.resize(img, dsize)
    cv2.warpAffine = lambda img, M, dsize=None, borderValue=(0, 0, 0): om.warp_affine_u8(img, M, dsize, borderValue)
    cv2.warpPerspective = lambda img, M, dsize=None, borderValue=(0, 0, 0): om.warp_perspective_u8(img, M, dsize, borderValue)
    cv2.getRotationMatrix2D = lambda angle=0, center=(0, 0), scale=1: om.get_rotation_matrix_2d(center, angle, scale)
    cv2.split = lambda a: [a[..., i] for i in range(a.shape[-1])]
    cv2.merge = lambda chans: np.stack(chans, -1)
    cv2.LUT = lambda a, lut: lut[a]

    def cvt(img, code, dst=None):
        if code == cv2.COLOR_BGR2HSV:
            h, s_, v = oa.bgr2hsv_u8(img)
            return np.stack([h, s_, v], -1).astype(np.uint8)
        out = oa.hsv2bgr_u8(img[..., 0].astype(np.int64), img[..., 1].astype(np.int64), img[..., 2].astype(np.int64))
        if dst is not None:
            dst[...] = out
        return out
    cv2.cvtColor = cvt
    sys.modules["cv2"] = cv2


def gen_mosaic():
    """Mosaic / random-affine / mixup samples from the REFERENCE's own MosaicDetection + TrainTransform
    (models/data/mosaic_detection.py, models/data/augmentation/data_augments.py).  Both files `import cv2`, which cannot be
    installed here: the import is served by a module whose resize / warpAffine / getRotationMatrix2D / cvtColor / LUT are the
    oracle's restatements of OpenCV's 8-bit algorithms (oracle/augment.py, oracle/mosaic.py).  What the fixture therefore PINS
    is everything the reference itself decides -- control flow, the order of the draws from `random` / `numpy.random`, label
    arithmetic, padding, blending -- given those pixel primitives; the primitives themselves stay unpinned against cv2."""
    import random
    _stub_cv2()
    from models.data.mosaic_detection import MosaicDetection
    from models.data.augmentation.data_augments import TrainTransform
    d = {}
    cases = [("mix", dict(mosaic_prob=1.0, mixup_prob=1.0), 11), ("nomix", dict(mosaic_prob=1.0, mixup_prob=0.0), 12),
             ("plain", dict(mosaic_prob=0.0, mixup_prob=1.0), 13), ("coin", dict(mosaic_prob=0.5, mixup_prob=0.5), 14),
             # perspective != 0: the reference switches to cv2.warpPerspective with the same affine matrix (:319-327)
             ("persp", dict(mosaic_prob=1.0, mixup_prob=0.0, perspective=0.001), 15)]
    for tag, kw, seed in cases:
        ds = toy_detection_dataset()
        md = MosaicDetection(ds, (48, 64), preprocess=TrainTransform(max_labels=20, flip_prob=0.5, hsv_prob=1.0), **kw)
        random.seed(seed)
        np.random.seed(seed)
        idxs = [0, 3, 5, 2]
        for k, idx in enumerate(idxs):
            img, lab, info, ids, name = md[idx]
            d["%s_%d_img" % (tag, k)] = np.asarray(img, dtype=np.float32)
            d["%s_%d_labels" % (tag, k)] = np.asarray(lab, dtype=np.float32)
            d["%s_%d_info" % (tag, k)] = np.asarray(info)
        d["%s_idx" % tag] = np.asarray(idxs)
        d["%s_seed" % tag] = np.asarray(seed)
        d["%s_state" % tag] = np.asarray(random.random())       # the position of the `random` stream after the four samples
    ds = toy_detection_dataset()
    for i, (im, an) in enumerate(zip(ds.imgs, ds.annotations)):
        d["ds_img%d" % i] = im
        d["ds_lab%d" % i] = an[0]
    save("mosaic_samples", d)
    del sys.modules["cv2"]


def gen_cutout():
    """Rounding cut-out (SURVEY 8f rank 3 remainder).  (1) The reference's own `cutout_rounding`
    (models/data/augmentation/cutout_round.py -- numpy only, it runs here unchanged) on seeded images and boxes: interior boxes,
    boxes on every image border, no boxes, many boxes, repeated calls on one numpy-random stream (holes over holes, rejected holes).
    (2) The reference's MosaicDetection with cutoutR_prob > 0 (cv2 served as in gen_mosaic): the mosaic branch, the plain branch, a
    mix of both."""
    import random
    sys.path.insert(0, REF)
    from models.data.augmentation.cutout_round import cutout_rounding
    nhole, ratio, mix, thr = (1, 3), [[0.1, 0.1], [0.3, 0.1], [0.1, 0.3], [0.2, 0.2], [0.3, 0.3]], 0.7, 0.2   # mosaic_detection.py:52-55
    d = {}
    rng = np.random.RandomState(77)

    def boxes(k, h, w, edge=False):
        x1 = rng.uniform(2, w * 0.6, k); y1 = rng.uniform(2, h * 0.6, k)
        bw = rng.uniform(3, w * 0.3, k); bh = rng.uniform(3, h * 0.3, k)
        lab = np.stack([x1, y1, np.minimum(x1 + bw, w - 2), np.minimum(y1 + bh, h - 2), rng.randint(0, 80, k).astype(np.float64)], 1)
        if edge:      # one box on the left / top borders, one on the right / bottom ones
            lab[0, :4] = [0.4, 0.0, w * 0.3, h * 0.25]
            lab[1, :4] = [w * 0.7, h * 0.7, float(w), h - 0.5]
        return lab
    cases = [("interior", 40, 56, boxes(3, 40, 56)), ("edges", 44, 60, boxes(4, 44, 60, edge=True)), ("none", 32, 48, np.zeros((0, 5))),
             ("many", 60, 84, boxes(11, 60, 84)), ("tiny", 9, 13, np.array([[3.2, 2.1, 7.9, 6.5, 1.0]]))]
    for tag, h, w, lab in cases:
        img = rng.randint(0, 256, (h, w, 3)).astype(np.uint8)
        d["f_%s_img" % tag], d["f_%s_labels" % tag] = img, lab
        np.random.seed(1000 + len(tag))
        d["f_%s_seed" % tag] = np.asarray(1000 + len(tag))
        cur = img.copy()
        for rep in range(4):      # four calls on one stream, each on the previous result
            cur = cutout_rounding(cur, lab, nhole, ratio, mix, thr)
            d["f_%s_out%d" % (tag, rep)] = cur.copy()
        d["f_%s_state" % tag] = np.asarray(np.random.randint(0, 1 << 30))
    _stub_cv2()
    from models.data.mosaic_detection import MosaicDetection
    from models.data.augmentation.data_augments import TrainTransform
    for tag, kw, seed in [("cmosaic", dict(mosaic_prob=1.0, mixup_prob=0.0, cutoutR_prob=1.0), 21), ("cplain", dict(mosaic_prob=0.0, cutoutR_prob=1.0), 22),
                          ("ccoin", dict(mosaic_prob=0.5, mixup_prob=0.5, cutoutR_prob=0.6), 23)]:
        ds = toy_detection_dataset()
        md = MosaicDetection(ds, (48, 64), preprocess=TrainTransform(max_labels=20, flip_prob=0.5, hsv_prob=1.0), **kw)
        random.seed(seed)
        np.random.seed(seed)
        idxs = [0, 3, 5, 2]
        for k, idx in enumerate(idxs):
            img, lab, info, ids, name = md[idx]
            d["%s_%d_img" % (tag, k)] = np.asarray(img, dtype=np.float32)
            d["%s_%d_labels" % (tag, k)] = np.asarray(lab, dtype=np.float32)
        d["%s_idx" % tag] = np.asarray(idxs)
        d["%s_seed" % tag] = np.asarray(seed)
        d["%s_state" % tag] = np.asarray([random.random(), float(np.random.randint(0, 1 << 30))])
    save("cutout_round", d)
    del sys.modules["cv2"]


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "cutout":
        gen_cutout()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "mosaic":
        gen_mosaic()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "format":
        gen_format_outputs()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "eyolox":
        gen_network_e()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "deploy":
        gen_deploy()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "warm_s":
        gen_network_warm_s()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "warm":
        gen_network_warm()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "cfg1":
        gen_cfg1()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "wide":
        gen_wide(tuple(sys.argv[2:]) or ("yolox_l", "yolox_x", "yolov7"))
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "repconv":
        gen_repconv()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "rect":
        gen_network_rect()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "loss_side":
        gen_loss_side_effects()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "loss_l1":
        gen_loss_cases(only=("F", "G"))
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "v7loss":
        gen_v7loss_cases()
        sys.exit(0)
    gen_v7loss_cases()
    gen_repconv()
    gen_network_v7()
    gen_loss_cases()
    gen_loss_side_effects()
    gen_blocks()
    gen_network()
    gen_network_rect()
    gen_network_warm()
    gen_network_e()
    gen_format_outputs()
    gen_deploy()
    gen_cfg1()
    gen_wide()
    gen_mosaic()
    gen_cutout()
    gen_schedule()

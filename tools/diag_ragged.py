"""Ragged column blocks inside the real graphs (GPU box): raw head maps of randomly initialised yolox_m / _x / _s / _l in TRAINING mode with PLYOLO_RAG=0
against PLYOLO_RAG=1, against a control that regroups the BatchNorm statistics the same way through a long-tested path (64-channel WHOLE blocks
forced), and the EVAL predictions (no batch statistics).  Finding: eval bit-identical; in training mode any regrouping of the statistics' fp32
partials -- ragged or not -- is amplified by a random 100-layer network to 2 ... 60 % at its outputs (warm networks track).
    python tools/diag_ragged.py"""
import os, sys, torch, yaml, numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests")
import pl_yolo_amd
def run(env, name, size=128, B=2, train=True):
    for k in ("PLYOLO_RAG", "PLYOLO_FORCE_BN"): os.environ.pop(k, None)
    os.environ.update(env)
    cfg = yaml.safe_load(open("configs/model/yolox/%s.yaml" % name))
    torch.manual_seed(96)
    m = pl_yolo_amd.build_model(cfg, 80); m.compute_dtype = "bf16"; m = m.to("cuda")
    m.train() if train else m.eval()
    gen = torch.Generator().manual_seed(4321)
    imgs = (torch.rand(B, 3, size, size, generator=gen) * 255).cuda()
    with torch.no_grad():
        if train: maps = [t.float().cpu() for t in m(imgs, None)]
        else: maps = [m(imgs, torch.zeros(B, 1, 5, device="cuda")).float().cpu()]
    return maps
rel = lambda a, b: [round(float((x - y).abs().max() / x.abs().max()), 5) for x, y in zip(a, b)]
for name in ("yolox_m", "yolox_x", "yolox_s", "yolox_l"):
    size, B = 256, 4
    a = run({"PLYOLO_RAG": "0"}, name, size, B); b = run({"PLYOLO_RAG": "1"}, name, size, B)
    f = run({"PLYOLO_RAG": "0", "PLYOLO_FORCE_BN": "64"}, name, size, B)
    print(name, "train  RAG0 vs RAG1", rel(a, b), " RAG0 vs RAG0 with 64-channel blocks forced (whole blocks, same outputs per convolution, another grouping of the statistics' fp32 partials)", rel(a, f))
    a = run({"PLYOLO_RAG": "0"}, name, size, B, False); b = run({"PLYOLO_RAG": "1"}, name, size, B, False)
    print(name, "eval   RAG0 vs RAG1", rel(a, b), "bit-identical:", all(torch.equal(x, y) for x, y in zip(a, b)))

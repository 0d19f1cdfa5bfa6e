#!/usr/bin/env python3
"""The training step issued from a USER-CREATED stream (plus a few idle streams the application may own): does the step time
survive?  (GPU box)   env: GPU_MAX_HW_QUEUES, PLYOLO_OWN_MAIN"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pl_yolo_amd  # noqa: E402  (sets the environment defaults before the HIP runtime starts)
import torch, yaml  # noqa: E402
import bench  # noqa: E402
dev = torch.device("cuda:0")
cfg = yaml.safe_load(open(os.path.join(ROOT, "configs", "model", "yolox", "yolox_s.yaml")))
torch.manual_seed(96)
model = pl_yolo_amd.build_model(cfg, 80); model.compute_dtype = "bf16"; model = model.to(dev).train()
imgs, labels = bench.synthetic(32, 640, 80, 1234)
imgs, labels = imgs.to(dev), labels.to(dev)
extra = [torch.cuda.Stream() for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 2)]
for st in extra:                       # make them real (a queue is bound on first use)
    with torch.cuda.stream(st):
        torch.zeros(16, device=dev).add_(1)
main = extra[0] if extra else torch.cuda.current_stream()
torch.cuda.synchronize()
with torch.cuda.stream(main):
    def step():
        out = model(imgs, labels); model.zero_grad(set_to_none=True); out["loss"].backward(); return out
    for _ in range(5): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(40): step()
    torch.cuda.synchronize()
print("queues %s own_main %s extra streams %d (step on %s): %.3f ms/step" % (os.environ.get("GPU_MAX_HW_QUEUES"), os.environ.get("PLYOLO_OWN_MAIN"), len(extra),
      "a user stream" if extra else "the default stream", (time.perf_counter() - t0) / 40 * 1e3))

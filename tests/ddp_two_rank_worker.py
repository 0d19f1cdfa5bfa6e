"""Worker of tests/test_gpu_ddp.py::test_two_processes_one_gpu_gloo: one of two ranks that share cuda:0 and exchange their
gradients through the data-parallel backward plan (bucketed host hooks) over the gloo backend.  Not collected by pytest."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import pl_yolo_amd  # noqa: E402
from pl_yolo_amd import ddp  # noqa: E402
from oracle import detector as odet  # noqa: E402


def main():
    out_path, steps = sys.argv[1], int(sys.argv[2])
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    # default: both ranks share cuda:0 and gloo carries the exchange; PLYOLO_TWO_RANK_BACKEND=nccl: one GPU per rank, RCCL over xGMI
    if os.environ.get("PLYOLO_TWO_RANK_BACKEND", "gloo") == "nccl":
        dev = torch.device("cuda", rank)
        torch.cuda.set_device(dev)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
        dev = torch.device("cuda:0")
    with open(os.path.join(ROOT, "configs", "model", "yolox", "yolox_test.yaml")) as f:
        cfg = yaml.safe_load(f)
    torch.manual_seed(96 + 7 * rank)          # DIFFERENT initial weights per rank: attach() must bring rank 0's everywhere
    model = pl_yolo_amd.build_model(cfg, 3)
    model.compute_dtype = "bf16"
    model = model.to(dev).train()
    ddp.attach(model)
    imgs, labels = odet.synthetic_batch(2, 64, 3, num_gt=3, max_gt=6, seed=5 + rank)
    imgs, labels = imgs.to(dev), labels.to(dev)
    for _ in range(steps):
        out = model(imgs, labels)
        model.zero_grad(set_to_none=True)
        out["loss"].backward()
    torch.cuda.synchronize()
    sess = [s for k, s in model.runner().sessions.items() if k[4] == "train"][0]
    d = {"loss": np.asarray(float(out["loss"].detach())), "buckets": np.asarray(len(sess.sched.buckets)), "hooks": np.asarray(sess.bwd.hooks())}
    for n, p in model.named_parameters():
        d["w/" + n] = p.detach().float().cpu().numpy()
        if p.grad is not None:
            d["g/" + n] = p.grad.detach().float().cpu().numpy()
    np.savez(out_path, **d)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()

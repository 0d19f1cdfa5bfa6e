import sys, copy, os, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import test_gpu_network as T
import hiputil as hu
seq = [c == "g" for c in sys.argv[1]]
g, model = T._golden_model("bf16")
model.train()
x = torch.from_numpy(g["x"]).to(hu.DEV); labels = torch.from_numpy(g["labels"]).to(hu.DEV)
sd0 = copy.deepcopy(model.state_dict())
res = []
for use_graph in seq:
    model.load_state_dict(sd0)
    model.runner().use_graph = use_graph
    model.zero_grad(set_to_none=True)
    out = model(x, labels); l = float(out["loss"].detach()); torch.cuda.synchronize()
    out["loss"].backward(); torch.cuda.synchronize()
    gr = torch.cat([p.grad.flatten() for p in model.parameters() if p.grad is not None]).clone()
    res.append((l, gr))
    print("graph" if use_graph else "eager", l, float(gr.abs().sum()), "%.6f" % hu.cossim(gr, res[0][1]))
if len(sys.argv) > 2:
    names = [n for n, p in model.named_parameters() if p.grad is not None]
    sizes = [p.numel() for n, p in model.named_parameters() if p.grad is not None]
    a, b = res[0][1], res[-1][1]
    o = 0
    for n, sz in zip(names, sizes):
        d = float((a[o:o+sz] - b[o:o+sz]).abs().max()); m = float(a[o:o+sz].abs().max())
        if d > 1e-6 * max(m, 1e-6): print("  diff", n, "max|d| %.3g of max %.3g" % (d, m))
        o += sz

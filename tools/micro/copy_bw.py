import torch, time
def t(fn, reps=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/reps*1e3
for mb in (26, 105, 210, 420, 840):
    n = mb*1000*1000//2
    x = torch.randn(n, device="cuda").to(torch.bfloat16); y = torch.empty_like(x)
    tc = t(lambda: y.copy_(x)); ts = t(lambda: torch.nn.functional.silu(x, inplace=False))
    print("%4d MB  copy %6.1f us %5.2f TB/s   silu %6.1f us %5.2f TB/s" % (mb, tc, 2*n*2/tc/1e6, ts, 2*n*2/ts/1e6))

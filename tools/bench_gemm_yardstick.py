"""hipBLASLt (torch.mm, bf16) on the GEMM shapes that the wide models' convolutions are equivalent to: a yardstick for what a library
GEMM reaches on this box at the same arithmetic intensity (no im2col cost counted: A is a plain [M, K] matrix).  Evidence tool only."""
import torch, time
dev = "cuda"
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
shapes = [("square 8192", 8192, 8192, 8192), ("square 4096", 4096, 4096, 4096),
          ("x 3x3 320->320 @160^2 B16", 409600, 320, 2880), ("x 3x3 640->640 @80^2 B16", 102400, 640, 5760), ("x 3x3 160->160 @320^2 B16", 1638400, 160, 1440),
          ("s 3x3 128->128 @80^2 B32", 204800, 128, 1152), ("s 3x3 256->256 @40^2 B32", 51200, 256, 2304),
          ("x 1x1 320->320 @160^2", 409600, 320, 320), ("x 1x1 640->640 @80^2", 102400, 640, 640), ("x 1x1 160->160 @160^2", 409600, 160, 160),
          ("s 1x1 128->128 @80^2", 204800, 128, 128), ("s 1x1 256->256 @40^2", 51200, 256, 256)]
print("%-32s %9s %9s %9s" % ("shape (M x N x K)", "us", "TFLOP/s", "GB/s"))
for name, M, N, K in shapes:
    a = torch.randn(M, K, device=dev, dtype=torch.bfloat16); b = torch.randn(K, N, device=dev, dtype=torch.bfloat16)
    us = t(lambda: torch.mm(a, b))
    fl = 2.0 * M * N * K; by = 2.0 * (M * K + K * N + M * N)
    print("%-32s %9.1f %9.1f %9.1f   [%d x %d x %d]" % (name, us, fl / us / 1e6, by / us / 1e3, M, N, K))
    del a, b

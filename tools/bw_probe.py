import torch, time
dev = torch.device("cuda:0")
n = 2 * 1024**3  # floats: 8 GB
a = torch.empty(n, device=dev, dtype=torch.float32); b = torch.empty(n, device=dev, dtype=torch.float32)
def t(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / reps
ms = t(lambda: a.fill_(1.0)); print("fill  (write only): %.3f ms  %.0f GB/s" % (ms, 4 * n / ms / 1e6))
ms = t(lambda: b.copy_(a)); print("copy  (read+write): %.3f ms  %.0f GB/s total" % (ms, 8 * n / ms / 1e6))
ms = t(lambda: a.sum()); print("sum   (read only) : %.3f ms  %.0f GB/s" % (ms, 4 * n / ms / 1e6))
# 1:2 read:write mix like the analysis bank: out[2n] from in[n]
c = torch.empty(n // 2, device=dev)
ms = t(lambda: torch.cat([c, c], out=a)); print("cat   (1 read : 2 write... reads cached?): %.3f ms  %.0f GB/s total" % (ms, (4 * n + 2 * n) / ms / 1e6))

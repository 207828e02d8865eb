"""HBM ceilings as plain torch kernels see them: fill (write only), sum (read only), copy (read + write)."""
import torch, time
dev = torch.device("cuda:0")
n = 2 * 1024**3 // 4
a = torch.empty(n, dtype=torch.int32, device=dev)
b = torch.empty(n, dtype=torch.int32, device=dev)
def t(f, k=20):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(k): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / k
GB = n * 4 / 1e9
ms = t(lambda: a.fill_(7)); print("fill  %.3f ms  %.0f GB/s written" % (ms, GB / ms * 1e3))
ms = t(lambda: a.zero_()); print("zero  %.3f ms  %.0f GB/s written" % (ms, GB / ms * 1e3))
ms = t(lambda: a.sum());   print("sum   %.3f ms  %.0f GB/s read" % (ms, GB / ms * 1e3))
ms = t(lambda: b.copy_(a)); print("copy  %.3f ms  %.0f GB/s read+written" % (ms, 2 * GB / ms * 1e3))

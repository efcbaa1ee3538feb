"""GPU: what torch's own streaming kernels reach on this box (sum / dot reads, copy, fill): reference
points for the read / write bandwidth the clipper kernels are compared with in DESIGN.md."""
import torch, time
a = torch.empty(1 << 28, dtype=torch.float32, device="cuda").normal_()   # 1 GiB
b = torch.empty(1 << 28, dtype=torch.float32, device="cuda").normal_()
def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
ms = t(lambda: a.sum()); print("sum 1GiB read:", 2**30 / ms / 1e9 * 1e3, "GB/s")
ms = t(lambda: torch.dot(a, b)); print("dot 2GiB read:", 2**31 / ms / 1e9 * 1e3, "GB/s")
ms = t(lambda: b.copy_(a)); print("copy r+w 2GiB:", 2**31 / ms / 1e9 * 1e3, "GB/s")
ms = t(lambda: a.fill_(1.0)); print("fill 1GiB write:", 2**30 / ms / 1e9 * 1e3, "GB/s")
a2 = a[: 1 << 25]; 
ms = t(lambda: a2.sum()); print("sum 128MiB read:", 2**27 / ms / 1e9 * 1e3, "GB/s")

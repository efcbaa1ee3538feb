"""PMC calibration (MI355X_MICROARCH.md, HBM section): known-byte kernels in OUR access patterns,
so FETCH_SIZE / WRITE_SIZE of the WDF kernels can be corrected.  Run under
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv ...   and again with WRITE_SIZE.
Buffers are 512 MiB (> the 256 MiB Infinity Cache) unless noted."""
import sys
import numpy as np
import torch
import os
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(_R, "differentiable-wdfs_amd", "lib"))
from wdf_hip import binding as wb, workload

n = 128 * 1024 * 1024                      # 512 MiB of fp32
a = torch.empty(n, dtype=torch.float32, device="cuda").normal_()
b = torch.empty_like(a)
torch.cuda.synchronize()
b.copy_(a)                                 # pattern A: 16 B/lane coalesced copy: reads 512 MiB, writes 512 MiB
w = wb.omega(a[: n // 4])                  # pattern B: wdf::omega_kernel, 4 B/lane coalesced: reads 128 MiB, writes 128 MiB
B, T = 8192, 4096                          # pattern C: the clipper's own loads (16 B per lane, row stride T*4 B)
x = torch.as_tensor(workload.sweep_batch(B, T), device="cuda")
th = torch.tensor(workload.clipper_theta(), dtype=torch.float32, device="cuda")
y, zs, _ = wb.clipper_fwd(x, th, workload.FS)          # sequential: reads 128 MiB (x), writes 256 MiB (y, zstash)
torch.cuda.synchronize()
print("done")

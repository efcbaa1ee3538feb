"""GPU: the MLP-root forward's warm-start controller inside the bench's training loop: every verdict the controller read
(warm-up in use, chunk boundaries that missed, largest miss, waves repaired chunk-locally / sequentially).
usage: WDF_MLP_TRACE_WARMUP=1 python tools/mlp_warm_trace.py [root=mlp2x16] [steps=260]"""
import sys, os, subprocess, json
os.environ["WDF_MLP_TRACE_WARMUP"] = "1"
root = sys.argv[1] if len(sys.argv) > 1 else "mlp2x16"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 260
sys.argv = ["bench.py", "--root", root, "--steps", str(steps), "--warmup", "3", "--no-cpu-baseline", "--no-parity"]
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
try:
    bench.main()
except SystemExit:
    pass
from wdf_hip import mlp_root
v = mlp_root._TRACE_VERDICTS
print("verdicts (warm-up: bad/max miss/gated/sequential), ten per line:")
for i in range(0, len(v), 10):
    print("  " + "  ".join(f"{w}:{b}/{m:.1e}/{g}/{q}" for w, b, m, g, q in v[i:i + 10]))

"""GPU: where the device tape interpreter (csrc/wdf_ss_dyn_rows.h) stops paying against torch running the same tape: trees of
one to four capacitors with a pot channel, forward + gradients through the element API at 1340 x 2048, with the rows made
by the device (WDF_DYN_ROWS_MAX_OPS=192) and by torch (=0).  Sets lowering.DYN_ROWS_MAX_OPS."""
import json, os, sys, time
import numpy as np, torch
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(_R, "differentiable-wdfs_amd", "lib"))
import tf_wdf as wdf
from tf_wdf import tf
from wdf_hip import lowering, workload

FS, B, T = 48000.0, 1340, 2048
x = workload.sweep_batch(B, T, seed=4) * 0.6
r = workload.dataset_resistance_batch(B, T, grid=(300.0, 1.0e3, 2.5e3, 5.0e3))
xin = torch.as_tensor(np.stack([x, r], axis=-1).astype(np.float32), device="cuda")
tgt = 0.2 * torch.randn((T, B), device="cuda", generator=torch.Generator(device="cuda").manual_seed(0))


def tree(ns):
    Vs = wdf.ResistiveVoltageSource(1.0e3)
    Cs = [wdf.Capacitor(22.0e-9 * (i + 1), FS, True) for i in range(ns)]
    Rs = [wdf.Resistor(1.0e3 * (i + 1), True) for i in range(max(ns - 1, 1))]
    top = wdf.Series(Vs, Cs[0])
    chain = None
    for i in range(1, ns):
        sec = wdf.Parallel(Rs[i - 1], Cs[i])
        chain = sec if chain is None else wdf.Series(chain, sec)
    top = wdf.Parallel(top, chain if chain is not None else Rs[0])
    rt = wdf.DiodePair(top, 4.352e-9, Vt=0.049, trainable=True)
    circ = wdf.Circuit(top, rt, Cs[-1], per_sample_R=Vs)
    return circ, [c.C for c in Cs] + [rt.Is, rt.nVt]


for ns in (1, 2, 3, 4):
    out = {"capacitors": ns}
    for name, cap in (("device", 192), ("torch", 0)):
        lowering.DYN_ROWS_MAX_OPS = cap
        circ, params = tree(ns)

        def step():
            with tf.GradientTape() as tape:
                y = circ(xin)
                loss = tf.reduce_mean(tf.square(y - tgt))
            return tape.gradient(loss, params)

        for _ in range(4):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(6):
            step()
        torch.cuda.synchronize()
        out[name + "_ms"] = round((time.perf_counter() - t0) / 6 * 1e3, 3)
        out["tape_ops"] = len(circ._dyn_tape[0].ops)
    print(json.dumps(out))

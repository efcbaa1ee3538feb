#!/usr/bin/env python3
"""tools/stress_nl_step.py [cases] [seed] -- randomized cases of the diode-root one-pass step (HPF clipper topology: component
values, diode counts, batch and length drawn at random; three calls with Adam steps in between) against the fp64 oracle."""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "differentiable-wdfs_amd", "lib"))
sys.path.insert(0, os.path.join(REPO, "oracle"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import tf_wdf as wdf  # noqa: E402
from tf_wdf import tf  # noqa: E402
import oracle as O  # noqa: E402

FS = 48000
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
worst = {"y": 0.0, "loss": 0.0, "grad": 0.0}
for case in range(cases):
    B = int(rng.choice([1, 3, 64, 65, 128, 130, 257, 512]))
    T = int(rng.choice([17, 64, 100, 130, 257, 1000, 1501, 2048, 3000]))
    n_up, n_down = int(rng.integers(1, 4)), int(rng.integers(1, 4))
    theta = np.array([10 ** rng.uniform(3.5, 5.0), 10 ** rng.uniform(2.5, 3.7), 10 ** rng.uniform(-8.5, -7.0), 10 ** rng.uniform(-9.5, -8.0),
                      25.85e-3 * rng.uniform(1.0, 2.0)], dtype=np.float32).astype(np.float64)
    x = (rng.standard_normal((B, T)) * rng.uniform(0.3, 2.0)).astype(np.float32)
    tgt = (0.3 * rng.standard_normal((T, B))).astype(np.float32)
    R = wdf.Resistor(float(theta[0]), True); Vs = wdf.ResistiveVoltageSource(float(theta[1]), trainable=True)   # noqa: E702
    C = wdf.Capacitor(float(theta[2]), FS, True)
    top = wdf.Parallel(R, wdf.Series(Vs, C))
    dp = wdf.DiodePair(top, float(theta[3]), Vt=float(theta[4]), nDiodes=1.0, N_up=n_up, N_down=n_down, trainable=True)
    circ, params = wdf.Circuit(top, dp, R), [R.R, Vs.R, C.C, dp.Is, dp.nVt]
    circ.to_device()
    opts = [tf.keras.optimizers.Adam(learning_rate=1.0e-3 * float(p)) for p in params]
    nodes = [(O.NODE_RESISTOR, -1, -1, 0, -1, -1), (O.NODE_RES_VSOURCE, -1, -1, 1, 0, -1), (O.NODE_CAPACITOR, -1, -1, 2, -1, -1),
             (O.NODE_SERIES, 1, 2, -1, -1, -1), (O.NODE_PARALLEL, 0, 3, -1, -1, -1)]
    oc = O.Circuit(nodes, top=4, probe=0, n_in=1, root_kind=O.ROOT_DIODE_PAIR, fs=FS, p_is=3, p_nvt=4, n_up=n_up, n_down=n_down)
    xd, td = torch.as_tensor(x, device="cuda"), torch.as_tensor(tgt, device="cuda")
    line = []
    for call in range(3):
        th = np.array([float(p) for p in params], dtype=np.float32).astype(np.float64)
        with tf.GradientTape() as tape:
            loss = circ.mse(xd, td)
        grads = tape.gradient(loss, params)
        g = np.array([float(v) for v in grads])
        y = circ.last_output.detach().cpu().numpy()
        yref = O.tree_fwd(oc, th, x.astype(np.float64))
        e = yref - tgt
        gref = O.tree_grad(oc, th, x.astype(np.float64), 2.0 * e / e.size)
        ey, el = float(np.max(np.abs(y - yref))), abs(float(loss) - float(np.mean(e * e))) / float(np.mean(e * e))
        eg = float(np.max(np.abs(g - gref) / np.maximum(np.abs(gref), 1e-3 * np.max(np.abs(gref)))))
        ctl = circ._tree.read_ctl(next(iter(circ._tree.cache.values())))
        worst = {"y": max(worst["y"], ey), "loss": max(worst["loss"], el), "grad": max(worst["grad"], eg)}
        line.append(f"{ey:.1e}/{el:.1e}/{eg:.1e} w{ctl['w_used']} r{ctl['gated_groups']}")
        for o, gr, p in zip(opts, grads, params):
            o.apply_gradients([(gr, p)])
    print(f"case {case}: B {B} T {T} diodes {n_up}/{n_down} theta {[f'{v:.3g}' for v in theta]}: " + "  ".join(line), flush=True)
print(f"worst: |y - oracle| {worst['y']:.2e}, loss {worst['loss']:.2e}, gradients {worst['grad']:.2e}")

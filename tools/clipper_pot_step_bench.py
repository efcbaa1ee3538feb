#!/usr/bin/env python3
"""tools/clipper_pot_step_bench.py [steps] -- clipper_pot.py:245-269's epoch as the script writes it (GradientTape ->
circ.mse_esr -> tape.gradient(model.trainable_variables) -> Adam(1e-4, beta_1 0.5).apply_gradients) at the reference's
training-set shape (1340 x 2048, pot value per sample, committed 2x16 weights): plain path against Circuit.to_device()."""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "differentiable-wdfs_amd", "lib"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import tf_wdf as wdf  # noqa: E402
from tf_wdf import tf  # noqa: E402
from layers import DenseRootModel  # noqa: E402
from wdf_hip import binding, workload  # noqa: E402

FS, B, T, skip = 48000, 1340, 2048, 50
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
x = workload.sweep_batch(B, T, seed=4) * 0.6
r = workload.dataset_resistance_batch(B, T)
xin = torch.as_tensor(np.stack([x, r], axis=-1), device="cuda")
target, _, _ = binding.clipper_fwd(torch.as_tensor(x, device="cuda"), torch.tensor(workload.clipper_theta(), dtype=torch.float32, device="cuda"),
                                   FS, r=torch.as_tensor(r, device="cuda"), want_stash=False)
wh, hidden, n_layers = workload.reference_mlp_weights("2x16_pre")


def model_json():
    layers, o, sizes = [], 0, [2] + [hidden] * n_layers + [1]
    for i in range(len(sizes) - 1):
        ni, no = sizes[i], sizes[i + 1]
        k = wh[o:o + ni * no].reshape(ni, no); o += ni * no
        b = wh[o:o + no]; o += no
        layers.append({"type": "dense", "activation": "tanh" if i < len(sizes) - 2 else "", "shape": [None, no], "weights": [k.tolist(), b.tolist()]})
    return {"in_shape": [None, 2], "layers": layers}


for resident in (False, True):
    Vs = wdf.ResistiveVoltageSource(45.0e3)
    C = wdf.Capacitor(workload.C_CLIPPER, FS)
    P1 = wdf.Parallel(Vs, C)
    model = DenseRootModel(model_json())
    circ = wdf.Circuit(P1, model, C, per_sample_R=Vs)
    if resident:
        circ.to_device()
    opt = tf.keras.optimizers.Adam(learning_rate=1.0e-4, beta_1=0.5)
    tv = model.trainable_variables

    def step():
        with tf.GradientTape() as tape:
            loss = circ.mse_esr(xin, target, skip)
        opt.apply_gradients(zip(tape.gradient(loss, tv), tv))
        return loss

    for _ in range(40):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = step()
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{'Circuit.to_device() (resident step)' if resident else 'plain path                         '}: {dt / steps * 1e3:.4f} ms per epoch = "
          f"{B * T / (dt / steps) / 1e9:.2f} G samples/s (host side {t_host / steps * 1e3:.4f} ms); loss {float(loss):.5e}")

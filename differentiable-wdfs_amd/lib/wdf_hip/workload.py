"""Synthetic workloads of BASELINE.json / SURVEY.md section 8(d) (all seeds fixed).

The measured dataset of the reference is absent (.MISSING_LARGE_BLOBS), so inputs are
per-sequence exponential sine sweeps 100 Hz -> 10 kHz at 48 kHz (the family lpf.py:58 uses)
with amplitude A_b ~ U(0.1, 5) V and start phase phi_b ~ U(0, 2 pi),
numpy.random.default_rng(1234); zero initial state (clipper_pot.py:110-111).
"""
import numpy as np

FS = 48000.0
# 1N4148 (diode_config.py:14-16), Vs resistance (clipper_pot.py:97), C (clipper_pot.py:50)
IS_1N4148 = 4.352e-9
VT = 25.85e-3
NABLA_1N4148 = 1.906
R_CLIPPER = 45.0e3
C_CLIPPER = 4.7e-9


def clipper_theta():
    """{Is, nVt, R, C} of the 1N4148 diode clipper."""
    return np.array([IS_1N4148, VT * NABLA_1N4148, R_CLIPPER, C_CLIPPER], dtype=np.float64)


def target_theta():
    """theta* the synthetic targets are generated at: (1.2 Is, 0.95 nVt, 0.9 R, 1.1 C)."""
    return clipper_theta() * np.array([1.2, 0.95, 0.9, 1.1])


def sweep_batch(B_global, T, fs=FS, seed=1234, b0=0, b1=None, f0=100.0, f1=10000.0, dtype=np.float32):
    """Rows b0:b1 of the global [B_global, T] input batch (so N ranks can each build their
    shard of the SAME global batch without materialising the rest)."""
    b1 = B_global if b1 is None else b1
    rng = np.random.default_rng(seed)
    amp = rng.uniform(0.1, 5.0, B_global)[b0:b1]
    phase = rng.uniform(0.0, 2.0 * np.pi, B_global)[b0:b1]
    t = np.arange(T, dtype=np.float64) / fs
    dur = T / fs
    k = np.log(f1 / f0)
    inst = 2.0 * np.pi * f0 * dur / k * (np.exp(t / dur * k) - 1.0)
    x = amp[:, None] * np.sin(inst[None, :] + phase[:, None])
    return np.ascontiguousarray(x, dtype=dtype)


def pot_resistance_batch(B_global, T, b0=0, b1=None, dtype=np.float32):
    """Per-sequence constant pot values on the reference's file-name grid
    (.MISSING_LARGE_BLOBS:1-5: 10.0k, 25.2k, 45.2k, 75.0k, 99.1k), as clipper_pot.py feeds them
    through input channel 1 (dataimport.py:96)."""
    b1 = B_global if b1 is None else b1
    grid = np.array([10.0e3, 25.2e3, 45.2e3, 75.0e3, 99.1e3])
    r = grid[np.arange(b0, b1) % len(grid)]
    return np.ascontiguousarray(np.repeat(r[:, None], T, axis=1), dtype=dtype)


def dataset_resistance_batch(B_global, T, b0=0, b1=None, grid=(10.0e3, 25.2e3, 75.0e3, 99.1e3), dtype=np.float32):
    """Pot values laid out as the reference's loader leaves them: load_diode_data concatenates the training
    recordings file by file and batch_data cuts the result into sequences (dataimport.py:82-137,
    clipper_pot.py:61-80), so the batch is four contiguous blocks of sequences, one pot value each
    (training files of .MISSING_LARGE_BLOBS:1-5: 10.0k, 25.2k, 75.0k, 99.1k; 45.2k validates)."""
    b1 = B_global if b1 is None else b1
    g = np.asarray(grid, dtype=np.float64)
    idx = np.minimum((np.arange(b0, b1) * len(g)) // max(B_global, 1), len(g) - 1)
    return np.ascontiguousarray(np.repeat(g[idx][:, None], T, axis=1), dtype=dtype)


def reference_mlp_weights(name="2x16", path=None):
    """(flat weights float32, hidden, n_tanh) of one of the reference's committed networks
    (wdf_py/diode_clipper/models/*.json: kernel[in][out] then bias[out] per layer) -- from the package's own data file
    wdf_hip/data/mlp_reference_weights.npz (the numbers tests/golden/gen_golden.py read out of those JSON files; the
    package does not reach into tests/), or from `path`: another .npz with `<name>_theta` / `<name>_sizes`, or a model JSON
    in the reference's schema (model_utils.save_model / layers.DenseRootModel)."""
    import os
    if path is not None and str(path).endswith(".json"):
        import json
        layers = json.load(open(path))["layers"]
        flat, sizes = [], [2]
        for layer in layers:
            k, b = np.asarray(layer["weights"][0], dtype=np.float32), np.asarray(layer["weights"][1], dtype=np.float32)
            flat += [k.reshape(-1), b.reshape(-1)]
            sizes.append(int(b.size))
        return np.concatenate(flat), sizes[1], len(sizes) - 2
    if path is None:
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "mlp_reference_weights.npz")
    g = np.load(path)
    sizes = [int(v) for v in g[f"{name}_sizes"]]
    return g[f"{name}_theta"].astype(np.float32), sizes[1], len(sizes) - 2

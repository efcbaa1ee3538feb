"""CPU: dataset CSV format / loader semantics (dataimport.py), batching (clipper_pot.py:61-80)
and the JSON weight interchange (model_utils.py, layers.py loader, clipper_pot.py:298-331)."""
import json
import os
from collections import namedtuple

import numpy as np
import pytest

REF = "/root/reference"
DiodeConfig = namedtuple("DiodeConfig", ["name", "Is", "nabla", "Vt", "N_up", "N_down"])
D1 = DiodeConfig("1N4148 (1U-1D)", 4.352e-9, 1.906, 25.85e-3, 1, 1)


def test_csv_roundtrip_and_split(tmp_path):
    import dataimport as di
    fs, secs = 1000.0, di.TIME_REMOVE_PRE + di.DUR_OF_DATA + 0.2
    files = di.write_synthetic_dataset(tmp_path, lambda x, R: np.tanh(x) * (1e4 / R) ** 0.1, fs=fs, seconds=secs)
    assert [p.name for p in files] == di.DATASET_FILES["1N4148/1up1down"]
    raw = di.createDataset(files[0])
    assert raw["FS"] == fs and raw["num_samples"] == int(np.ceil(di.DUR_OF_DATA * fs))      # 2.5 s dropped, 14.3 s kept
    x_full = di.sweep_signal(fs, secs, 0)
    assert np.allclose(raw["dataset"][:, 0], x_full[int(di.TIME_REMOVE_PRE * fs):][: raw["num_samples"]], atol=1e-7)
    train, train_n, val, val_n, FS = di.load_diode_data(D1, tmp_path)
    assert FS == fs and train.shape == (3, train_n) and val.shape == (3, val_n)
    assert train_n == 4 * raw["num_samples"] and val_n == raw["num_samples"]                # 45.2k validates
    assert set(np.unique(train[1])) == {10000.0, 25200.0, 75000.0, 99100.0}
    assert set(np.unique(val[1])) == {45200.0}
    X, Y = di.batch_data(train, train_n, batch_size=2048)
    assert X.shape == (train_n // 2048, 2048, 2) and Y.shape == (train_n // 2048, 2048, 1)
    assert np.array_equal(X[0, :, 0], train[0, :2048]) and np.array_equal(Y[1, :, 0], train[2, 2048:4096])


def test_header_parsers():
    import dataimport as di
    rows = ["#a", "#b", "#c", "#d", "#Sample rate: 96000.0Hz", "#Samples: 1234", "#", "#", "#"]
    assert di.getSampleRate(rows) == 96000.0 and di.getDatasetSize(rows) == 1234.0
    assert di.is_training_resistance(25.2) and di.is_training_resistance(75.0) and not di.is_training_resistance(45.2)


def _model_json(sizes=(2, 8, 8, 8, 1), seed=0, with_input_layer=True):
    rng = np.random.default_rng(seed)
    layers = [{"type": "unknown", "activation": "", "shape": [[None, 2]], "weights": []}] if with_input_layer else []
    for i in range(len(sizes) - 1):
        layers.append({"type": "dense", "activation": "tanh" if i < len(sizes) - 2 else "", "shape": [None, sizes[i + 1]],
                       "weights": [rng.standard_normal((sizes[i], sizes[i + 1])).tolist(), rng.standard_normal(sizes[i + 1]).tolist()]})
    return {"in_shape": [None, 2], "layers": layers}


def test_json_weight_roundtrip(tmp_path):
    from layers import DenseRootModel, DenseLayer
    import model_utils as mu
    js = _model_json()
    m = DenseRootModel(js)
    assert [type(l).__name__ if isinstance(l, DenseLayer) else l.__name__ for l in m.layers] == \
        ["DenseLayer", "tanh", "DenseLayer", "tanh", "DenseLayer", "tanh", "DenseLayer"]
    assert tuple(m.layers[0].kernel.shape) == (1, 2, 8) and tuple(m.layers[0].bias.shape) == (1, 8)   # layers.py:18-21
    out = mu.save_model_json(m)
    assert out["in_shape"] == (None, 2) and [l["activation"] for l in out["layers"]] == ["tanh", "tanh", "tanh", ""]
    dense_in = [l for l in js["layers"] if l["type"] == "dense"]
    for a, b in zip(out["layers"], dense_in):
        assert np.allclose(a["weights"][0], np.array(b["weights"][0], np.float32)) and a["shape"][1] == b["shape"][1]
    f = tmp_path / "m.json"
    mu.save_model(m, f)
    m2 = DenseRootModel(mu.load_model_json(f))
    for l1, l2 in zip(m.layers, m2.layers):
        if isinstance(l1, DenseLayer):
            assert np.array_equal(l1.kernel.numpy(), l2.kernel.numpy()) and np.array_equal(l1.bias.numpy(), l2.bias.numpy())


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference checkout not present")
def test_all_committed_reference_models_load():
    """Every current-schema JSON under wdf_py/diode_clipper/models (72 files, SURVEY #13) loads."""
    from layers import DenseRootModel
    from wdf_hip import mlp_root
    root = os.path.join(REF, "wdf_py/diode_clipper/models")
    n = 0
    for dp, dn, fns in os.walk(root):
        if os.path.basename(dp) == "old":
            continue                                            # legacy schema, unused by the scripts
        for fn in fns:
            if fn.endswith(".json"):
                js = json.load(open(os.path.join(dp, fn)))
                if "layers" not in js or "in_shape" not in js:
                    continue
                dense_shapes = [l["shape"] for l in js["layers"] if l["type"] == "dense"]
                if not isinstance(js["in_shape"][-1], int) or \
                        any(not (isinstance(sh, list) and isinstance(sh[-1], int)) for sh in dense_shapes):
                    continue      # legacy variants (e.g. 1N4148_clipper_pot.json) the reference loader cannot read either
                m = DenseRootModel(js)
                dense, hidden, n_tanh = mlp_root.describe(m)
                assert (hidden in (4, 8, 16) and n_tanh == 3) or (hidden in (4, 8) and n_tanh in (4, 5))
                assert mlp_root.flat_weights(dense).numel() == 3 * hidden + (n_tanh - 1) * (hidden * hidden + hidden) + hidden + 1
                n += 1
    assert n >= 60


def test_audio_dspy_helpers():
    import audio_dspy as adsp
    s = adsp.sweep_log(100, 10000, 1280 / 48000, 48000)
    assert len(s) == 1280 and abs(s[0]) < 1e-12 and np.max(np.abs(s)) <= 1.0
    b, a = adsp.design_LPF1(720, 48000)
    w = 2 * np.pi * 720 / 48000
    H = (b[0] + b[1] * np.exp(-1j * w)) / (a[0] + a[1] * np.exp(-1j * w))
    assert abs(abs(H) - 1 / np.sqrt(2)) < 1e-9 and abs(sum(b) / sum(a) - 1.0) < 1e-12


def test_pretraining_module_has_no_cpu_path():
    """diode_config mirrors the reference's named configurations; diode_pretraining imports on a
    CPU-only box but every computing entry point raises (no fallback)."""
    import torch
    import diode_config as dc
    import diode_pretraining as dp
    from wdf_hip.binding import WdfHipError
    d = dc.diode_1n4148_2u3d
    assert (d.name, d.Is, d.nabla, d.Vt, d.N_up, d.N_down) == ("1N4148 (2U-3D)", 4.352e-9, 1.906, 25.85e-3, 2, 3)
    assert dc.default_diode == dc.DiodeConfig("DefaultDiode", 1.0e-9, 1.0, 25.85e-3, 1, 1)
    m = dp.build_model(2, 4, seed=0)                        # host-side construction works anywhere
    assert [tuple(v.shape) for v in m.trainable_variables][:2] in ([(1, 2, 4), (1, 4)], [(1, 4), (1, 2, 4)])
    if not torch.cuda.is_available():
        import pytest
        with pytest.raises(WdfHipError):
            dp.synthetic_table(d)


def test_loader_matches_the_reference_loader_fixture(tmp_path, golden):
    """f1 pinned to the reference: tests/golden/gen_golden.py g8 wrote the five CSVs of 1up1down with
    THIS writer, loaded them with the REFERENCE's dataimport.createDataset / load_diode_data
    (dataimport.py:10-59,82-137) and cut them with clipper_pot.py's own batch_data (:61-80).  The drop-in
    loader, on files written the same way, must return the same arrays: trimmed lengths (2.5 s dropped),
    R from the file name, the train / validation split, shapes, first and last rows, checksums."""
    from collections import namedtuple
    import dataimport as di
    g = golden("g8_dataimport.npz")
    cfg = namedtuple("DiodeConfig", ["name", "Is", "nabla", "Vt", "N_up", "N_down"])("1N4148 (1U-1D)", 4.352e-9, 1.906,
                                                                                    25.85e-3, 1, 1)
    sim = lambda x, R: np.tanh(2.0 * x) * (0.3 + 1.0e-6 * R)          # the generator's "measurement"
    files = di.write_synthetic_dataset(tmp_path, sim, fs=float(g["fs"]), seconds=float(g["seconds"]))
    assert sorted(os.path.basename(str(f)) for f in files) == list(g["files"])
    train, train_N, val, val_N, FS = di.load_diode_data(cfg, tmp_path)
    assert FS == float(g["fs"]) and train_N == int(g["train_N"]) and val_N == int(g["val_N"])
    assert tuple(train.shape) == tuple(g["train_data_shape"]) and tuple(val.shape) == tuple(g["val_data_shape"])
    batch = int(g["batch"])
    tX, tY = di.batch_data(train, train_N, batch)
    vX, vY = di.batch_data(val, val_N, batch)
    for got, name in ((tX, "train_X"), (tY, "train_Y"), (vX, "val_X"), (vY, "val_Y")):
        assert tuple(got.shape) == tuple(g[f"{name}_shape"]), name
    assert np.array_equal(tX[:, :4, :], g["train_X_head"]) and np.array_equal(tX[:, -4:, :], g["train_X_tail"])
    assert np.array_equal(tY[:, :4], g["train_Y_head"]) and np.array_equal(tY[:, -4:], g["train_Y_tail"])
    assert np.array_equal(vX[:, :4, :], g["val_X_head"]) and np.array_equal(vY[:, -4:], g["val_Y_tail"])
    assert np.array_equal(tX[:, 0, 1], g["train_R_per_sequence"]) and np.array_equal(vX[:, 0, 1], g["val_R_per_sequence"])
    for got, ref in ((tX[..., 0], g["train_X_sum"][0]), (tX[..., 1], g["train_X_sum"][1]), (tY, g["train_Y_sum"]),
                     (vX[..., 0], g["val_X_sum"][0]), (vX[..., 1], g["val_X_sum"][1]), (vY, g["val_Y_sum"])):
        assert got.astype(np.float64).sum() == float(ref)


def test_package_weight_file_is_the_goldens_and_json_path(golden, tmp_path):
    """wdf_hip/data/mlp_reference_weights.npz (what bench.py --root mlp* and the tools read: the package does not reach into
    tests/) holds the same numbers as the golden the reference's JSON models were read into; a model JSON written by
    model_utils.save_model loads to the same flat vector through the `path` argument."""
    from wdf_hip import workload
    import layers, model_utils
    g3 = golden("g3_mlp_clipper.npz")
    for net in ("2x4", "2x8", "2x16", "2x16_pre", "4x4", "4x8"):
        w, hidden, n_layers = workload.reference_mlp_weights(net)
        assert np.array_equal(w, g3[f"{net}_theta"].astype(np.float32))
        assert [2] + [hidden] * n_layers + [1] == [int(v) for v in g3[f"{net}_sizes"]]
    w, hidden, n_layers = workload.reference_mlp_weights("2x8")
    js = {"in_shape": [None, 2], "layers": []}
    o, n_in = 0, 2
    for i in range(n_layers + 1):
        n_out = hidden if i < n_layers else 1
        k = w[o:o + n_in * n_out].reshape(n_in, n_out); o += n_in * n_out
        b = w[o:o + n_out]; o += n_out
        js["layers"].append({"type": "dense", "activation": "tanh" if i < n_layers else "", "shape": [None, n_out],
                             "weights": [k.tolist(), b.tolist()]})
        n_in = n_out
    import json
    path = tmp_path / "m.json"
    json.dump(js, open(path, "w"))
    w2, h2, n2 = workload.reference_mlp_weights(path=str(path))
    assert (h2, n2) == (hidden, n_layers) and np.array_equal(w2, w)

"""GPU: the tanh-MLP root (layers.DenseRootModel inside clipper_pot.py's ClipperModel) against
goldens recorded by running the reference's ClipperModel / loss_func with the committed
2x4, 2x8, 2x16 weight files (tests/golden/gen_golden.py: g3)."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

FS = 48000


def cuda(a):
    return torch.as_tensor(np.ascontiguousarray(a, dtype=np.float32), device="cuda")


def model_json(g, name):
    """Rebuild the reference JSON structure (in_shape + dense layers) from the golden's flat weights."""
    sizes, acts, th = [int(s) for s in g[f"{name}_sizes"]], [int(a) for a in g[f"{name}_acts"]], g[f"{name}_theta"]
    layers, o = [], 0
    for i in range(len(sizes) - 1):
        ni, no = sizes[i], sizes[i + 1]
        k = th[o:o + ni * no].reshape(ni, no)
        o += ni * no
        b = th[o:o + no]
        o += no
        layers.append({"type": "dense", "activation": {1: "tanh", 2: "relu", 0: ""}[acts[i]], "shape": [None, no],
                       "weights": [k.tolist(), b.tolist()]})
    return {"in_shape": [None, 2], "layers": [{"type": "unknown", "activation": "", "shape": [[None, 2]], "weights": []}] + layers}


@pytest.mark.parametrize("name", ["2x4", "2x8", "2x16", "2x16_pre"])
def test_clipper_model_forward_loss_grads(golden, name):
    import tf_wdf as wdf
    from tf_wdf import tf
    from layers import DenseRootModel, DenseLayer
    g = golden("g3_mlp_clipper.npz")
    skip = int(g["skip"])
    # clipper_pot.py:94-101
    Vs = wdf.ResistiveVoltageSource(45.0e3)
    C = wdf.Capacitor(float(g["C"]), FS)
    P1 = wdf.Parallel(Vs, C)
    model = DenseRootModel(model_json(g, name))
    assert sum(isinstance(l, DenseLayer) for l in model.layers) == len(g[f"{name}_sizes"]) - 1
    circ = wdf.Circuit(P1, model, C, per_sample_R=Vs)
    x = cuda(g["x"])                                            # [B,T,2]: Vin, R  (clipper_pot.py:68-70)
    target = cuda(g["target"])                                  # [B,T,1]
    with tf.GradientTape() as tape:
        y = circ(x)                                             # [T,B]
        outs = tf.transpose(y, perm=[1, 0])[:, :, None]         # clipper_pot.py:247 -> [B,T,1]
        o, t = outs[:, skip:, :], target[:, skip:, :]
        # loss_func(outs, train_Y) = MSE + ESR with (outs, target) passed as (target, pred)
        mse = tf.reduce_mean(tf.square(o - t))
        energy = tf.reduce_sum(tf.square(o)) + np.finfo(float).eps
        n = float(o.shape[0] * o.shape[1])
        esr = tf.sqrt(tf.reduce_sum(tf.square(o - t)) / energy / n)
        loss = mse + esr
    tv = model.trainable_variables
    assert len(tv) == 2 * (len(g[f"{name}_sizes"]) - 1)
    dense = [l for l in model.layers if isinstance(l, DenseLayer)]
    order = []
    for d in dense:
        order += [d.kernel, d.bias]
    grads = tape.gradient(loss, order)
    assert np.max(np.abs(y.numpy() - g[f"{name}_y_f64"])) < 3e-5
    assert abs(float(loss) - float(g[f"{name}_loss_f64"])) < 2e-5
    got = np.concatenate([gr.numpy().ravel() for gr in grads])
    ref = g[f"{name}_grad_f64"]
    assert np.max(np.abs(got - ref)) < 2e-3 * np.max(np.abs(ref)), np.max(np.abs(got - ref)) / np.max(np.abs(ref))


def test_static_resistance_and_capacitor_gradient(oracle, golden):
    """Scalar trainable R and C with the MLP root against the oracle's complex-step gradient."""
    import tf_wdf as wdf
    from tf_wdf import tf
    from layers import DenseRootModel
    g = golden("g3_mlp_clipper.npz")
    name = "2x8"
    Vs = wdf.ResistiveVoltageSource(45.0e3, trainable=True)
    C = wdf.Capacitor(float(g["C"]), FS, trainable=True)
    P1 = wdf.Parallel(Vs, C)
    model = DenseRootModel(model_json(g, name))
    circ = wdf.Circuit(P1, model, C)
    x = g["x"][:, :, 0]
    rng = np.random.default_rng(0)
    gy = (rng.standard_normal((x.shape[1], x.shape[0])) / x.size).astype(np.float32)
    y = circ(cuda(x))
    grads = tf.GradientTape().gradient(tf.reduce_sum(y * cuda(gy)), [Vs.R, C.C])
    sizes, acts = [int(s) for s in g[f"{name}_sizes"]], [int(a) for a in g[f"{name}_acts"]]
    O = oracle
    nodes = [(O.NODE_RES_VSOURCE, -1, -1, 0, 0, -1), (O.NODE_CAPACITOR, -1, -1, 1, -1, -1),
             (O.NODE_PARALLEL, 0, 1, -1, -1, -1)]
    oc = O.Circuit(nodes, top=2, probe=1, n_in=1, root_kind=O.ROOT_MLP, fs=FS, mlp_off=2, mlp_sizes=sizes, mlp_act=acts)
    theta = np.concatenate([[45.0e3, float(np.float32(g["C"]))], g[f"{name}_theta"].astype(np.float32).astype(np.float64)])
    yref = O.tree_fwd(oc, theta, x)
    assert np.max(np.abs(y.numpy() - yref)) < 3e-5
    gref = O.tree_grad(oc, theta, x, gy.astype(np.float64), params=[0, 1])
    got = np.array([float(v) for v in grads])
    assert np.max(np.abs(got - gref) / np.abs(gref)) < 3e-3, (got, gref)


def test_unsupported_network_is_an_error(golden):
    import tf_wdf as wdf
    from layers import DenseRootModel
    from wdf_hip import binding
    js = {"in_shape": [None, 2], "layers": [
        {"type": "dense", "activation": "relu", "shape": [None, 4], "weights": [np.zeros((2, 4)).tolist(), [0.0] * 4]},
        {"type": "dense", "activation": "", "shape": [None, 1], "weights": [np.zeros((4, 1)).tolist(), [0.0]]}]}
    Vs = wdf.ResistiveVoltageSource(45.0e3)
    C = wdf.Capacitor(4.7e-9, FS)
    P1 = wdf.Parallel(Vs, C)
    circ = wdf.Circuit(P1, DenseRootModel(js), C)
    with pytest.raises(binding.WdfHipError):
        circ(cuda(np.zeros((2, 16))))

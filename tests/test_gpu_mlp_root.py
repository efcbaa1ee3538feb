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


@pytest.mark.parametrize("name", ["2x4", "2x8", "2x16", "2x16_pre", "4x4", "4x8"])
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
    ey = np.max(np.abs(y.numpy() - g[f"{name}_y_f64"]))
    got = np.concatenate([gr.numpy().ravel() for gr in grads])
    ref = g[f"{name}_grad_f64"]
    # per component: relative to its own size, plus a floor of 1e-5 of the largest component (weights
    # whose gradient is a cancelling sum carry the fp32 rounding of its terms)
    eg = np.max(np.abs(got - ref) / (np.abs(ref) + 1e-1 * np.max(np.abs(ref))))
    print(f"{name}: max|y - ref| {ey:.2e}, loss diff {abs(float(loss) - float(g[f'{name}_loss_f64'])):.2e}, grad {eg:.2e}")
    assert ey < 5e-6                                            # observed 1.1e-6 .. 2.1e-6 (fp32 tanh chain)
    assert abs(float(loss) - float(g[f"{name}_loss_f64"])) < 2e-6
    assert np.all(np.abs(got - ref) <= 1e-4 * np.abs(ref) + 1e-5 * np.max(np.abs(ref))), eg


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


def test_segmented_time_parallel_matches_sequential(golden):
    """Data-level time parallelism of the MLP path: K overlapping segments == the sequential run
    (outputs within the verified 1e-6, gradients within 1e-4 relative), and a circuit whose
    memory outlasts the segments falls back to the sequential call."""
    import tf_wdf as wdf
    from tf_wdf import tf
    from layers import DenseRootModel, DenseLayer
    from wdf_hip import mlp_root, workload
    g = golden("g3_mlp_clipper.npz")
    B, T = 96, 2048
    x = workload.sweep_batch(B, T, seed=9) * 0.6
    r = workload.pot_resistance_batch(B, T)
    xin = cuda(np.stack([x, r], axis=-1))
    gy = cuda(np.random.default_rng(1).standard_normal((T, B)) / (B * T))

    def run(tp):
        Vs = wdf.ResistiveVoltageSource(45.0e3)
        C = wdf.Capacitor(float(g["C"]), FS, trainable=True)
        P1 = wdf.Parallel(Vs, C)
        model = DenseRootModel(model_json(g, "2x8"))
        circ = wdf.Circuit(P1, model, C, per_sample_R=Vs, time_parallel=tp)
        y = circ(xin)
        dense = [l for l in model.layers if isinstance(l, DenseLayer)]
        grads = tf.GradientTape().gradient(tf.reduce_sum(y * gy), [C.C, dense[0].kernel, dense[2].kernel, dense[3].bias])
        return y, [gr.numpy().ravel() for gr in grads]

    assert mlp_root.segment_plan(B, T, 99.1e3, float(g["C"]), FS) is not None
    mlp_root.LAST_SEGMENT_MISS["miss"] = None
    y_seq, g_seq = run(None)
    y_tp, g_tp = run("segments")       # the data-level path (truncated backward); "auto" = the in-kernel path below
    assert mlp_root.LAST_SEGMENT_MISS["miss"] is not None and mlp_root.LAST_SEGMENT_MISS["miss"] <= 1e-6
    assert float((y_tp - y_seq).abs().max()) <= 2e-6
    for a, b in zip(g_tp, g_seq):
        assert np.max(np.abs(a - b)) <= 1e-4 * np.max(np.abs(b)) + 1e-12
    # too-short segments are caught by the verification and the result is still the sequential one
    y_bad, _, miss = mlp_root.clipper_mlp_segmented(
        cuda([45.0e3, float(g["C"])]), mlp_root.flat_weights([l for l in DenseRootModel(model_json(g, "2x8")).layers
                                                             if isinstance(l, DenseLayer)]).float().cuda(),
        cuda(x), cuda(r), float(FS), 8, 3, (8, 256, 16))
    assert miss > 1e-6 and float((y_bad - y_seq).abs().max()) <= 2e-6


@pytest.mark.parametrize("hidden,n_tanh", [(4, 3), (8, 3), (16, 3), (8, 4), (4, 5), (8, 5)])
@pytest.mark.parametrize("dyn", [False, True])
@pytest.mark.parametrize("S", [64 * 700 + 37, 1, 17, 16 * 8 * 4 * 3 + 1])   # ragged: the tail lanes must contribute nothing; one sample;
def test_wgrad_kernel_matches_dense_autograd(hidden, n_tanh, dyn, S):       # one partial block; one sample past a workgroup's share
    """wdf_clipper_mlp_wgrad == -sum_n gb[n] dMLP(a[n], lr[n])/dw by float64 torch autograd of the
    same network (tolerance 2e-5 of the largest entry: fp32 accumulation on the matrix cores, double reduction)."""
    import torch
    from wdf_hip import binding as wb
    rng = np.random.default_rng(hidden * 10 + n_tanh)
    nw = wb.lib().wdf_mlp_weight_count(hidden, n_tanh)
    w = cuda(rng.standard_normal(nw) * 0.4)
    a = cuda(rng.standard_normal(S) * 2.0)
    gb = cuda(rng.standard_normal(S) / S)
    th2 = cuda([45.0e3, 4.7e-9])
    lr = cuda(np.log(rng.uniform(100.0, 2.0e3, S))) if dyn else None
    gw = wb.clipper_mlp_wgrad(a, lr, gb, th2, w, hidden, n_tanh, FS)
    wd = w.double().requires_grad_(True)
    lr_d = lr.double() if dyn else torch.log(1.0 / (1.0 / th2[0].double() + 2.0 * th2[1].double() * FS)).expand(S)
    h, o, n_in = torch.stack([a.double(), lr_d], dim=1), 0, 2
    for _ in range(n_tanh):
        k = wd[o:o + n_in * hidden].reshape(n_in, hidden); o += n_in * hidden
        h = torch.tanh(h @ k + wd[o:o + hidden]); o += hidden
        n_in = hidden
    out = (h @ wd[o:o + hidden].reshape(hidden, 1))[:, 0] + wd[o + hidden]
    want = torch.autograd.grad(-(gb.double() * out).sum(), wd)[0]
    err = float((gw.double() - want).abs().max())
    assert err <= 2e-5 * float(want.abs().max()), (err, float(want.abs().max()))


@pytest.mark.parametrize("hidden,n_tanh", [(4, 3), (8, 3), (16, 3), (8, 4), (4, 5), (8, 5)])
@pytest.mark.parametrize("B,T,dyn", [(1, 16, False), (5, 100, True), (7, 257, False), (130, 515, True)])
def test_row_kernels_equal_lane_kernels(hidden, n_tanh, B, T, dyn):
    """The 16-lane-row kernels (default) against the one-lane-per-sequence kernels, random weights,
    ragged shapes (B not a multiple of 4, T not a multiple of 16), static and per-sample R: forward
    to 2e-6 of the state's scale, {R, C} and weight gradients to 2e-5 of their largest entry (summation order differs)."""
    from wdf_hip import binding as wb, workload
    rng = np.random.default_rng(B * 1000 + T + hidden)
    x = cuda(workload.sweep_batch(B, T, seed=3) * 0.5)
    r = cuda(workload.pot_resistance_batch(B, T)) if dyn else None
    th2 = cuda([45.0e3, 4.7e-9])
    nw = wb.lib().wdf_mlp_weight_count(hidden, n_tanh)
    w = cuda(rng.standard_normal(nw) * 0.5)
    z0 = cuda(rng.uniform(-0.2, 0.2, B))
    gy = cuda(rng.standard_normal((T, B)) / (B * T))
    try:
        wb.MLP_LANE_PER_SEQUENCE = True
        y_l, zs_l, zT_l = wb.clipper_mlp_fwd(x, th2, w, hidden, n_tanh, FS, r=r, z0=z0, want_zT=True)
        gth_l, gb, ain, lrin = wb.clipper_mlp_bwd(x, th2, w, hidden, n_tanh, FS, zs_l, gy, r=r)
        gw_l = wb.clipper_mlp_wgrad(ain, lrin, gb, th2, w, hidden, n_tanh, FS)
        wb.MLP_LANE_PER_SEQUENCE = False
        y_r, zs_r, zT_r = wb.clipper_mlp_fwd(x, th2, w, hidden, n_tanh, FS, r=r, z0=z0, want_zT=True)
        gth_r, gw_r = wb.clipper_mlp_bwd_w(x, th2, w, hidden, n_tanh, FS, zs_l, gy, r=r)
        gth_r2, gb_r, ain_r, _ = wb.clipper_mlp_bwd(x, th2, w, hidden, n_tanh, FS, zs_l, gy, r=r)
    finally:
        wb.MLP_LANE_PER_SEQUENCE = False
    tol = 2e-6 * max(1.0, float(zs_l.abs().max()))           # random 0.5-sigma weights drive |z| to a few volts
    assert float((y_r - y_l).abs().max()) <= tol and float((zs_r - zs_l).abs().max()) <= tol
    assert float((zT_r - zT_l).abs().max()) <= tol
    for a, b in ((gth_r, gth_l), (gth_r2, gth_l), (gw_r, gw_l), (gb_r, gb), (ain_r, ain)):
        assert float((a - b).abs().max()) <= 2e-5 * float(b.abs().max()) + 1e-12


@pytest.fixture(autouse=True, params=["by-batch-size", "row", "matrix-cores"])
def forward_kernel(request, monkeypatch):
    """Every test of this file with the forward kernel chosen by batch size (the default: the row kernel at these
    sizes), forced to the row kernel and forced to the matrix-core kernel (csrc/wdf_mlp_mfma.h)."""
    if request.param == "by-batch-size":
        monkeypatch.delenv("WDF_MLP_FWD_ROW", raising=False)
    else:
        monkeypatch.setenv("WDF_MLP_FWD_ROW", "1" if request.param == "row" else "0")
    return request.param


@pytest.fixture(params=["by-grid-size", "matrix-cores"])
def wgrad_kernel(request, monkeypatch):
    """The weight-gradient pass (C) of the step-parallel reverse sweep: chosen by grid size (the row kernel at the small
    batches of these tests) and forced onto the matrix-core kernel clipper_mlp_mfma_wgrad_tp_kernel (csrc/wdf_mlp_mfma.h;
    WDF_MLP_WGRAD_MFMA = its chunk count), which by itself only runs once 16-sequence waves fill half the chip."""
    if request.param == "matrix-cores":
        monkeypatch.setenv("WDF_MLP_WGRAD_MFMA", "4")
    else:
        monkeypatch.delenv("WDF_MLP_WGRAD_MFMA", raising=False)
    return request.param


def test_mlp_forward_on_matrix_cores_equals_the_row_kernel_at_a_large_batch(monkeypatch):
    """B = 12300 >= 12288: wdf_clipper_mlp_fwd runs the matrix-core kernel by itself; y, stash and final state
    equal the row kernel's (forced by WDF_MLP_FWD_ROW = 1) to 1e-5: the two sum a layer in different orders, and
    with the trained roots EACH sits 3-5e-6 from an fp64 evaluation of the recursion (tools/mlp_fwd_accuracy.py:
    row 3.5-4.2e-6, matrix cores 3.1-4.7e-6, 4.1-6.4e-6 apart) -- neither is the more accurate one."""
    from wdf_hip import binding as wb, workload
    B, T = 12300, 96
    rng = np.random.default_rng(3)
    x = cuda(np.tile(workload.sweep_batch(128, T, seed=5) * 0.5, (-(-B // 128), 1))[:B])
    r = cuda(np.tile(workload.dataset_resistance_batch(128, T), (-(-B // 128), 1))[:B])
    th2 = cuda([45.0e3, 4.7e-9])
    z0 = cuda(rng.uniform(-0.2, 0.2, B))
    for net in ("2x16", "2x8", "4x8"):
        wh, hidden, n_tanh = workload.reference_mlp_weights(net)
        w = cuda(wh)
        for rr in (r, None):
            monkeypatch.setenv("WDF_MLP_FWD_ROW", "1")
            y0, zs0, zT0 = wb.clipper_mlp_fwd(x, th2, w, hidden, n_tanh, FS, r=rr, z0=z0, want_zT=True)
            monkeypatch.delenv("WDF_MLP_FWD_ROW")
            y1, zs1, zT1 = wb.clipper_mlp_fwd(x, th2, w, hidden, n_tanh, FS, r=rr, z0=z0, want_zT=True)
            assert not torch.equal(y0, y1)                       # (another kernel did run)
            for a, b in ((y0, y1), (zs0, zs1), (zT0, zT1)):
                assert float((a - b).abs().max()) <= 1e-5, (net, float((a - b).abs().max()))


# ---- in-kernel time-parallel MLP-root kernels (csrc/wdf_mlp_tp.h) ----------------------------------------
@pytest.mark.parametrize("hidden,n_tanh", [(8, 3), (16, 3), (4, 5), (8, 5)])
@pytest.mark.parametrize("B,T,K,dyn", [(5, 100, 3, True), (7, 257, 2, False), (130, 515, 4, True), (96, 2048, 8, True)])
def test_mlp_time_parallel_kernels_equal_sequential(hidden, n_tanh, B, T, K, dyn, wgrad_kernel, monkeypatch):
    """Forward: chunks warmed up per wave (pot-dependent), verified on the device -> the sequential kernel's y,
    stash and final state to 2e-6 with a clean status.  Reverse sweep (kappa / adjoint scan / weight gradient,
    parallel over all steps): EXACT -- {R, C} and weight gradients equal the sequential sweep's to summation
    order (2e-5 of the largest entry).  Ragged shapes: B not a multiple of 4, T not a multiple of 16."""
    from wdf_hip import binding as wb, mlp_root, workload
    rng = np.random.default_rng(B * 1000 + T + hidden)
    x = cuda(workload.sweep_batch(B, T, seed=5) * 0.5)
    r = cuda(workload.pot_resistance_batch(B, T)) if dyn else None
    th2 = cuda([45.0e3, 4.7e-9])
    nw = wb.lib().wdf_mlp_weight_count(hidden, n_tanh)
    w = cuda(rng.standard_normal(nw) * 0.3)
    z0 = cuda(rng.uniform(-0.2, 0.2, B))
    gy = cuda(rng.standard_normal((T, B)) / (B * T))
    y, zs, zT = wb.clipper_mlp_fwd(x, th2, w, hidden, n_tanh, FS, r=r, z0=z0, want_zT=True)
    gth, gw = wb.clipper_mlp_bwd_w(x, th2, w, hidden, n_tanh, FS, zs, gy, r=r)
    wrow, wmax = (mlp_root.warmup_per_wave(r, 4.7e-9, FS) if dyn else (None, 448))     # random small weights: fast forgetting
    y2, zs2, zT2, st = wb.clipper_mlp_fwd_tp(x, th2, w, hidden, n_tanh, FS, K, wmax, r=r, warmup_per_wave=wrow, z0=z0,
                                             want_zT=True)
    s = wb.mlp_tp_status(st)
    assert s["n_bad"] == 0 and s["gated_waves"] == 0 and s["max_miss"] <= 1e-6, s
    scale = max(1.0, float(zs.abs().max()))
    assert float((y2 - y).abs().max()) <= 2e-6 * scale and float((zs2 - zs).abs().max()) <= 2e-6 * scale
    assert float((zT2 - zT).abs().max()) <= 2e-6 * scale
    gth2, gw2 = wb.clipper_mlp_bwd_w_tp(x, th2, w, hidden, n_tanh, FS, zs, gy, 2 * K, r=r)
    for a, b in ((gth2, gth), (gw2, gw)):
        assert float((a - b).abs().max()) <= 2e-5 * float(b.abs().max()) + 1e-12, (a, b)
    if wgrad_kernel == "matrix-cores":                       # (another kernel did run: the row pass sums in another order)
        monkeypatch.setenv("WDF_MLP_WGRAD_MFMA", "0")
        _, gw_row = wb.clipper_mlp_bwd_w_tp(x, th2, w, hidden, n_tanh, FS, zs, gy, 2 * K, r=r)
        assert not torch.equal(gw_row, gw2)
        assert float((gw_row - gw2).abs().max()) <= 2e-5 * float(gw_row.abs().max()) + 1e-12


def test_mlp_time_parallel_forward_repairs_a_short_warmup(forward_kernel):
    """A warm-up far too short for the 99.1 kOhm sequences: the verify kernel gates exactly the waves that
    missed and the gated sequential launch restores their rows -- the result is the sequential kernel's."""
    from wdf_hip import binding as wb, workload
    B, T, K = 40, 1024, 4
    x = cuda(workload.sweep_batch(B, T, seed=6) * 0.5)
    r = cuda(workload.dataset_resistance_batch(B, T))            # waves of 10k sequences converge in 32 steps, 99.1k ones do not
    th2 = cuda([45.0e3, 4.7e-9])
    wh, hidden, n_tanh = workload.reference_mlp_weights("2x16")  # a trained root: diode-like, slow to forget when off
    w = cuda(wh)
    y, zs, zT = wb.clipper_mlp_fwd(x, th2, w, hidden, n_tanh, FS, r=r, want_zT=True)
    y2, zs2, zT2, st = wb.clipper_mlp_fwd_tp(x, th2, w, hidden, n_tanh, FS, K, 32, r=r, want_zT=True)
    s = wb.mlp_tp_status(st)
    assert s["n_bad"] > 0 and 0 < s["gated_waves"] <= 10 and 0 <= s["sequential_waves"] <= s["gated_waves"], s
    # chunks on the matrix cores against the sequential ROW kernel: two fp32 evaluations, each 3-5e-6 from fp64
    tol = 1e-5 if forward_kernel == "matrix-cores" else 2e-6
    assert float((y2 - y).abs().max()) <= tol and float((zs2 - zs).abs().max()) <= tol
    assert float((zT2 - zT).abs().max()) <= tol


@pytest.mark.parametrize("hidden,n_tanh", [(8, 3), (16, 3), (8, 5)])
@pytest.mark.parametrize("B,T,K,dyn,warm", [(5, 100, 3, True, 448), (7, 257, 2, False, 448), (130, 515, 4, True, 448),
                                            (96, 2048, 8, True, 448), (40, 1024, 4, True, 32)])
def test_mlp_forward_stored_kappa_equals_the_recomputed_one(hidden, n_tanh, B, T, K, dyn, warm, wgrad_kernel):
    """wdf_clipper_mlp_fwd_tp_kappa / wdf_clipper_mlp_bwd_w_tp_kappa: the same y, stash and verdict as the plain
    pair (2e-6: the compiler contracts the two instantiations differently), and gradients equal to the recomputing sweep's to 2e-5 of the
    largest entry (kappa from the forward's registers vs from the stored stash: the same formula, a few ulps).
    warm = 32 with the dataset's resistances: some waves are re-run and take their kappa from the gated pass."""
    from wdf_hip import binding as wb, workload
    rng = np.random.default_rng(B + T + hidden)
    x = cuda(workload.sweep_batch(B, T, seed=5) * 0.5)
    r = cuda(workload.dataset_resistance_batch(B, T) if warm == 32 else workload.pot_resistance_batch(B, T)) if dyn else None
    th2 = cuda([45.0e3, 4.7e-9])
    if warm == 32:
        wh, hidden, n_tanh = workload.reference_mlp_weights("2x16")
        w = cuda(wh)
    else:
        w = cuda(rng.standard_normal(wb.lib().wdf_mlp_weight_count(hidden, n_tanh)) * 0.3)
    gy = cuda(rng.standard_normal((T, B)) / (B * T))
    y, zs, zT, st = wb.clipper_mlp_fwd_tp(x, th2, w, hidden, n_tanh, FS, K, warm, r=r, want_zT=True)
    y2, zs2, zT2, st2, kap = wb.clipper_mlp_fwd_tp(x, th2, w, hidden, n_tanh, FS, K, warm, r=r, want_zT=True, want_kappa=True)
    scale = max(1.0, float(zs.abs().max()))
    # warm = 32: the flagged waves of BOTH calls went through the chunk-local repair, which is verified to tol (1e-6
    # here) rather than exact -- two such results may sit 2 tol + rounding apart
    lim = (4e-6 if warm == 32 else 2e-6) * scale
    for a, b in ((y, y2), (zs, zs2), (zT, zT2)):
        assert float((a - b).abs().max()) <= lim, float((a - b).abs().max())
    s, s2 = wb.mlp_tp_status(st), wb.mlp_tp_status(st2)
    assert s["sequential_waves"] <= s["gated_waves"]
    assert s["gated_waves"] == s2["gated_waves"] and (s["gated_waves"] > 0) == (warm == 32), (s, s2)
    assert bool(torch.isfinite(kap).all()) and float(kap.abs().max()) < 1.0      # |dz'/dz| < 1: the circuit forgets
    gth, gw = wb.clipper_mlp_bwd_w_tp(x, th2, w, hidden, n_tanh, FS, zs, gy, 2 * K, r=r)
    gth2, gw2 = wb.clipper_mlp_bwd_w_tp(x, th2, w, hidden, n_tanh, FS, zs2, gy, 2 * K, r=r, kappa=kap)
    for a, b in ((gth2, gth), (gw2, gw)):
        assert float((a - b).abs().max()) <= 2e-5 * float(b.abs().max()) + 1e-12, (a, b)


def test_device_loss_kernels_equal_autograd_of_the_loss():
    """wdf_loss_sums -> wdf_esr_coef -> wdf_loss_esr_grad (the MLP-root training loop's loss, bench.py --root mlp*):
    loss and d loss / d y equal torch autograd of  mse + esr  past `skip` in float64 (clipper_pot.py:146-156,177,232)
    to 1e-6 relative; the rows before skip get exactly 0."""
    from wdf_hip import binding as wb
    rng = np.random.default_rng(8)
    T, B, skip = 300, 37, 50
    y = cuda(rng.standard_normal((T, B)) * 0.3)
    t = cuda(rng.standard_normal((T, B)) * 0.3)
    n, eps = float(B * (T - skip)), float(np.finfo(float).eps)
    sums = wb.loss_sums(y, t, skip)
    gcoef, loss3 = wb.esr_coef(sums, n, eps)
    gy = wb.loss_esr_grad(y, t, gcoef, skip)
    with pytest.raises(wb.WdfHipError, match="skip"):
        wb.loss_esr_grad(y, t, gcoef, T)
    yd = y.double().requires_grad_(True)
    o, tt = yd[skip:], t.double()[skip:]
    S, E = ((o - tt) ** 2).sum(), (o ** 2).sum()
    loss = S / n + torch.sqrt(S / (E + eps) / n)
    (g_ref,) = torch.autograd.grad(loss, [yd])
    assert abs(float(loss3[2]) - float(loss)) <= 1e-6 * float(loss)
    assert float(gy[:skip].abs().max()) == 0.0
    assert float((gy.double() - g_ref).abs().max()) <= 1e-6 * float(g_ref.abs().max())


def test_mlp_warm_started_chunks_follow_a_training_loop(wgrad_kernel):
    """A training loop on one batch: from the second call on every chunk starts from the previous call's state at
    its first sample (secant-extrapolated from the third), with a fraction of the cold warm-up; the weights move by
    3e-5 per step (sign steps on every weight).  Every call's y stays within 1e-5 of the sequential
    kernel's for the same weights -- arrival states are verified to 4e-6, and a learned root is not a strict
    contraction: inside the chunk the miss was seen to grow to 1.6x (6.3e-6) before it decays; the kernels themselves
    sit 3-5e-6 from fp64 -- and its weight gradient within 2e-5 of the sequential sweep's."""
    from wdf_hip import binding as wb, mlp_root, workload
    B, T = 96, 2048
    x = cuda(workload.sweep_batch(B, T, seed=11) * 0.6)
    r = cuda(workload.dataset_resistance_batch(B, T))
    th2 = cuda([45.0e3, workload.C_CLIPPER])
    wh, hidden, n_tanh = workload.reference_mlp_weights("2x16_pre")
    w = cuda(wh).requires_grad_(True)
    gy = cuda(np.random.default_rng(2).standard_normal((T, B)) / (B * T))
    plan = mlp_root.plan_mlp_time_parallel(B, T, r, None, workload.C_CLIPPER, FS)
    assert plan is not None and plan.k_fwd > 1
    mlp_root._WARM_START.clear()
    used = []
    for it in range(10):
        y, _ = mlp_root.clipper_mlp(th2, w, x, r, None, FS, hidden, n_tanh, workload.C_CLIPPER, time_parallel=plan)
        used.append(mlp_root.LAST_TP_STATUS["warmup_used"])
        (gw,) = torch.autograd.grad(y, [w], grad_outputs=gy)
        y_seq, zs, _ = wb.clipper_mlp_fwd(x, th2, w.detach(), hidden, n_tanh, FS, r=r)
        _, gw_seq = wb.clipper_mlp_bwd_w(x, th2, w.detach(), hidden, n_tanh, FS, zs, gy, r=r)
        assert float((y.detach() - y_seq).abs().max()) <= 1e-5, (it, float((y.detach() - y_seq).abs().max()))
        assert float((gw - gw_seq).abs().max()) <= 2e-5 * float(gw_seq.abs().max()), it
        with torch.no_grad():
            w -= 3.0e-5 * torch.sign(gw)
    # warm calls ran a shorter warm-up (the controller lengthens it by a quarter whenever a wave had to be re-run)
    assert used[0] >= plan.warmup and used[1] < used[0] and sorted(used[1:])[len(used) // 2] < used[0], used
    mlp_root._WARM_START.clear()


def test_mlp_clipper_auto_plan_trains_like_the_sequential_path(golden, wgrad_kernel):
    """Circuit(..., time_parallel="auto") on a dataset-shaped batch: the planner picks the in-kernel
    time-parallel kernels; y and every gradient equal the sequential path's."""
    import tf_wdf as wdf
    from tf_wdf import tf
    from layers import DenseRootModel, DenseLayer
    from wdf_hip import mlp_root, workload, binding as wb
    g = golden("g3_mlp_clipper.npz")
    B, T = 96, 2048
    xin = cuda(np.stack([workload.sweep_batch(B, T, seed=9) * 0.6, workload.pot_resistance_batch(B, T)], axis=-1))
    gy = cuda(np.random.default_rng(1).standard_normal((T, B)) / (B * T))

    def run(tp):
        Vs = wdf.ResistiveVoltageSource(45.0e3)
        C = wdf.Capacitor(float(g["C"]), FS, trainable=True)
        P1 = wdf.Parallel(Vs, C)
        model = DenseRootModel(model_json(g, "2x16"))
        circ = wdf.Circuit(P1, model, C, per_sample_R=Vs, time_parallel=tp)
        y = circ(xin)
        dense = [l for l in model.layers if isinstance(l, DenseLayer)]
        grads = tf.GradientTape().gradient(tf.reduce_sum(y * gy), [C.C] + [d.kernel for d in dense] + [d.bias for d in dense])
        return y, [gr.numpy().ravel() for gr in grads]

    mlp_root.LAST_TP_STATUS["status"] = None
    y_seq, g_seq = run(None)
    assert mlp_root.LAST_TP_STATUS["status"] is None
    y_tp, g_tp = run("auto")
    s = wb.mlp_tp_status(mlp_root.LAST_TP_STATUS["status"])
    assert s["n_bad"] == 0, s
    # the planner verifies chunk arrivals to 4e-6 (mlp_root.plan_mlp_time_parallel: above the path's own 1-2e-6
    # rounding floor), so y may sit that far from the sequential kernel's
    assert float((y_tp - y_seq).abs().max()) <= 4e-6
    for a, b in zip(g_tp, g_seq):
        assert np.max(np.abs(a - b)) <= 2e-5 * np.max(np.abs(b)) + 1e-12


# ---- the reference's own training shape: the kernels the default dispatch picks there -----------------------------
def _oracle_mlp_problem(oracle, net, pick, x, r, C):
    from wdf_hip import workload
    wh, hidden, n_tanh = workload.reference_mlp_weights(net)
    sizes, acts = [2] + [hidden] * n_tanh + [1], [oracle.ACT_TANH] * n_tanh + [oracle.ACT_NONE]
    oc = oracle.clipper_mlp_circuit(FS, sizes, acts)
    theta = np.concatenate([[45.0e3, float(np.float32(C))], wh.astype(np.float32).astype(np.float64)])
    xin = np.stack([x[pick].astype(np.float64), r[pick].astype(np.float64)], axis=-1)
    return oc, theta, xin


def mlp_components(hidden, n_tanh, stride=4):
    """Indices into the flat weight vector: every bias, the whole first and last layer, every `stride`-th entry of the
    hidden kernels (the oracle differentiates by complex step, one pass per component)."""
    idx, o, n_in = [], 0, 2
    for layer in range(n_tanh):
        nk = n_in * hidden
        idx += list(range(o, o + nk, 1 if layer == 0 else stride))
        idx += list(range(o + nk, o + nk + hidden))
        o += nk + hidden
        n_in = hidden
    idx += list(range(o, o + hidden + 1))
    return np.array(idx)


@pytest.mark.parametrize("net", ["2x16_pre", "2x8"])
def test_training_step_at_the_reference_shape_against_the_oracle(oracle, forward_kernel, monkeypatch, net):
    """clipper_pot.py's training set: 1340 sequences x 2048 samples, the pot value per sample (clipper_pot.py:58,94-127,
    245-269).  At THIS shape the default dispatch leaves the row kernels: for the 2x16 net the forward's 12 chunks run on
    the matrix cores (clipper_mlp_mfma_fwd_tp_kernel) and, for every net, the weight-gradient pass of the reverse sweep
    does (clipper_mlp_mfma_wgrad_tp_kernel: 84 x 22 waves of 16 sequences).  Checked: (1) those kernels did run (their
    results are not bit-equal to the row kernels'); (2) y, dL/dC and dL/dw of 8 picked sequences (two per pot value)
    against the fp64 oracle -- ORC_ROOT_MLP, complex-step derivative per component; (3) the whole-batch gradient against
    the sequential row sweep (another kernel family, no chunks, no stored kappa)."""
    if forward_kernel != "by-batch-size":
        pytest.skip("this test pins what the DEFAULT dispatch runs")
    from wdf_hip import binding as wb, mlp_root, workload
    monkeypatch.delenv("WDF_MLP_WGRAD_MFMA", raising=False)
    B, T, C = 1340, 2048, workload.C_CLIPPER
    x = workload.sweep_batch(B, T, seed=4) * 0.6
    r = workload.dataset_resistance_batch(B, T)
    xd, rd = cuda(x), cuda(r)
    wh, hidden, n_tanh = workload.reference_mlp_weights(net)
    th2 = cuda([45.0e3, C]).requires_grad_(True)
    w = cuda(wh).requires_grad_(True)
    plan = mlp_root.plan_mlp_time_parallel(B, T, rd, None, C, FS, hidden=hidden, n_tanh=n_tanh)
    assert plan is not None and plan.k_bwd == 6 and plan.k_fwd == (12 if hidden == 16 else 6), plan
    pick = np.array([3, 200, 401, 640, 700, 1000, 1100, 1339])
    rng = np.random.default_rng(12)
    gy_pick = np.zeros((T, B), dtype=np.float32)
    gy_pick[:, pick] = rng.standard_normal((T, len(pick))) / (len(pick) * T)
    gy_all = cuda(rng.standard_normal((T, B)) / (B * T))
    mlp_root._WARM_START.clear()
    y, _ = mlp_root.clipper_mlp(th2, w, xd, rd, None, FS, hidden, n_tanh, C, time_parallel=plan)
    s = wb.mlp_tp_status(mlp_root.LAST_TP_STATUS["status"])
    assert s["n_bad"] == 0 or s["sequential_waves"] <= s["gated_waves"], s
    gth_p, gw_p = torch.autograd.grad(y, [th2, w], grad_outputs=cuda(gy_pick), retain_graph=True)
    gth_a, gw_a = torch.autograd.grad(y, [th2, w], grad_outputs=gy_all)
    mlp_root._WARM_START.clear()
    # (1) which kernels: the sequential row forward / the row weight-gradient pass give other bits
    y_seq, zs_seq, _ = wb.clipper_mlp_fwd(xd, th2.detach(), w.detach(), hidden, n_tanh, FS, r=rd)
    e_seq = float((y.detach() - y_seq).abs().max())
    if hidden == 16:
        assert not torch.equal(y.detach(), y_seq)
    monkeypatch.setenv("WDF_MLP_WGRAD_MFMA", "0")
    gth_r, gw_r = wb.clipper_mlp_bwd_w_tp(xd, th2.detach(), w.detach(), hidden, n_tanh, FS, zs_seq, gy_all, plan.k_bwd, r=rd)
    monkeypatch.delenv("WDF_MLP_WGRAD_MFMA")
    gth_m, gw_m = wb.clipper_mlp_bwd_w_tp(xd, th2.detach(), w.detach(), hidden, n_tanh, FS, zs_seq, gy_all, plan.k_bwd, r=rd)
    assert not torch.equal(gw_m, gw_r)
    e_mr = float((gw_m - gw_r).abs().max() / gw_r.abs().max())
    # (3) the whole batch against the sequential sweep
    gth_s, gw_s = wb.clipper_mlp_bwd_w(xd, th2.detach(), w.detach(), hidden, n_tanh, FS, zs_seq, gy_all, r=rd)
    e_ws = float((gw_a - gw_s).abs().max() / gw_s.abs().max())
    e_cs = abs(float(gth_a[1] - gth_s[1])) / abs(float(gth_s[1]))
    # (2) the oracle on the picked sequences
    oc, theta, xin = _oracle_mlp_problem(oracle, net, pick, x, r, C)
    y_ref = oracle.tree_fwd(oc, theta, xin)
    e_y = float(np.max(np.abs(y.detach()[:, pick].cpu().numpy() - y_ref)))
    comp = mlp_components(hidden, n_tanh)
    g_ref = oracle.tree_grad(oc, theta, xin, gy_pick[:, pick].astype(np.float64), params=[1] + list(2 + comp))
    got = np.concatenate([[float(gth_p[1])], gw_p.cpu().numpy()[comp].astype(np.float64)])
    e_g = float(np.max(np.abs(got - g_ref) / (np.abs(g_ref) + 1e-1 * np.max(np.abs(g_ref[1:])))))
    print(f"{net}: |y - seq| {e_seq:.2e}  |y - oracle| {e_y:.2e}  grad vs oracle {e_g:.2e} ({len(comp) + 1} components)  "
          f"matrix-core vs row wgrad {e_mr:.2e}  whole batch vs sequential sweep: w {e_ws:.2e}, C {e_cs:.2e}")
    assert e_seq <= 1e-5 and e_y <= 1e-5                       # each fp32 path sits 3-5e-6 from fp64 with the trained roots
    assert e_mr <= 1e-5 and e_ws <= 2e-5 and e_cs <= 1e-4          # measured 7e-7, 3e-6, 1.2e-5
    # per component: relative to its own size, plus 1e-5 of the largest weight-gradient component (as the g3 test)
    assert np.all(np.abs(got[1:] - g_ref[1:]) <= 1e-4 * np.abs(g_ref[1:]) + 1e-5 * np.max(np.abs(g_ref[1:]))), e_g
    assert abs(got[0] - g_ref[0]) <= 3e-3 * abs(g_ref[0]), (got[0], g_ref[0])   # dL/dC: as test_static_resistance_and_capacitor_gradient

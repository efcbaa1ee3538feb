"""GPU: the resident training step of the MLP-root pot clipper (csrc/wdf_mlp_step.h, wdf_clipper_mlp_step) -- what one
epoch of clipper_pot.py:245-269 does (ClipperModel.forward :103-127, MSE + ESR past skip_samples :141-177 with the
(outs, target) swap :248, gradient to the DenseRootModel weights layers.py:72-82, Adam :180) in five launches --
against the fp64 oracle (tree interpreter, MLP root, complex-step gradient) and against the sequential kernels."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

FS = 48000
EPS = float(np.finfo(float).eps)


def cuda(a, dtype=np.float32):
    return torch.as_tensor(np.ascontiguousarray(a, dtype=dtype), device="cuda")


def problem(B, T, net, seed=4, dyn=True, amp=0.6):
    from wdf_hip import binding as wb, workload
    x = workload.sweep_batch(B, T, seed=seed) * amp
    r = workload.dataset_resistance_batch(B, T) if dyn else None
    wh, hidden, n_layers = workload.reference_mlp_weights(net)
    xd = cuda(x)
    rd = None if r is None else cuda(r)
    th4 = cuda(workload.clipper_theta())
    target, _, _ = wb.clipper_fwd(xd, th4, FS, r=rd, want_stash=False)       # the analytic diode pair "measured" it
    return x, r, xd, rd, wh, hidden, n_layers, target


def loss_and_gy(y, target, skip, n_global):
    """clipper_pot.py:146-156,177 in float64 on the device: (mse, esr), dLoss/dy [T,B]."""
    o, t = y.double()[skip:], target.double()[skip:]
    S, E = ((o - t) ** 2).sum(), (o ** 2).sum() + EPS
    esr = torch.sqrt(S / E / n_global)
    ga, gb = 2.0 / n_global + 1.0 / (esr * E * n_global), -esr / E
    gy = torch.zeros_like(y, dtype=torch.float64)
    gy[skip:] = ga * (o - t) + gb * o
    return float(S / n_global), float(esr), gy.float().contiguous()


def sequential_reference(xd, rd, w, hidden, n_layers, target, skip, C):
    """The same step through the sequential row kernels (no chunks, no matrix cores) and torch for the loss."""
    from wdf_hip import binding as wb
    th2 = cuda([45.0e3, C])
    y, zs, _ = wb.clipper_mlp_fwd(xd, th2, w, hidden, n_layers, FS, r=rd)
    n = float(y.shape[1] * (y.shape[0] - skip))
    mse, esr, gy = loss_and_gy(y, target, skip, n)
    _, gw = wb.clipper_mlp_bwd_w(xd, th2, w, hidden, n_layers, FS, zs, gy, r=rd)
    return y, mse, esr, gy, gw


def mlp_components(hidden, n_layers, stride=4):
    idx, o, n_in = [], 0, 2
    for layer in range(n_layers):
        nk = n_in * hidden
        idx += list(range(o, o + nk, 1 if layer == 0 else stride))
        idx += list(range(o + nk, o + nk + hidden))
        o += nk + hidden
        n_in = hidden
    return np.array(idx + list(range(o, o + hidden + 1)))


@pytest.mark.parametrize("net,dyn", [("2x16_pre", True), ("2x16", False), ("2x8", True), ("4x8", True), ("2x4", True)])
def test_step_against_the_oracle_and_the_sequential_kernels(oracle, net, dyn):
    from wdf_hip import mlp_root, workload
    B, T, skip, C = 40, 512, 50, workload.C_CLIPPER                           # 2 full columns + a ragged one
    x, r, xd, rd, wh, hidden, n_layers, target = problem(B, T, net, dyn=dyn)
    w = cuda(wh)
    st = mlp_root.MlpTrainStep(xd, rd, target, w, hidden, n_layers, FS, C, R_static=45.0e3, skip=skip, adam=None,
                               n_items=12, wgrad_chunks=4)
    assert st.n_items == 12 and st.items[:, 3].max() == T
    y_seq, mse_s, esr_s, gy, gw_seq = sequential_reference(xd, rd, w, hidden, n_layers, target, skip, C)
    for call in range(3):                                                   # cold, then twice from the snapshots
        st.step()
        torch.cuda.synchronize()
        info, wc = st.read()
        assert info["calls"] == call + 1 and info["sequential_columns"] == 0, info
        e_y = float((st.y - y_seq).abs().max())
        e_g = float((st.gw - gw_seq).abs().max() / gw_seq.abs().max())
        l3 = st.loss3.cpu().numpy()
        print(f"{net} call {call}: |y - seq| {e_y:.2e}  gw vs seq {e_g:.2e}  loss {l3}  verdict {info}  W {wc}")
        assert e_y <= 1e-5 and e_g <= 5e-5
        # the loss arithmetic itself: the same formulas in float64 on the step's own y
        mse_o, esr_o, gy_o = loss_and_gy(st.y, target, skip, float(B * (T - skip)))
        assert abs(l3[0] - mse_o) <= 2e-6 * mse_o and abs(l3[1] - esr_o) <= 2e-6 * esr_o and abs(l3[2] - (mse_o + esr_o)) <= 2e-6 * (mse_o + esr_o)
        assert abs(l3[2] - (mse_s + esr_s)) <= 1e-3 * (mse_s + esr_s)          # (two fp32 forwards 4e-6 apart)
        # ... and the reverse sweep alone: the sequential sweep fed with the step's own dLoss/dy
        from wdf_hip import binding as wb
        _, gw_own = wb.clipper_mlp_bwd_w(xd, cuda([45.0e3, C]), w, hidden, n_layers, FS, st.zstash, gy_o, r=rd)
        e_go = float((st.gw - gw_own).abs().max() / gw_own.abs().max())
        print(f"    reverse sweep alone vs the sequential sweep: {e_go:.2e}")
        assert e_go <= 1e-5
    # the oracle: y of every sequence, the whole-batch gradient of the real loss on a spread of components
    sizes, acts = [2] + [hidden] * n_layers + [1], [oracle.ACT_TANH] * n_layers + [oracle.ACT_NONE]
    oc = oracle.clipper_mlp_circuit(FS, sizes, acts)
    theta = np.concatenate([[45.0e3, float(np.float32(C))], wh.astype(np.float32).astype(np.float64)])
    rr = r if r is not None else np.full((B, T), 45.0e3)
    xin = np.stack([x.astype(np.float64), rr.astype(np.float64)], axis=-1)
    y_ref = oracle.tree_fwd(oc, theta, xin)
    e_yo = float(np.max(np.abs(st.y.cpu().numpy() - y_ref)))
    o, t = y_ref[skip:], target.cpu().numpy().astype(np.float64)[skip:]
    n = float(B * (T - skip))
    S, E = np.sum((o - t) ** 2), np.sum(o ** 2) + EPS
    esr = np.sqrt(S / E / n)
    gy_ref = np.zeros_like(y_ref)
    gy_ref[skip:] = (2.0 / n + 1.0 / (esr * E * n)) * (o - t) - esr / E * o
    comp = mlp_components(hidden, n_layers)
    g_ref = oracle.tree_grad(oc, theta, xin, gy_ref, params=list(2 + comp))
    got = st.gw.cpu().numpy().astype(np.float64)[comp]
    e_go = float(np.max(np.abs(got - g_ref) / (np.abs(g_ref) + 1e-1 * np.max(np.abs(g_ref)))))
    print(f"{net}: |y - oracle| {e_yo:.2e}  loss vs oracle {abs(st.loss3[2].item() - (S / n + esr)):.2e}  grad vs oracle {e_go:.2e}")
    assert e_yo <= 1e-5
    assert abs(st.loss3[2].item() - (S / n + esr)) <= 2e-4 * (S / n + esr)     # (fp32 y, 3e-6 from fp64, against residuals of 1e-2)
    assert np.all(np.abs(got - g_ref) <= 2e-4 * np.abs(g_ref) + 2e-5 * np.max(np.abs(g_ref))), e_go


def test_step_training_loop_follows_the_sequential_loop():
    """40 Adam steps (clipper_pot.py:180: Adam(1e-4, beta_1 0.5)): the fused step's weights track a loop built from the
    sequential kernels, the controller moves the per-column warm-ups, nothing goes sequential."""
    from wdf_hip import binding as wb, mlp_root, workload
    B, T, skip, C = 72, 1024, 50, workload.C_CLIPPER
    x, r, xd, rd, wh, hidden, n_layers, target = problem(B, T, "2x16_pre")
    w = cuda(wh)
    adam = wb.Adam(w.numel(), lr=1.0e-4, beta_1=0.5, device="cuda")
    st = mlp_root.MlpTrainStep(xd, rd, target, w, hidden, n_layers, FS, C, skip=skip, adam=adam, n_items=40, wgrad_chunks=8)
    w2 = cuda(wh)
    adam2 = wb.Adam(w2.numel(), lr=1.0e-4, beta_1=0.5, device="cuda")
    _, w_first = st.read()
    losses = []
    for i in range(40):
        st.step()
        _, mse2, esr2, _, gw = sequential_reference(xd, rd, w2, hidden, n_layers, target, skip, C)
        adam2.apply(w2, gw)
        losses.append(float(st.loss3[2]))
        assert abs(losses[-1] - (mse2 + esr2)) <= 5e-3 * (mse2 + esr2)          # the loss curve is the sequential loop's
        if i == 20:
            assert st.replan()                                              # a re-plan in the middle of training
    torch.cuda.synchronize()
    info, w_last = st.read()
    dw = float((w - w2).abs().max())
    moved = float((w - cuda(wh)).abs().max())
    print(f"|w - w_seq| {dw:.2e} of a total move {moved:.2e}; loss {losses[0]:.4e} -> {losses[-1]:.4e}; W {w_first} -> {w_last}; {info}")
    assert int(adam.step) == 40 and info["calls"] == 40
    assert moved > 5e-4 and dw <= 2e-2 * moved                             # (Adam's sign-like steps amplify rounding differences)
    assert info["total_sequential"] == 0
    assert not np.array_equal(w_first, w_last)


def test_step_repairs_what_a_short_warm_up_misses():
    """Warm-ups pinned at one unit and the weights pushed between calls: chunk boundaries miss, the flagged chunks are
    re-run (and, where a re-run chunk no longer ends where it did, the column goes sequential) -- the results stay the
    sequential ones."""
    from wdf_hip import mlp_root, workload
    B, T, skip, C = 64, 1024, 50, workload.C_CLIPPER
    x, r, xd, rd, wh, hidden, n_layers, target = problem(B, T, "2x16_pre", amp=1.0)
    w = cuda(wh)
    st = mlp_root.MlpTrainStep(xd, rd, target, w, hidden, n_layers, FS, C, skip=skip, adam=None, n_items=64, wgrad_chunks=8)
    st.step()                                                               # cold call: leaves snapshots
    st.set_wcol(np.ones(st.ncol, dtype=np.int32))
    st.freeze(True)
    rng = np.random.default_rng(5)
    w += cuda(0.02 * rng.standard_normal(w.numel()))
    st.step()
    torch.cuda.synchronize()
    info, wc = st.read()
    y_seq, mse_s, esr_s, gy, gw_seq = sequential_reference(xd, rd, w, hidden, n_layers, target, skip, C)
    e_y = float((st.y - y_seq).abs().max())
    e_g = float((st.gw - gw_seq).abs().max() / gw_seq.abs().max())
    print(f"{info}  |y - seq| {e_y:.2e}  gw {e_g:.2e}")
    assert info["n_bad"] > 0 and info["flagged_columns"] > 0
    assert np.all(wc == 1)                                                   # frozen
    assert e_y <= 1e-5 and e_g <= 3e-5
    assert abs(float(st.loss3[2]) - (mse_s + esr_s)) <= 2e-5 * (mse_s + esr_s)
    # and the call after it starts from repaired snapshots
    st.freeze(False)
    st.step()
    torch.cuda.synchronize()
    assert float((st.y - y_seq).abs().max()) <= 1e-5


def test_step_in_phases_equals_the_whole_step():
    """The multi-rank split (forward + sums | reverse sweep with the global sums | separate Adam) gives the single
    call's bits when the 'all-reduce' is the identity."""
    from wdf_hip import binding as wb, mlp_root, workload
    B, T, skip, C = 48, 512, 50, workload.C_CLIPPER
    x, r, xd, rd, wh, hidden, n_layers, target = problem(B, T, "2x16_pre")
    runs = []
    for split in (False, True):
        w = cuda(wh)
        adam = wb.Adam(w.numel(), lr=1.0e-4, beta_1=0.5, device="cuda")
        calls = []
        st = mlp_root.MlpTrainStep(xd, rd, target, w, hidden, n_layers, FS, C, skip=skip, adam=adam, n_items=12, wgrad_chunks=4,
                                   sums_allreduce=(lambda t: calls.append(tuple(t.shape))) if split else None,
                                   grad_allreduce=(lambda t: calls.append(tuple(t.shape))) if split else None)
        for _ in range(4):
            st.step()
        torch.cuda.synchronize()
        runs.append((w.clone(), st.y.clone(), st.gw.clone(), st.loss3.clone()))
        if split:
            assert calls == [(2,), (w.numel(),)] * 4
    for a, b in zip(*runs):
        assert torch.equal(a, b)


def test_step_replayed_as_a_hip_graph_equals_eager_launches():
    from wdf_hip import binding as wb, mlp_root, workload
    B, T, skip, C = 48, 512, 50, workload.C_CLIPPER
    x, r, xd, rd, wh, hidden, n_layers, target = problem(B, T, "2x16_pre")
    out = []
    for graph in (False, True):
        w = cuda(wh)
        adam = wb.Adam(w.numel(), lr=1.0e-4, beta_1=0.5, device="cuda")
        st = mlp_root.MlpTrainStep(xd, rd, target, w, hidden, n_layers, FS, C, skip=skip, adam=adam, n_items=12, wgrad_chunks=4)
        st.step()
        if graph:
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                with torch.cuda.graph(g, stream=side):
                    st.step()
            torch.cuda.current_stream().wait_stream(side)
            for _ in range(5):
                g.replay()
        else:
            for _ in range(5):                                               # (the capture itself launches nothing)
                st.step()
        torch.cuda.synchronize()
        info, _ = st.read()
        assert info["calls"] == 6
        out.append((w.clone(), st.y.clone(), st.loss3.clone()))
    for a, b in zip(*out):
        assert torch.equal(a, b)


def test_step_with_relu_hidden_layers(oracle):
    """layers.py:63-65 accepts activation "relu": the step's kernels carry it as a template flag."""
    from wdf_hip import mlp_root, workload
    B, T, skip, C, hidden, n_layers = 32, 256, 50, workload.C_CLIPPER, 8, 3
    rng = np.random.default_rng(3)
    count = 3 * hidden + (n_layers - 1) * (hidden * hidden + hidden) + hidden + 1
    wh = (0.4 * rng.standard_normal(count)).astype(np.float32)
    x, r, xd, rd, _, _, _, target = problem(B, T, "2x8")
    w = cuda(wh)
    st = mlp_root.MlpTrainStep(xd, rd, target, w, hidden, n_layers, FS, C, skip=skip, adam=None, activation="relu",
                               n_items=8, wgrad_chunks=4)
    st.step()
    st.step()
    torch.cuda.synchronize()
    sizes, acts = [2] + [hidden] * n_layers + [1], [oracle.ACT_RELU] * n_layers + [oracle.ACT_NONE]
    oc = oracle.clipper_mlp_circuit(FS, sizes, acts)
    theta = np.concatenate([[45.0e3, float(np.float32(C))], wh.astype(np.float64)])
    xin = np.stack([x.astype(np.float64), r.astype(np.float64)], axis=-1)
    y_ref = oracle.tree_fwd(oc, theta, xin)
    o, t = y_ref[skip:], target.cpu().numpy().astype(np.float64)[skip:]
    n = float(B * (T - skip))
    S, E = np.sum((o - t) ** 2), np.sum(o ** 2) + EPS
    esr = np.sqrt(S / E / n)
    gy_ref = np.zeros_like(y_ref)
    gy_ref[skip:] = (2.0 / n + 1.0 / (esr * E * n)) * (o - t) - esr / E * o
    comp = mlp_components(hidden, n_layers, stride=2)
    g_ref = oracle.tree_grad(oc, theta, xin, gy_ref, params=list(2 + comp))
    got = st.gw.cpu().numpy().astype(np.float64)[comp]
    e_y = float(np.max(np.abs(st.y.cpu().numpy() - y_ref)))
    e_g = float(np.max(np.abs(got - g_ref) / (np.abs(g_ref) + 1e-1 * np.max(np.abs(g_ref)))))
    print(f"relu: |y - oracle| {e_y:.2e}  grad vs oracle {e_g:.2e}  {st.read()[0]}")
    assert e_y <= 1e-5
    # (a relu kink crossed by fp32 rounding moves a sample's gradient: looser than tanh)
    assert np.all(np.abs(got - g_ref) <= 2e-3 * np.abs(g_ref) + 2e-4 * np.max(np.abs(g_ref))), e_g


def test_step_rejects_what_it_does_not_cover():
    from wdf_hip import binding as wb, mlp_root, workload
    x, r, xd, rd, wh, hidden, n_layers, target = problem(16, 40, "2x16_pre")   # T not a multiple of 16
    with pytest.raises(wb.WdfHipError):
        mlp_root.MlpTrainStep(xd, rd, target, cuda(wh), hidden, n_layers, FS, workload.C_CLIPPER)


def _clipper_pot_model(name="2x16_pre"):
    """clipper_pot.py:94-101 with a committed reference network (weights from the golden fixture)."""
    import tf_wdf as wdf
    from layers import DenseRootModel
    from wdf_hip import workload
    wh, hidden, n_layers = workload.reference_mlp_weights(name)
    layers_json, o, n_in = [], 0, 2
    sizes = [2] + [hidden] * n_layers + [1]
    for i in range(len(sizes) - 1):
        ni, no = sizes[i], sizes[i + 1]
        k = wh[o:o + ni * no].reshape(ni, no); o += ni * no
        b = wh[o:o + no]; o += no
        layers_json.append({"type": "dense", "activation": "tanh" if i < len(sizes) - 2 else "", "shape": [None, no],
                            "weights": [k.tolist(), b.tolist()]})
    Vs = wdf.ResistiveVoltageSource(45.0e3)
    C = wdf.Capacitor(workload.C_CLIPPER, FS)
    P1 = wdf.Parallel(Vs, C)
    model = DenseRootModel({"in_shape": [None, 2], "layers": layers_json})
    return wdf, wdf.Circuit(P1, model, C, per_sample_R=Vs), model


def test_clipper_pot_loop_through_the_element_api_on_the_resident_step():
    """clipper_pot.py:245-269 as the script writes it -- GradientTape, the MSE + ESR loss past 50 samples, tape.gradient over
    model.trainable_variables, Adam(1e-4, beta_1 0.5).apply_gradients -- on a circuit moved to the device: every epoch is
    one resident training step + one Adam launch; the weights follow the loop on the plain path."""
    from wdf_hip import workload, binding as wb
    B, T, skip = 96, 1024, 50
    x = workload.sweep_batch(B, T, seed=4) * 0.6
    r = workload.dataset_resistance_batch(B, T)
    xin = cuda(np.stack([x, r], axis=-1))                       # [B,T,2]: Vin, R (clipper_pot.py:68-70)
    target, _, _ = wb.clipper_fwd(cuda(x), cuda(workload.clipper_theta()), FS, r=cuda(r), want_stash=False)
    ends = []
    for resident in (False, True):
        wdf, circ, model = _clipper_pot_model()
        tf = wdf.tf
        if resident:
            circ.to_device()
        opt = tf.keras.optimizers.Adam(learning_rate=1.0e-4, beta_1=0.5)          # clipper_pot.py:180
        tv = model.trainable_variables
        assert len(tv) == 8
        w_init = torch.cat([v.as_subclass(torch.Tensor).detach().reshape(-1).cuda() for v in tv]).clone()
        losses = []
        for _ in range(12):
            with tf.GradientTape() as tape:
                loss = circ.mse_esr(xin, target, skip)
            grads = tape.gradient(loss, tv)
            opt.apply_gradients(zip(grads, tv))
            losses.append(float(loss))
        if resident:
            assert all(v.is_cuda for v in tv) and ("flat", id(circ._mlp)) in opt._resident      # one Adam launch per epoch
            assert next(iter(circ._mlp.cache.values()))["st"].read()[0]["calls"] == 12
            y_val = circ(xin)                                    # a validation forward reads the same (moved) weights
            assert y_val.shape == (T, B)
        ends.append((torch.cat([v.as_subclass(torch.Tensor).detach().reshape(-1).cuda() for v in tv]), losses, w_init))
    (w0, l0, wi), (w1, l1, _) = ends
    moved = float((w0 - wi).abs().max())
    print(f"loss {l0[0]:.5e} -> {l0[-1]:.5e} (plain) / {l1[-1]:.5e} (resident); |w - w'| {float((w0 - w1).abs().max()):.2e} of {moved:.2e}")
    assert all(abs(a - b) <= 5e-3 * a for a, b in zip(l0, l1))
    assert float((w0 - w1).abs().max()) <= 0.05 * moved


def test_step_with_relu_network_against_the_reference_golden(golden):
    """g9: the reference's own ClipperModel (clipper_pot.py:94-127) run on a ReLU network (layers.py:63-67; the pretrained
    2x16 JSON with its activations switched and its weights halved), loss_func and tape.gradient of clipper_pot.py:246-249 --
    outputs, loss and the gradient of every weight from the resident step."""
    from wdf_hip import mlp_root
    g = golden("g9_mlp_relu.npz")
    name = "2x16_relu"
    x, r = g["x"][:, :, 0], g["x"][:, :, 1]
    target = cuda(g["target"][:, :, 0].T)                                   # [T,B]
    sizes = [int(v) for v in g[f"{name}_sizes"]]
    assert [int(a) for a in g[f"{name}_acts"]] == [2] * (len(sizes) - 2) + [0]
    hidden, n_layers, skip = sizes[1], len(sizes) - 2, int(g["skip"])
    w = cuda(g[f"{name}_theta"])
    st = mlp_root.MlpTrainStep(cuda(x), cuda(r), target, w, hidden, n_layers, FS, float(g["C"]), skip=skip, adam=None,
                               activation="relu", n_items=4, wgrad_chunks=4)
    for call in range(2):                                                   # cold, then from the snapshots
        st.step()
    torch.cuda.synchronize()
    y, gw, loss = st.y.cpu().numpy(), st.gw.cpu().numpy().astype(np.float64), float(st.loss3[2])
    y64, g64, l64 = g[f"{name}_y_f64"], g[f"{name}_grad_f64"], float(g[f"{name}_loss_f64"])
    e_y = float(np.max(np.abs(y - y64)))
    e_g = float(np.max(np.abs(gw - g64)) / np.max(np.abs(g64)))
    print(f"relu golden: |y - reference| {e_y:.2e} (the reference's own f32 run: {np.max(np.abs(g[f'{name}_y_f32'] - y64)):.2e}), "
          f"loss {loss:.7f} / {l64:.7f}, gradient {e_g:.2e} (reference f32: "
          f"{np.max(np.abs(g[f'{name}_grad_f32'] - g64)) / np.max(np.abs(g64)):.2e})")
    assert e_y <= 1e-5 and abs(loss - l64) <= 1e-5 * l64
    # (a relu kink crossed by fp32 rounding moves a sample's gradient: looser than tanh)
    assert np.all(np.abs(gw - g64) <= 2e-3 * np.abs(g64) + 2e-4 * np.max(np.abs(g64))), e_g


def test_relu_network_through_the_element_api_against_the_reference_golden(golden):
    """The drop-in surface on a ReLU DenseRootModel (layers.py:63-67): Circuit.to_device() + circ.mse_esr + tape.gradient over
    model.trainable_variables -- loss and every weight's gradient against g9 (the reference's ClipperModel executed)."""
    import tf_wdf as wdf
    from layers import DenseRootModel
    tf = wdf.tf
    g = golden("g9_mlp_relu.npz")
    name = "2x16_relu"
    sizes = [int(v) for v in g[f"{name}_sizes"]]
    wh, o, layers_json = g[f"{name}_theta"].astype(np.float32), 0, []
    for i in range(len(sizes) - 1):
        ni, no = sizes[i], sizes[i + 1]
        k = wh[o:o + ni * no].reshape(ni, no); o += ni * no          # noqa: E702
        b = wh[o:o + no]; o += no                                    # noqa: E702
        layers_json.append({"type": "dense", "activation": "relu" if i < len(sizes) - 2 else "", "shape": [None, no],
                            "weights": [k.tolist(), b.tolist()]})
    Vs, C = wdf.ResistiveVoltageSource(45.0e3), wdf.Capacitor(float(g["C"]), FS)
    model = DenseRootModel({"in_shape": [None, 2], "layers": layers_json})
    circ = wdf.Circuit(wdf.Parallel(Vs, C), model, C, per_sample_R=Vs).to_device()
    tv = model.trainable_variables
    with tf.GradientTape() as tape:
        loss = circ.mse_esr(cuda(g["x"]), cuda(g["target"][:, :, 0].T), int(g["skip"]))
    grads = tape.gradient(loss, tv)
    # golden order: kernel, bias per layer (trainable_variables lists bias first)
    dl = [l for l in model.layers if hasattr(l, "kernel")]
    got = np.concatenate([np.concatenate([next(gr for gr, v in zip(grads, tv) if v is l.kernel).detach().cpu().numpy().ravel(),
                                          next(gr for gr, v in zip(grads, tv) if v is l.bias).detach().cpu().numpy().ravel()]) for l in dl])
    g64, l64 = g[f"{name}_grad_f64"], float(g[f"{name}_loss_f64"])
    print(f"element API, relu: loss {float(loss):.7f} / {l64:.7f}, gradient {np.max(np.abs(got - g64)) / np.max(np.abs(g64)):.2e}")
    assert abs(float(loss) - l64) <= 1e-5 * l64
    assert np.all(np.abs(got - g64) <= 2e-3 * np.abs(g64) + 2e-4 * np.max(np.abs(g64)))
    assert float(np.max(np.abs(circ.last_output.detach().cpu().numpy() - g[f"{name}_y_f64"]))) <= 1e-5


def test_relu_network_forward_call_against_the_reference_golden(golden):
    """circ(x) on a ReLU DenseRootModel with host-resident weights: the forward phase of the resident step's kernels; and the
    gradient is refused there with a pointer to mse_esr (the plain path's kernels are tanh-only)."""
    import tf_wdf as wdf
    from layers import DenseRootModel
    from wdf_hip import binding as wb
    g = golden("g9_mlp_relu.npz")
    name = "2x16_relu"
    sizes = [int(v) for v in g[f"{name}_sizes"]]
    wh, o, layers_json = g[f"{name}_theta"].astype(np.float32), 0, []
    for i in range(len(sizes) - 1):
        ni, no = sizes[i], sizes[i + 1]
        k = wh[o:o + ni * no].reshape(ni, no); o += ni * no          # noqa: E702
        b = wh[o:o + no]; o += no                                    # noqa: E702
        layers_json.append({"type": "dense", "activation": "relu" if i < len(sizes) - 2 else "", "shape": [None, no],
                            "weights": [k.tolist(), b.tolist()]})
    Vs, C = wdf.ResistiveVoltageSource(45.0e3), wdf.Capacitor(float(g["C"]), FS)
    circ = wdf.Circuit(wdf.Parallel(Vs, C), DenseRootModel({"in_shape": [None, 2], "layers": layers_json}), C, per_sample_R=Vs)
    with torch.no_grad():
        y = circ(cuda(g["x"]))
    assert tuple(y.shape) == g[f"{name}_y_f64"].shape
    assert float(np.max(np.abs(y.cpu().numpy() - g[f"{name}_y_f64"]))) <= 1e-5
    with pytest.raises(wb.WdfHipError):
        circ(cuda(g["x"]))

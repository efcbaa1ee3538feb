"""CPU: the probed step as a tape (lib/wdf_hip/probe_tape.py) against the host probe (lowering.Circuit.matrices: the elements'
own calc_impedance / reflected / incident code on float64 torch scalars with autograd) -- values and Jacobian; and the
chunk planner of the MLP-root training step (mlp_root.plan_step_items)."""
import math

import numpy as np
import pytest

torch = pytest.importorskip("torch")

FS = 48000


def _lpf(wdf):
    Vs, R1, C1 = wdf.IdealVoltageSource(), wdf.Resistor(1000.0, True), wdf.Capacitor(1.0e-6, FS, True)
    return wdf.Circuit(wdf.Inverter(wdf.Series(R1, C1)), Vs, C1), [C1.C, R1.R]            # lpf.py:20-28


def _divider(wdf):
    R1, R2 = wdf.Resistor(2.0e3, True), wdf.Resistor(100.0, True)
    return wdf.Circuit(wdf.Inverter(wdf.Series(R1, R2)), wdf.IdealVoltageSource(), R1), [R1.R, R2.R]   # voltage_divider.py:19-26


def _hpf(wdf):
    R, Vs, C = wdf.Resistor(33.0e3, True), wdf.ResistiveVoltageSource(1.0e3, trainable=True), wdf.Capacitor(22.0e-9, FS, True)
    top = wdf.Parallel(R, wdf.Series(Vs, C))                                              # HPFDiodeClipper.h:28-32
    return wdf.Circuit(top, wdf.DiodePair(top, 4.352e-9, Vt=25.85e-3, nDiodes=1.906, trainable=True), R), [R.R, Vs.R, C.C]


def _ladder(wdf):
    Ra, Rb = wdf.Resistor(1.0e3, True), wdf.Resistor(2.2e3, True)
    Ca, Cb = wdf.Capacitor(1.0e-7, FS, True), wdf.Capacitor(2.2e-7, FS, True)
    top = wdf.Inverter(wdf.Series(Ra, wdf.Parallel(Ca, wdf.Series(Rb, Cb))))
    return wdf.Circuit(top, wdf.IdealVoltageSource(), Cb), [Ra.R, Rb.R, Ca.C, Cb.C]


@pytest.mark.parametrize("build", [_lpf, _divider, _hpf, _ladder])
def test_tape_reproduces_the_host_probe(build):
    import tf_wdf as wdf
    from wdf_hip import probe_tape
    circ, params = build(wdf)
    before = [(e, dict(e.__dict__)) for e in circ.elements]
    tape, outs, rport = probe_tape.record(circ, params)
    for e, d in before:                                          # recording leaves the elements as it found them
        assert all(e.__dict__.get(k) is v for k, v in d.items() if k in ("R", "C", "z", "a", "b", "Vs"))
    assert len(tape.ops) <= probe_tape.MAX_OPS
    val, jac = tape.evaluate([float(p) for p in params], outs + [rport])
    coef, r_port = circ.matrices()
    ref = np.concatenate([coef.detach().numpy(), [float(r_port)]])
    # (the host probe does part of its arithmetic in float32 -- the Variables' own dtype; the tape is float64 throughout)
    assert np.allclose(val, ref, rtol=2e-6, atol=1e-12)
    rows = []
    for c in list(coef) + [r_port]:
        g = torch.autograd.grad(c, params, retain_graph=True, allow_unused=True) if c.requires_grad else [None] * len(params)
        rows.append([0.0 if gi is None else float(gi) for gi in g])
    ref_j = np.array(rows)
    assert np.allclose(jac, ref_j, rtol=2e-5, atol=1e-9 * np.max(np.abs(ref_j)))
    ops, consts = tape.packed()
    assert ops.dtype == np.int32 and ops.shape[1] == 3 and consts.dtype == np.float64


def test_step_plan_tiles_every_column_and_balances_the_waves():
    from wdf_hip import mlp_root
    T = 2048
    w = [32] * 21 + [64] * 21 + [192] * 21 + [256] * 21          # warm-up steps per column: four pot values
    items = mlp_root.plan_step_items(T, w, 1024)
    assert items.shape == (1024, 4)
    cost = []
    for c in range(84):
        rows = items[items[:, 0] == c]
        assert list(rows[:, 1]) == list(range(len(rows))) and rows[0, 2] == 0 and rows[-1, 3] == T
        assert np.all(rows[1:, 2] == rows[:-1, 3]) and np.all(rows[:, 2:] % 16 == 0)
        assert rows[0, 3] - rows[0, 2] >= max(rows[1:, 3] - rows[1:, 2])          # chunk 0 (no warm-up) is the longest
        cost.append(max(rows[0, 3], (rows[1, 3] - rows[1, 2]) + mlp_root.WARM_COST * w[c]))
    k = [int((items[:, 0] == c).sum()) for c in (0, 21, 42, 63)]
    assert k[0] <= k[1] <= k[2] <= k[3] and k[3] >= k[0] + 6     # slow columns get more, shorter chunks
    assert max(cost) <= 1.25 * min(cost)                         # ... and every wave about the same number of steps


def test_tape_evaluated_in_torch_over_a_resistance_channel():
    """probe_tape.Tape.evaluate_torch (what the rows of wdf_ss_dyn_* are made of): with scalar parameters it reproduces
    Circuit.matrices() (the host probe) and its gradient; with a [B,T] resistance channel in place of one parameter every
    row equals the scalar evaluation at that sample's resistance, and the gradient to a static component is the sum over
    the samples' gradients."""
    import torch
    import tf_wdf as wdf
    from wdf_hip import probe_tape
    FS = 48000.0
    R, Vs, C = wdf.Resistor(33.0e3, True), wdf.ResistiveVoltageSource(1.0e3, trainable=True), wdf.Capacitor(22.0e-9, FS, True)
    top = wdf.Parallel(R, wdf.Series(Vs, C))
    dp = wdf.DiodePair(top, 4.352e-9, Vt=0.0493, trainable=True)
    circ = wdf.Circuit(top, dp, R, per_sample_R=Vs)
    own = {"Resistor": "R", "ResistiveVoltageSource": "R", "Capacitor": "C"}
    params = [(e, own[type(e).__name__]) for e in circ.elements if type(e).__name__ in own]
    pvars = [e.__dict__[n] for e, n in params]
    tape, outs, rport = probe_tape.record(circ, pvars, device_limits=False)
    coef, rp = circ.matrices()
    vals = [torch.tensor(float(v), dtype=torch.float64, requires_grad=True) for v in pvars]
    nodes = tape.evaluate_torch(vals, outs + [rport])
    got = torch.stack([n.reshape(()) for n in nodes])
    assert torch.allclose(got[:-1], coef.detach(), rtol=1e-6, atol=1e-12) and abs(float(got[-1]) - float(rp)) < 1e-6 * float(rp)
    # a channel in place of the source resistance (parameter 1 in tree order: R, Vs, C)
    i_pot = [k for k, (e, _) in enumerate(params) if e is Vs][0]
    r = torch.tensor([[300.0, 1.0e3, 5.0e3], [2.0e3, 750.0, 1.2e4]], dtype=torch.float64)
    vch = list(vals)
    vch[i_pot] = r
    rows = torch.stack([torch.broadcast_to(n, r.shape) for n in tape.evaluate_torch(vch, outs + [rport])], dim=-1)   # [B,T,n]
    gC = torch.autograd.grad(rows.sum(), vals[2], retain_graph=True)[0]
    gsum = 0.0
    for b in range(2):
        for t in range(3):
            vs = [torch.tensor(float(v), dtype=torch.float64, requires_grad=True) for v in pvars]
            vs[i_pot] = torch.tensor(float(r[b, t]), dtype=torch.float64)
            one = torch.stack([n.reshape(()) for n in tape.evaluate_torch(vs, outs + [rport])])
            assert torch.allclose(rows[b, t], one.detach(), rtol=1e-12, atol=0)
            gsum += float(torch.autograd.grad(one.sum(), vs[2])[0])
    assert abs(float(gC) - gsum) <= 1e-9 * abs(gsum)


def test_resident_entries_are_found_by_storage_not_by_object():
    """lowering.tensor_key / EntryCache: two slices of the same rows are the same batch, another slice or an in-place change
    is another; the cache keeps at most max_entries, and beyond four entries at most max_bytes."""
    import torch
    from wdf_hip import lowering
    X = torch.arange(24, dtype=torch.float32).reshape(6, 4)
    a, b, c = X[0:2], X[0:2], X[2:4]
    assert a is not b and lowering.tensor_key(a) == lowering.tensor_key(b) != lowering.tensor_key(c)
    k0 = lowering.tensor_key(a)
    X[0, 0] += 1.0
    assert lowering.tensor_key(X[0:2]) != k0                     # the version counter moved
    cache = lowering.EntryCache(max_entries=6, max_bytes=100)
    for i in range(10):
        cache.put(i, {"i": i}, nbytes=40)
    assert len(cache) == 4 and cache.get(9) is not None and cache.get(5) is None        # 4 entries of 40 B: above 100 B only 4 stay
    small = lowering.EntryCache(max_entries=6, max_bytes=10_000)
    for i in range(10):
        small.put(i, i, nbytes=40)
    assert len(small) == 6 and small.get(3) is None and small.get(4) == 4


def test_network_root_slope_range_is_the_networks_derivative():
    """mlp_root.slope_range (what lowering._plan_dyn sizes a network root's warm-up with): min / max of d(-MLP(a, log R))/da over
    the a grid, against central differences of the same float64 network."""
    from types import SimpleNamespace

    from wdf_hip import mlp_root
    rng = np.random.default_rng(3)
    sizes = [2, 8, 8, 1]
    dense = [SimpleNamespace(kernel=torch.tensor(rng.standard_normal((1, i, o)) * 0.9), bias=torch.tensor(rng.standard_normal((1, o)) * 0.3))
             for i, o in zip(sizes[:-1], sizes[1:])]

    def net(a, lr):
        h = np.stack([a, np.full_like(a, lr)], axis=1)
        for n, d in enumerate(dense):
            h = h @ d.kernel[0].numpy() + d.bias[0].numpy()
            if n + 1 < len(dense):
                h = np.tanh(h)
        return -h[:, 0]

    a = np.linspace(-3.0, 3.0, 513)
    lo, hi = float("inf"), float("-inf")
    for lr in (math.log(500.0), math.log(2.0e4)):
        d = (net(a + 1e-6, lr) - net(a - 1e-6, lr)) / 2e-6
        lo, hi = min(lo, d.min()), max(hi, d.max())
    got = mlp_root.slope_range(dense, [math.log(500.0), math.log(2.0e4)], 3.0, n=513)
    assert abs(got[0] - lo) < 1e-6 and abs(got[1] - hi) < 1e-6, (got, lo, hi)

"""GPU: the pre-training stage (lib/diode_pretraining.py).

Known answer: the reference documents the loss its committed pre-trained 2x16 1N4148 (1U-1D)
network reaches on the synthetic table, "MSE = 7.98e-6, ESR = 9.49e-5"
(diode_pretraining.py:196-200); the table built here by the HIP diode-pair kernel and the
network evaluated by wdf_mlp_eval must reproduce those two numbers from the same weights
(golden g3 "2x16_pre" holds them)."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from test_gpu_mlp_root import model_json  # noqa: E402


def test_table_matches_scipy_restatement():
    """synthetic_table == eqn (45) with scipy.special.wrightomega in float64, cast to float32
    (diode_pretraining.py:39-60,64-74,97-104); fp32 kernel tolerance 2e-6 absolute on |b| <= 2.5."""
    from scipy.special import wrightomega
    import diode_pretraining as dp
    from diode_config import diode_1n4148_1u1d, diode_1n4148_2u3d
    for d in (diode_1n4148_1u1d, diode_1n4148_2u3d):
        x, y = dp.synthetic_table(d)
        assert x.shape == (20000, 2) and y.shape == (20000,)
        a, R = x[:, 0].astype(np.float64), np.exp(x[:, 1].astype(np.float64))
        V = d.Vt * d.nabla
        mu0 = np.where(a >= 0, d.N_down, d.N_up)
        mu1 = np.where(a >= 0, d.N_up, d.N_down)
        lam = np.sign(a)
        b = a - 2 * V * lam * (mu0 * wrightomega(np.log(R * d.Is / V / mu0) + lam * a / (mu0 * V)).real
                               - mu1 * wrightomega(np.log(R * d.Is / V / mu1) - lam * a / (mu1 * V)).real)
        assert np.max(np.abs(y + b)) < 2e-6
    assert isinstance(dp.diode_pair_func(0.5, 1.0e4, diode_1n4148_1u1d), np.float32)


def test_pretrained_2x16_reproduces_documented_losses(golden):
    import diode_pretraining as dp
    from diode_config import diode_1n4148_1u1d
    from layers import DenseRootModel
    g = golden("g3_mlp_clipper.npz")
    model = DenseRootModel(model_json(g, "2x16_pre"))
    x, y = dp.synthetic_table(diode_1n4148_1u1d)
    yt = torch.as_tensor(y, device="cuda")
    out = dp.model_apply(model, x)
    mse, esr = float(dp.mse_loss(yt, out)), float(dp.esr_loss(yt, out))
    assert abs(mse - 7.98e-6) < 0.02e-6, mse             # diode_pretraining.py:199  "2x16: MSE = 7.98e-6"
    assert abs(esr - 9.49e-5) < 0.02e-5, esr             #                            "ESR = 9.49e-5"


@pytest.mark.parametrize("n_layers,size", [(2, 8), (4, 4), (2, 16), (3, 8)])
def test_fit_follows_float64_adam(n_layers, size):
    """fit() (HIP eval + HIP weight gradient + Adam on the device) against the same loop in
    float64 torch autograd, unshuffled so both see the same batches: after 40 steps the weights
    agree to 2e-4 of their scale, and the loss went down."""
    import diode_pretraining as dp
    from diode_config import diode_1n4148_1u1d
    from wdf_hip import mlp_root
    x, y = dp.synthetic_table(diode_1n4148_1u1d, n_points=64, R_orders=np.linspace(2, 6, 20))
    rng = np.random.default_rng(0)
    pick = rng.permutation(len(y))                       # mix the impedances into every batch
    x, y = x[pick], y[pick]
    model = dp.build_model(n_layers, size, seed=3)
    dense, hidden, n_tanh = mlp_root.describe(model)
    assert (hidden, n_tanh) == (size, n_layers + 1)
    w0 = mlp_root.flat_weights(dense).detach().double()
    lr, bs, epochs = 1e-3, 32, 1
    hist = dp.fit(model, x, y, epochs, learning_rate=lr, batch_size=bs, shuffle=False, fused=False)
    w_hip = mlp_root.flat_weights(mlp_root.describe(model)[0]).detach().double()
    # the one-launch-per-epoch kernel (wdf_mlp_fit_epoch) walks the same batches
    model_f = dp.build_model(n_layers, size, seed=3)
    hist_f = dp.fit(model_f, x, y, epochs, learning_rate=lr, batch_size=bs, shuffle=False, fused=True)
    w_fused = mlp_root.flat_weights(mlp_root.describe(model_f)[0]).detach().double()

    w = w0.clone().requires_grad_(True)
    m, v = torch.zeros_like(w), torch.zeros_like(w)
    xt, yt = torch.as_tensor(x).double(), torch.as_tensor(y).double()
    it, losses = 0, []
    for s in range(0, len(y), bs):
        h, o, n_in = xt[s:s + bs], 0, 2
        for _ in range(n_tanh):
            h = torch.tanh(h @ w[o:o + n_in * hidden].reshape(n_in, hidden) + w[o + n_in * hidden:o + n_in * hidden + hidden])
            o += n_in * hidden + hidden
            n_in = hidden
        out = h @ w[o:o + hidden] + w[o + hidden]
        t = yt[s:s + bs]
        loss = torch.mean((t - out) ** 2) + torch.sqrt(torch.sum((t - out) ** 2) / (torch.sum(t ** 2) + dp.eps) / dp.N)
        (gr,) = torch.autograd.grad(loss, [w])
        it += 1
        lr_t = lr * np.sqrt(1 - 0.999 ** it) / (1 - 0.9 ** it)
        with torch.no_grad():
            m.mul_(0.9).add_(gr, alpha=0.1)
            v.mul_(0.999).addcmul_(gr, gr, value=0.001)
            w.sub_(lr_t * m / (torch.sqrt(v) + 1e-7))
        losses.append(float(loss))
    assert it == 40
    # a short last batch (Keras keeps it): 1280 points in batches of 48 -> 26 full + one of 32
    m_a, m_b = dp.build_model(n_layers, size, seed=4), dp.build_model(n_layers, size, seed=4)
    h_a = dp.fit(m_a, x, y, 2, learning_rate=lr, batch_size=48, shuffle=True, seed=7, fused=False)
    h_b = dp.fit(m_b, x, y, 2, learning_rate=lr, batch_size=48, shuffle=True, seed=7, fused=True)
    wa = mlp_root.flat_weights(mlp_root.describe(m_a)[0]).detach()
    wb_ = mlp_root.flat_weights(mlp_root.describe(m_b)[0]).detach()
    assert float((wa - wb_).abs().max()) <= 5e-4 * float(wa.abs().max())
    assert np.allclose(h_a, h_b, rtol=2e-4)
    assert float((w_hip - w.detach()).abs().max()) <= 2e-4 * float(w.detach().abs().max())
    assert abs(hist[0] - np.mean(losses)) <= 1e-4 * np.mean(losses)
    assert float((w_fused - w.detach()).abs().max()) <= 2e-4 * float(w.detach().abs().max())
    assert abs(hist_f[0] - np.mean(losses)) <= 1e-4 * np.mean(losses)
    assert np.mean(losses[-8:]) < np.mean(losses[:8])


def test_pretrain_end_to_end_and_json_round_trip(tmp_path):
    """pretrain() -> model_utils.save_model -> layers.DenseRootModel(json) gives the same network,
    and a short full run lowers both losses."""
    import diode_pretraining as dp
    from diode_config import diode_1n4148_1u2d
    from layers import DenseRootModel
    from model_utils import save_model, load_model_json
    model, stats = dp.pretrain(diode_1n4148_1u2d, n_layers=2, layer_size=8, epochs=2, learning_rate=5e-4,
                               batch_size=250, seed=1)
    assert stats["name"] == "1N4148 (1U-2D)_2x8_pretrained"
    assert stats["after"][0] < stats["before"][0] and stats["after"][1] < stats["before"][1]
    path = tmp_path / f"{stats['name']}_model.json"
    save_model(model, str(path))
    again = DenseRootModel(load_model_json(str(path)))
    x, _ = dp.synthetic_table(diode_1n4148_1u2d, n_points=50)
    assert float((dp.model_apply(again, x) - dp.model_apply(model, x)).abs().max()) < 1e-6


def test_device_adam_matches_keras_rule():
    """wdf_adam_step against the TF 2.5 Adam update in float64 numpy, with per-parameter learning
    rates and the clip constraint of tf_wdf.py:74,104, over 50 steps."""
    from wdf_hip import binding as wb
    rng = np.random.default_rng(5)
    n = 609
    th = rng.standard_normal(n)
    lr = np.abs(rng.standard_normal(n)) * 1e-2
    lo, hi = np.full(n, -1.0), np.full(n, 1.2)
    b1, b2, eps = 0.5, 0.999, 1e-7                       # clipper_pot.py:179 uses beta_1 = 0.5
    opt = wb.Adam(n, lr, beta_1=b1, beta_2=b2, epsilon=eps, lo=lo, hi=hi)
    th_dev = torch.as_tensor(th, dtype=torch.float32, device="cuda")
    ref = th.astype(np.float32).astype(np.float64)
    m, v = np.zeros(n), np.zeros(n)
    for t in range(1, 51):
        g = rng.standard_normal(n) * (1.0 + 0.1 * t)
        opt.apply(th_dev, torch.as_tensor(g, dtype=torch.float32, device="cuda"))
        g = g.astype(np.float32).astype(np.float64)
        m = b1 * m + (1 - b1) * g
        v = b2 * v + (1 - b2) * g * g
        ref = np.clip(ref - lr.astype(np.float32) * np.sqrt(1 - b2 ** t) / (1 - b1 ** t) * m / (np.sqrt(v) + eps), lo, hi)
    assert int(opt.step.cpu()[0]) == 50
    assert np.max(np.abs(th_dev.cpu().numpy() - ref)) < 2e-5
    assert np.any(ref == 1.2) or np.any(ref == -1.0)      # the constraint was active somewhere

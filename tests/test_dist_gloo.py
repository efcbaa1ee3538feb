"""CPU, world_size 2, gloo: the data-parallel sharding + single fused all-reduce of
wdf_hip.dist gives the same loss / gradient as the unsharded batch.  The local compute in
this test is the CPU oracle (allowed in tests); on the GPU box bench.py plugs the HIP kernels
into the same helper with the nccl (= RCCL) backend."""
import os
import socket
import sys

import numpy as np
import pytest

torch = pytest.importorskip("torch")
import torch.multiprocessing as mp  # noqa: E402

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, B, T, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    for p in (os.path.join(REPO, "oracle"), os.path.join(REPO, "differentiable-wdfs_amd", "lib")):
        sys.path.insert(0, p)
    import oracle as O
    from wdf_hip import dist as wdist, workload
    w, r, _ = wdist.init(backend="gloo")
    assert (w, r) == (world, rank)
    b0, b1 = wdist.shard_range(B, rank, world)
    x = workload.sweep_batch(B, T, b0=b0, b1=b1, dtype=np.float64)       # this rank's rows of the global batch
    theta, fs = workload.clipper_theta(), workload.FS
    tgt = O.clipper_fwd(workload.target_theta(), fs, x)
    y = O.clipper_fwd(theta, fs, x)
    d = y - tgt
    _, g = O.clipper_fwd_bwd(theta, fs, x, 2.0 * d)                       # dSSE_local/dtheta
    loss, grad = wdist.mse_step_allreduce(torch.tensor(float(np.sum(d * d)), dtype=torch.float64),
                                          torch.tensor(g, dtype=torch.float64), float(B * T))
    if rank == 0:
        out.put((float(loss), grad.numpy().copy()))
    wdist.barrier()
    torch.distributed.destroy_process_group()


def _worker_esr(rank, world, port, B, T, skip, out):
    """MSE + ESR (clipper_pot.py:177): two exchanges per step -- the loss sums after the forward,
    the gradient after the reverse sweep."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    for p in (os.path.join(REPO, "oracle"), os.path.join(REPO, "differentiable-wdfs_amd", "lib")):
        sys.path.insert(0, p)
    import oracle as O
    from wdf_hip import dist as wdist, workload
    wdist.init(backend="gloo")
    b0, b1 = wdist.shard_range(B, rank, world)
    x = workload.sweep_batch(B, T, b0=b0, b1=b1, dtype=np.float64)
    theta, fs = workload.clipper_theta(), workload.FS
    tgt = O.clipper_fwd(workload.target_theta(), fs, x)
    y = O.clipper_fwd(theta, fs, x)
    S, E = wdist.esr_sums_allreduce(np.sum((y[skip:] - tgt[skip:]) ** 2), np.sum(y[skip:] ** 2))
    ga, gb, mse, esr = wdist.esr_coefficients(S, E, float(B * (T - skip)), np.finfo(float).eps)
    gy = ga * (y - tgt) + gb * y
    gy[:skip] = 0.0
    _, g = O.clipper_fwd_bwd(theta, fs, x, gy)
    grad = wdist.allreduce_sum_(torch.tensor(g, dtype=torch.float64))
    if rank == 0:
        out.put((mse + esr, grad.numpy().copy()))
    wdist.barrier()
    torch.distributed.destroy_process_group()


def _worker_esr_one_exchange(rank, world, port, B, T, skip, out):
    """The one-pass MSE + ESR step (wdf_clipper_step_esr_tp): ONE exchange of ten numbers per step."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    for p in (os.path.join(REPO, "oracle"), os.path.join(REPO, "differentiable-wdfs_amd", "lib")):
        sys.path.insert(0, p)
    import oracle as O
    from wdf_hip import dist as wdist, workload
    wdist.init(backend="gloo")
    b0, b1 = wdist.shard_range(B, rank, world)
    x = workload.sweep_batch(B, T, b0=b0, b1=b1, dtype=np.float64)
    theta, fs = workload.clipper_theta(), workload.FS
    tgt = O.clipper_fwd(workload.target_theta(), fs, x)
    y = O.clipper_fwd(theta, fs, x)
    d, yy = y - tgt, y.copy()
    d[:skip] = 0.0
    yy[:skip] = 0.0
    _, gP = O.clipper_fwd_bwd(theta, fs, x, d)                           # d(S_local / 2)/dtheta
    _, gQ = O.clipper_fwd_bwd(theta, fs, x, yy)                          # d(E_local / 2)/dtheta
    loss, grad = wdist.esr_step_allreduce(np.sum(d * d), np.sum(yy * yy), gP, gQ, float(B * (T - skip)), np.finfo(float).eps)
    if rank == 0:
        out.put((float(loss), grad.numpy().copy()))
    wdist.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("worker", [_worker_esr, _worker_esr_one_exchange])
def test_two_rank_gloo_mse_esr_matches_single_rank(oracle, worker):
    """Sharded MSE + ESR == the unsharded loss and gradient (torch float64 autograd of the loss as the
    script writes it, through the oracle's adjoint)."""
    from wdf_hip import workload
    B, T, skip, world = 9, 260, 50, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=worker, args=(r, world, port, B, T, skip, q)) for r in range(world)]
    for p in procs:
        p.start()
    loss, grad = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    x = workload.sweep_batch(B, T, dtype=np.float64)
    theta, fs = workload.clipper_theta(), workload.FS
    tgt = torch.tensor(oracle.clipper_fwd(workload.target_theta(), fs, x))
    y = torch.tensor(oracle.clipper_fwd(theta, fs, x), requires_grad=True)
    o, t = y[skip:], tgt[skip:]
    n = o.numel()
    S = ((o - t) ** 2).sum()
    l1 = S / n + torch.sqrt(S / ((o ** 2).sum() + np.finfo(float).eps) / n)      # clipper_pot.py:146-156,177
    (gy,) = torch.autograd.grad(l1, [y])
    _, g1 = oracle.clipper_fwd_bwd(theta, fs, x, gy.numpy())
    assert abs(loss - float(l1)) < 1e-12 * max(1.0, abs(float(l1)))
    assert np.max(np.abs(grad - g1) / np.abs(g1)) < 1e-9


def test_shard_range_covers_batch():
    sys.path.insert(0, os.path.join(REPO, "differentiable-wdfs_amd", "lib"))
    from wdf_hip import dist as wdist
    for n, w in [(8192, 8), (10, 3), (7, 8), (1340, 8)]:
        spans = [wdist.shard_range(n, r, w) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
        assert max(b - a for a, b in spans) - min(b - a for a, b in spans) <= 1


@pytest.mark.timeout(300)
def test_two_rank_gloo_matches_single_rank(oracle):
    from wdf_hip import workload
    B, T, world = 10, 300, 2                                              # ragged split 5 + 5; also 3 ranks below
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, B, T, q)) for r in range(world)]
    for p in procs:
        p.start()
    loss, grad = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    x = workload.sweep_batch(B, T, dtype=np.float64)
    theta, fs = workload.clipper_theta(), workload.FS
    tgt = oracle.clipper_fwd(workload.target_theta(), fs, x)
    l1, g1, _ = oracle.clipper_mse_step(theta, fs, x, tgt, dtype=np.float64)
    assert abs(loss - l1) < 1e-12 * max(1.0, abs(l1))
    assert np.max(np.abs(grad - g1) / np.abs(g1)) < 1e-9


def _worker_mlp(rank, world, port, B, T, skip, out):
    """The MLP-root training step's protocol (dist.esr_two_exchange, what mlp_root.MlpTrainStep.step runs between its
    phases): this rank's loss sums -> all-reduce -> reverse sweep with the global sums -> all-reduce of the weight gradient."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    for p in (os.path.join(REPO, "oracle"), os.path.join(REPO, "differentiable-wdfs_amd", "lib")):
        sys.path.insert(0, p)
    import oracle as O
    from wdf_hip import dist as wdist, workload
    wdist.init(backend="gloo")
    b0, b1 = wdist.shard_range(B, rank, world)
    x = workload.sweep_batch(B, T, b0=b0, b1=b1, dtype=np.float64) * 0.6
    r = workload.dataset_resistance_batch(B, T, b0=b0, b1=b1, dtype=np.float64)
    wh, hidden, n_layers = workload.reference_mlp_weights("2x8")
    oc = O.clipper_mlp_circuit(workload.FS, [2] + [hidden] * n_layers + [1], [O.ACT_TANH] * n_layers + [O.ACT_NONE])
    theta = np.concatenate([[45.0e3, workload.C_CLIPPER], wh.astype(np.float64)])
    xin = np.stack([x, r], axis=-1)
    tgt = O.clipper_fwd(workload.clipper_theta(), workload.FS, x, r=r)
    y = O.tree_fwd(oc, theta, xin)
    n_global, eps = float(B * (T - skip)), np.finfo(float).eps
    params = list(range(2, 2 + 40)) + [len(theta) - 1]

    def forward_sums():
        return torch.tensor([np.sum((y[skip:] - tgt[skip:]) ** 2), np.sum(y[skip:] ** 2)], dtype=torch.float64)

    def backward(sums):
        ga, gb, _, _ = wdist.esr_coefficients(float(sums[0]), float(sums[1]), n_global, eps)
        gy = ga * (y - tgt) + gb * y
        gy[:skip] = 0.0
        return torch.tensor(O.tree_grad(oc, theta, xin, gy, params=params), dtype=torch.float64)

    sums, grad = wdist.esr_two_exchange(forward_sums, backward)
    _, _, mse, esr = wdist.esr_coefficients(float(sums[0]), float(sums[1]), n_global, eps)
    if rank == 0:
        out.put((mse + esr, grad.numpy().copy()))
    wdist.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_gloo_mlp_root_step_matches_single_rank(oracle):
    """clipper_pot.py's loss and weight gradient (MLP root, pot value per sample), sharded over two ranks == unsharded."""
    from wdf_hip import workload
    B, T, skip, world = 7, 200, 50, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_mlp, args=(r, world, port, B, T, skip, q)) for r in range(world)]
    for p in procs:
        p.start()
    loss, grad = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    x = workload.sweep_batch(B, T, dtype=np.float64) * 0.6
    r = workload.dataset_resistance_batch(B, T, dtype=np.float64)
    wh, hidden, n_layers = workload.reference_mlp_weights("2x8")
    oc = oracle.clipper_mlp_circuit(workload.FS, [2] + [hidden] * n_layers + [1], [oracle.ACT_TANH] * n_layers + [oracle.ACT_NONE])
    theta = np.concatenate([[45.0e3, workload.C_CLIPPER], wh.astype(np.float64)])
    xin = np.stack([x, r], axis=-1)
    tgt = torch.tensor(oracle.clipper_fwd(workload.clipper_theta(), workload.FS, x, r=r))
    y = torch.tensor(oracle.tree_fwd(oc, theta, xin), requires_grad=True)
    o, t = y[skip:], tgt[skip:]
    n = o.numel()
    S = ((o - t) ** 2).sum()
    l1 = S / n + torch.sqrt(S / ((o ** 2).sum() + np.finfo(float).eps) / n)      # clipper_pot.py:146-156,177
    (gy,) = torch.autograd.grad(l1, [y])
    params = list(range(2, 2 + 40)) + [len(theta) - 1]
    g1 = oracle.tree_grad(oc, theta, xin, gy.numpy(), params=params)
    assert abs(loss - float(l1)) < 1e-12 * max(1.0, abs(float(l1)))
    assert np.max(np.abs(grad - g1)) < 1e-9 * np.max(np.abs(g1))

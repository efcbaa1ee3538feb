"""GPU: the clipper_pot.py workflow end to end on synthetic stand-ins for the missing dataset:
CSV files in the reference's format and names -> load_diode_data -> batch_data -> the pot clipper
with an MLP root (a hand-written per-sample loop, recorded: tests/loops.py) -> MSE+ESR loss with the
script's argument order -> Adam ->
save the weights as JSON in the plugin's schema.  The "measurement" is the GPU diode-pair
clipper (north-star root); the model being trained is the reference's tanh-MLP root."""
import json
from collections import namedtuple

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

DiodeConfig = namedtuple("DiodeConfig", ["name", "Is", "nabla", "Vt", "N_up", "N_down"])
D1 = DiodeConfig("1N4148 (1U-1D)", 4.352e-9, 1.906, 25.85e-3, 1, 1)


def test_clipper_pot_workflow(tmp_path, golden):
    import dataimport as di
    import model_utils as mu
    import tf_wdf as wdf
    from tf_wdf import tf
    from layers import DenseRootModel
    from wdf_hip import binding as wb
    from test_gpu_mlp_root import model_json

    FS_DATA = 48000.0
    C_val = 4.7e-9

    # ---- "measure" the circuit: diode-pair clipper on the GPU, written as the 5 CSVs of 1up1down
    def simulate(x, R):
        th = torch.tensor([D1.Is, D1.Vt * D1.nabla, R, C_val], dtype=torch.float32, device="cuda")
        xd = torch.as_tensor(x[None, :], dtype=torch.float32, device="cuda").contiguous()
        y, _, _ = wb.clipper_fwd(xd, th, FS_DATA, want_stash=False)
        return y[:, 0].cpu().numpy()

    secs = di.TIME_REMOVE_PRE + 0.45                          # short recordings keep the test fast
    files = di.write_synthetic_dataset(tmp_path, simulate, fs=FS_DATA, seconds=secs)
    assert len(files) == 5

    # ---- clipper_pot.py:48-85
    train_data, train_N, val_data, val_N, FS = di.load_diode_data(D1, tmp_path)
    assert FS == FS_DATA and train_N == 4 * val_N
    batch_size = 2048
    train_X, train_Y = di.batch_data(train_data, train_N, batch_size)
    val_X, val_Y = di.batch_data(val_data, val_N, batch_size)
    assert train_X.shape[1:] == (2048, 2) and val_X.shape[0] >= 1

    # ---- the model: pot clipper with a DenseRootModel root, its loop written by hand (tests/loops.py)
    from loops import PotClipper, mse_plus_esr
    eps = np.finfo(float).eps
    optimizer = tf.keras.optimizers.Adam(learning_rate=0.0001, beta_1=0.5, beta_2=0.999)   # clipper_pot.py:180

    g = golden("g3_mlp_clipper.npz")
    model = PotClipper(wdf, FS, C_val, mlp_json=model_json(g, "2x16_pre"))   # warm start from the PRE-trained weights (clipper_pot.py:132-137)
    skip_samples = 50
    tY, vY = tf.constant(train_Y).cuda(), tf.constant(val_Y).cuda()
    losses = []
    for epoch in range(8):                                     # the epoch of clipper_pot.py:245-269: train step + validation pass
        with tf.GradientTape() as tape:
            outs = tf.transpose(model.run(train_X)[..., 0], perm=[1, 0, 2])
            loss = mse_plus_esr(tf, outs[:, skip_samples:, :], tY[:, skip_samples:, :], eps)
        val_outs = tf.transpose(model.run(val_X)[..., 0], perm=[1, 0, 2])
        val_loss = mse_plus_esr(tf, val_outs[:, skip_samples:, :], vY[:, skip_samples:, :], eps)
        grads = tape.gradient(loss, model.trainable_variables)
        optimizer.apply_gradients(zip(grads, model.trainable_variables))
        losses.append((float(loss), float(val_loss)))
    assert all(np.isfinite(l) and np.isfinite(v) for l, v in losses)
    # The warm start already fits to 2e-4; Adam's first normalised step (lr per weight on all 609
    # weights, beta_1 = 0.5) kicks the loss up, after which it must come back down.
    assert losses[-1][0] < 0.1 * losses[1][0] and losses[-1][1] < 0.1 * losses[1][1], losses
    # ---- clipper_pot.py:298-331: write the trained root as JSON, reload it
    out = tmp_path / "model.json"
    mu.save_model(model.mlp, out)
    js = json.load(open(out))
    assert [l["activation"] for l in js["layers"]] == ["tanh", "tanh", "tanh", ""] and js["in_shape"] == [None, 2]
    m2 = DenseRootModel(js)
    assert np.array_equal(m2.layers[0].kernel.numpy(), model.mlp.layers[0].kernel.numpy())


def test_bench_two_rank_path_rehearsal():
    """bench.py's multi-rank path (batch shards per rank, the fused [SSE, grads] all-reduce, Adam on
    every rank, max-over-ranks timing, one JSON line from rank 0) run as the driver launches it --
    torch.distributed.run, 2 ranks -- but with both ranks on cuda:0 over gloo, since the test box has
    one GPU.  Checks the contract fields and that both ranks really trained on a 2 x 8192 batch."""
    import json, os, socket, subprocess, sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(repo, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1",
           "--batch", "1024", "--seq-len", "2048", "--rehearse-on-one-gpu"]
    out = subprocess.run(cmd, cwd=repo, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 4 and d["warmup"] == 1 and d["scaling"] == "weak"
    assert d["parity"] is None and d["value_batch_major"] > 0          # parity runs on rank 0 at N = 1 only
    assert d["step_kernels"].startswith("one pass") and set(d["kernel_ms"]["fused_step"]) == {"min", "median", "max", "n"}
    assert d["unit"] == "samples/s" and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert d["config"]["global_batch"] == 2048 and d["config"]["parallelism"] == "dp2"
    assert abs(d["value"] - 2048 * 2048 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    assert d["roofline"]["bound"] == "hbm" and 0.0 < d["roofline"]["hbm"]["frac_moved"] < 1.0
    # frac is what the kernel moves (12 B/sample: a rate the memory system really sees); SURVEY 8(d)'s two-pass figure beside it
    assert abs(d["roofline"]["frac"] - d["roofline"]["hbm"]["frac_moved"]) < 1e-9 and d["roofline"]["algorithmic_bytes_per_sample"] == 12
    assert abs(d["roofline"]["survey_8d_two_pass_equivalent"]["frac"] - 2.0 * d["roofline"]["frac"]) < 1e-9
    assert d["ranks_seen"] == 2 and d["ranks"]["collective_backend"].startswith("gloo") and len(d["ranks"]["ms_per_step_per_rank"]["all"]) == 2
    assert "external launcher" in d["ranks"]["launcher"]
    # both curves from the one invocation: the weak headline (1024 per rank) and SURVEY 8e's strong split of ONE 1024 batch
    sc = d["scaling_curves"]
    assert sc["weak"]["global_batch"] == 2048 and sc["strong"]["global_batch"] == 1024 and sc["strong"]["sequences_per_rank"] == 512
    assert abs(sc["strong"]["value"] - 1024 * 2048 / (sc["strong"]["ms_per_step"] * 1e-3)) < 1e-6 * sc["strong"]["value"]
    assert "cpu_baseline" not in d                           # rank 0 at N = 1 only
    assert d["config"]["optimizer"]["loss_last_step"] < d["config"]["optimizer"]["loss_first_step"]


def test_bench_gpus_2_starts_its_own_ranks():
    """`python bench.py --gpus 2` with NO launcher around it (the form a driver uses for N = 1): bench.py starts the two
    ranks itself, the collective counts them (`ranks_seen`), rank 0 prints the one line.  Both ranks on cuda:0 over gloo
    here (one-GPU box); on a node with N GPUs the same entry point runs one rank per GPU over RCCL."""
    env_clean = {k: v for k, v in __import__("os").environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    d = _run_bench(["--gpus", "2", "--steps", "4", "--warmup", "1", "--batch", "1024", "--seq-len", "2048", "--rehearse-on-one-gpu",
                    "--no-batch-major"], env=env_clean)
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2 and d["ranks"]["world_size"] == 2
    assert "started the ranks itself" in d["ranks"]["launcher"]
    assert d["scaling"] == "weak" and d["config"]["global_batch"] == 2048
    assert set(d["scaling_curves"]) == {"weak", "strong"} and d["scaling_curves"]["strong"]["global_batch"] == 1024
    assert d["config"]["optimizer"]["loss_last_step"] < d["config"]["optimizer"]["loss_first_step"]
    # more ranks than GPUs without the rehearsal flag: refused with a message, not a hang
    import os, subprocess, sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    import torch
    n = torch.cuda.device_count() + 1
    out = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", str(n)], cwd=repo, env=env_clean,
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 2 and "GPU(s)" in out.stderr


def _run_bench(extra, launcher=None, timeout=900, env=None):
    import os, subprocess, sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable] + (launcher or []) + [os.path.join(repo, "bench.py")] + extra
    out = subprocess.run(cmd, cwd=repo, capture_output=True, text=True, timeout=timeout, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_force_dist_runs_the_rccl_branch():
    """bench.py --force-dist: a one-rank nccl (= RCCL) process group, the fused [SSE, grads] device buffer
    all-reduced on it every step, Adam as its own launch afterwards -- the N > 1 step on the one GPU of the
    test box.  Same numbers as the folded single-rank step (the all-reduce of one rank is the identity)."""
    common = ["--steps", "6", "--warmup", "2", "--batch", "2048", "--seq-len", "2048", "--plan", "8,192,16",
              "--no-cpu-baseline", "--no-batch-major"]
    d = _run_bench(common + ["--force-dist"])
    ref = _run_bench(common)
    assert "RCCL" in d["config"]["collective"] and ref["config"]["collective"] is None
    assert d["n_gpus"] == 1 and d["scaling"] == "weak"
    a, b = np.array(d["config"]["optimizer"]["theta_final"]), np.array(ref["config"]["optimizer"]["theta_final"])
    assert np.max(np.abs(a - b) / np.abs(b)) < 1e-6, (a, b)
    assert d["parity"]["max_abs_y"] < 1e-6 and d["parity"]["max_rel_grad"] < 1e-4, d["parity"]
    assert ref["parity"]["max_abs_y"] < 1e-6 and ref["parity"]["max_rel_grad"] < 1e-4, ref["parity"]
    # and the two-kernel form of the same step (forward kernel + reverse-sweep kernel) trains to the same parameters
    two = _run_bench(common + ["--two-kernel"])
    assert two["step_kernels"].startswith("two kernels") and set(two["kernel_ms"]) == {"fwd", "bwd"}
    c = np.array(two["config"]["optimizer"]["theta_final"])
    assert np.max(np.abs(c - b) / np.abs(b)) < 1e-5, (c, b)
    assert two["parity"]["max_abs_y"] < 1e-6 and two["parity"]["max_rel_grad"] < 1e-4, two["parity"]


def test_bench_strong_scaling_rehearsal():
    """--scaling strong: ONE batch split over the ranks (two ranks on cuda:0 over gloo here); the line says so
    and `value` counts the global batch once."""
    import socket, sys
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    launcher = ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                "--master-port", str(port)]
    d = _run_bench(["--gpus", "2", "--steps", "4", "--warmup", "1", "--batch", "2048", "--seq-len", "2048", "--scaling", "strong",
                    "--rehearse-on-one-gpu", "--no-batch-major"], launcher=launcher)
    assert d["scaling"] == "strong" and d["config"]["global_batch"] == 2048 and d["config"]["parallelism"] == "dp2"
    assert "1024 sequences" in d["config"]["workload"] and d["ranks_seen"] == 2
    assert d["scaling_curves"]["weak"]["global_batch"] == 4096 and d["scaling_curves"]["strong"]["value"] == d["value"]
    assert abs(d["value"] - 2048 * 2048 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]


@pytest.mark.parametrize("kind", ["lpf", "hpf"])
def test_bench_tree_step_lines(kind):
    """bench.py --config lpf / hpf: lpf.py's loop through the element API with resident components, as a bench line; the
    parity block holds the same kernels against the fp64 oracle."""
    d = _run_bench(["--config", kind, "--batch", "512", "--seq-len", "2048", "--steps", "6", "--warmup", "3"])
    assert d["unit"] == "samples/s" and d["n_gpus"] == 1 and d["value"] > 0 and d["roofline"]["frac"] > 0
    assert kind in d["config"]["workload"] and len(d["kernel_ms"]) == 1
    p = d["parity"]
    assert p["max_abs_y"] < 3e-6 and p["loss_rel"] < 2e-6 and p["grad_max_rel"] < 3e-4, p
    if kind == "hpf":
        assert d["control"]["gated_groups"] == 0


def test_bench_c4_line():
    """bench.py --config c4: BASELINE configs[3]'s per-GPU share (dataset-shaped batch, one pot value per sequence streamed as
    channel 1, MSE + ESR past 50 samples, Adam in the step) as a line with its own parity block against the oracle."""
    d = _run_bench(["--config", "c4", "--batch", "1340", "--steps", "6", "--warmup", "3"])
    assert d["unit"] == "samples/s" and d["value"] > 0 and d["config"]["seq_len"] == 2048
    assert d["config"]["resistance_channel"].startswith("one value per sequence")
    assert d["config"]["verify_status"]["n_bad"] == 0
    assert d["config"]["loss_last_step"] < d["config"]["loss_first_step"]
    p = d["parity"]
    assert p["max_abs_y"] < 2e-6 and p["max_rel_grad"] < 1e-4 and p["rel_loss"] < 2e-5, p

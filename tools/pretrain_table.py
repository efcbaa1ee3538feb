"""GPU: the reference's pre-training table (diode_pretraining.py:190-201) re-run with the reference's recipe
(lib/diode_pretraining.pretrain: orthogonal kernels, zero biases, Adam(2e-5), mini-batches of 32 reshuffled every epoch,
MSE + ESR with the script's N = 1000, 2000 epochs), several seeds per row: an epoch is one launch of wdf_mlp_fit_epoch --
one workgroup -- so the rows run side by side as separate processes on the one GPU.
usage: python tools/pretrain_table.py [seeds per row, default 3] [epochs, default 2000] [rows net:diode,...] [first seed] > table.jsonl
One JSON line per (row, seed), then one summary line per row: best / median / worst of MSE and ESR against the documented pair."""
import json, os, subprocess, sys, statistics
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DOC = {  # diode_pretraining.py:190-201  (net, diode) -> (MSE, ESR)
    ("2x4", "1u1d"): (1.34e-3, 1.23e-3), ("2x8", "1u1d"): (5.51e-5, 2.49e-4), ("2x16", "1u1d"): (7.98e-6, 9.49e-5),
    ("4x4", "1u1d"): (6.38e-4, 8.48e-4), ("4x8", "1u1d"): (4.43e-5, 2.24e-4),
    ("2x16", "3u3d"): (6.14e-5, 2.46e-4), ("2x16", "2u3d"): (7.65e-6, 9.29e-5), ("2x16", "2u2d"): (1.79e-5, 1.53e-4),
    ("2x16", "1u3d"): (1.15e-5, 1.10e-4), ("2x16", "1u2d"): (1.87e-5, 1.51e-4)}

if len(sys.argv) > 1 and sys.argv[1] == "--one":
    sys.path.insert(0, os.path.join(_R, "differentiable-wdfs_amd", "lib"))
    import time, torch
    import diode_pretraining as dp
    import diode_config as dc
    net, diode, seed, epochs = sys.argv[2], sys.argv[3], int(sys.argv[4]), int(sys.argv[5])
    n_layers, size = (int(v) for v in net.split("x"))
    t0 = time.time()
    model, stats = dp.pretrain(getattr(dc, f"diode_1n4148_{diode}"), n_layers, size, epochs=epochs, seed=seed)
    torch.cuda.synchronize()
    print(json.dumps({"net": net, "diode": diode, "seed": seed, "epochs": epochs, "seconds": time.time() - t0,
                      "mse": stats["after"][0], "esr": stats["after"][1], "last_epoch_mean_batch_loss": stats["history"][-1]}))
    sys.exit(0)

n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
epochs = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
only = sys.argv[3].split(",") if len(sys.argv) > 3 else None          # e.g. 2x8:1u1d,2x16:1u1d
seed0 = int(sys.argv[4]) if len(sys.argv) > 4 else 0
jobs = [(net, diode, seed0 + seed) for (net, diode) in DOC if only is None or f"{net}:{diode}" in only for seed in range(n_seeds)]
procs = [(j, subprocess.Popen([sys.executable, os.path.abspath(__file__), "--one", j[0], j[1], str(j[2]), str(epochs)],
                              stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)) for j in jobs]
rows = {}
for j, p in procs:
    out, _ = p.communicate()
    for line in out.splitlines():
        if line.startswith("{"):
            d = json.loads(line)
            rows.setdefault((d["net"], d["diode"]), []).append(d)
            print(line, flush=True)
for key, rs in rows.items():
    mse, esr = sorted(r["mse"] for r in rs), sorted(r["esr"] for r in rs)
    doc = DOC[key]
    print(json.dumps({"row": f"{key[0]} 1N4148 ({key[1].upper()})", "documented_mse_esr": doc, "seeds": len(rs),
                      "mse_best_median_worst": [mse[0], statistics.median(mse), mse[-1]],
                      "esr_best_median_worst": [esr[0], statistics.median(esr), esr[-1]],
                      "best_over_documented": [mse[0] / doc[0], esr[0] / doc[1]],
                      "median_over_documented": [statistics.median(mse) / doc[0], statistics.median(esr) / doc[1]]}), flush=True)

"""CPU: `python bench.py --gpus N` with no launcher around it starts its own ranks (torch.distributed.run on 127.0.0.1), the
collective counts them and rank 0 prints one line -- the entry point's pre-flight (`--launch-check`: no kernels, gloo when
there is no GPU), so the path a first 8-GPU lease takes is exercised in the CPU suite of every round."""
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env=None, timeout=300):
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py")] + args, cwd=REPO, capture_output=True, text=True,
                         timeout=timeout, env=env)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    return out, lines


def _clean_env():
    return {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}


def test_bench_starts_its_own_ranks_and_counts_them():
    out, lines = _run(["--gpus", "2", "--launch-check"], env=_clean_env())
    assert out.returncode == 0, out.stderr[-2000:]
    assert len(lines) == 1, out.stdout[-1000:]
    d = json.loads(lines[0])
    assert d["launch_check"] and d["ok"] and d["n_gpus"] == 2 and d["ranks_seen"] == 2
    assert d["ranks"]["world_size"] == 2 and "started the ranks itself" in d["ranks"]["launcher"]
    assert len(d["ranks"]["ms_per_step_per_rank"]["all"]) == 2


def test_bench_under_an_external_launcher_is_left_alone():
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "3", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(REPO, "bench.py"), "--gpus", "3", "--launch-check"]
    out = subprocess.run(cmd, cwd=REPO, capture_output=True, text=True, timeout=300, env=_clean_env())
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["ranks_seen"] == 3 and "external launcher" in d["ranks"]["launcher"]


def test_bench_refuses_a_world_size_that_contradicts_gpus():
    env = _clean_env()
    env.update({"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    out, lines = _run(["--gpus", "2", "--launch-check"], env=env)
    assert out.returncode != 0 and not lines and "WORLD_SIZE=1" in out.stderr


def test_cpu_baseline_reports_quota_table_and_both_figures():
    """bench.py's CPU leg (the oracle as the reported baseline): every thread count measured warm, the all-cores figure and the
    best figure both on the line, the cgroup's quota printed next to the affinity count."""
    sys.path.insert(0, REPO)
    import bench
    txt, cores = bench.cgroup_cpu_quota()
    assert cores is None or cores > 0
    r = bench.cpu_baseline(256, 48000.0, hold_s=0.2, warm_floor_s=0.1)
    avail = len(os.sched_getaffinity(0))
    assert r["kind"] == "port" and r["logical_cpus"] == avail and r["all_cores"]["threads"] == avail
    assert {e["threads"] for e in r["calibration"]} >= {1, avail}
    assert all(e["sequences"] >= 32 * e["threads"] or e["sequences"] == 8192 for e in r["calibration"])
    assert all(e["warm_passes_discarded"] >= 1 and e["best"] >= e["median"] > 0 for e in r["calibration"])
    assert r["value"] >= max(e["best"] for e in r["calibration"]) and r["value"] >= r["all_cores"]["value"]
    assert r["cores"] in {e["threads"] for e in r["calibration"]} and "cgroup_cpu_max" in r

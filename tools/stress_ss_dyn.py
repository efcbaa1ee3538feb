"""GPU: randomized trees for the streamed-coefficient kernels (csrc/wdf_ss_dyn.h, lowering.Circuit._run_dyn) against the
oracle's tree interpreter.  Each case: a random binary tree of Series / Parallel / Inverter over resistors, capacitors (<= 4)
and resistive sources (<= 2), a random root (ideal source, diode pair with random N_up / N_down, DenseRootModel 2x4 / 2x8 /
2x16 / 4x4), optionally a per-sample resistance channel on a random resistor / source; random batch and length.  y and the
gradient of sum(y gy) to every live component (and Is, nVt) against oracle.tree_fwd / tree_grad (complex step).
A case whose fp32 CPU restatement (the oracle's float build) is itself far from the fp64 one is ill-conditioned (a random
network root on a random tree can have a loop gain near one): its bounds scale with that distance.
usage: python tools/stress_ss_dyn.py [cases, default 40] [seed, default 0]   -> worst errors; exit status 1 on a violation."""
import os, sys
import numpy as np, torch
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(_R, "differentiable-wdfs_amd", "lib")); sys.path.insert(0, os.path.join(_R, "oracle"))
import tf_wdf as wdf
from tf_wdf import tf
from layers import DenseRootModel
from wdf_hip import workload
import oracle as O

FS = 48000.0
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)


def net_json(name):
    wh, hidden, n_layers = workload.reference_mlp_weights(name)
    layers, o, n_in = [], 0, 2
    for i in range(n_layers + 1):
        n_out = hidden if i < n_layers else 1
        k = wh[o:o + n_in * n_out].reshape(n_in, n_out); o += n_in * n_out
        b = wh[o:o + n_out]; o += n_out
        layers.append({"type": "dense", "activation": "tanh" if i < n_layers else "", "shape": [None, n_out], "weights": [k.tolist(), b.tolist()]})
        n_in = n_out
    return {"in_shape": [None, 2], "layers": layers}, wh.astype(np.float64), [2] + [hidden] * n_layers + [1]


def cuda(a):
    return torch.as_tensor(np.ascontiguousarray(a, dtype=np.float32), device="cuda")


worst = {"y": 0.0, "grad": 0.0}
bad = 0
done = 0
while done < n_cases:
    n_caps, n_src = int(rng.integers(1, 5)), int(rng.integers(0, 3))
    root_kind = rng.choice(["ideal", "diode", "mlp"])
    if root_kind != "ideal" and n_src == 0:
        n_src = 1
    if root_kind == "ideal" and n_src == 2:
        n_src = 1                                      # (ni <= 2: the ideal source is a channel too)
    n_res = int(rng.integers(1, 4))
    leaves = [("C", float(np.exp(rng.uniform(np.log(2e-9), np.log(2e-7))))) for _ in range(n_caps)] + \
             [("V", float(np.exp(rng.uniform(np.log(3e2), np.log(2e4))))) for _ in range(n_src)] + \
             [("R", float(np.exp(rng.uniform(np.log(1e3), np.log(1e5))))) for _ in range(n_res)]
    rng.shuffle(leaves)
    nodes, theta, params, elems = [], [], [], []
    vin = [0]

    def leaf(kind, val):
        p = len(theta)
        theta.append(float(np.float32(val)))
        if kind == "C":
            e = wdf.Capacitor(val, FS, True); nodes.append([O.NODE_CAPACITOR, -1, -1, p, -1, -1]); params.append(e.C)
        elif kind == "R":
            e = wdf.Resistor(val, True); nodes.append([O.NODE_RESISTOR, -1, -1, p, -1, -1]); params.append(e.R)
        else:
            e = wdf.ResistiveVoltageSource(val, trainable=True); nodes.append([O.NODE_RES_VSOURCE, -1, -1, p, -1, -1]); params.append(e.R)
        elems.append((e, len(nodes) - 1, kind))
        return e, len(nodes) - 1

    # post-order construction: combine a shuffled list of subtrees pairwise
    items = [leaf(k, v) for k, v in leaves]
    while len(items) > 1:
        i = int(rng.integers(0, len(items) - 1))
        (ea, na), (eb, nb) = items[i], items[i + 1]
        if rng.random() < 0.5:
            e = wdf.Series(ea, eb); nodes.append([O.NODE_SERIES, na, nb, -1, -1, -1])
        else:
            e = wdf.Parallel(ea, eb); nodes.append([O.NODE_PARALLEL, na, nb, -1, -1, -1])
        items[i:i + 2] = [(e, len(nodes) - 1)]
        if rng.random() < 0.15:
            e2 = wdf.Inverter(items[i][0]); nodes.append([O.NODE_INVERTER, items[i][1], -1, -1, -1, -1]); items[i] = (e2, len(nodes) - 1)
    top, ntop = items[0]
    # channel numbers: the Circuit numbers the sources in POST-ORDER of its own walk
    order = wdf._lowering._walk(top)
    src_elems = [e for e in order if type(e).__name__ == "ResistiveVoltageSource"]
    for e, ni_, kind in elems:
        if kind == "V":
            nodes[ni_][4] = src_elems.index(e)
    ni = len(src_elems) + (1 if root_kind == "ideal" else 0)
    caps = [e for e, _, k in elems if k == "C"]
    probe_e, probe_n, _ = elems[int(rng.integers(0, len(elems)))]
    pot = None
    if rng.random() < 0.6:
        cands = [(e, n_, k) for e, n_, k in elems if k in ("R", "V")]
        pot = cands[int(rng.integers(0, len(cands)))]
        nodes[pot[1]][5] = ni
    n_in = ni + (1 if pot else 0)
    kw = {}
    extra_params = []
    if root_kind == "ideal":
        root = wdf.IdealVoltageSource(); kw = dict(root_kind=O.ROOT_IDEAL_VSOURCE, root_vin=ni - 1)
    elif root_kind == "diode":
        n_up, n_down = int(rng.integers(1, 4)), int(rng.integers(1, 4))
        Is, nVt = 4.352e-9 * float(np.exp(rng.uniform(-1, 1))), 0.0493 * float(rng.uniform(0.8, 1.3))
        root = wdf.DiodePair(top, Is, Vt=nVt, nDiodes=1.0, N_up=n_up, N_down=n_down, trainable=True)
        kw = dict(root_kind=O.ROOT_DIODE_PAIR, p_is=len(theta), p_nvt=len(theta) + 1, n_up=n_up, n_down=n_down)
        theta += [float(np.float32(Is)), float(np.float32(nVt))]
        extra_params = [root.Is, root.nVt]
    else:
        name = str(rng.choice(["2x4", "2x8", "2x16", "4x4"]))
        js, w64, sizes = net_json(name)
        root = DenseRootModel(js)
        kw = dict(root_kind=O.ROOT_MLP, mlp_off=len(theta), mlp_sizes=sizes, mlp_act=[O.ACT_TANH] * (len(sizes) - 2) + [O.ACT_NONE])
        theta += [float(np.float32(v)) for v in w64]
    is_clipper = (type(top).__name__ == "Parallel" and type(top.P1).__name__ == "ResistiveVoltageSource" and type(top.P2).__name__ == "Capacitor")
    if pot is None and root_kind != "mlp":
        continue                                       # (static diode / ideal trees are the other kernels' business)
    try:
        circ = wdf.Circuit(top, root, probe_e, per_sample_R=pot[0] if pot else None, force_generic=True,
                           **({"time_parallel": None} if os.environ.get("STRESS_SEQUENTIAL") else {}))   # (A/B: the sequential kernels only)
    except Exception as exc:                           # e.g. no source in the tree
        continue
    if not circ._dyn:
        continue
    B, T = int(rng.integers(1, 90)), int(rng.integers(8, 400))
    x = (rng.standard_normal((B, T, ni)) * rng.uniform(0.3, 2.0)).astype(np.float32)
    if pot:
        lo, hi = (3e2, 2e4) if pot[2] == "V" else (1e3, 1e5)
        rr = np.exp(rng.uniform(np.log(lo), np.log(hi), (B, 1))) * (1.0 + 0.3 * np.sin(np.arange(T)[None, :] * rng.uniform(0.005, 0.05, (B, 1))))
        xin = np.concatenate([x, rr[:, :, None].astype(np.float32)], axis=-1)
    else:
        xin = x
    gy = (rng.standard_normal((T, B)) / (B * T)).astype(np.float32)
    with tf.GradientTape() as tape:
        y = circ(cuda(xin))
        loss = tf.reduce_sum(y * cuda(gy))
    live = [i for i, (e, n_, k) in enumerate(elems) if pot is None or e is not pot[0]]
    plist = [params[i] for i in live] + extra_params
    grads = tape.gradient(loss, plist)
    oc = O.Circuit([tuple(n) for n in nodes], top=ntop, probe=probe_n, n_in=n_in, fs=FS, **kw)
    th = np.array(theta, dtype=np.float64)
    y_ref = O.tree_fwd(oc, th, xin.astype(np.float64))
    pidx = [nodes[elems[i][1]][3] for i in live] + ([kw["p_is"], kw["p_nvt"]] if root_kind == "diode" else [])
    g_ref = O.tree_grad(oc, th, xin.astype(np.float64), gy.astype(np.float64), params=pidx)
    got = np.array([0.0 if g is None else float(g) for g in grads])
    ey = float(np.max(np.abs(y.cpu().numpy() - y_ref)))
    # how far the SAME recursion in fp32 on the CPU (the oracle's float build) lands from the fp64 one: a tree + root whose step
    # amplifies rounding (loop gain near or above one: a random network root on a random tree can do that) is no measure of the kernels
    y_ref32 = O.tree_fwd(oc, th.astype(np.float32), xin.astype(np.float32), dtype=np.float32)
    ey32 = float(np.max(np.abs(y_ref32.astype(np.float64) - y_ref)))
    if float(np.max(np.abs(y_ref))) < 1e-6 or not np.isfinite(ey32):
        continue                                       # (a probe that sees nothing / a diverging recursion: nothing to compare)
    gmax = float(np.max(np.abs(g_ref * th[pidx])))
    scale = np.abs(g_ref) + 1e-3 * gmax / np.abs(th[pidx])        # (a component the output barely feels)
    eg = 0.0 if gmax < 1e-12 else float(np.max(np.abs(got - g_ref) / scale))     # (an output no component moves: y = x exactly)
    yscale = max(1.0, float(np.max(np.abs(y_ref))))
    worst["y"], worst["grad"] = max(worst["y"], ey / yscale), max(worst["grad"], eg)
    done += 1
    flag = ""
    ill = ey32 > 1e-6 * yscale                          # the fp32 restatement itself is that far off: ill-conditioned case
    tol_y = max(5e-6 * yscale, 8.0 * ey32)
    # (the gradient bound allows for what ORDER of summation alone does: the same fp32 network arithmetic evaluated per lane and in
    #  16-lane rows, both builds over these 150 trees on one box, lands 0.1x ... 39x apart in gradient error, geometric mean 1.00:
    #  profiles/README.md, round 5)
    tol_g = 2e-3 if not ill else 0.2            # (ill-conditioned: y is held to the fp32 restatement's own deviation; of the gradient
                                                #  only signs and sizes -- a weak component there is a difference of amplified roundings)
    if not (ey <= tol_y and eg <= tol_g):
        bad += 1
        flag = "  <-- VIOLATION\n      got   " + np.array2string(got, precision=4) + "\n      oracle " + np.array2string(g_ref, precision=4) + \
               "\n      theta " + np.array2string(th[pidx], precision=4)
    flag = (" (ill-conditioned: fp32 oracle %.1e off)" % ey32 if ill else "") + flag
    print(f"case {done}: ns={circ.ns} ni={circ.ni} root={root_kind} pot={'-' if pot is None else pot[2]} clipper={is_clipper} B={B} T={T} "
          f"|y-ref|={ey:.2e} grad rel={eg:.2e}{flag}", flush=True)
print("worst:", worst, "violations:", bad)
sys.exit(1 if bad else 0)

"""Lowering of a tf_wdf element tree + root to the HIP kernels (the fast tier).

Circuit(top, root, probe) walks the tree the user built from tf_wdf elements, probes ONE
time step of it with unit vectors -- every adaptor / one-port is linear in the waves
(tf_wdf.py:31-214), so that step is a = ca.z + da.x ; b = root(a) ; z' = A z + Bx x + E b ;
y = cy.z + dy.x + fy b -- and runs the T-step recursion for the whole batch in one launch of
csrc/wdf_statespace.h (or csrc/wdf_clipper.h for the diode-clipper topology, which also
takes the per-sample resistance channel of clipper_pot.py:116-117).  The probe runs the
elements' own calc_impedance / reflected / incident code on tiny float64 torch vectors, so
the matrices carry the autograd graph back to R and C: the reverse-sweep kernel returns
dL/d(matrices) and torch chains it to the component values -- which is what
tape.gradient(loss, model.trainable_variables) is in the reference (lpf.py:87-90).
"""
import collections
import ctypes
import math
import os
import weakref
from collections import namedtuple

import numpy as np
import torch

from . import binding
from . import compat_tf as tf
from . import warmstart


# ------------------------------------------------------------------------------ autograd glue
SsTpPlan = namedtuple("SsTpPlan", ["k_fwd", "warmup", "tol", "k_bwd"])
LAST_SS_TP_STATUS = {"status": None}
N_SIMD = 1024            # MI355X: 256 CUs x 4


def plan_ss_time_parallel(coef64, ns, ni, root_kind, B, T, tol=1.0e-6):
    """SsTpPlan for the time-parallel state-space kernels (csrc/wdf_statespace.h), or None when the batch already fills the
    chip / the tree has no state.  Host arithmetic on the step's small matrices only (no device work, no sync).
    Reverse sweep: exact, so chunks are added until every SIMD holds ~2 waves (more only add record traffic).  Forward with a diode root: a chunk warms up
    from z = 0 for W steps; W outlasts the slowest mode of the step's Jacobian  A + Da E ca^T  at both ends of the diode's
    slope Da in [-1, 1] (off and fully conducting), with the same 0.01 tol margin as the clipper planner; chunks only while
    a chunk is at least as long as its warm-up.  The device verifies every boundary whatever the estimate."""
    if ns < 1:
        return None
    waves = max(1, -(-B // 64))
    k_bwd = min(T // 64, (2 * N_SIMD) // waves)           # (measured at 8192 x 4096: 16 chunks 0.179 ms, 32: 0.182, 64: 0.221)
    k_fwd, W = 1, 0
    if root_kind == binding.ROOT_DIODE_PAIR:
        c = coef64.detach().double().cpu().numpy()
        A = c[:ns * ns].reshape(ns, ns)
        oE = ns * ns + ns * ni
        E, ca = c[oE:oE + ns], c[oE + ns:oE + 2 * ns]
        rho = max(float(np.max(np.abs(np.linalg.eigvals(A + sgn * np.outer(E, ca))))) for sgn in (1.0, -1.0))
        if rho < 1.0 - 1e-9:
            W = 16 if rho <= 0.0 else max(16, -(-int(math.ceil(math.log(0.01 * tol) / math.log(rho))) // 8) * 8)
            k_fwd = min(T // max(W, 64), (2 * N_SIMD) // waves)
    if k_fwd < 2 and k_bwd < 2:
        return None
    return SsTpPlan(max(1, k_fwd), W, float(tol), max(1, k_bwd))


class SsWarmStart:
    """Warm-started chunks for the time-parallel state-space forward when the SAME batch is visited again with slightly
    different coefficients (a training loop: lpf.py:86-99 re-runs data_in every epoch): chunk k starts from the state the
    previous call had at the sample its warm-up begins -- the secant through the last two calls once there are two -- so a
    fraction of the cold warm-up closes the gap, and with the shorter warm-up more chunks pay.  The device verifies every
    boundary as always; warmstart.WarmUpController steers the warm-up from its verdicts.  One object per (batch, circuit);
    made by Circuit.__call__, keyed on the caller's tensor and its version."""

    def __init__(self, T, B, ns, plan, secant=True):
        self.T, self.B, self.ns, self.plan, self.secant = int(T), int(B), int(ns), plan, bool(secant)
        self.k_max = max(2, (2 * N_SIMD) // max(1, -(-self.B // 64)))
        # (misses here are real -- a diode root contracts -- so one repaired wave counts; far inside the tolerance the
        #  controller takes two units at a time)
        self.ctl = warmstart.WarmUpController(plan.warmup, plan.warmup // 4, unit=16, floor=16, miss_waves=1, wait_calls=16,
                                              bold_below=4.0, tol=plan.tol)
        self.rows, self.prev = {}, {}
        self._idx = {}
        self.trace = collections.deque(maxlen=256)      # warm-up used by the last calls (probing)

    def chunks(self, W):
        return binding.lib().wdf_ss_tp_chunks(self.T, max(2, min(self.T // max(W, 64), self.k_max)))

    def start(self):
        """-> (chunks, warm-up, zinit) for this call; (plan's, cold, None) when there is nothing to start from yet."""
        self.ctl.cold = self.plan.warmup
        W = self.ctl.begin(self.rows)
        if W is None:
            return self.plan.k_fwd, self.plan.warmup, None
        r1, r2 = self.rows[W], (self.prev.get(W) if self.secant else None)
        return self.chunks(W), W, (r1 if r2 is None else torch.lerp(r2, r1, 2.0))

    def finish(self, zs, status, was_warm, w_used):
        """After the forward: keep this call's states at the chunk starts the next call may use; queue the verdict."""
        if was_warm:
            self.ctl.end(status, w_used)
        cands = self.ctl.candidates()
        key = tuple(cands)
        if key not in self._idx:
            if len(self._idx) > 16:
                self._idx.clear()
            starts = [binding.ss_tp_starts(self.T, self.chunks(w), w) for w in cands]
            self._idx[key] = (torch.tensor([t for s_ in starts for t in s_], dtype=torch.int64, device=zs.device),
                              [len(s_) for s_ in starts])
        idx, lens = self._idx[key]
        rows = zs.index_select(0, idx).split(lens, 0)            # [sum of chunk counts, ns, B] floats; the stash is not kept
        self.prev = {w: self.rows[w] for w, n in zip(cands, lens) if w in self.rows and self.rows[w].shape[0] == n}
        self.rows = dict(zip(cands, rows))
        self.trace.append(w_used)


class DynWarmStart:
    """Warm-started chunks for the streamed-coefficient forward (csrc/wdf_ss_dyn.h) when the SAME batch is visited again with
    components one optimizer step apart: the previous call's whole state stash is kept (T ns B floats), so chunk k may start
    any number of steps W before its first owned step from the state that call had there; warmstart.WarmUpController steers
    W from the device's verdicts (a miss costs a sequential re-run of the waves concerned, never a wrong result), and with
    the shorter warm-up more chunks pay.  A tree whose cold warm-up outlasts its chunks (HPFDiodeClipper.h's: 672 steps) runs
    its first call sequentially and is chunked from the second on."""

    def __init__(self, T, B, ns, k_max, cold_W, tol):
        self.T, self.B, self.ns, self.k_max = int(T), int(B), int(ns), int(k_max)
        cap = max(32, min(int(cold_W), self.T // 2) // 16 * 16)
        self.ctl = warmstart.WarmUpController(cap, max(32, cap // 2), unit=16, floor=16, miss_waves=1, wait_calls=16, bold_below=4.0,
                                              tol=tol)
        self.prev = None
        self.trace = collections.deque(maxlen=256)

    def start(self):
        """-> (chunks, warm-up, zinit [K, ns, B]) or None (no earlier call to start from)."""
        if self.prev is None:
            return None
        W = self.ctl.begin(range(0, 1 << 30))
        K = binding.dyn_chunks(self.T, max(2, min(self.k_max, self.T // max(W, 64))))
        if K < 2:
            return None
        Lc = binding.dyn_chunk_len(self.T, K)
        key = (K, W)
        if getattr(self, "_idx_key", None) != key:
            self._idx_key = key
            self._idx = torch.tensor([max(0, k * Lc - W) for k in range(K)], dtype=torch.int64, device=self.prev.device)
        return K, W, self.prev.index_select(0, self._idx).contiguous()

    def finish(self, zs, status, was_warm, w_used):
        if was_warm and status is not None:
            self.ctl.end(status, w_used)
        self.prev = zs
        self.trace.append(w_used)


class _StateSpaceFn(torch.autograd.Function):
    """y [T,B] = statespace(coef, rootp, x [B,T,ni], z0).  tp: an SsTpPlan (time-parallel kernels) or None.
    warm: an SsWarmStart for this batch (training loops) or None."""

    @staticmethod
    def forward(ctx, coef, rootp, x, z0, ns, ni, root_kind, n_up, n_down, want_zT, tp=None, warm=None):
        need = coef.requires_grad or (rootp is not None and rootp.requires_grad) or (z0 is not None and z0.requires_grad)
        c = coef.detach().contiguous()
        rp = None if rootp is None else rootp.detach().contiguous()
        z0d = None if z0 is None else z0.detach().contiguous()
        B, T = x.shape[0], x.shape[1]
        k_lin = min(T // 64, (2 * 1024) // max(1, -(-B // 64))) if (root_kind == binding.ROOT_NONE and ns > 0) else 0
        if k_lin >= 2:      # linear tree, few sequences: the exact chunked scan (csrc/wdf_statespace.h) fills the chip
            y, zs, zT = binding.ss_fwd_lin_tp(x, c, ns, ni, k_lin, want_stash=need, z0=z0d, want_zT=want_zT)
        elif tp is not None and tp.k_fwd >= 2 and root_kind == binding.ROOT_DIODE_PAIR:
            # nonlinear root: chunks warmed up from z = 0, verified on the device, missed waves re-run sequentially
            k, W, zinit = (tp.k_fwd, tp.warmup, None) if (warm is None or not need or z0d is not None) else warm.start()
            y, zs, zT, st = binding.ss_fwd_tp(x, c, ns, ni, rp, k, W, tp.tol, n_up, n_down, want_stash=need,
                                              z0=z0d, want_zT=want_zT, zinit=zinit)
            LAST_SS_TP_STATUS["status"] = st
            LAST_SS_TP_STATUS["warmup_used"], LAST_SS_TP_STATUS["chunks_used"] = W, k
            if warm is not None and need and z0d is None:
                warm.finish(zs, st, zinit is not None, W)
        else:
            y, zs, zT = binding.ss_fwd(x, c, ns, ni, root_kind, rp, n_up, n_down, want_stash=need, z0=z0d, want_zT=want_zT)
        ctx.cfg = (ns, ni, root_kind, n_up, n_down, z0 is not None, tp)
        ctx.save_for_backward(c, rp, x, zs)
        if want_zT:
            ctx.mark_non_differentiable(zT)
            return y, zT
        return y, None

    @staticmethod
    def backward(ctx, gy, _gzT):
        ns, ni, root_kind, n_up, n_down, has_z0, tp = ctx.cfg
        c, rp, x, zs = ctx.saved_tensors
        if tp is not None and tp.k_bwd >= 2 and ns >= 1:      # exact chunked reverse sweep (any root)
            gcoef, groot, gz0 = binding.ss_bwd_tp(x, c, ns, ni, zs, gy.contiguous(), tp.k_bwd, root_kind, rp, n_up, n_down,
                                                  want_gz0=has_z0)
        else:
            gcoef, groot, gz0 = binding.ss_bwd(x, c, ns, ni, zs, gy.contiguous(), root_kind, rp, n_up, n_down,
                                               want_gz0=has_z0)
        return gcoef, groot, None, gz0, None, None, None, None, None, None, None, None


# The device tape interpreter keeps a sample's node values in LDS: its time grows with the square of the tape's length, torch's
# (one launch per operation over the whole channel) linearly -- past this many operations torch runs the tape
# (tools/ss_dyn_rows_crossover.py, profiles/r05_rows_crossover.txt: 32 operations 1.6 vs 2.2 ms per step, 48: 2.5 vs 3.3, 89: 5.5 vs 5.2, 138: 12.9 vs 7.5).
DYN_ROWS_MAX_OPS = int(os.environ.get("WDF_DYN_ROWS_MAX_OPS", "80"))


class _DynRowsFn(torch.autograd.Function):
    """rows [T,n,B] (or the static row [n]) of a probed step from its tape, on the device (csrc/wdf_ss_dyn_rows.h):
    calc_impedance for every sample in one launch; backward: dLoss/d(component value) from the rows' adjoint in two."""

    @staticmethod
    def forward(ctx, rtape, chan, r, *vals):
        params = torch.stack([v.detach().double().reshape(()) for v in vals]) if vals else torch.zeros((0,), dtype=torch.float64, device=r.device)
        rows = binding.ss_dyn_rows(rtape, params, chan, r)
        ctx.rtape, ctx.chan, ctx.r, ctx.params = rtape, chan, r, params
        ctx.dtypes = [v.dtype for v in vals]
        return rows if r is not None else rows.reshape(-1)

    @staticmethod
    def backward(ctx, grows):
        g = grows.float().contiguous()
        if ctx.r is None:
            g = g.reshape(1, -1, 1)
        gp = binding.ss_dyn_rows_bwd(ctx.rtape, ctx.params, ctx.chan, ctx.r, g)
        need = ctx.needs_input_grad[3:]
        return (None, None, None) + tuple(gp[i].to(ctx.dtypes[i]) if need[i] else None for i in range(len(need)))


class _SsDynFn(torch.autograd.Function):
    """y [T,B] = the state-space recursion with streamed coefficient rows and / or the MLP root (csrc/wdf_ss_dyn.h).
    rows: [T,n,B] (per-sample impedance) or [n] (static), n = A | Bx | E | ca | da | cy | dy | fy | R_port.
    rootvec: {Is, nVt} (diode root), the flat weights (MLP root) or None.  The reverse sweep hands back dL/d(row entry) for
    every sample; torch chains it through the rows' own graph (the probe tape evaluated over the resistance channel) to the
    component values -- calc_impedance's chain rule per sample (tf_wdf.py:114-115,139-145,168-177)."""

    @staticmethod
    def forward(ctx, rows, rootvec, x, z0, ns, ni, kind, hidden, n_tanh, n_up, n_down, want_zT, tp=None, warm=None):
        need = rows.requires_grad or (rootvec is not None and rootvec.requires_grad) or (z0 is not None and z0.requires_grad)
        need = need or (warm is not None and ns >= 1)                         # (the next call's chunks start from this call's states)
        hot = warm.start() if (warm is not None and z0 is None) else None
        if hot is not None:
            tp = SsTpPlan(hot[0], hot[1], tp.tol if tp is not None else 1.0e-6, tp.k_bwd if tp is not None else 1)
        r = rows.detach().float().contiguous()
        rv = None if rootvec is None else rootvec.detach().float().contiguous()
        z0d = None if z0 is None else z0.detach().float().contiguous()
        rootp, w = (rv, None) if kind == binding.ROOT_DIODE_PAIR else (None, rv)
        if tp is not None and tp.k_fwd >= 2 and ns >= 1:       # verified time chunks; the waves that missed re-run sequentially
            y, zs, zT, st = binding.ss_dyn_fwd_tp(x, r, ns, ni, tp.k_fwd, tp.warmup, tp.tol, kind, rootp=rootp, w=w, hidden=hidden,
                                                  n_tanh=n_tanh, n_up=n_up, n_down=n_down, want_stash=need, z0=z0d, want_zT=want_zT,
                                                  zinit=None if hot is None else hot[2])
            LAST_SS_TP_STATUS["status"] = st
            LAST_SS_TP_STATUS["warmup_used"], LAST_SS_TP_STATUS["chunks_used"] = tp.warmup, tp.k_fwd
        else:
            st = None
            y, zs, zT = binding.ss_dyn_fwd(x, r, ns, ni, kind, rootp=rootp, w=w, hidden=hidden, n_tanh=n_tanh, n_up=n_up, n_down=n_down,
                                           want_stash=need, z0=z0d, want_zT=want_zT)
        if warm is not None and z0 is None and zs is not None:
            warm.finish(zs, st, hot is not None, tp.warmup if (tp is not None and st is not None) else 0)
        ctx.tp = tp
        ctx.cfg = (ns, ni, kind, hidden, n_tanh, n_up, n_down, z0 is not None, rows.dim() == 3)   # (3 dims: [T,n,B] or [1,n,B])
        ctx.save_for_backward(r, rv, x, zs)
        if want_zT:
            ctx.mark_non_differentiable(zT)
            return y, zT
        return y, None

    @staticmethod
    def backward(ctx, gy, _gzT):
        ns, ni, kind, hidden, n_tanh, n_up, n_down, has_z0, per_sample = ctx.cfg
        r, rv, x, zs = ctx.saved_tensors
        rootp, w = (rv, None) if kind == binding.ROOT_DIODE_PAIR else (None, rv)
        if zs is None:
            zs = torch.zeros((x.shape[1], 1, x.shape[0]), dtype=torch.float32, device=x.device)      # (ns = 0: nothing to read)
        tp = ctx.tp
        if tp is not None and tp.k_bwd >= 2 and ns >= 1:       # exact time chunks
            grows, groot, gz0 = binding.ss_dyn_bwd_tp(x, r, ns, ni, zs, gy.contiguous(), tp.k_bwd, kind, rootp=rootp, w=w, hidden=hidden,
                                                      n_tanh=n_tanh, n_up=n_up, n_down=n_down, want_gz0=has_z0)
        else:
            grows, groot, gz0 = binding.ss_dyn_bwd(x, r, ns, ni, zs, gy.contiguous(), kind, rootp=rootp, w=w, hidden=hidden, n_tanh=n_tanh,
                                                   n_up=n_up, n_down=n_down, want_gz0=has_z0)
        if not per_sample:
            # static row: the kernel summed over the steps itself (float64 accumulators, grows [1,n,B]: no [T,n,B] array of
            # per-sample products exists any more); what is left is the sum over the sequences
            grows = grows.sum(dim=(0, 2), dtype=torch.float64).float()
        return (grows, None if groot is None else groot.float(), None, (gz0[:ns] if has_z0 else None), None, None, None, None, None,
                None, None, None, None, None)


# ------------------------------------------------------------------------------ resident entries: which batch is this?
def tensor_key(t):
    """What names a batch for the resident paths' caches: the STORAGE a tensor looks at (address, shape, strides, dtype,
    device) and its version counter -- not the Python object.  A training loop that cuts the same mini-batches out of its
    dataset every epoch (`X[i:j]`: a fresh tensor object each time, the same memory) then finds its stepper, its resident
    time-major copy and its warm-start state again; an in-place change of the data bumps the version (shared by all views
    of a storage) and misses.  The entry keeps the first tensor it was made for alive, so the address cannot be handed to
    other data while the entry exists."""
    return (t.data_ptr(), tuple(t.shape), tuple(t.stride()), t.dtype, t._version, t.device.type, t.device.index)


class EntryCache:
    """Insertion-ordered cache of resident entries: at most `max_entries`, and -- beyond the first four -- at most
    `max_bytes` of device buffers in total (a full-batch loop keeps a training and a validation set; a mini-batch loop
    keeps every mini-batch of an epoch while they are small)."""

    def __init__(self, max_entries=256, max_bytes=16 << 30):
        self.d, self.nbytes, self.max_entries, self.max_bytes = {}, {}, int(max_entries), int(max_bytes)

    def get(self, key):
        return self.d.get(key)

    def put(self, key, ent, nbytes=0):
        self.d[key] = ent
        self.nbytes[key] = int(nbytes)
        while len(self.d) > self.max_entries or (len(self.d) > 4 and sum(self.nbytes.values()) > self.max_bytes):
            old = next(iter(self.d))
            del self.d[old], self.nbytes[old]
        return ent

    def values(self):
        return self.d.values()

    def __len__(self):
        return len(self.d)


def _nbytes(*ts):
    return sum(t.numel() * t.element_size() for t in ts if isinstance(t, torch.Tensor))


# ------------------------------------------------------------------------------ resident linear trees
_LIN_OCC = int(os.environ.get("WDF_LIN_OCC", "2"))       # chunks are cut so that every SIMD gets this many waves
_NL_OCC = int(os.environ.get("WDF_NL_OCC", "1"))
_NL_WMIN = int(os.environ.get("WDF_NL_WMIN", "16"))      # the shortest warm-up the device's controller may settle on
_ROWS = 1024                                             # result rows per slab (see _LinResident.entry)
NL_TOL = 1.0e-6                                          # boundary tolerance of the diode-root one-pass step (plan_ss_time_parallel's)


class _LinResident:
    """A linear tree (ideal-source root) whose component values live on the device: the probed step as a device tape
    (probe_tape.py), and per (x, target) pair the buffers of the one-pass MSE step (csrc/wdf_ss_step.h)."""

    def __init__(self, circ, device):
        from . import probe_tape
        self.circ = circ
        own = {"Resistor": "R", "ResistiveVoltageSource": "R", "Capacitor": "C"}
        self.params = []                                          # (element, attribute) of every component value, tree order
        for e in circ.elements:
            name = own.get(_kind(e))
            if name is not None:
                self.params.append((e, name))
        pvars = [e.__dict__[n] for e, n in self.params]
        self.n_tree = len(pvars)                                  # the component values the tape sees; then the root's own
        if circ.root_kind == "DiodePair":
            self.params += [(circ.root, "Is"), (circ.root, "nVt")]
            pvars += [circ.root.Is, circ.root.nVt]
        if not 1 <= len(pvars) <= probe_tape.MAX_PARAMS:
            raise binding.WdfHipError(f"Circuit.to_device: 1..{probe_tape.MAX_PARAMS} component values (this tree has {len(pvars)})")
        for (e, n), v in zip(self.params, pvars):
            # what the resident step would silently freeze is refused (as on the clipper path): a value computed from other
            # Variables would get no gradient and never move; a Variable another circuit's block already holds
            if isinstance(v, torch.Tensor) and not getattr(v, "_is_tf_variable", False) and (v.requires_grad or v.grad_fn is not None):
                raise binding.WdfHipError(f"Circuit.to_device: {type(e).__name__}.{n} is a tensor computed from other Variables; the "
                                          "resident step would freeze it -- keep this circuit on the host path")
            if isinstance(v, torch.Tensor) and getattr(v, "_wdf_block", None) is not None:
                raise binding.WdfHipError("Circuit.to_device: a component Variable already lives in another circuit's block")
        # the tape is recorded (and its size checked) BEFORE any Variable moves: a circuit the device probe cannot hold
        # leaves its Variables where they were
        tape, outs, rport = probe_tape.record(circ, pvars[:self.n_tree])
        self.captured = list(pvars)                               # what every (element, attribute) held at to_device()
        self.captured_val = [float(v) for v in pvars]
        self.pb = tf.ParamBlock(self.captured_val, torch.device(device))
        for i, v in enumerate(pvars):
            if isinstance(v, torch.Tensor) and getattr(v, "_is_tf_variable", False) and v.numel() == 1:
                self.pb.adopt(i, v)
        self.adopted = {i: v for i, v in self.pb.members.items()}
        self.host_tape, self.host_outs = tape, outs + [rport]
        ops, consts = tape.packed()
        dev = self.pb.block.device
        self.n_ops, self.n_out = int(ops.shape[0]), len(outs) + 1
        self.tape = torch.as_tensor(ops, device=dev).contiguous()
        self.consts = torch.as_tensor(consts if len(consts) else np.zeros(1), device=dev)
        self.outs = torch.as_tensor(np.asarray(outs + [rport], dtype=np.int32), device=dev)
        self.coef = torch.zeros(self.n_out, dtype=torch.float32, device=dev)
        self.coef64 = torch.zeros(self.n_out, dtype=torch.float64, device=dev)
        self.jac = torch.zeros((self.n_out, self.n_tree), dtype=torch.float64, device=dev)
        self.cache = EntryCache()

    def check(self):
        """Every component value must still be what to_device() captured: an adopted Variable the same object, still in the
        block; a frozen (non-Variable) value the same object or the same number -- set_resistance / `e.R = ...` afterwards
        would otherwise be silently ignored by the resident step."""
        for i, (e, n) in enumerate(self.params):
            v = e.__dict__.get(n)
            if i in self.adopted:
                if v is not self.adopted[i] or getattr(v, "_wdf_block", (None,))[0] is not self.pb:
                    raise binding.WdfHipError(f"Circuit.to_device: {type(e).__name__}.{n} was replaced after to_device(); build a new Circuit")
            elif v is not self.captured[i]:
                same = False
                try:
                    same = (not isinstance(v, torch.Tensor) or v.numel() == 1) and float(v) == self.captured_val[i]
                except (TypeError, ValueError):
                    pass
                if not same:
                    raise binding.WdfHipError(f"Circuit.to_device: {type(e).__name__}.{n} was changed after to_device() (the resident "
                                              "step holds the value it had then); build a new Circuit")
                self.captured[i] = v

    def probe(self):
        """The step's coefficients and their Jacobian from the block, on the device -- behind the optimizer updates the
        script queued since the last step (compat_tf.ParamBlock.defer), in the SAME launch."""
        L = binding.lib()
        jobs = self.pb.take_pending()
        with torch.no_grad(), torch._C.DisableTorchFunctionSubclass():
            arr = binding.adam_jobs(jobs) if jobs else None
            rc = L.wdf_ss_probe_adam(None if arr is None else ctypes.cast(arr, ctypes.c_void_p), len(jobs), binding._ptr(self.tape),
                                     self.n_ops, binding._ptr(self.consts), binding._ptr(self.pb.block), self.n_tree,
                                     binding._ptr(self.outs), self.n_out, binding._ptr(self.coef), binding._ptr(self.coef64),
                                     binding._ptr(self.jac), binding._stream())
        binding._check(rc, "wdf_ss_probe_adam")

    def host_coef(self):
        """The coefficient vector (float64 numpy, and the port resistance) at the block's lagging host mirror: what the
        time-parallel planner needs, without a synchronisation."""
        hv = tuple(self.pb.host_values()[:self.n_tree])
        if getattr(self, "_hc_key", None) != hv:                  # (the mirror moves every few dozen steps: evaluated then only)
            vals, _ = self.host_tape.evaluate(hv, self.host_outs)
            self._hc_key, self._hc_val, self.plans = hv, (vals[:-1], float(vals[-1])), {}
        return self._hc_val

    def entry(self, x, target):
        key = (tensor_key(x), tensor_key(target))
        ent = self.cache.get(key)
        if ent is None:
            circ, dev = self.circ, self.pb.block.device
            xd = x.as_subclass(torch.Tensor).to(dev).float()
            if xd.dim() == 2:
                xd = xd.unsqueeze(-1)
            B, T, ni = xd.shape
            if ni != circ.ni:
                raise binding.WdfHipError(f"x must be [B,T,{circ.ni}] (or [B,T] for one channel), got {tuple(x.shape)}")
            x_tm = xd.permute(1, 2, 0).contiguous()               # [T][ni][B]: once per training set
            tgt = target.as_subclass(torch.Tensor).to(dev).float().reshape(T, B).contiguous()
            per_wave = 128 if B % 2 == 0 else 64                  # (two sequences per lane when the rows pair up)
            L_ = binding.lib()
            y = torch.empty((T, B), dtype=torch.float32, device=dev)
            if circ.root_kind == "DiodePair":
                # the tangent-carried step of a diode-pair root: chunks no shorter than 64 steps, one wave per SIMD and up
                k = max(1, min(T // 64, (_NL_OCC * N_SIMD) // max(1, -(-B // per_wave))))
                ws = self._plan_nl(B, T, k, dev)
            else:
                k = max(1, min(T // 64, (_LIN_OCC * N_SIMD) // max(1, -(-B // per_wave))))
                nbytes = L_.wdf_ss_lin_step_ws_bytes(circ.ns, circ.ni, B, T, k)
                if nbytes == 0:
                    raise binding.WdfHipError(L_.wdf_last_error().decode() or "wdf_ss_lin_step_ws_bytes: unsupported tree")
                ws = torch.zeros((nbytes,), dtype=torch.uint8, device=dev)
            # results of a call: {SSE, gradients} and the loss, each call in a row of its OWN (rows are handed out from a
            # slab of _ROWS; a full slab is replaced by a fresh one and lives on for as long as anything still refers to
            # it): the loss history a script keeps (lpf.py:99 `losses.append(loss)`) and gradients read after the loop stay
            # what they were -- no copy per call, one allocation per _ROWS calls
            ent = self.cache.put(key, {"x": x_tm, "t": tgt, "y": y, "ws": ws,
                                       "ring": torch.zeros((_ROWS, 2 + self.pb.n), dtype=torch.float32, device=dev), "turn": 0,
                                       "B": B, "T": T, "k": k, "hold": (x, target), "calls": 0, "watch": None, "replans": 0},
                                 _nbytes(x_tm, tgt, y, ws))
        return ent

    def _plan_nl(self, B, T, k, dev, cold_floor=0):
        """Workspace of the diode-root step for k chunks, planned: cold first call, then the device steers the warm-up."""
        L_ = binding.lib()
        circ = self.circ
        nbytes = L_.wdf_ss_nl_step_ws_bytes(circ.ns, circ.ni, B, T, k)
        if nbytes == 0:
            raise binding.WdfHipError(L_.wdf_last_error().decode() or "wdf_ss_nl_step_ws_bytes: unsupported tree")
        ws = torch.empty((nbytes,), dtype=torch.uint8, device=dev)
        Lc = L_.wdf_ss_nl_step_chunk_len(T, k)
        cold = min(4096, max(self.cold_warmup(), int(cold_floor) // 16 * 16))
        warm = max(16, min(Lc // 16 * 16, 64))
        binding._check(L_.wdf_ss_nl_step_plan(binding._ptr(ws), circ.ns, circ.ni, B, T, k, cold, warm, _NL_WMIN,
                                              max(16, Lc // 16 * 16), float(NL_TOL), binding._stream()), "wdf_ss_nl_step_plan")
        return ws

    def _watch(self, ent):
        """Every 16th call, WITHOUT a synchronisation: look at the control block as it was 16 calls ago.  Groups that keep
        missing although the warm-up already is as long as a chunk (a circuit whose memory outlasts the chunks: every miss
        costs a sequential pass of its group) -> half as many chunks, twice the room; down to one chunk, which has no
        boundary to miss."""
        ent["calls"] += 1
        if ent["calls"] % 16:
            return
        w = ent["watch"]
        if w is None:
            w = ent["watch"] = {"buf": torch.zeros(32, dtype=torch.int32).pin_memory(), "event": None, "last": 0}
        if w["event"] is not None:
            if not w["event"].query():
                return
            raw = w["buf"].numpy()
            total, w_snap, w_max = int(raw[15]), int(raw[4]), int(raw[6])
            delta, w["last"] = total - w["last"], total
            w["event"] = None
            Lc = binding.lib().wdf_ss_nl_step_chunk_len(ent["T"], ent["k"])
            if delta >= 2 and w_snap >= min(w_max, Lc // 16 * 16) and ent["k"] > 1:
                ent["k"] = max(1, ent["k"] // 2)
                ent["ws"] = self._plan_nl(ent["B"], ent["T"], ent["k"], ent["ws"].device, cold_floor=4 * w_snap)
                ent["watch"], ent["replans"] = None, ent["replans"] + 1
                return
        w["buf"].copy_(ent["ws"][:128].view(torch.int32), non_blocking=True)
        w["event"] = torch.cuda.Event()
        w["event"].record()

    def cold_warmup(self):
        """Warm-up of a chunk that starts from z = 0 (the first call on a batch): outlasts the slowest mode of the step's
        Jacobian A + Da E ca^T over the diode's slope Da in [-1, 1] (plan_ss_time_parallel's estimate), from the host mirror."""
        c, _ = self.host_coef()
        ns, ni = self.circ.ns, self.circ.ni
        A = np.asarray(c[:ns * ns]).reshape(ns, ns)
        oE = ns * ns + ns * ni
        E, ca = np.asarray(c[oE:oE + ns]), np.asarray(c[oE + ns:oE + 2 * ns])
        rho = max(float(np.max(np.abs(np.linalg.eigvals(A + sgn * np.outer(E, ca))))) for sgn in (1.0, -1.0))
        if rho <= 0.0:
            return 16
        if rho >= 1.0 - 1e-9:
            return 4096
        return int(min(4096, max(16, -(-int(math.ceil(math.log(0.01 * NL_TOL) / math.log(rho))) // 16) * 16)))

    def read_ctl(self, ent):
        """The step's control block (csrc/wdf_ss_nl_step.h, NlStepCtl) as a dict -- synchronises."""
        raw = np.zeros(32, dtype=np.int32)
        binding._check(binding.lib().wdf_ss_nl_step_read(binding._ptr(ent["ws"]), raw.ctypes.data, binding._stream()), "wdf_ss_nl_step_read")
        f = raw.view(np.float32)
        return {"call": int(raw[0]), "parity": int(raw[1]), "have_snap": int(raw[2]), "w_cur": int(raw[3]), "w_snap": int(raw[4]),
                "w_min": int(raw[5]), "w_max": int(raw[6]), "cool": int(raw[7]), "tol": float(f[8]), "n_bad": int(raw[12]),
                "max_miss": float(f[13]), "gated_groups": int(raw[14]), "total_gated": int(raw[15]), "w_used": int(raw[19])}

    def step(self, ent, z0=None, want_zT=False):
        """probe + one-pass step -> (out = {SSE, d(mean squared error)/d component value}, loss = the mean squared error): views
        of this call's own result row (never written again).  z0 [ns,B] float32 on the device: the capacitor states the call
        starts from (linear trees; None: zero); want_zT: the states it ends in are left in ent["zT"] (a buffer of its own per
        call parity, never the one z0 may be)."""
        circ = self.circ
        self.probe()
        B, T, n = ent["B"], ent["T"], self.pb.n
        if ent["turn"] >= ent["ring"].shape[0]:
            ent["ring"], ent["turn"] = torch.zeros_like(ent["ring"]), 0
        row = ent["ring"][ent["turn"]]
        ent["turn"] += 1
        out, loss = row[:1 + n], row[1 + n]
        if circ.root_kind == "DiodePair":
            rc = binding.lib().wdf_ss_nl_step_mse(binding._ptr(ent["x"]), binding._ptr(self.coef), binding._ptr(self.pb.block),
                                                  binding._ptr(self.jac), self.n_tree, circ.ns, circ.ni, int(circ.root.N_up),
                                                  int(circ.root.N_down), binding._ptr(ent["t"]), 2.0 / float(B * T),
                                                  binding._ptr(ent["y"]), binding._ptr(ent["ws"]), binding._ptr(out),
                                                  binding._ptr(loss), B, T, ent["k"], binding._stream())
            binding._check(rc, "wdf_ss_nl_step_mse")
            self._watch(ent)
            return out, loss
        zT = None
        if want_zT and circ.ns > 0:
            bufs = ent.setdefault("zT_bufs", [torch.empty((circ.ns, B), dtype=torch.float32, device=ent["y"].device) for _ in range(2)])
            zT = bufs[0] if (z0 is None or z0.data_ptr() != bufs[0].data_ptr()) else bufs[1]
        ent["zT"] = zT
        rc = binding.lib().wdf_ss_lin_step_mse(binding._ptr(ent["x"]), binding._ptr(self.coef), binding._ptr(self.jac), self.pb.n,
                                               circ.ns, circ.ni, binding._ptr(ent["t"]), 2.0 / float(B * T), binding._ptr(ent["y"]),
                                               binding._ptr(ent["ws"]), binding._ptr(out), binding._ptr(loss), None, B, T, ent["k"],
                                               None if z0 is None else binding._ptr(z0), None if zT is None else binding._ptr(zT),
                                               binding._stream())
        binding._check(rc, "wdf_ss_lin_step_mse")
        return out, loss


class _DynResident:
    """Circuit.to_device() of a circuit that runs on the streamed-coefficient kernels (per-sample impedance / an MLP root outside
    the clipper topology, csrc/wdf_ss_dyn.h): the scalar component Variables (and a diode root's Is, nVt) become 0-dim views of one
    device block (compat_tf.ParamBlock: same objects, same constraints, same autograd leaves; tf.keras.optimizers.Adam updates
    them on the device), a DenseRootModel's kernels and biases views of one flat device vector.  _run_dyn is unchanged -- it
    only ever handed these values to device code; what goes is the host round trip per Variable and step."""

    def __init__(self, circ, device):
        own = {"Resistor": "R", "ResistiveVoltageSource": "R", "Capacitor": "C"}
        params = [(e, own[_kind(e)]) for e in circ.elements if _kind(e) in own]
        if circ.root_kind == "DiodePair":
            params += [(circ.root, "Is"), (circ.root, "nVt")]
        pvars = [e.__dict__[n] for e, n in params]
        for (e, n), v in zip(params, pvars):
            if isinstance(v, torch.Tensor) and not getattr(v, "_is_tf_variable", False) and (v.requires_grad or v.grad_fn is not None):
                raise binding.WdfHipError(f"Circuit.to_device: {type(e).__name__}.{n} is a tensor computed from other Variables -- "
                                          "keep this circuit on the host path")
            if isinstance(v, torch.Tensor) and getattr(v, "_wdf_block", None) is not None:
                raise binding.WdfHipError("Circuit.to_device: a component Variable already lives in another circuit's block")
        self.pb = tf.ParamBlock([float(v) for v in pvars], torch.device(device))
        for i, v in enumerate(pvars):
            if isinstance(v, torch.Tensor) and getattr(v, "_is_tf_variable", False) and v.numel() == 1:
                self.pb.adopt(i, v)
        self.w = None
        if circ.root_kind == "DenseRootModel":
            from . import mlp_root
            dense, _, _ = mlp_root.describe(circ.root)
            self.w = mlp_root.flat_weights(dense).detach().float().to(device).contiguous()
            o = 0
            for d in dense:
                for v in (d.kernel, d.bias):
                    n = v.numel()
                    if getattr(v, "_wdf_flat", None) is not None:
                        raise binding.WdfHipError("Circuit.to_device: this network's weights already live in another circuit's vector")
                    with torch.no_grad():
                        v.data = self.w[o:o + n].view(v.shape)
                    o += n


class _ProbeFn(torch.autograd.Function):
    """(coef float32 [ncoef], rootp float32 [3] | None) of a resident tree from its parameter block, on the device
    (wdf_ss_probe); backward contracts dLoss/d coef with the probe's Jacobian -- calc_impedance's chain rule
    (tf_wdf.py:114-115,139-145,168-177) without the host."""
    _wdf_sends_pending = True        # (res.probe() carries the queued optimizer updates: compat_tf.Tensor.__torch_function__)

    @staticmethod
    def forward(ctx, res, idx, *live):
        res.probe()
        ncoef = res.n_out - 1
        coef = res.coef[:ncoef].clone()
        ctx.res, ctx.idx, ctx.ncoef = res, idx, ncoef
        ctx.save_for_backward(res.jac.clone())
        if res.circ.root_kind == "DiodePair":
            rootp = torch.cat([res.pb.block[res.n_tree:res.n_tree + 2], res.coef[ncoef:ncoef + 1]])
            return coef, rootp
        return coef, coef.new_zeros(3)

    @staticmethod
    def backward(ctx, gcoef, grootp):
        (jac,) = ctx.saved_tensors
        res = ctx.res
        g = torch.cat([gcoef, grootp[2:3]]).double() @ jac                    # [n_tree]
        if res.circ.root_kind == "DiodePair":
            g = torch.cat([g, grootp[0:2].double()])
        g = g.float()
        return (None, None) + tuple(g[i] for i in ctx.idx)


class _LinResidentMseFn(torch.autograd.Function):
    """loss = mean((y - target)^2) of a resident linear tree: the one-pass step produced d loss / d Variable with it."""
    _wdf_sends_pending = True        # (res.step() begins with res.probe(), which carries the queued optimizer updates)

    @staticmethod
    def forward(ctx, res, ent, inv_n, idx, z0, want_zT, *live):
        out, loss = res.step(ent, z0, want_zT)                   # (this call's own result row: nothing to copy)
        ctx.save_for_backward(out)
        ctx.idx = idx
        ctx.mark_non_differentiable(out)
        return loss, out

    @staticmethod
    def backward(ctx, gl, _):
        (out,) = ctx.saved_tensors
        g = gl * out
        return (None,) * 6 + tuple(g[1 + i] for i in ctx.idx)


# ------------------------------------------------------------------------------ tree walking
def _kind(e):
    return type(e).__name__


def _walk(top):
    """Post-order list of the elements under `top` (P1 before P2, children before parents)."""
    order, seen = [], set()

    def visit(e):
        if id(e) in seen:
            raise ValueError("a WDF element appears twice in the tree")
        seen.add(id(e))
        for child in (getattr(e, "P1", None), getattr(e, "P2", None)):
            if child is not None:
                visit(child)
        order.append(e)

    visit(top)
    return order


_STATE_ATTRS = ("a", "b", "z", "Vs", "R", "p1R", "p2R", "b_diff", "b_temp")


class _Saved:
    """Saves / restores the wave attributes the probe overwrites."""

    def __init__(self, elements):
        self.snap = [(e, {k: e.__dict__[k] for k in _STATE_ATTRS if k in e.__dict__}) for e in elements]

    def restore(self):
        for e, d in self.snap:
            for k in _STATE_ATTRS:
                if k in d:
                    e.__dict__[k] = d[k]
                elif k in e.__dict__ and k not in ("R",):
                    del e.__dict__[k]


def diode_pair_reflected(dp):
    """DiodePair.reflected(): recorded by the loop recorder, or -- on concrete GPU tensors --
    evaluated element-wise by the HIP kernel (no autograd; the differentiable path is Circuit)."""
    a = dp.a
    if hasattr(a, "__wdf_root__"):
        return a.__wdf_root__(dp)
    if isinstance(a, torch.Tensor) and a.is_cuda:
        R = tf.convert(dp.R, dtype=torch.float32, device=a.device).expand_as(a).contiguous()
        return binding.diode_pair(a.contiguous().float(), R, float(dp.Is), float(dp.nVt), dp.N_up, dp.N_down)
    raise binding.WdfHipError(
        "DiodePair.reflected() needs the GPU: pass CUDA tensors, or run the loop through "
        "tf_wdf.Circuit / a recorded TensorArray loop (there is no CPU fallback)")


# ------------------------------------------------------------------------------ Circuit
class Circuit:
    """The fast tier.  top: element connected to the root (e.g. the Inverter of lpf.py:28 or
    P1 of clipper_pot.py:99); root: IdealVoltageSource, DiodePair or layers.DenseRootModel;
    probe: element whose voltage() is the output (C1 in lpf.py:44, C in clipper_pot.py:123).

    Input channels: channel k of x feeds the k-th voltage source found walking the tree in
    post-order (ResistiveVoltageSource leaves), then the ideal-source root if there is one.
    per_sample_R: a ResistiveVoltageSource whose resistance is streamed from the NEXT input
    channel (clipper_pot.py:114-116: channel 0 = Vin, channel 1 = R); diode-clipper
    topology only.
    """

    def __init__(self, top, root, probe, per_sample_R=None, force_generic=False, time_parallel="auto", warm_start=True):
        self.top, self.root, self.probe = top, root, probe
        self.time_parallel = time_parallel          # "auto" | None | engine.TpPlan
        self.warm_start = bool(warm_start)          # generic trees with a diode root: SsWarmStart when a batch is re-visited
        self.force_generic = bool(force_generic)    # tests: run the clipper tree through the generic kernel
        self.elements = _walk(top)
        if probe not in self.elements:
            raise ValueError("probe must be an element of the tree under `top`")
        self.caps = [e for e in self.elements if _kind(e) == "Capacitor"]
        self.sources = [e for e in self.elements if _kind(e) == "ResistiveVoltageSource"]
        self.root_kind = _kind(root)
        if self.root_kind not in ("IdealVoltageSource", "DiodePair", "DenseRootModel"):
            raise ValueError(f"unsupported root {self.root_kind}")
        self.ns = len(self.caps)
        self.ni = len(self.sources) + (1 if self.root_kind == "IdealVoltageSource" else 0)
        self.per_sample_R = per_sample_R
        if per_sample_R is not None and (per_sample_R not in self.elements or _kind(per_sample_R) not in ("ResistiveVoltageSource", "Resistor")):
            raise ValueError("per_sample_R must be a ResistiveVoltageSource or a Resistor of the tree (set_resistance: "
                             "tf_wdf.py:51-52,80-81)")
        if self.ni < 1:
            raise ValueError("the circuit has no voltage source")
        # Outside the clipper topology a per-sample impedance or a DenseRootModel root runs on the streamed-coefficient kernels
        # (csrc/wdf_ss_dyn.h) -- and so does, since round 6, ANY tree of five to eight capacitors (the static-coefficient kernels
        # of csrc/wdf_statespace.h are compiled for at most four: a larger tree hands the streamed kernels one static row)
        self._dyn = ((per_sample_R is not None or self.root_kind == "DenseRootModel") and (self.force_generic or not self._is_clipper())) \
            or self.ns > 4
        if self._dyn and (self.ns > 8 or self.ni > 2):
            raise binding.WdfHipError("trees of at most eight capacitors and two sources run on the GPU kernels "
                                      f"(this one has {self.ns} and {self.ni})")

    # -- device-resident component values
    def to_device(self, device="cuda"):
        """Move the circuit's component Variables {Is, nVt, R, C} into one device-resident parameter block
        (compat_tf.ParamBlock): from here on mse() reads the values where they live, tape.gradient returns device
        scalars and tf.keras.optimizers.Adam.apply_gradients is one kernel launch -- the training loops of
        lpf.py:86-99 / clipper_pot.py:245-269 run without a host round trip per step (print / .numpy() / float() on a
        Variable still work: they copy back on demand).  Diode-clipper topology only: the generic lowering differentiates
        its float64 probe on the host and needs the Variables there.  Returns self."""
        binding.require_gpu()
        if any(getattr(self, a, None) is not None for a in ("_lin", "_pblock", "_tree", "_mlp", "_dynres")):
            return self
        if self._dyn:
            # (round 6) the streamed-coefficient path has no one-pass step, but nothing in it needs the component values on the
            # host either: they move into a device block (the tape interpreter, the kernels and the optimizers read them there),
            # a network root's weights into one flat device vector -- a training loop then makes no host round trip per step
            self._dynres = _DynResident(self, device)
            return self
        if self.root_kind == "DenseRootModel":
            if not self._is_clipper():
                raise binding.WdfHipError("the MLP root is supported on the diode-clipper topology (clipper_pot.py:94-101)")
            from . import mlp_root
            self._mlp = mlp_root.MlpResident(self, device)       # clipper_pot.py's model: weights in one device vector
            return self
        if self.root_kind == "IdealVoltageSource" and self.ns <= 2 and self.ni <= 2 and self.per_sample_R is None:
            # a linear tree (lpf.py, voltage_divider.py): the probed step becomes a device tape, mse() the one-pass step
            self._lin = _LinResident(self, device)
            return self
        if self.root_kind in ("IdealVoltageSource", "DiodePair") and self.per_sample_R is None and \
                not (self._is_clipper() and self.root_kind == "DiodePair"):
            # any other tree the state-space kernels run (HPFDiodeClipper.h:28-32 ...): the probe moves to the device, the
            # forward / reverse-sweep kernels stay -- __call__ no longer touches the host per step
            self._tree = _LinResident(self, device)
            return self
        if not (self._is_clipper() and self.root_kind == "DiodePair"):
            raise binding.WdfHipError("Circuit.to_device: the diode-pair clipper and linear trees (ideal-source root, at most two "
                                      "capacitors and two sources) keep their component values on the device")
        dp, vs, cap = self.root, self.top.P1, self.top.P2
        Rv = 1.0 if self.per_sample_R is not None else vs.R
        parts = [dp.Is, dp.nVt, Rv, cap.C]
        for p in parts:
            if isinstance(p, torch.Tensor) and not getattr(p, "_is_tf_variable", False) and (p.requires_grad or p.grad_fn is not None):
                raise binding.WdfHipError("Circuit.to_device: a component value is a tensor computed from other Variables; the "
                                          "resident step would freeze it -- keep this circuit on the host path")
            if isinstance(p, torch.Tensor) and getattr(p, "_wdf_block", None) is not None:
                raise binding.WdfHipError("Circuit.to_device: a component Variable already lives in another circuit's block")
        pb = tf.ParamBlock([float(p) for p in parts], torch.device(device))
        self._adopted_parts = {}
        for i, p in enumerate(parts):
            if isinstance(p, torch.Tensor) and getattr(p, "_is_tf_variable", False) and p.numel() == 1:
                pb.adopt(i, p)
                self._adopted_parts[i] = p
        self._pblock, self._res_cache = pb, EntryCache()
        return self

    def _theta(self, parts, dev):
        """{Is, nVt, R, C} as one float32[4] on the device, differentiable w.r.t. the Variables among them."""
        pb = getattr(self, "_pblock", None)
        if pb is not None:      # resident: the adopted Variables already live on the device, the rest are the block's constants
            pb.flush()
            return torch.stack([(pb.members[i] if i in pb.members else pb.block[i]).as_subclass(torch.Tensor).reshape(())
                                for i in range(4)]).to(dev)
        return torch.stack([p.as_subclass(torch.Tensor).float().reshape(()) for p in parts]).to(dev)

    def _host_RC(self, parts):
        """R and C as host numbers for the planner: the Variables themselves when they live on the host, the block's
        lagging mirror when resident (no synchronisation either way)."""
        pb = getattr(self, "_pblock", None)
        if pb is not None:
            h = pb.host_values()
            return h[2], h[3]
        return float(parts[2]), float(parts[3])

    def _loss_resident(self, x, target, kind="mse", skip=0):
        """mse() / mse_esr() on a resident circuit: inputs prepared once per (x, target) pair -- a training set and a
        validation set alternate in clipper_pot.py:245-262, each keeps its own stepper and warm-start state -- then one pass
        of the one-pass training step per call."""
        from . import engine
        pb = self._pblock
        pb.flush()                                               # (queued optimizer updates of the block go out first)
        dp_, vs_, cap_ = self.root, self.top.P1, self.top.P2
        now = [dp_.Is, dp_.nVt, (1.0 if self.per_sample_R is not None else vs_.R), cap_.C]
        for i, v in getattr(self, "_adopted_parts", {}).items():
            if now[i] is not v or getattr(v, "_wdf_block", (None,))[0] is not pb:
                raise binding.WdfHipError("Circuit.to_device: a component Variable was replaced (set_resistance?) or moved to another "
                                          "block after to_device(); build a new Circuit")
        with torch._C.DisableTorchFunctionSubclass():
            key = (tensor_key(x), tensor_key(target), kind, int(skip))
        ent = self._res_cache.get(key)
        if ent is None:
            dp, cap = self.root, self.top.P2
            xd = x.as_subclass(torch.Tensor).to(pb.block.device).float()
            xv, r = engine.split_channels(xd, self.per_sample_R is not None, time_major=True, anchor=x)
            tgt = target.as_subclass(torch.Tensor).to(pb.block.device).float().reshape(xv.shape).contiguous()
            host = pb.host_values()
            R_plan = host[2] if r is None else engine.resistance_max(r)
            tp = self.time_parallel
            if tp == "auto":
                tp = engine.tuned_plan(pb.block, xv, r, float(cap.FS), R_plan, host[3], n_up=dp.N_up, n_down=dp.N_down,
                                       time_major=True, R_min=None if r is None else engine.resistance_min(r), fused=True)
            T, B = xv.shape
            st = engine.MseStep(B, T, float(cap.FS), tp, pb.block.device, n_up=dp.N_up, n_down=dp.N_down, time_major=True,
                                warm=True, loss=kind, skip=int(skip))
            live = [(i, v) for i, v in sorted(pb.members.items()) if v.requires_grad]
            ent = self._res_cache.put(key, (st, xv, r, tgt, 1.0 / float(B * T) if kind == "mse" else 1.0,
                                            [i for i, _ in live], [v for _, v in live],
                                            x, target,                    # (x, target held: their storage stays theirs)
                                            {id(v): i for i, v in live}),
                                      3 * _nbytes(xv, r, tgt))
        st, xv, r, tgt, inv_n, idx, live = ent[:7]
        engine.LAST_TP_STATUS["status"] = st.status
        loss, out = engine._ResidentMseFn.apply(st, pb.block, xv, r, tgt, inv_n, idx, *live)
        loss = loss.as_subclass(tf.Tensor)
        # d loss / d Variable is already known: tape.gradient(loss, ...) on THIS tensor reads it (compat_tf.GradientTape)
        loss._wdf_fused = (out, ent[9])
        return loss

    def mse_esr(self, x, target, skip=0, z0=None, carry_state=False):
        """The training loss of clipper_pot.py:146-156,177 on this circuit's output past `skip` samples (:232,248):
        mean((y - t)^2) + sqrt(sum((y - t)^2) / (sum(y^2) + eps) / n)  -- the scripts call esr_loss(outs, train_Y) on a
        function declared (target, predicted), so the normalising energy is the OUTPUT's.  target: [T,B] like the output.
        A resident diode-pair clipper (to_device()) evaluates loss and gradient in one pass over the data
        (wdf_clipper_step_esr_tp); anything else composes it from the forward.
        z0 / carry_state: as in mse() -- the state the call starts from, or the one the previous carry_state call ended in
        (`circ.last_state`); composed from __call__(x, z0, return_state) (the resident one-pass steps start from zero state,
        which is what clipper_pot.py:110-111 does before every forward)."""
        binding.require_gpu()
        stateful = (z0 is not None or carry_state) and self.ns > 0
        if stateful:
            if carry_state and z0 is None:
                z0 = getattr(self, "last_state", None)
            y, zT = self(x, z0=z0, return_state=True)
            self.last_state, self.last_output = zT.detach(), y.detach()
            o = y[int(skip):]
            t = tf.convert(target, device=o.device).reshape(y.shape)[int(skip):]
            S, E = tf.reduce_sum(tf.square(o - t)), tf.reduce_sum(tf.square(o)) + float(np.finfo(float).eps)
            n = float(o.numel())
            return S / n + tf.sqrt(S / E / n)
        mres = getattr(self, "_mlp", None)
        if mres is not None and isinstance(x, torch.Tensor) and isinstance(target, torch.Tensor):
            from . import mlp_root
            if int(x.shape[1]) % 16 == 0:                        # (the resident step's blocks are 16 steps)
                ent = mres.entry(x, target, skip)
                live = [v for v in mres.vars if v.requires_grad]
                loss, out = mlp_root.MlpResidentFn.apply(mres, ent, *live)
                loss = loss.as_subclass(tf.Tensor)
                loss._wdf_fused = (out, {id(v): mres.spec[id(v)] for v in live})
                self.last_output = ent["st"].y
                return loss
        if (getattr(self, "_pblock", None) is not None and isinstance(x, torch.Tensor) and isinstance(target, torch.Tensor)
                and not self.force_generic):
            return self._loss_resident(x, target, "mse+esr", skip)
        y = self(x)
        o = y[int(skip):]
        t = tf.convert(target, device=o.device).reshape(y.shape)[int(skip):]
        S, E = tf.reduce_sum(tf.square(o - t)), tf.reduce_sum(tf.square(o)) + float(np.finfo(float).eps)
        n = float(o.numel())
        return S / n + tf.sqrt(S / E / n)

    # -- topology tests
    def mse(self, x, target, z0=None, carry_state=False):
        """tf.reduce_mean(tf.square(self(x) - target)) as ONE fused evaluation where the kernels allow
        it (diode-pair clipper: forward kernel + MSE-fused reverse sweep, the gradient of every
        trainable component ready when tape.gradient asks); any other circuit takes the plain path.
        target: [T,B] like the output.

        z0 [ns,B]: the capacitor states the call starts from (default: zero, clipper_pot.py:110-111).  carry_state=True: the
        call starts from the states the PREVIOUS carry_state call on this circuit ended in (zero the first time) -- lpf.py:30-49
        never resets C1, so its epoch n starts where epoch n - 1 ended; the state handed over is a constant of the new call
        (the reference's stored tensor belongs to the previous tape).  `circ.last_state` [ns,B] is what the call ended in
        (set whenever z0 / carry_state is used); `circ.reset_state()` forgets it.  A resident linear tree runs this inside the
        one-pass step (wdf_ss_lin_step_mse's z0 / zT); every other circuit composes it from __call__(x, z0, return_state)."""
        binding.require_gpu()
        stateful = z0 is not None or carry_state
        if carry_state and z0 is None:
            z0 = getattr(self, "last_state", None)
        if stateful and self.ns == 0:
            stateful, z0 = False, None                            # (no capacitor: nothing to carry)
        lin = getattr(self, "_lin", None)
        if lin is None and getattr(self, "_tree", None) is not None and self.root_kind == "DiodePair" and 1 <= self.ns <= 2 \
                and 1 <= self.ni <= 2 and not self.force_generic:
            lin = self._tree                                     # diode-pair root: the tangent-carried step (csrc/wdf_ss_nl_step.h)
        if stateful and lin is not None and lin is not getattr(self, "_lin", None):
            lin = None                                           # (diode-root one-pass step: zero state only -> the plain path)
        if lin is not None and isinstance(x, torch.Tensor) and isinstance(target, torch.Tensor):
            with torch._C.DisableTorchFunctionSubclass():        # (asking a tensor for its version or requires_grad must not
                lin.check()                                      #  send the queued optimizer updates: the probe carries them)
                ent = lin.entry(x, target)
                live = [(i, v) for i, v in sorted(lin.pb.members.items()) if v.requires_grad]
                z0d = None
                if z0 is not None:                               # the linear step takes the state in and hands it out
                    z0d = torch.as_tensor(z0).as_subclass(torch.Tensor).detach().to(ent["y"].device).float().reshape(self.ns, ent["B"]).contiguous()
            loss, out = _LinResidentMseFn.apply(lin, ent, 1.0 / float(ent["B"] * ent["T"]), [i for i, _ in live], z0d, stateful,
                                                *[v for _, v in live])
            loss = loss.as_subclass(tf.Tensor)
            loss._wdf_fused = (out, {id(v): i for i, v in live})
            self.last_output = ent["y"]                          # the forward's y [T,B] of this call (TensorArray.stack() layout)
            if stateful:
                self.last_state = ent["zT"]
            return loss
        if stateful:
            y, zT = self(x, z0=z0, return_state=True)
            self.last_state, self.last_output = zT.detach(), y.detach()
            return tf.reduce_mean(tf.square(y - tf.convert(target, device=y.device).reshape(y.shape)))
        if not (self._is_clipper() and self.root_kind == "DiodePair" and not self.force_generic):
            y = self(x)
            return tf.reduce_mean(tf.square(y - target))
        from . import engine
        if getattr(self, "_pblock", None) is not None and isinstance(x, torch.Tensor) and isinstance(target, torch.Tensor):
            return self._loss_resident(x, target)
        anchor = x if isinstance(x, torch.Tensor) else None
        x = torch.as_tensor(x).as_subclass(torch.Tensor)
        x = (x if x.is_cuda else x.cuda()).float()
        dp, vs, cap = self.root, self.top.P1, self.top.P2
        # a streamed pot resistance overrides the source's own R (which a recorded loop may have left symbolic)
        Rv = torch.tensor(1.0) if self.per_sample_R is not None else (vs.R if isinstance(vs.R, torch.Tensor) else torch.tensor(float(vs.R)))
        parts = [dp.Is, dp.nVt, Rv, cap.C]
        theta = self._theta(parts, x.device)
        # the sweep reads its inputs time-major: the transposed copy is made once per input tensor
        xv, r = engine.split_channels(x, self.per_sample_R is not None, time_major=True, anchor=anchor)
        tgt = torch.as_tensor(target).as_subclass(torch.Tensor).to(x.device).float().reshape(xv.shape).contiguous()
        R_host, C_host = self._host_RC(parts)
        R_plan = R_host if r is None else engine.resistance_max(r)
        tp = self.time_parallel
        if tp == "auto":
            tp = engine.tuned_plan(theta, xv, r, float(cap.FS), R_plan, C_host, n_up=dp.N_up, n_down=dp.N_down,
                                   time_major=True, R_min=None if r is None else engine.resistance_min(r), fused=True)
        loss = engine.clipper_mse(theta, xv, tgt, float(cap.FS), r=r, n_up=dp.N_up, n_down=dp.N_down, tp=tp,
                                  time_major=True)
        return loss.as_subclass(tf.Tensor)

    def reset_state(self):
        """Forget the state carried by mse(..., carry_state=True): the next such call starts from zero (Capacitor.reset,
        tf_wdf.py:117-118)."""
        self.last_state = None

    def _is_clipper(self):
        t = self.top
        return (_kind(t) == "Parallel" and _kind(t.P1) == "ResistiveVoltageSource" and _kind(t.P2) == "Capacitor"
                and self.probe is t.P2)

    # -- one probed step -> state-space matrices (float64 CPU tensors with autograd graph)
    def matrices(self):
        ns, ni = self.ns, self.ni
        K = ns + ni + 1
        eye = torch.eye(K, dtype=torch.float64)
        from . import trace
        saved = _Saved(self.elements + [self.root])
        rec, trace._current = trace._current, None      # probing is never part of a recorded loop
        plain = torch._C.DisableTorchFunctionSubclass()   # ~35 scalar / [K] torch ops: plain dispatch halves their host time
        plain.__enter__()
        try:
            for s, cap in enumerate(self.caps):
                cap.z = eye[s]
            for i, src in enumerate(self.sources):
                src.Vs = eye[ns + i]
            self.top.calc_impedance()
            up = self.top.reflected() + torch.zeros(K, dtype=torch.float64)        # broadcast scalars
            self.top.incident(eye[K - 1])
            znew = [cap.z + torch.zeros(K, dtype=torch.float64) for cap in self.caps]
            yv = (self.probe.a + self.probe.b) * 0.5 + torch.zeros(K, dtype=torch.float64)
            r_port = self.top.R
        finally:
            plain.__exit__(None, None, None)
            saved.restore()
            trace._current = rec
        up, yv = up.as_subclass(torch.Tensor), yv.as_subclass(torch.Tensor)
        Z = torch.stack([z.as_subclass(torch.Tensor) for z in znew]) if ns else torch.zeros(0, K, dtype=torch.float64)
        A, Bx, E = Z[:, :ns], Z[:, ns:ns + ni], Z[:, K - 1]
        ca, da = up[:ns], up[ns:ns + ni]
        cy, dy, fy = yv[:ns], yv[ns:ns + ni], yv[K - 1]
        if self.root_kind == "IdealVoltageSource":
            # b = -a + 2 Vs (tf_wdf.py:26-28) is linear: fold it in.  Vs is the LAST channel.
            er = torch.zeros(ni, dtype=torch.float64)
            er[ni - 1] = 1.0
            A = A - torch.outer(E, ca)
            Bx = Bx - torch.outer(E, da) + 2.0 * torch.outer(E, er)
            cy = cy - fy * ca
            dy = dy - fy * da + 2.0 * fy * er
            E, ca, da, fy = torch.zeros_like(E), torch.zeros_like(ca), torch.zeros_like(da), torch.zeros_like(fy)
        coef = torch.cat([A.reshape(-1), Bx.reshape(-1), E, ca, da, cy, dy, fy.reshape(1)])
        return coef, r_port.as_subclass(torch.Tensor) if isinstance(r_port, torch.Tensor) else torch.tensor(float(r_port))

    # -- run
    def __call__(self, x, z0=None, return_state=False):
        """x: [B,T] or [B,T,n_in] float32 CUDA tensor (batch-major, the reference's
        input[:, i, c] layout).  Returns y [T,B] (and the final capacitor states [ns,B])."""
        binding.require_gpu()
        self._anchor = x if isinstance(x, torch.Tensor) else None      # the caller's object: cache key for its copies
        if not isinstance(x, torch.Tensor):
            x = torch.as_tensor(x)
        x = x.as_subclass(torch.Tensor)
        if not x.is_cuda:
            x = x.cuda()
        x = x.float()
        if x.dim() == 2:
            x = x.unsqueeze(-1)
        nchan = self.ni + (1 if self.per_sample_R is not None else 0)
        if x.dim() != 3 or x.shape[2] != nchan:
            raise binding.WdfHipError(f"x must be [B,T,{nchan}] (or [B,T] for one channel), got {tuple(x.shape)}")
        dev = x.device

        if self._dyn:
            return self._run_dyn(x, z0, return_state)
        if self._is_clipper() and self.root_kind == "DiodePair" and not self.force_generic:
            return self._run_clipper(x, z0, return_state)
        if self.root_kind == "DenseRootModel":
            from . import mlp_root
            return mlp_root.run_clipper_mlp(self, x, z0, return_state)

        res = getattr(self, "_tree", None) or getattr(self, "_lin", None)
        if res is not None:
            # resident component values: coefficients (and their chain rule) from the device probe, the plan from the
            # block's lagging host mirror
            res.check()
            live = [(i, v) for i, v in sorted(res.pb.members.items()) if v.requires_grad]
            coef, rootp_r = _ProbeFn.apply(res, [i for i, _ in live], *[v for _, v in live])
            c64, _ = res.host_coef()
            coef64 = torch.as_tensor(c64)
            if self.root_kind == "DiodePair":
                rootp, kind, n_up, n_down = rootp_r, binding.ROOT_DIODE_PAIR, self.root.N_up, self.root.N_down
            else:
                rootp, kind, n_up, n_down = None, binding.ROOT_NONE, 1, 1
        else:
            coef64, r_port = self.matrices()
            coef = coef64.to(device=dev, dtype=torch.float32)
            if self.root_kind == "DiodePair":
                dp = self.root
                rootp = torch.stack([dp.Is.as_subclass(torch.Tensor).double().reshape(()),
                                     dp.nVt.as_subclass(torch.Tensor).double().reshape(()),
                                     r_port.double().reshape(())]).to(device=dev, dtype=torch.float32)
                kind, n_up, n_down = binding.ROOT_DIODE_PAIR, dp.N_up, dp.N_down
            else:
                rootp, kind, n_up, n_down = None, binding.ROOT_NONE, 1, 1
        z0t = None if z0 is None else z0.as_subclass(torch.Tensor).to(dev).float().reshape(self.ns, -1).contiguous()
        tp = getattr(self, "time_parallel", None)
        if tp == "auto" and res is not None:                      # (the plan follows the host mirror: cached with it)
            pkey = (kind, int(x.shape[0]), int(x.shape[1]))
            tp = res.plans.get(pkey, "miss")
            if tp == "miss":
                tp = res.plans[pkey] = plan_ss_time_parallel(coef64, self.ns, self.ni, kind, x.shape[0], x.shape[1])
        elif tp == "auto":
            tp = plan_ss_time_parallel(coef64, self.ns, self.ni, kind, x.shape[0], x.shape[1])
        elif not isinstance(tp, SsTpPlan):
            tp = None
        warm = None
        if tp is not None and tp.k_fwd >= 2 and kind == binding.ROOT_DIODE_PAIR and self._anchor is not None and self.warm_start:
            # the caller's tensor object and version name the batch: the same one again -> its chunks start warm
            ws = self.__dict__.setdefault("_ss_warm", {})
            wkey = (id(self._anchor), self._anchor._version, tuple(x.shape))
            hit = ws.get(wkey)
            if hit is None or hit[0]() is not self._anchor:
                for k_ in [k_ for k_, v_ in ws.items() if v_[0]() is None]:
                    del ws[k_]
                if len(ws) >= 4:
                    ws.pop(next(iter(ws)))                         # oldest out
                hit = ws[wkey] = (weakref.ref(self._anchor), SsWarmStart(x.shape[1], x.shape[0], self.ns, tp))
            warm = hit[1]
            warm.plan = tp                  # (the cold plan follows the components as they train; the warm state stays)
        y, zT = _StateSpaceFn.apply(coef, rootp, x.contiguous(), z0t, self.ns, self.ni, kind, n_up, n_down,
                                    bool(return_state), tp, warm)
        y = y.as_subclass(tf.Tensor)
        return (y, zT) if return_state else y

    def _run_dyn(self, x, z0, return_state):
        """Per-sample impedance on any small tree and / or the MLP root on any small tree (csrc/wdf_ss_dyn.h).
        The probed step is recorded once as a scalar tape over the component values (probe_tape.record: the elements' own
        calc_impedance / reflected / incident code); with a resistance channel the tape is evaluated over the whole channel
        in torch -- one coefficient row per (sample, sequence), with the autograd graph back to the static components -- which
        is set_resistance + calc_impedance every step (clipper_pot.py:116-117) hoisted out of the time loop."""
        from . import probe_tape
        dev = x.device
        B, T = int(x.shape[0]), int(x.shape[1])
        own = {"Resistor": "R", "ResistiveVoltageSource": "R", "Capacitor": "C"}
        res = getattr(self, "_dynres", None)
        if res is not None:
            res.pb.flush()                                   # (queued optimizer updates of the block go out before anything reads it)
        if getattr(self, "_dyn_tape", None) is None:
            params = [(e, own[_kind(e)]) for e in self.elements if _kind(e) in own]
            pvars = [e.__dict__[n] for e, n in params]
            tape, outs, rport = probe_tape.record(self, pvars, device_limits=False)
            self._dyn_tape = (tape, outs + [rport], params)
        tape, outs, params = self._dyn_tape
        chan = next((i for i, (e, _) in enumerate(params) if e is self.per_sample_R), -1)
        rtape = self.__dict__.get("_dyn_rtape")
        if rtape is None:
            rtape = self._dyn_rtape = binding.RowsTape(*tape.packed(), outs)
        on_device = rtape.fits(len(params)) and len(rtape.ops) <= DYN_ROWS_MAX_OPS and not x.requires_grad
        # A pot that keeps its value along every sequence (the reference's recordings: dataimport.py:96 repeats the file's
        # resistance down the whole channel, batch_data cuts sequences out of it): calc_impedance gives ONE row per sequence --
        # the tape runs over B values instead of B x T, the kernels read rows [1,n,B] and the sweep sums dL/d(row) over the
        # steps itself.  Looked at once per input tensor (one comparison pass, cached on the storage).
        per_seq = False
        if chan >= 0 and not x.requires_grad and T > 1 and getattr(self, "per_sequence_rows", True):
            with torch._C.DisableTorchFunctionSubclass():
                ckey = tensor_key(self._anchor) if isinstance(getattr(self, "_anchor", None), torch.Tensor) else None
            cache = self.__dict__.setdefault("_dyn_chan_const", {})
            if ckey is not None and ckey in cache:
                per_seq = cache[ckey]
            else:
                with torch.no_grad():
                    rch = x[:, :, self.ni]
                    per_seq = bool((rch == rch[:, :1]).all())
                if ckey is not None:
                    if len(cache) > 16:
                        cache.clear()
                    cache[ckey] = per_seq
        vals = []
        for i, (e, n) in enumerate(params):
            if i == chan:
                # the resistance channel: [T,B] for the device tape, [B,T] float64 for torch's
                if per_seq:
                    vals.append(x[:, :1, self.ni].t().contiguous() if on_device else x[:, :1, self.ni].double())
                    continue
                vals.append(x[:, :, self.ni].t().contiguous() if on_device else x[:, :, self.ni].double())
            else:
                v = e.__dict__[n]
                v = v.as_subclass(torch.Tensor) if isinstance(v, torch.Tensor) else torch.tensor(float(v))
                vals.append(v.reshape(()).to(dev) if on_device else v.double().reshape(()).to(dev))
        with torch._C.DisableTorchFunctionSubclass():
            if on_device:
                # calc_impedance of every sample in one launch (csrc/wdf_ss_dyn_rows.h); the channel's slot gets a placeholder
                r = vals[chan].float() if chan >= 0 else None
                rows = _DynRowsFn.apply(rtape, chan, r, *[v if i != chan else v.new_zeros(()) for i, v in enumerate(vals)])
            else:
                # (a tape past the device evaluator's sizes, or a channel that wants its own gradient: torch runs the tape)
                nodes = tape.evaluate_torch(vals, outs)
                if self.per_sample_R is not None:
                    Tr = 1 if per_seq else T
                    rows = torch.stack([torch.broadcast_to(v, (B, Tr)) for v in nodes], dim=0)      # [n,B,T]
                    rows = rows.permute(2, 0, 1).float().contiguous()                                # [T,n,B] ([1,n,B]: per sequence)
                else:
                    rows = torch.stack([v.reshape(()) for v in nodes]).float()                       # one static row [n]
        hidden = n_tanh = 0
        n_up = n_down = 1
        if self.root_kind == "DiodePair":
            dp = self.root
            rootvec = torch.stack([dp.Is.as_subclass(torch.Tensor).double().reshape(()),
                                   dp.nVt.as_subclass(torch.Tensor).double().reshape(())]).to(device=dev, dtype=torch.float32)
            kind, n_up, n_down = binding.ROOT_DIODE_PAIR, dp.N_up, dp.N_down
        elif self.root_kind == "DenseRootModel":
            from . import mlp_root
            dense, hidden, n_tanh, act = mlp_root.describe(self.root, with_activation=True)
            if act != "tanh":
                raise binding.WdfHipError("the MLP root outside the clipper topology: tanh networks (the ReLU kernels are the "
                                          "clipper's resident step)")
            rootvec = mlp_root.flat_weights(dense).float().to(dev)
            kind = binding.ROOT_MLP
        else:
            rootvec, kind = None, binding.ROOT_NONE
        z0t = None if z0 is None else z0.as_subclass(torch.Tensor).to(dev).float().reshape(self.ns, -1).contiguous()
        xs = x[:, :, :self.ni].contiguous()
        tp = self._plan_dyn(tape, outs, vals, kind, B, T, xs=xs) if self.time_parallel == "auto" else \
            (self.time_parallel if isinstance(self.time_parallel, SsTpPlan) else None)
        warm = None
        if tp is not None and self.warm_start and self.ns >= 1 and z0t is None and isinstance(getattr(self, "_anchor", None), torch.Tensor):
            # the same batch again (a training loop: lpf.py:86-99 re-runs data_in every epoch) -> its chunks start warm
            wsd = self.__dict__.setdefault("_dyn_warm", EntryCache(max_entries=4))
            with torch._C.DisableTorchFunctionSubclass():
                wkey = (tensor_key(self._anchor), kind)
            hit = wsd.get(wkey)
            if hit is None:
                hit = wsd.put(wkey, (self._anchor, DynWarmStart(T, B, self.ns, tp.k_bwd, tp.warmup if tp.warmup > 0 else 64, tp.tol)))
            warm = hit[1]
        y, zT = _SsDynFn.apply(rows, rootvec, xs, z0t, self.ns, self.ni, kind, hidden, n_tanh, n_up, n_down, bool(return_state), tp, warm)
        y = y.as_subclass(tf.Tensor)
        return (y, zT[:self.ns]) if return_state else y

    def _plan_dyn(self, tape, outs, vals, kind, B, T, tol=1.0e-6, xs=None):
        """SsTpPlan for the streamed-coefficient kernels (or None: the batch fills the chip / no state).  The reverse sweep is
        exact in chunks: as many as give every SIMD ~2 waves, none shorter than 64 steps.  The forward warms a chunk up from
        z = 0: W outlasts the slowest mode of the step's Jacobian A + Da E ca^T over the root's slope Da -- [-1, 1] for a diode
        pair (passive), for a network root what its weights give over the batch's amplitude (mlp_root.slope_range: a learned
        b(a) is not bounded by 1) -- taken at BOTH ends of the resistance channel (the tape evaluated at its smallest and largest
        value -- the adaptor coefficients are monotone in one resistance); every boundary is verified on the device whatever the
        estimate.  Re-derived every 32 calls (the components train slowly; a stale W costs a re-run of the waves that missed,
        never a wrong result)."""
        ns, ni = self.ns, self.ni
        if ns < 1:
            return None
        waves = max(1, -(-B // 64))
        k_max = min(T // 64, (2 * N_SIMD) // waves)
        if k_max < 2:
            return None
        cache = self.__dict__.setdefault("_dyn_plans", {})
        key = (B, T, kind)
        hit = cache.get(key)
        if hit is not None and hit[1] < 32:
            cache[key] = (hit[0], hit[1] + 1)
            return hit[0]
        with torch.no_grad(), torch._C.DisableTorchFunctionSubclass():
            ends = [[(v if v.dim() == 0 else sel(v)).double() for v in vals] for sel in (torch.amin, torch.amax)] \
                if self.per_sample_R is not None else [[v.double() for v in vals]]
            rho = 0.0
            for pv in ends:
                c = torch.stack([v.reshape(()) for v in tape.evaluate_torch(pv, outs)]).double().cpu().numpy()
                A = c[:ns * ns].reshape(ns, ns)
                oE = ns * ns + ns * ni
                E, ca = c[oE:oE + ns], c[oE + ns:oE + 2 * ns]
                if kind == binding.ROOT_NONE:
                    slopes = (0.0,)
                elif kind == binding.ROOT_MLP:
                    from . import mlp_root
                    # the a-range the network is asked about: four times the batch's largest input (the incident wave is
                    # ca.z + da.x, both of the input's order); one read-back per re-derivation
                    a_max = 4.0 * max(float(xs.abs().max()) if xs is not None else 1.0, 0.25)
                    s_lo, s_hi = mlp_root.slope_range(mlp_root.describe(self.root, with_activation=True)[0],
                                                      [math.log(max(float(c[-1]), 1e-30))], a_max)
                    slopes = tuple(np.linspace(s_lo, s_hi, 5))          # (the radius need not peak at an end of the range)
                else:
                    slopes = (1.0, -1.0)
                rho = max(rho, max(float(np.max(np.abs(np.linalg.eigvals(A + sgn * np.outer(E, ca))))) for sgn in slopes))
        k_fwd, W = 1, 0
        if rho < 1.0 - 1e-9:
            W = 8 if rho <= 0.0 else max(8, -(-int(math.ceil(math.log(0.01 * tol) / math.log(rho))) // 8) * 8)
            k_fwd = max(1, min(k_max, T // max(W, 64)))
        plan = SsTpPlan(k_fwd, W, float(tol), k_max)
        cache[key] = (plan, 0)
        return plan

    def _run_clipper(self, x, z0, return_state):
        from . import engine
        dp, vs, cap = self.root, self.top.P1, self.top.P2
        dev = x.device
        # a streamed pot resistance overrides the source's own R (which a recorded loop may have left symbolic)
        Rv = torch.tensor(1.0) if self.per_sample_R is not None else (vs.R if isinstance(vs.R, torch.Tensor) else torch.tensor(float(vs.R)))
        parts = [dp.Is, dp.nVt, Rv, cap.C]
        theta = self._theta(parts, dev)
        xv, r = engine.split_channels(x, self.per_sample_R is not None, anchor=getattr(self, "_anchor", None))
        if z0 is not None or return_state:
            z0t = None if z0 is None else z0.as_subclass(torch.Tensor).to(dev).float().reshape(-1).contiguous()
            y, zT = engine.clipper_stateful(theta, xv, float(cap.FS), r=r, n_up=dp.N_up, n_down=dp.N_down, z0=z0t)
            y = y.as_subclass(tf.Tensor)
            return (y, zT.reshape(1, -1)) if return_state else y
        tp = self.time_parallel
        if tp == "auto":
            # component values live on the host (tiny CPU variables): planning costs no sync
            # per-sample R: the warm-up must outlast the slowest (largest-R) sequence in the batch
            R_host, C_host = self._host_RC(parts)
            R_plan = R_host if r is None else engine.resistance_max(r)
            tp = engine.tuned_plan(theta, xv, r, float(cap.FS), R_plan, C_host, n_up=dp.N_up, n_down=dp.N_down,
                                   R_min=None if r is None else engine.resistance_min(r))
        y = engine.clipper(theta, xv, float(cap.FS), r=r, n_up=dp.N_up, n_down=dp.N_down, tp=tp)
        return y.as_subclass(tf.Tensor)

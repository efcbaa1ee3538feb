"""Host side of warm-started chunks for the time-parallel forwards that keep their controller on the host (the MLP-root
clipper, mlp_root.py; generic state-space trees with a diode root, lowering.py).  The diode-pair clipper's own kernels
steer themselves on the device (csrc/wdf_clipper.h, TpCtl) -- this is the same policy for callers that go through torch.

A training loop re-visits its batch with coefficients one optimizer step apart (lpf.py:86-99, clipper_pot.py:245-269), so a
chunk can start from the state an earlier call had at the sample its warm-up begins; a fraction of the cold warm-up then
closes the gap.  How small a fraction is found by trying: the device verifies every chunk boundary anyway and repairs
what missed, the count of repaired waves is summed ON THE DEVICE over all calls and copied to pinned memory behind a
forward now and then (one copy in flight, never waited for -- the host runs a dozen calls ahead of the device, so the
policy counts CALLS, not verdicts):
  * repairs since the last verdict  -> two units more, and the warm-up that failed is not tried again for 256 calls;
  * clean for `wait_calls` calls    -> one unit less (two, and after 4 calls, while the sampled miss is `bold_below` x
                                       inside the tolerance).
The caller keeps start states for the warm-ups candidates() names, so a change never costs a cold call.
"""
import collections

import torch


class WarmUpController:
    def __init__(self, cold, start, unit=16, floor=16, miss_waves=1, wait_calls=16, bold_below=None, tol=None):
        """cold: the planner's cold warm-up (the cap); start: the first warm warm-up tried; miss_waves: repaired waves per
        verdict interval that count as 'too short' (the MLP root's fp32 floor sends a wave or two back now and then
        whatever the warm-up); bold_below (with tol): sampled miss x bold_below < tol -> the bolder shrink."""
        self.cold, self.unit, self.floor = int(cold), int(unit), int(floor)
        self.miss_waves, self.wait_calls, self.bold_below, self.tol = int(miss_waves), int(wait_calls), bold_below, tol
        self.W = max(self.floor, min(-(-int(start) // self.unit) * self.unit, self.cold))
        self.want = None
        self.calls, self.since, self.bad, self.bad_at = 0, 0, 0, -10**9
        self.gated = self.pin = self.pending = None
        self.gated_seen = 0
        self.verdicts = collections.deque(maxlen=256)   # (probing) the last verdicts read: (warm-up, n_bad, max miss, gated waves, sequential waves)

    def _read_verdict(self):
        if self.pending is None or not self.pending[0].query():
            return
        _, issued, w_then = self.pending
        self.pending = None
        buf = self.pin
        miss, total = float(buf[1:2].view(torch.float32)[0]), int(buf[4])
        self.verdicts.append((w_then, int(buf[0]), miss, int(buf[2]), int(buf[3])))
        missed = total - self.gated_seen >= self.miss_waves
        self.gated_seen = total
        if missed:
            if w_then >= self.bad:
                self.bad, self.bad_at = w_then, issued
            if w_then >= self.W:
                self.want, self.since = min(w_then + 2 * self.unit, self.cold), self.calls
            return
        if self.calls - self.bad_at > 256:
            self.bad = 0
        bold = self.bold_below is not None and miss * self.bold_below < self.tol
        lower = max(self.floor, self.W - (2 if bold else 1) * self.unit)
        if w_then == self.W and issued - self.since >= (4 if bold else self.wait_calls) and self.bad < lower < self.W:
            self.want, self.since = lower, self.calls

    def begin(self, available):
        """Start of a call.  available: the warm-ups the caller holds start states for.  -> the warm-up to run warm, or
        None (nothing to start from: run cold)."""
        self.calls += 1
        self._read_verdict()
        if self.want is not None and self.want in available:
            self.W, self.want = self.want, None
        return self.W if self.W in available else None

    def candidates(self):
        """The warm-ups the NEXT call may ask for (the caller keeps this call's states at their chunk starts)."""
        c = (self.want, self.W - 2 * self.unit if self.bold_below is not None else None, self.W - self.unit, self.W,
             self.W + 2 * self.unit)
        return list(dict.fromkeys(w for w in c if w is not None and self.floor <= w <= self.cold))

    def end(self, status, w_used):
        """After a WARM call: add its repaired waves to the device-side sum and, when none is in flight, queue a verdict
        (status: the forward's device int32[4] = {n_bad, max-miss bits, gated waves, sequential waves})."""
        if self.gated is None:
            self.gated = torch.zeros((1,), dtype=torch.int32, device=status.device)
            self.pin = torch.empty((5,), dtype=torch.int32, pin_memory=True)
        self.gated.add_(status[2:3])
        if self.pending is None:
            self.pin[:4].copy_(status, non_blocking=True)
            self.pin[4:].copy_(self.gated, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            self.pending = (ev, self.calls, int(w_used))

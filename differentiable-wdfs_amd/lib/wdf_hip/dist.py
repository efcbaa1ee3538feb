"""Data-parallel sharding of the sequence batch.

Sequences are independent (zero initial state per sequence, clipper_pot.py:110-111) and the
parameters are a handful of replicated scalars, so the batch dimension shards contiguously
across ranks with NO data-path collective.  The only exchange is one all-reduce (RCCL over
xGMI via torch.distributed backend "nccl"; gloo in the CPU tests) of a single fused fp32
buffer per step: [loss partial sums..., parameter-gradient vector].  The buffer is 5 floats
for the diode clipper (609+ for an MLP root): latency-bound, so exactly one collective per
step, issued on the compute stream right behind the reduce kernel.
The reference has no distributed path at all; this is new.
"""
import os

import torch
import torch.distributed as dist


def env_world():
    return int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))


_FORCED = False          # init(force=True): the collective runs even at world size 1


def init(backend=None, force=False):
    """Initialise torch.distributed from the torchrun environment (no-op for world size 1 unless
    force: then a one-rank group is created so that the RCCL all-reduce of the fused gradient buffer
    really executes -- the code path of N > 1 on a one-GPU box).  Returns (world, rank, local_rank)."""
    global _FORCED
    world, rank, local = env_world()
    if force and world == 1:
        os.environ.setdefault("MASTER_PORT", "29531")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        _FORCED = True
    if (world > 1 or force) and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        import datetime
        limit = datetime.timedelta(minutes=5)            # (a rank that never arrives fails the job in minutes, not in half an hour)
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group(backend, device_id=torch.device("cuda", local), timeout=limit)
        else:
            dist.init_process_group(backend, timeout=limit)
    elif torch.cuda.is_available():
        torch.cuda.set_device(local)
    return world, rank, local


def shard_range(n_global, rank, world):
    """Contiguous [b0, b1) of rank's shard; remainders go to the lowest ranks."""
    q, r = divmod(n_global, world)
    b0 = rank * q + min(rank, r)
    return b0, b0 + q + (1 if rank < r else 0)


def allreduce_sum_(buf):
    """In-place SUM all-reduce of one fused buffer (no-op for world size 1)."""
    if dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or _FORCED):
        dist.all_reduce(buf, op=dist.ReduceOp.SUM)
    return buf


def mse_step_allreduce(sse_local, gtheta_local_unnormalised, n_global):
    """Combine per-rank results of an MSE step.
    sse_local: sum of squared errors over the local shard (0-dim tensor);
    gtheta_local_unnormalised: dSSE_local/dtheta.  Returns (loss, grad) of the GLOBAL mean."""
    buf = torch.cat([sse_local.reshape(1), gtheta_local_unnormalised.reshape(-1)])
    allreduce_sum_(buf)
    return buf[0] / n_global, buf[1:] / n_global


def barrier():
    if dist.is_available() and dist.is_initialized():
        dist.barrier()


def esr_coefficients(S, E, n, eps):
    """Host mirror of esr_coef_kernel (csrc/wdf_clipper.h): from the GLOBAL sums S = sum (y-t)^2,
    E = sum y^2 over n samples -> (ga, gb, mse, esr) with dL/dy = ga (y - t) + gb y for
    L = S/n + sqrt(S / (E + eps) / n)   (clipper_pot.py:146-156,177)."""
    E = E + eps
    mse = S / n
    esr = (S / E / n) ** 0.5
    ga = 2.0 / n + (1.0 / (esr * E * n) if esr > 0.0 else 0.0)
    return ga, -esr / E, mse, esr


def esr_sums_allreduce(S_local, E_local):
    """Stage 1 of a sharded MSE + ESR step: the two loss sums made global (one 16-byte all-reduce)."""
    buf = torch.tensor([float(S_local), float(E_local)], dtype=torch.float64)
    allreduce_sum_(buf)
    return float(buf[0]), float(buf[1])


def esr_step_allreduce(S_local, E_local, gP_local, gQ_local, n_global, eps):
    """The one-pass MSE + ESR step's single exchange (wdf_clipper_step_esr_tp, finish=False): every rank contributes
    sums10 = {S, E, gP[4], gQ[4]} of its shard (gP = d(S/2)/dtheta, gQ = d(E/2)/dtheta); after ONE all-reduce of the ten
    numbers every rank forms the global loss and gradient (wdf_esr_finish's formulas).  -> (mse + esr, grad[4])."""
    buf = torch.cat([torch.as_tensor([float(S_local), float(E_local)], dtype=torch.float64),
                     torch.as_tensor(gP_local, dtype=torch.float64).reshape(-1), torch.as_tensor(gQ_local, dtype=torch.float64).reshape(-1)])
    allreduce_sum_(buf)
    ga, gb, mse, esr = esr_coefficients(float(buf[0]), float(buf[1]), n_global, eps)
    return mse + esr, ga * buf[2:6] + gb * buf[6:10]


def esr_two_exchange(forward_sums, backward, allreduce=None):
    """The protocol of a sharded MSE + ESR step whose gradient is a long vector (the MLP-root step, mlp_root.MlpTrainStep: 609
    weights -- the one-exchange form above would double the reverse sweep's matrix work to save one 16-byte message):
        sums = forward_sums()        this rank's {S, E} (tensor [2], float64) after its forward
        all-reduce(sums)             16 bytes
        grad = backward(sums)        the reverse sweep with the GLOBAL sums (they decide dLoss/dy = ga (y - t) + gb y)
        all-reduce(grad)             the weight gradient
    -> (sums, grad), both global.  allreduce: the in-place SUM (default: allreduce_sum_)."""
    ar = allreduce_sum_ if allreduce is None else allreduce
    sums = forward_sums()
    ar(sums)
    grad = backward(sums)
    ar(grad)
    return sums, grad

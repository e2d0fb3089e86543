"""Single-node data parallelism: one process per GPU, torch.distributed over RCCL (backend "nccl"
on ROCm) for the GPUs, gloo for CPU-side tests.

The hot path shards by utterance: inference needs no data-path collective; training all-reduces the
gradients once per step (4.15 M / 4.42 M floats), in per-layer buckets of the packed gradient buffer that
overlap with the backward pass (StackedBLSTMModel._backward).
The loss is a mean over B*T*F, so summing per-rank gradients and dividing by the world size
reproduces the single-GPU gradient of the global batch when ranks hold equal batches of equal padded
length -- the plain a / v / av models on fixed-length clips, which is what BASELINE config 4 trains.
The variants whose objective is ``loss_hole = sum|err|(1-m) / sum(1-m)`` (-ssnn / -emb / -ctc) all-reduce the
denominator -- the ranks' gap element counts, one float before the backward pass -- and rescale their gradient by
``gap_r * world / G``, so the summed buckets divided by the world size are the gradient of the GLOBAL ratio of sums
whatever the ranks' shares of the gap frames (``model.loss_hole_global`` reports that ratio).  What stays an
approximation: ragged batches of the plain models (T = the rank's own longest clip, so the ranks' means are over
different element counts; the ranks then weigh equally) -- knowing the global count would cost a host round trip per step.
``model.gradients`` is the LOCAL gradient before ``train_op`` and the world-SUM (not yet divided) after it.
"""
import os

import torch
import torch.distributed as dist


def env_world():
    """(rank, local_rank, world_size) from the torch.distributed.run environment."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def rehearsing():
    """AVSI_DP_REHEARSE=1: a ONE-rank process group counts as data parallel -- the bucketed asynchronous all-reduce, the
    guard words behind the last bucket, the CU reserve of the cooperative kernels and the 1 / world inside the fused Adam
    all run against the communicator (backend nccl = RCCL: its stream, its events, its ordering against the launch
    stream) on a box that has ONE GPU.  RCCL moves no data between ranks then; everything on this side of it is the
    production path (tests/test_dp_gpu.py::test_world1_rccl_rehearsal, bench.py `dp_train`)."""
    return os.environ.get('AVSI_DP_REHEARSE', '0') == '1'


def init(backend=None):
    """Initialise the default process group when launched with WORLD_SIZE > 1 (or rehearsing()); returns (rank, world)."""
    rank, local_rank, world = env_world()
    if (world > 1 or rehearsing()) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            # AVSI_DIST_BACKEND=gloo: ranks that share one GPU (tests on a single-GPU box; RCCL refuses two ranks
            # on one device) -- the gradient all-reduce is then staged through the host
            backend = os.environ.get("AVSI_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if torch.cuda.is_available():
            torch.cuda.set_device(local_rank % torch.cuda.device_count())
        dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world


def collectives_share_the_gpu():
    """True when this process runs RCCL collectives on its GPU (world > 1, backend nccl): their kernels hold compute
    units while the cooperative recurrent kernels run (the bucketed gradient all-reduce overlaps the BPTT of the
    layers below), so those kernels are sized to leave ops.COOP_CU_RESERVE CUs free (ops.coop_cu_budget)."""
    return dp_active() and dist.get_backend() == 'nccl'


_SOLO = 0


class solo(object):
    """Context manager: inside it this process behaves as a single-process job (world_size() == 1, rank() == 0: no
    gradient all-reduce, no CU reserve for collectives) although a process group exists.  For the self-check of a
    data-parallel run -- one rank repeats the global batch alone while its peers wait at a barrier (bench.py,
    `dp_train.check`) -- never for training proper: the ranks' variables diverge under it."""

    def __enter__(self):
        global _SOLO
        _SOLO += 1
        return self

    def __exit__(self, *exc):
        global _SOLO
        _SOLO -= 1
        return False


def dp_active():
    """True when this process's gradients go through the process group's collectives: a group exists, this is not a
    solo() section, and there is more than one rank (or the one-rank rehearsal is switched on)."""
    return dist.is_available() and dist.is_initialized() and not _SOLO and (dist.get_world_size() > 1 or rehearsing())


def world_size():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() and not _SOLO else 1


def rank():
    return dist.get_rank() if dist.is_available() and dist.is_initialized() and not _SOLO else 0


def shard_range(n, rank_, world):
    """Contiguous [lo, hi) slice of n items for `rank_` of `world` (sizes differ by at most 1)."""
    base, extra = divmod(n, world)
    lo = rank_ * base + min(rank_, extra)
    return lo, lo + base + (1 if rank_ < extra else 0)


def all_reduce_sum_(flat):
    """In-place sum over ranks of one flat buffer (no-op on a single process)."""
    if dp_active():
        if flat.is_cuda and dist.get_backend() != 'nccl':
            host = flat.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM)
            flat.copy_(host)
        else:
            dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    return flat


def all_reduce_mean_scalars(values):
    """Mean over ranks of a few python floats (validation losses); returns a list of floats."""
    if not dp_active():
        return list(values)
    dev = 'cuda' if dist.get_backend() == 'nccl' else 'cpu'
    t = torch.tensor(list(values), dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return (t / world_size()).tolist()


def all_reduce_sum_scalars(values):
    """Sum over ranks of a few python floats; returns a list of floats."""
    if not dp_active():
        return list(values)
    dev = 'cuda' if dist.get_backend() == 'nccl' else 'cpu'
    t = torch.tensor(list(values), dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.tolist()


def all_reduce_max_scalar(value):
    """Maximum over ranks of one python float (the 'some rank saw a non-finite loss' verdict of the trainer)."""
    if not dp_active():
        return float(value)
    dev = 'cuda' if dist.get_backend() == 'nccl' else 'cpu'
    t = torch.tensor([float(value)], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def all_reduce_sum_async(flat):
    """Start the in-place sum over ranks of one buffer and return a handle to wait() on (None when there is nothing
    to wait for).  With RCCL the collective runs on the communicator's stream, ordered after the work already
    enqueued on the current stream: called right after the kernels that produced `flat`, it overlaps with whatever
    the caller enqueues next (the BPTT of the layers below)."""
    if not dp_active():
        return None
    if flat.is_cuda and dist.get_backend() == 'nccl':
        return dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True)
    all_reduce_sum_(flat)
    return None


def shards_step(model, feeds):
    """One optimiser step over ``feeds`` -- one feed dict per would-be rank -- in ONE process: every shard runs through the
    model's own forward and backward at the ranks' batch size (hence on the kernels a rank of that batch size runs), the
    gradients are summed in rank order and applied once with 1 / len(feeds), exactly what ``train_op`` does behind the
    all-reduce.  What differs from a data-parallel step of that world size is the collective alone (and the order in
    which it adds the ranks' terms): the like-with-like reference of bench.py's `dp_train.check` and tests/test_dp_gpu.py.
    Also plain gradient accumulation over micro-batches.  Call inside ``solo()`` when a process group exists.
    Returns (the shards' loss_func values as device scalars, the summed gradient)."""
    gsum, losses = None, []
    for feed in feeds:
        model.feed(**feed)
        losses.append(model.loss_func.detach().clone())
        g = model._backward()
        gsum = g.clone() if gsum is None else gsum.add_(g)
    model.apply_gradients(gsum, world=len(feeds), guard=model.step_guard)
    model._cache['trained'] = True
    return losses, gsum

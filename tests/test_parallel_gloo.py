"""Data-parallel contract on CPU with gloo, world_size 2 and 8 (no GPU): summing per-rank gradients of
equal shards and dividing by the world size, then the TF-Adam update, reproduces the single-process
step on the global batch; shard helpers partition the work; the step-guard words of one rank reach every rank."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import avsi_amd  # noqa: F401
from avsi_amd import parallel
from oracle import blstm as O


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _problem(B=4):
    rng = np.random.default_rng(0)
    N = 1536
    wav = np.round(rng.normal(0, 3000, size=(B, N)))
    T = N // 192
    masks = np.ones((B, T, 257))
    masks[:, 2:4] = 0
    params = O.cast_params(O.init_params(1, 257, (6, 6), 257), np.float64)
    return wav, masks, np.zeros(257), np.ones(257), np.full(B, T), params


def _grads(wav, masks, mean, std, seq, params):
    fwd = O.model_forward(wav, masks, mean, std, seq, params, keep=True)
    g = O.model_backward(fwd, masks, seq)
    return np.concatenate([v.reshape(-1) for _, v in O.flatten_params(g)]), fwd['loss']


def _worker(rank, world, port, out, B=4):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    r, w = parallel.init(backend="gloo")
    assert (r, w) == (rank, world) and parallel.world_size() == world and parallel.rank() == rank
    wav, masks, mean, std, seq, params = _problem(B)
    lo, hi = parallel.shard_range(len(wav), rank, world)
    g, loss = _grads(wav[lo:hi], masks[lo:hi], mean, std, seq[lo:hi], params)
    flat = torch.from_numpy(g)
    parallel.all_reduce_sum_(flat)
    flat /= world
    (mean_loss,) = parallel.all_reduce_mean_scalars([loss])
    # the step guard of models.StackedBLSTMModel travels as two words behind the gradients: rank 5 (world 8) reports a
    # non-finite loss and rank 3 a cooperative timeout -- every rank must read NaN and 1 after the sum
    guard = torch.tensor([float('nan') if rank == 5 else 0.0, 1.0 if rank == 3 else 0.0])
    parallel.all_reduce_sum_(guard)
    with parallel.solo():
        assert parallel.world_size() == 1 and parallel.rank() == 0
        alone = parallel.all_reduce_sum_(torch.tensor([float(rank)]))       # no collective under solo()
        assert float(alone) == float(rank)
    if rank == 0:
        np.save(out, np.concatenate([flat.numpy(), [mean_loss]]))
    np.save(out + ".guard%d.npy" % rank, guard.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gradient_average_equals_global_batch(tmp_path):
    out = str(tmp_path / "g.npy")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    got = np.load(out)
    wav, masks, mean, std, seq, params = _problem()
    ref, loss = _grads(wav, masks, mean, std, seq, params)
    np.testing.assert_allclose(got[:-1], ref, rtol=1e-10, atol=1e-14)
    assert got[-1] == pytest.approx(loss, rel=1e-12)
    # identical Adam step from identical averaged gradients
    p1 = np.concatenate([v.reshape(-1) for _, v in O.flatten_params(params)])
    p2 = p1.copy()
    O.adam_tf_step(p1, ref, np.zeros_like(p1), np.zeros_like(p1), 1)
    O.adam_tf_step(p2, got[:-1], np.zeros_like(p2), np.zeros_like(p2), 1)
    np.testing.assert_allclose(p1, p2, rtol=0, atol=1e-12)


def test_eight_ranks_at_global_batch_256_equal_the_global_batch(tmp_path):
    """BASELINE configs[3]'s world: 8 ranks x 32 utterances = global batch 256 (a narrow network and short clips keep
    the float64 oracle quick): the averaged gradient equals the single-process gradient of all 256, and the guard words
    of ranks 5 and 3 arrive on every rank."""
    out = str(tmp_path / "g8.npy")
    mp.spawn(_worker, args=(8, _free_port(), out, 256), nprocs=8, join=True)
    got = np.load(out)
    wav, masks, mean, std, seq, params = _problem(256)
    ref, loss = _grads(wav, masks, mean, std, seq, params)
    np.testing.assert_allclose(got[:-1], ref, rtol=1e-9, atol=1e-13)
    assert got[-1] == pytest.approx(loss, rel=1e-12)
    for rank in range(8):
        guard = np.load(out + ".guard%d.npy" % rank)
        assert np.isnan(guard[0]) and guard[1] == 1.0, rank


def test_shard_range_partitions_everything():
    for n in (0, 1, 7, 8, 100):
        for world in (1, 2, 3, 8):
            spans = [parallel.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def test_single_process_helpers_are_noops():
    t = torch.arange(4.0)
    assert parallel.world_size() == 1 and parallel.rank() == 0
    assert torch.equal(parallel.all_reduce_sum_(t.clone()), t)
    assert parallel.all_reduce_mean_scalars([1.5, 2.5]) == [1.5, 2.5]


def _sum_worker(rank, world, port, out):
    import os
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import avsi_amd  # noqa: F401
    from avsi_amd import parallel
    parallel.init(backend="gloo")
    # rank 1 saw no batch at all: frame-weighted means combine to rank 0's value, not to half of it
    vals, frames = ([0.5, 0.25], 40.0) if rank == 0 else ([float('nan'), float('nan')], 0.0)
    sums = parallel.all_reduce_sum_scalars([v * frames if frames else 0.0 for v in vals] + [frames])
    if rank == 0:
        np.save(out, np.array([s / sums[-1] for s in sums[:-1]]))
    import torch.distributed as dist
    dist.destroy_process_group()


def test_weighted_combination_ignores_ranks_without_batches(tmp_path):
    out = str(tmp_path / "sums.npy")
    mp.spawn(_sum_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    np.testing.assert_allclose(np.load(out), [0.5, 0.25])


def _rehearsal_worker(port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
                      AVSI_DP_REHEARSE="1")
    assert parallel.dp_active() is False            # no group yet
    r, w = parallel.init(backend="gloo")
    assert (r, w) == (0, 1) and dist.is_initialized() and parallel.world_size() == 1
    # a one-rank group counts as data parallel while rehearsing: the collectives are issued, and sum over one rank is identity
    assert parallel.dp_active() is True and parallel.collectives_share_the_gpu() is False      # gloo: no kernels on the GPU
    calls = []
    plain = dist.all_reduce
    dist.all_reduce = lambda t, *a, **kw: (calls.append(t.numel()), plain(t, *a, **kw))[1]
    flat = torch.arange(10, dtype=torch.float32)
    parallel.all_reduce_sum_(flat)
    assert parallel.all_reduce_sum_async(flat[:4]) is None and torch.equal(flat, torch.arange(10, dtype=torch.float32))
    assert parallel.all_reduce_sum_scalars([1.5, 2.0]) == [1.5, 2.0] and parallel.all_reduce_max_scalar(3.0) == 3.0
    assert calls == [10, 4, 2, 1]
    with parallel.solo():
        assert parallel.dp_active() is False
        parallel.all_reduce_sum_(flat)
    assert calls == [10, 4, 2, 1]
    os.environ["AVSI_DP_REHEARSE"] = "0"
    assert parallel.dp_active() is False             # a one-rank group without the switch: a plain single process
    dist.destroy_process_group()
    q.put("ok")


def test_one_rank_rehearsal_switch():
    """AVSI_DP_REHEARSE=1 (parallel.rehearsing): a one-rank process group issues every collective of the data-parallel
    path; without the switch, or inside solo(), it issues none."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rehearsal_worker, args=(_free_port(), q))
    p.start()
    p.join(120)
    assert p.exitcode == 0 and q.get(timeout=5) == "ok"


def test_bench_plain_launch_starts_a_child_launcher_and_relays_its_status():
    """`python bench.py --gpus 2` with no launcher around it (the form the driver uses at N = 1) must start its ranks
    itself instead of exiting with a usage message.  Without a GPU the ranks cannot get far: what is checked here is that
    the parent started `torch.distributed.run` with two ranks, that both reached bench.py's own code (each fails at the
    first GPU call), that the parent neither imported torch nor hung, and that it passes the failure on as its status."""
    import subprocess
    import sys
    if torch.cuda.is_available():
        pytest.skip("covered on the GPU by tests/test_bench_contract_gpu.py")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--batch", "64", "--steps", "1",
                          "--warmup", "0"], env=env, capture_output=True, text=True, timeout=300, cwd=root)
    assert out.returncode != 0
    assert "must be launched with" not in out.stderr
    assert "local_rank: 0" in out.stderr and "local_rank: 1" in out.stderr, out.stderr[-2000:]       # torchrun's failure report


def _hole_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    parallel.init(backend="gloo")
    err, m, w = _hole_problem()
    lo, hi = parallel.shard_range(len(err), rank, world)
    wr = w.clone().requires_grad_(True)
    num = ((err[lo:hi] * wr).abs() * (1 - m[lo:hi])).sum()
    gap = (1 - m[lo:hi]).sum()
    (num / gap).backward()                                 # what the loss kernel leaves: d num_r / gap_r
    # models.StackedBLSTMModel._forward (blend variants): rescale by gap_r * world / G, sum the buckets, 1 / world in Adam
    total = parallel.all_reduce_sum_(gap.clone().reshape(1))
    g = wr.grad * (gap * world / total)
    parallel.all_reduce_sum_(g)
    g /= world
    both = parallel.all_reduce_sum_(torch.stack([num.detach(), gap]))
    if rank == 0:
        np.savez(out, g=g.numpy(), loss=float(both[0] / both[1]), local=float(num / gap))
    dist.destroy_process_group()


def _hole_problem():
    gen = torch.Generator().manual_seed(3)
    err = torch.randn(4, 6, 5, generator=gen, dtype=torch.float64)
    m = torch.ones(4, 6, 5, dtype=torch.float64)
    for b, n in enumerate((1, 2, 4, 5)):                   # gap frames per utterance: 3 on rank 0, 9 on rank 1
        m[b, :n] = 0
    return err, m, torch.rand(5, generator=gen, dtype=torch.float64) + 0.5


def test_loss_hole_objective_is_the_ratio_of_global_sums(tmp_path):
    """under data parallelism loss_hole's numerator and denominator are summed over the ranks separately.  Two
    ranks with 3 against 9 gap frames reproduce the single-process gradient and loss of the four utterances; the mean of
    the ranks' own ratios differs."""
    out = str(tmp_path / "hole.npz")
    mp.spawn(_hole_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    got = np.load(out)
    err, m, w = _hole_problem()
    wr = w.clone().requires_grad_(True)
    loss = ((err * wr).abs() * (1 - m)).sum() / (1 - m).sum()
    loss.backward()
    np.testing.assert_allclose(got["g"], wr.grad.numpy(), rtol=1e-12)
    np.testing.assert_allclose(got["loss"], float(loss.detach()), rtol=1e-12)
    assert abs(float(got["local"]) - float(loss.detach())) > 1e-3 * float(loss.detach())

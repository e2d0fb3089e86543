"""Data-parallel training with the real kernels (BASELINE config 4's contract: the same maths as one GPU at the
same global batch).  Two ranks launched with torch.distributed.run share the one GPU of the test box, so the
process group is gloo (AVSI_DIST_BACKEND; RCCL refuses two ranks on a device) and the gradient all-reduce is
staged through the host -- everything else (sharding, gradient scaling inside the fused Adam, identical
variables on all ranks) is the production path."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def test_two_ranks_equal_one_process_at_the_global_batch(tmp_path):
    env = dict(os.environ, AVSI_DIST_BACKEND='gloo', HSA_ENABLE_IPC_MODE_LEGACY='0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), os.path.join(HERE, 'dp_worker.py'), str(tmp_path)]
    run = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert run.returncode == 0, run.stderr[-3000:]
    f0, f1 = np.load(str(tmp_path / 'flat_rank0.npy')), np.load(str(tmp_path / 'flat_rank1.npy'))
    assert np.array_equal(f0, f1)                          # every rank applies the same update
    sys.path.insert(0, HERE)
    import dp_worker
    ref, ref_losses = dp_worker.run(0, 1, steps=3)
    init, _ = dp_worker.run(0, 1, steps=0)
    # three Adam steps move every weight by ~3e-3; the two runs differ by summation order only
    assert np.abs(ref - init).max() > 1e-3
    np.testing.assert_allclose(f0, ref, rtol=0, atol=2e-5)
    # loss_func is a mean over B*T*F: the global loss is the mean of the two ranks' losses
    l0, l1 = np.load(str(tmp_path / 'loss_rank0.npy')), np.load(str(tmp_path / 'loss_rank1.npy'))
    np.testing.assert_allclose((l0 + l1) / 2, ref_losses, rtol=2e-4)


def test_two_ranks_at_config_4_shape_equal_one_process(tmp_path):
    """BASELINE configs[3]'s per-GPU share: 32 utterances per rank, 3 s clips (T = 250), AV model, two Adam steps; the result
    equals one process at 64 utterances."""
    env = dict(os.environ, AVSI_DIST_BACKEND='gloo', HSA_ENABLE_IPC_MODE_LEGACY='0', AVSI_COOP_CUS='128')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), os.path.join(HERE, 'dp_worker.py'), str(tmp_path), '64', '48000', '2']
    run = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert run.returncode == 0, run.stderr[-3000:]
    f0, f1 = np.load(str(tmp_path / 'flat_rank0.npy')), np.load(str(tmp_path / 'flat_rank1.npy'))
    assert np.array_equal(f0, f1)
    sys.path.insert(0, HERE)
    import dp_worker
    ref, ref_losses = dp_worker.run(0, 1, steps=2, B_global=64, N=48000)
    init, _ = dp_worker.run(0, 1, steps=0, B_global=64, N=48000)
    assert np.abs(ref - init).max() > 1e-3
    np.testing.assert_allclose(f0, ref, rtol=0, atol=2e-5)
    l0, l1 = np.load(str(tmp_path / 'loss_rank0.npy')), np.load(str(tmp_path / 'loss_rank1.npy'))
    np.testing.assert_allclose((l0 + l1) / 2, ref_losses, rtol=2e-4)


@pytest.mark.parametrize("ranks,N", [(4, 48000), (3, 9600)])
def test_four_and_three_ranks_at_32_per_rank_equal_one_process(tmp_path, ranks, N):
    """configs[3]'s per-GPU share (32 utterances, AV model) at world 4 (3 s clips, T = 250; with this process and the
    launcher the most the box allows on its one GPU) and at an odd world of 3 (T = 50); world 8 is rehearsed on the CPU
    (tests/test_parallel_gloo.py).  Every
    rank holds the same bits after two Adam steps, equal to one process at 32 x ranks utterances up to summation order.
    Each rank sizes its cooperative launches for its share of the chip (AVSI_COOP_CUS = 256 // ranks)."""
    G = 32 * ranks
    env = dict(os.environ, AVSI_DIST_BACKEND='gloo', HSA_ENABLE_IPC_MODE_LEGACY='0', AVSI_COOP_CUS=str(256 // ranks // 8 * 8))
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(ranks), '--master-addr',
           '127.0.0.1', '--master-port', str(_free_port()), os.path.join(HERE, 'dp_worker.py'), str(tmp_path), str(G), str(N), '2']
    run = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1200)
    assert run.returncode == 0, run.stderr[-3000:]
    flats = [np.load(str(tmp_path / ('flat_rank%d.npy' % r))) for r in range(ranks)]
    assert all(np.array_equal(flats[0], f) for f in flats[1:])
    sys.path.insert(0, HERE)
    import dp_worker
    ref, ref_losses = dp_worker.run(0, 1, steps=2, B_global=G, N=N)
    init, _ = dp_worker.run(0, 1, steps=0, B_global=G, N=N)
    assert np.abs(ref - init).max() > 1e-3
    np.testing.assert_allclose(flats[0], ref, rtol=0, atol=2e-5)
    losses = np.mean([np.load(str(tmp_path / ('loss_rank%d.npy' % r))) for r in range(ranks)], axis=0)
    np.testing.assert_allclose(losses, ref_losses, rtol=2e-4)


def _gpu_count():
    try:
        import torch
        return torch.cuda.device_count()
    except Exception:
        return 0


@pytest.mark.skipif(_gpu_count() < 2, reason="needs two GPUs: the RCCL branch (backend nccl) refuses two ranks on one device")
@pytest.mark.parametrize("B_global", [4, 192])
def test_two_ranks_over_rccl_equal_one_process(tmp_path, B_global):
    """The production branch: backend nccl (= RCCL), asynchronous all-reduce of the per-layer gradient buckets issued
    behind each layer's weight-gradient kernels -- from the side stream at Bp <= 64 (B_global 4), from the main
    stream above (B_global 192: 96 utterances per rank).  Bitwise-identical variables on both ranks, equal to one
    process at the global batch up to summation order."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    env.pop('AVSI_DIST_BACKEND', None)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), os.path.join(HERE, 'dp_worker.py'), str(tmp_path), str(B_global)]
    run = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert run.returncode == 0, run.stderr[-3000:]
    f0, f1 = np.load(str(tmp_path / 'flat_rank0.npy')), np.load(str(tmp_path / 'flat_rank1.npy'))
    assert np.array_equal(f0, f1)
    sys.path.insert(0, HERE)
    import dp_worker
    ref, ref_losses = dp_worker.run(0, 1, steps=3, B_global=B_global)
    np.testing.assert_allclose(f0, ref, rtol=0, atol=2e-5)
    l0, l1 = np.load(str(tmp_path / 'loss_rank0.npy')), np.load(str(tmp_path / 'loss_rank1.npy'))
    np.testing.assert_allclose((l0 + l1) / 2, ref_losses, rtol=2e-4)


@pytest.mark.parametrize("B", [32, 192])
def test_world1_rccl_rehearsal(tmp_path, B):
    """The `nccl` branch on a ONE-GPU box: a one-rank RCCL communicator with AVSI_DP_REHEARSE=1 (parallel.rehearsing), so that
    the per-layer asynchronous all-reduce buckets (issued from the side stream at 32 utterances, from the main stream at
    192), the two guard words behind the last bucket, the guarded Adam's 1 / world and the cooperative kernels' CU budget
    of 256 - COOP_CU_RESERVE all run against RCCL's communicator, its stream and its event ordering.  RCCL has no peer to
    move data to, so the result must equal a process without any group -- to summation order only where the smaller CU
    budget cuts a launch differently."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0', AVSI_DP_REHEARSE='1', MASTER_ADDR='127.0.0.1',
               MASTER_PORT=str(_free_port()), RANK='0', LOCAL_RANK='0', WORLD_SIZE='1')
    env.pop('AVSI_DIST_BACKEND', None)
    cmd = [sys.executable, os.path.join(HERE, 'dp_worker.py'), str(tmp_path), str(B), '48000', '3']
    run = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert run.returncode == 0, run.stderr[-3000:]
    import re
    m = re.search(r'RANK 0 BACKEND nccl SHARES_GPU True COOP_CUS (\d+) ALL_REDUCE async (\d+) sync (\d+)', run.stdout)
    assert m, run.stdout[-2000:]
    from avsi_amd import ops
    assert int(m.group(1)) == 256 - ops.COOP_CU_RESERVE
    assert int(m.group(2)) >= 3 * 4                      # three layers' buckets and the head's, every step
    f0 = np.load(str(tmp_path / 'flat_rank0.npy'))
    sys.path.insert(0, HERE)
    import dp_worker
    ref, ref_losses = dp_worker.run(0, 1, steps=3, B_global=B, N=48000)
    init, _ = dp_worker.run(0, 1, steps=0, B_global=B, N=48000)
    assert np.abs(ref - init).max() > 1e-3
    np.testing.assert_allclose(f0, ref, rtol=0, atol=2e-5)
    np.testing.assert_allclose(np.load(str(tmp_path / 'loss_rank0.npy')), ref_losses, rtol=2e-4)


@pytest.mark.parametrize("gaps", ["uneven", "zero"])
def test_two_ranks_with_unequal_gaps_train_the_global_loss_hole(tmp_path, gaps):
    """The variants' objective is loss_hole = sum|err|(1-m) / sum(1-m) (models.py:1006-1029): under data parallelism numerator
    and denominator are summed over the ranks separately, not the ranks' ratios averaged.  Four utterances with
    gaps of 2, 5, 8 and 11 frames, two per rank (7 against 19 gap frames): two Adam steps on two ranks equal one process at
    the four utterances to summation order, `loss_hole_global` is the single process's loss_hole on both ranks, and the plain
    mean of the ranks' own ratios is NOT (the test would not notice the difference otherwise).
    ``zero``: rank 0's two utterances have NO gap element (0 against 19 gap frames): its own ratio is 0 / 0, the global one
    is well defined -- the step must not be voided by a NaN on that rank, its local loss_hole reads 0."""
    env = dict(os.environ, AVSI_DIST_BACKEND='gloo', HSA_ENABLE_IPC_MODE_LEGACY='0', AVSI_COOP_CUS='128')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), os.path.join(HERE, 'dp_worker.py'), str(tmp_path), '4', '2880', '2', 'emb', gaps]
    run = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert run.returncode == 0, run.stderr[-3000:]
    f0, f1 = np.load(str(tmp_path / 'flat_rank0.npy')), np.load(str(tmp_path / 'flat_rank1.npy'))
    assert np.array_equal(f0, f1) and np.isfinite(f0).all()
    sys.path.insert(0, HERE)
    import dp_worker
    ref, ref_hole, ref_hole_g = dp_worker.run_variant(0, 1, steps=2, B_global=4, N=2880, gaps=gaps)
    init, _, _ = dp_worker.run_variant(0, 1, steps=0, B_global=4, N=2880, gaps=gaps)
    assert np.abs(ref - init).max() > 1e-3 and ref_hole == ref_hole_g
    np.testing.assert_allclose(f0, ref, rtol=0, atol=2e-5)
    l0, l1 = np.load(str(tmp_path / 'loss_rank0.npy')), np.load(str(tmp_path / 'loss_rank1.npy'))
    np.testing.assert_allclose(l0[1], ref_hole, rtol=2e-4)
    np.testing.assert_allclose(l1[1], ref_hole, rtol=2e-4)
    if gaps == 'zero':
        assert np.all(l0[0] == 0.0)
    assert np.abs((l0[0] + l1[0]) / 2 - np.array(ref_hole)).max() > 1e-3 * np.abs(ref_hole).max()

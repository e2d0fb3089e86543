"""bench.py prints ONE JSON line with the fields the driver reads (small batches here; the defaults are the
BASELINE configuration)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*args):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True, timeout=600,
                         cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


def test_default_mode_line():
    d = _run("--batch", "256", "--steps", "2", "--warmup", "1", "--cpu-sample", "32")
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "f32" and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s") and 0 < r["frac"] < 1
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and r["avg_launch_ms"] > 0
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["value"] > 0 and c["cores"] >= 1 and "sample" in c
    assert c["batch_8"]["value"] > 0 and "batches of 8" in c["batch_8"]["sample"]          # 8 and 32
    assert d["logmel_rms_vs_cpu_oracle"] < 1e-3                  # the BASELINE parity bar, checked in the bench itself
    # the sizes the reference and BASELINE.json name, each with its own ms_per_step
    also = d["also"]
    assert "error" not in also, also
    for k, b in (("infer_b100", 100), ("infer_b32", 32), ("infer_b128", 128), ("infer_b1024", 1024), ("train_b32", 32),
                 ("lws_b32", 32)):
        assert also[k]["per_gpu_batch"] == b and also[k]["ms_per_step"] > 0
        assert abs(also[k]["value"] - b / also[k]["ms_per_step"] * 1e3) < 1e-6 * also[k]["value"]
    assert abs(d["value"] - 256 * 2 / (d["ms_per_step"] * 2e-3)) < 1e-6 * d["value"]
    # the other configurations and the steps either side of the path (N = 1 only), each with an algorithmic roofline figure
    for k, b in (("train_b8192", 8192), ("unet_b512", 512), ("unet_b32", 32), ("unet_train_b512", 512), ("unet_train_b32", 32),
                 ("istft_b4096", 4096), ("lws_b1024", 1024),
                 ("infer_b8192_hostfed", 8192)):
        assert "error" not in also[k], also[k]
        assert also[k]["per_gpu_batch"] == b and also[k]["ms_per_step"] > 0 and also[k]["value"] > 0
    # the reference's drivers end to end (records under /dev/shm in, WAV files out / training epochs)
    assert "e2e" not in also, also.get("e2e")
    for k in ("e2e_train_b32", "e2e_infer_b1024_oracle_phase", "e2e_infer_b1024", "e2e_infer_b32_oracle_phase", "e2e_infer_b32"):
        assert also[k]["value"] > 0 and also[k]["ms_per_step"] > 0, k
    assert 0 < also["train_b8192"]["kernels"]["blstm_rec_bwd_pp_kernel"]["frac"] < 1
    assert 0 < also["unet_b512"]["frac_of_fp32_mfma_peak"] < 1 and 0 < also["istft_b4096"]["frac_of_hbm_peak"] < 1
    assert 0 < also["unet_train_b512"]["frac_of_fp32_mfma_peak"] < 1 and also["istft_b4096"]["from_stored_stft"]["ms_per_step"] > 0
    assert also["infer_b8192_hostfed"]["serial_upload_then_compute"]["ms_per_step"] >= also["infer_b8192_hostfed"]["ms_per_step"] * 0.9
    # configs[3], the same keys at every N: weak (32 per GPU) and fixed global 256 (256 / N per GPU)
    dp = d["dp_train"]
    assert "error" not in dp, dp
    assert dp["rccl_ranks"] == 1 and dp["weak_32_per_gpu"]["per_gpu_batch"] == 32 and dp["fixed_global_256"]["per_gpu_batch"] == 256
    assert dp["check"]["max_abs_update"] > 1e-3
    fe = d["roofline"]["others"]["frontend_kernel"]
    lo, med, hi = fe["launch_ms_min_median_max"]
    assert 0 < lo <= med <= hi


def test_train_and_unet_modes():
    t = _run("--mode", "train", "--batch", "64", "--steps", "1", "--warmup", "1")
    assert t["unit"] == "utterances/s" and t["value"] > 0 and "blstm_rec_bwd" in t["kernel_ms_per_step"]
    u = _run("--mode", "unet", "--batch", "64", "--steps", "2", "--warmup", "1")
    assert u["unit"] == "clips/s" and u["value"] > 0 and u["vs_baseline"] is None


@pytest.mark.parametrize("mode,ranks", [("infer", 2), ("train", 2), ("infer", 4)])
def test_multi_rank_launch_as_the_driver_does(mode, ranks):
    """`python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 ... bench.py --gpus 2`: rank 0 alone prints
    the line, the value is the whole-job aggregate, no CPU baseline leg.  The two ranks share this box's one GPU, so
    the process group is gloo here (AVSI_DIST_BACKEND); the driver's 8-GPU node uses RCCL."""
    import socket
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, AVSI_DIST_BACKEND='gloo', HSA_ENABLE_IPC_MODE_LEGACY='0', AVSI_COOP_CUS=str(256 // ranks))
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(ranks), '--master-addr',
           '127.0.0.1', '--master-port', str(port), os.path.join(ROOT, 'bench.py'), '--gpus', str(ranks), '--steps', '2',
           '--warmup', '1', '--batch', '64', '--mode', mode]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1500, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == ranks and d["scaling"] == "weak" and d["config"]["global_batch"] == 64 * ranks
    assert d["config"]["parallelism"] == "dp%d" % ranks
    assert abs(d["value"] - ranks * 64 * 2 / (d["ms_per_step"] * 2e-3)) < 1e-6 * d["value"]
    assert "cpu_baseline" not in d or d["cpu_baseline"] is None
    if mode == "infer":
        assert "error" not in d["also"], d["also"]
        assert d["also"]["train_b32"]["global_batch"] == 32 * ranks and d["also"]["train_b32"]["ms_per_step"] > 0
        # the data-parallel self-check of the bench line: every rank ends with the same bits, equal to one process at
        # the global batch up to summation order (gloo here; the driver's multi-GPU run is the same code over RCCL)
        dp = d["dp_train"]
        assert "error" not in dp, dp
        assert dp["rccl_ranks"] == ranks and dp["backend"] == "gloo"
        # ranks that share ONE GPU can starve each other's 32-way cooperative groups (each wants XCD 0 and 1 to itself): the
        # bench then falls back like the trainer does and still delivers every number
        assert dp["coop_fallbacks"] in (0, 1, 2)
        c = dp["check"]
        assert c["ranks_bit_identical"] is True and c["global_batch"] == 32 * ranks and c["frames"] == 250
        # r6: the reference takes the same steps shard by shard on the ranks' kernels (parallel.shards_step): bar 2e-5 again,
        # and the summed gradient of the first step is compared as well (Adam's variables cannot show a missed bucket)
        assert c["max_abs_diff_vs_single_process"] < c["bar"] <= (2e-4 if "bar_note" in c else 2e-5) < 1e-3 < c["max_abs_update"]
        assert c["ok"] is True
        assert c["first_step_summed_gradient_max_abs_diff_rel"] < c["gradient_bar"] <= 1e-4
        assert c["loss_max_rel_diff_vs_single_process"] < 2e-4
        assert dp["weak_32_per_gpu"]["global_batch"] == 32 * ranks and dp["fixed_global_256"]["per_gpu_batch"] == 256 // ranks
        assert dp["fixed_global_256"]["global_batch"] == 256
        for k in ("weak_32_per_gpu", "fixed_global_256"):
            assert dp[k]["ms_per_step"] > 0 and dp[k]["ms_per_step_no_collective"] > 0


def test_plain_launch_with_gpus_2_starts_its_own_ranks():
    """`python bench.py --gpus 2 ...` exactly as the driver launches N = 1 (no torch.distributed.run around it): the
    process starts its two ranks as a child launcher before it touches a GPU, relays rank 0's one line and its status."""
    env = dict(os.environ, AVSI_DIST_BACKEND='gloo', HSA_ENABLE_IPC_MODE_LEGACY='0', AVSI_COOP_CUS='128')
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--batch", "64", "--steps", "2",
                          "--warmup", "1"], env=env, capture_output=True, text=True, timeout=1500, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 128 and d["config"]["parallelism"] == "dp2"
    dp = d["dp_train"]
    assert "error" not in dp, dp
    assert dp["rccl_ranks"] == 2 and dp["check"]["ok"] is True and dp["check"]["ranks_bit_identical"] is True


def test_plain_launch_passes_a_failing_status_on():
    """Ranks that exit non-zero make the plain form exit non-zero too (here: a process-group backend that does not exist)."""
    env = dict(os.environ, AVSI_DIST_BACKEND='no-such-backend')
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--batch", "64", "--steps", "1",
                          "--warmup", "0"], env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode != 0
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]


def test_world1_rccl_rehearsal_in_the_bench_line():
    """AVSI_DP_REHEARSE=1 at N = 1: `dp_train` runs over a one-rank RCCL communicator (backend nccl) -- the asynchronous
    buckets, the guard words and the reduced cooperative CU budget against RCCL's stream -- and its self-check compares
    that run with the same process without collectives."""
    env = dict(os.environ, AVSI_DP_REHEARSE='1', HSA_ENABLE_IPC_MODE_LEGACY='0')
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "AVSI_DIST_BACKEND"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--batch", "64", "--steps", "2", "--warmup", "1",
                          "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    dp = d["dp_train"]
    assert "error" not in dp, dp
    assert dp["backend"] == "nccl" and dp["rccl_ranks"] == 1 and "rehearsal" in dp
    c = dp["check"]
    assert c["ranks_bit_identical"] is True and c["ok"] is True and c["max_abs_diff_vs_single_process"] < c["bar"]
    assert dp["weak_32_per_gpu"]["ms_per_step_no_collective"] > 0

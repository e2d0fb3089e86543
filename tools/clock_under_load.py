"""Shader clock and socket power while ONE kernel runs back to back for ~2.5 s each: the layer-input GEMM, the ping-pong forward
recurrence without and with the BPTT reserve, the ping-pong BPTT kernel (all at 8192 utterances), and a streaming copy.
rocm-smi is polled from a thread (a child process each time; nothing is exec'ed).  python tools/clock_under_load.py [Bp]

Why: the fractions of the 157.3 TFLOP/s fp32 MFMA peak quoted for these kernels assume the 2.4 GHz peak clock; GRBM_GUI_ACTIVE
of the round-5 counter passes says the chip runs them at 2.38 / 2.26 / 2.13 / 2.10 GHz."""
import json, os, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import avsi_amd  # noqa: F401
from avsi_amd import ops, _lib

Bp = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
T = 250
samples = []
stop = threading.Event()


def poll():
    while not stop.is_set():
        try:
            out = subprocess.run(["rocm-smi", "-c", "-P", "--json"], capture_output=True, text=True, timeout=10).stdout
            d = json.loads(out[out.index("{"):])
            card = d[sorted(d)[0]]
            samples.append((time.perf_counter(), card))
        except Exception as e:          # noqa: BLE001
            samples.append((time.perf_counter(), {"error": str(e)[:100]}))
        time.sleep(0.15)


def run(name, fn, flops, seconds=2.5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    while time.perf_counter() - t0 < seconds:
        fn(); fn(); fn()
        torch.cuda.synchronize()
        n += 3
    e1.record(); torch.cuda.synchronize()
    t1 = time.perf_counter()
    ms = e0.elapsed_time(e1) / n
    mine = [c for (t, c) in samples if t0 + 0.5 < t < t1 and "error" not in c]

    def num(c, *keys):
        for k in c:
            if all(w in k.lower() for w in keys):
                s = str(c[k])
                digits = "".join(ch for ch in s.replace("Mhz", "").replace("MHz", "") if ch.isdigit() or ch == ".")
                try:
                    return float(digits)
                except ValueError:
                    pass
        return float("nan")
    sclk = [num(c, "sclk", "clock") for c in mine]
    pwr = [num(c, "power") for c in mine]
    line = {"kernel": name, "ms": round(ms, 3), "TFLOP/s": round(flops / ms / 1e9, 1) if flops else None, "samples": len(mine),
            "sclk_MHz_min_mean_max": [min(sclk), round(sum(sclk) / len(sclk)), max(sclk)] if sclk else None,
            "power_W_mean": round(sum(pwr) / len(pwr)) if pwr else None}
    if flops and sclk:
        peak_at_clock = 65536 * (sum(sclk) / len(sclk)) * 1e6 / 1e12       # 1024 SIMDs x 64 flop / clock (v_mfma_f32_32x32x2_f32)
        line["frac_of_mfma_peak_at_the_measured_clock"] = round(flops / ms / 1e9 / peak_at_clock, 3)
        line["frac_of_157.3"] = round(flops / ms / 1e9 / 157.3, 3)
    print(json.dumps(line), flush=True)
    if mine and not sclk:
        print("keys:", list(mine[0])[:20], flush=True)


th = threading.Thread(target=poll, daemon=True)
th.start()
dev = 'cuda'
M, N, K = T * Bp, 2048, 512
a = torch.randn(M, K, device=dev); b = torch.randn(K, N, device=dev); out = torch.empty(M, N, device=dev)
run("gemm_dma_kernel<...,256> %d x %d x %d" % (M, N, K), lambda: ops.gemm(a, b, out=out, m=M, n=N, k=K), 2.0 * M * N * K)
del a, b
xproj = out.view(T, Bp, 2048).mul_(0.01)
whp = torch.randn(2 * 262144, device=dev) * 0.05
hout = torch.empty(T, Bp, 512, device=dev)
resv = torch.empty(T, Bp, 2, 5, 256, device=dev)
rec_flops = 2.0 * 256 * 1024 * 2 * T * Bp
run("blstm_rec_fwd_pp_kernel<false>", lambda: ops.blstm_rec_fwd(xproj, whp, hout, None, split=0), rec_flops)
run("blstm_rec_fwd_pp_kernel<true>", lambda: ops.blstm_rec_fwd(xproj, whp, hout, resv, split=0), rec_flops)
resv.uniform_(0.05, 0.95)
hout.normal_()
run(_lib.lib().avsi_blstm_rec_bwd_kernel_name(Bp).decode(), lambda: ops.blstm_rec_bwd(hout, resv, whp, xproj, split=0), rec_flops)
src = resv.view(-1)[: 1 << 30]
dst = torch.empty_like(src)
run("avsi_diag_copy_f32 4 GiB", lambda: _lib.check(_lib.lib().avsi_diag_copy_f32(_lib.ptr(src), _lib.ptr(dst), src.numel(), _lib.stream_ptr()), "copy"), 0)
print("copy GB/s: see ms above: %.0f" % 0, flush=True)
stop.set()
if samples:
    print("one raw sample:", json.dumps(samples[len(samples) // 2][1])[:600], flush=True)

"""Re-time the single-box thresholds of DESIGN 4.3 on THIS box: both sides of every switch of the recurrent-kernel policy, per
layer (T = 250), with the box's identity (GPU serial / unique id from rocm-smi) in the first line -- run on two boxes, compare.
python tools/thresholds_sweep.py"""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import avsi_amd  # noqa: F401
from avsi_amd import ops

try:
    ident = subprocess.run(["rocm-smi", "--showuniqueid", "--showserial"], capture_output=True, text=True, timeout=20).stdout
    ident = " ".join(l.split(":", 2)[-1].strip() for l in ident.splitlines() if "Unique" in l or "Serial" in l)
except Exception as e:      # noqa: BLE001
    ident = "unknown (%s)" % e
print("box:", ident, "|", os.uname().nodename, flush=True)
T = 250


def timed(fn, reps=3):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


whp = torch.randn(2 * 262144, device='cuda') * 0.05
# ---- forward: 32-way against 16-way (switch at 128), 16-way against column split by 16 (256), column split by 32 against the
#      batch-stationary kernel (AVSI_REC_CS_MAX = 3584)
for Bp, splits in ((96, (32, 16)), (128, (32, 16)), (160, (32, 16)), (192, (16, -16)), (256, (16, -16)), (320, (16, -16)),
                   (512, (-16, -32)), (576, (-16, -32)), (3072, (-32, 0)), (3584, (-32, 0)), (4096, (-32, 0))):
    xproj = torch.randn(T, Bp, 2048, device='cuda') * 0.3
    hout = torch.empty(T, Bp, 512, device='cuda')
    out = []
    for sp in splits:
        try:
            out.append("split %4d: %7.3f ms" % (sp, timed(lambda: ops.blstm_rec_fwd(xproj, whp, hout, None, split=sp))))
        except Exception as e:      # noqa: BLE001
            out.append("split %4d: %s" % (sp, str(e)[:40]))
    ops.coop_check()
    print("fwd  Bp=%5d  %s   policy: %d" % (Bp, "   ".join(out), ops.coop_split(Bp)), flush=True)
    del xproj, hout
# ---- BPTT: the cooperative splits (32 / 16 / 8 / 4 at 128 / 256 / 512 / 2048), the ping-pong window (4096 < Bp <= 8192)
for Bp, splits in ((128, (32, 16)), (160, (32, 16)), (256, (16, 8)), (320, (16, 8)), (512, (8, 4)), (640, (8, 4)), (2048, (4, 0)),
                   (2560, (4, 0))):
    dh = torch.randn(T, Bp, 512, device='cuda')
    resv = torch.rand(T, Bp, 2, 5, 256, device='cuda') * 0.9 + 0.05
    dz = torch.empty(T, Bp, 2048, device='cuda')
    out = []
    for sp in splits:
        try:
            out.append("split %4d: %7.3f ms" % (sp, timed(lambda: ops.blstm_rec_bwd(dh, resv, whp, dz, split=sp))))
        except Exception as e:      # noqa: BLE001
            out.append("split %4d: %s" % (sp, str(e)[:40]))
    ops.coop_check()
    print("bptt Bp=%5d  %s   policy: %d" % (Bp, "   ".join(out), ops.coop_split(Bp, True)), flush=True)
    del dh, resv, dz
for Bp in (4096, 4160, 6144, 8192, 8256):
    dh = torch.randn(T, Bp, 512, device='cuda')
    resv = torch.rand(T, Bp, 2, 5, 256, device='cuda') * 0.9 + 0.05
    dz = torch.empty(T, Bp, 2048, device='cuda')
    out = []
    for pp in ("0", "1"):
        os.environ["AVSI_BWD_PP"] = pp
        out.append("%s: %7.3f ms" % ("ping-pong" if pp == "1" else "K-halved ", timed(lambda: ops.blstm_rec_bwd(dh, resv, whp, dz, split=0))))
    del os.environ["AVSI_BWD_PP"]
    from avsi_amd import _lib
    print("bptt Bp=%5d  %s   policy: %s" % (Bp, "   ".join(out), _lib.lib().avsi_blstm_rec_bwd_kernel_name(Bp).decode()), flush=True)
    del dh, resv, dz

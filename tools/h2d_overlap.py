"""Does a pinned host-to-device copy on a side stream overlap with kernels on the main stream?  python tools/h2d_overlap.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
dev = torch.device('cuda', 0)
n = 1 << 29                      # 2 GiB of float32
host = torch.empty(n, dtype=torch.float32, pin_memory=True)
dst = torch.empty(n, dtype=torch.float32, device=dev)
a = torch.randn(8192, 8192, device=dev)
copy = torch.cuda.Stream(device=dev)

def compute(k=12):
    x = a
    for _ in range(k):
        x = x @ a
    return x

def timed(fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); return (time.perf_counter() - t0) * 1e3

def do_copy():
    with torch.cuda.stream(copy):
        dst.copy_(host, non_blocking=True)

compute(2); do_copy(); torch.cuda.synchronize()
tc = timed(compute)
th = timed(do_copy)
tb = timed(lambda: (do_copy(), compute()))
print("compute %.1f ms, copy %.1f ms (%.1f GB/s), both %.1f ms (sum %.1f)" % (tc, th, n * 4 / th / 1e6, tb, tc + th), flush=True)
# the same with the copy issued from a second THREAD's stream and with hipMemcpyAsync in pieces of 64 MiB
def do_copy_pieces(p=1 << 24):
    with torch.cuda.stream(copy):
        for i in range(0, n, p):
            dst[i:i + p].copy_(host[i:i + p], non_blocking=True)
tb2 = timed(lambda: (do_copy_pieces(), compute()))
print("pieces of 64 MiB: both %.1f ms" % tb2, flush=True)

# the double-buffered pattern of bench.py's host-fed entry: step i computes on set i % 2 while set (i + 1) % 2 is uploaded
main = torch.cuda.current_stream(dev)
sets = [torch.empty(n, dtype=torch.float32, device=dev) for _ in range(2)]
def pattern(steps, gate):
    ready = None
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(steps):
        if ready is not None:
            main.wait_event(ready)
        if gate == 'wait_stream':
            copy.wait_stream(main)
        elif gate == 'event':
            ev0 = torch.cuda.Event(); ev0.record(main); copy.wait_event(ev0)
        with torch.cuda.stream(copy):
            sets[(i + 1) % 2].copy_(host, non_blocking=True)
            ready = torch.cuda.Event(); ready.record(copy)
        compute()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3 / steps
for gate in ('none', 'wait_stream', 'event'):
    print("pattern, gate=%s: %.1f ms per step" % (gate, pattern(6, gate)), flush=True)

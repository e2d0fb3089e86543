"""Reader throughput on GRID-sized records (one file per sample): python tools/reader_throughput.py [n] [batch]"""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import avsi_amd  # noqa: F401
from avsi_amd import tfrecord_io as tio
from avsi_amd.dataset_reader import DataManager

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 32
N, T = 48000, 250
root = tempfile.mkdtemp(prefix='avsi_rd_')
rng = np.random.default_rng(0)
files = []
for i in range(n):
    wav = np.round(rng.normal(0, 3000, N)).astype(np.float32)
    mask = np.ones((T, 257), np.float32)
    video = rng.normal(size=(T, 136)).astype(np.float32)
    rec = tio.serialize_sample_fixed(T, 20, wav, video, mask, np.zeros(50, np.float32), "clip_%05d" % i)
    f = os.path.join(root, "data_%05d.tfrecord" % (i + 1))
    tio.write_records(f, [rec])
    files.append(f)
dm = DataManager()
for label, kw in (("python parser, caller's thread", dict(native=False, prefetch=0)),
                  ("native, caller's thread", dict(prefetch=0)),
                  ("native + prefetch", dict()),
                  ("native + prefetch + upload", dict(device='cuda'))):
    if 'device' in kw:
        import torch
        if not torch.cuda.is_available():
            continue
    for rep in range(2):
        ds = dm.get_dataset(files, shuffle=True, seed=1)
        t0 = time.time()
        _, it = dm.get_iterator(ds, batch_size=batch, n_epochs=1, **kw)
        cnt = sum(len(b[0]) for b in it)
        dt = time.time() - t0
    print("%-34s %5d records in %.2f s: %.0f records/s, %.2f ms per record" % (label, cnt, dt, cnt / dt, dt / cnt * 1e3), flush=True)

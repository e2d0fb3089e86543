"""Records per second out of the TFRecord reader (one GRID-sized record per file, on tmpfs).
python tools/reader_throughput.py [n] [batch] [device: 0|1]"""
import os, sys, tempfile, time, shutil
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import avsi_amd  # noqa: F401
from avsi_amd import tfrecord_io as tio
from avsi_amd.dataset_reader import DataManager
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
mode = sys.argv[3] if len(sys.argv) > 3 else '1'     # 0: host only, 1: pinned arenas + upload, 2: pinned arenas, no upload
dev = mode != '0'
N, T = 48000, 250
base = tempfile.mkdtemp(prefix='avsi_rd_', dir='/dev/shm' if os.path.isdir('/dev/shm') else None)
rng = np.random.default_rng(0)
wav = np.round(rng.normal(0, 3000, N)).astype(np.float32)
mask = np.ones((T, 257), np.float32); mask[100:133] = 0
video = rng.normal(size=(T, 136)).astype(np.float32)
rec = tio.serialize_sample_fixed(T, 20, wav, video, mask, np.zeros(50, np.float32), "clip")
for i in range(n):
    tio.write_records(os.path.join(base, "data_%05d.tfrecord" % (i + 1)), [rec])
files = sorted(os.path.join(base, f) for f in os.listdir(base))
dm = DataManager()
for rep in range(3):
    _, it = dm.get_iterator(dm.get_dataset(files, shuffle=False), batch_size=batch, n_epochs=1,
                            device=torch.device('cuda', 0) if dev else None)
    if mode == '2':
        it.upload = lambda batch, arena=None: batch
    if mode == '3':         # time the uploader's call and the copies it queues
        plain = it.upload
        def timed(batch, arena=None):
            t0 = time.time(); out = plain(batch, arena); t1 = time.time(); out.ready.synchronize(); t2 = time.time()
            print("  upload call %.1f ms, copies done after %.1f ms more" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3), flush=True)
            return out
        it.upload = timed
    t0 = time.time(); k = 0
    for b in it:
        k += len(b[0])
    if dev:
        torch.cuda.synchronize()
    dt = time.time() - t0
    print("reader%s: %d records in %.2f s: %.0f records/s (batch %d, %s threads)" % (
        " + upload" if dev else "", k, dt, k / dt, batch, os.environ.get('AVSI_READER_THREADS', 'default')), flush=True)
shutil.rmtree(base, ignore_errors=True)

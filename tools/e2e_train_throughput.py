"""End-to-end trainer throughput on GRID-sized records: synthetic TFRecords (48000 samples, 250 frames, one file per
sample like the reference's datasets) -> training.train() -> utterances/s including reading, parsing and upload.
python tools/e2e_train_throughput.py [n_train] [batch] [model]"""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import avsi_amd  # noqa: F401
from avsi_amd import tfrecord_io as tio, training
from avsi_amd import audio_processing as ap

n_train = int(sys.argv[1]) if len(sys.argv) > 1 else 256
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 32
model = sys.argv[3] if len(sys.argv) > 3 else 'av-blstm'
N, T = 48000, 250
base = tempfile.mkdtemp(prefix='avsi_e2e_', dir='/dev/shm' if os.path.isdir('/dev/shm') else None)
rng = np.random.default_rng(0)


def make(root, n):
    os.makedirs(root)
    wavs = []
    for i in range(n):
        wav = np.round(rng.normal(0, 3000, N)).astype(np.float32)
        mask = np.ones((T, 257), np.float32)
        s = rng.integers(0, T - 33)
        mask[s:s + 33] = 0
        video = rng.normal(size=(T, 136)).astype(np.float32)
        labels = np.pad(rng.integers(0, 33, 20), (0, 30)).astype(np.float32)
        rec = tio.serialize_sample_fixed(T, 20, wav, video, mask, labels, "clip_%05d" % i)
        tio.write_records(os.path.join(root, "data_%05d.tfrecord" % (i + 1)), [rec])
        if i < 32:
            wavs.append(wav)
    np.save(os.path.join(root, "seq_lengths.npy"), np.full(n, T))
    return wavs


t0 = time.time()
wavs = make(os.path.join(base, "data", "training-set"), n_train)
make(os.path.join(base, "data", "validation-set"), batch)
print("dataset written in %.1f s" % (time.time() - t0), flush=True)
spec = ap.frontend(torch.from_numpy(np.stack(wavs)).cuda(), want_spec=True)['spec']
np.save(os.path.join(base, "mean.npy"), spec.mean(dim=(0, 1)).cpu().numpy().astype(np.float64))
np.save(os.path.join(base, "std.npy"), spec.std(dim=(0, 1), unbiased=False).cpu().numpy().astype(np.float64))
cfg = os.path.join(base, "train.config")
open(cfg, "w").write("\n".join([
    "model = %s" % model, "audio_feat_dim = 257", "video_feat_dim = 136", "audio_len = %d" % N, "batch_size = %d" % batch,
    "net_dim = [250, 250, 250]", "dropout_rate = 0.0", "max_n_epochs = 3", "n_earlystop_epochs = 5", "optimizer_type = adam",
    "starter_learning_rate = 0.001", "lr_decay = 1.0", "lr_updating_steps = 10000", "l2 = 0.0",
    "num_asr_labels = 33", "ctc_loss = 0.001",
    "root_folder = %s" % os.path.join(base, "data"), "exp_folder = %s" % os.path.join(base, "logs", "exp"), "device = /gpu:0",
    "audio_feat_mean = %s" % os.path.join(base, "mean.npy"), "audio_feat_std = %s" % os.path.join(base, "std.npy"), ""]))
import contextlib, io, re
t0 = time.time()
out = io.StringIO()
prof = None
if os.environ.get('AVSI_PROFILE'):
    import cProfile
    prof = cProfile.Profile()
with contextlib.redirect_stdout(out):
    m = prof.runcall(training.train, cfg) if prof else training.train(cfg)
if prof:
    import pstats
    st = io.StringIO(); pstats.Stats(prof, stream=st).sort_stats('tottime').print_stats(28); print(st.getvalue()[:6000])
dt = time.time() - t0
steps = m.global_step
epochs = [float(x) for x in re.findall(r"Epoch training time \(seconds\) = ([0-9.]+)", out.getvalue())]
per_epoch = steps // max(1, len(epochs))
print("inside an epoch (reading, parsing, upload, step, loss bookkeeping): " + ", ".join(
    "%.2f ms / step" % (e / per_epoch * 1e3) for e in epochs) + "  (%d steps per epoch)" % per_epoch, flush=True)
print("model %s: %d steps of %d in %.2f s (3 epochs incl. validation and checkpoints): %.1f ms / step, %.0f utterances/s end to end"
      % (model, steps, batch, dt, dt / steps * 1e3, steps * batch / dt))

"""cProfile of infer() end to end (oracle phase, then the default LWS phase).  python tools/e2e_infer_profile.py [n] [batch]
AVSI_E2E_PLAIN=1: no profiler -- two plain calls per phase path with AVSI_INFER_TIMING=1 (infer()'s own stage stamps on stderr)."""
import cProfile, contextlib, io, os, pstats, shutil, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import avsi_amd  # noqa: F401
from avsi_amd import tfrecord_io as tio, training, inference
from avsi_amd import audio_processing as ap
from avsi_amd.config_utils import check_trainconfiguration, load_configfile
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
N, T = 48000, 250
base = tempfile.mkdtemp(prefix='avsi_e2ep_', dir='/dev/shm' if os.path.isdir('/dev/shm') else None)
rng = np.random.default_rng(0)
root = os.path.join(base, "test-set"); os.makedirs(root)
wav = np.round(rng.normal(0, 3000, N)).astype(np.float32)
mask = np.ones((T, 257), np.float32); mask[100:133] = 0
video = rng.normal(size=(T, 136)).astype(np.float32)
t0 = time.time()
for i in range(n):
    tio.write_records(os.path.join(root, "data_%05d.tfrecord" % (i + 1)),
                      [tio.serialize_sample_fixed(T, 20, wav, video, mask, np.zeros(50, np.float32), "clip_%05d" % i)])
print("wrote %d records in %.1f s" % (n, time.time() - t0), flush=True)
net = os.path.join(base, "netmodel"); os.makedirs(net)
np.save(os.path.join(net, "audio_features_mean.npy"), np.zeros(257)); np.save(os.path.join(net, "audio_features_std.npy"), np.ones(257))
open(os.path.join(net, "config.txt"), "w").write("\n".join([
    "model = av-blstm", "audio_feat_dim = 257", "video_feat_dim = 136", "audio_len = %d" % N, "batch_size = %d" % batch,
    "net_dim = [250, 250, 250]", "dropout_rate = 0.0", "max_n_epochs = 1", "n_earlystop_epochs = 5", "optimizer_type = adam",
    "starter_learning_rate = 0.001", "lr_decay = 1.0", "lr_updating_steps = 10000", "l2 = 0.0",
    "root_folder = %s" % base, "exp_folder = %s" % base, "device = /gpu:0", "audio_feat_mean = x", "audio_feat_std = x", ""]))
config = check_trainconfiguration(load_configfile(os.path.join(net, "config.txt")))
m = training.build_model(config, np.zeros(257), np.ones(257), is_training=False)
m.variables.save(os.path.join(net, "sinet"))
plain = os.environ.get('AVSI_E2E_PLAIN') == '1'
for oracle_phase in (True, False):
    if plain:
        os.environ['AVSI_INFER_TIMING'] = '1'
        for rep in range(3):
            t0 = time.time()
            with contextlib.redirect_stdout(io.StringIO()):
                inference.infer(net, root, os.path.join(base, "plain%d_%d" % (oracle_phase, rep)), "enh", norm=True, oracle_phase=oracle_phase, batch_size=batch)
            dt = time.time() - t0
            print("infer(oracle_phase=%s) call %d: %d utterances in %.3f s: %.0f utterances/s (batch %d)" % (oracle_phase, rep, n, dt, n / dt, batch), flush=True)
            shutil.rmtree(os.path.join(base, "plain%d_%d" % (oracle_phase, rep)), ignore_errors=True)
        continue
    with contextlib.redirect_stdout(io.StringIO()):
        inference.infer(net, root, os.path.join(base, "warm%d" % oracle_phase), "enh", norm=True, oracle_phase=oracle_phase, batch_size=batch)
    pr = cProfile.Profile()
    t0 = time.time()
    with contextlib.redirect_stdout(io.StringIO()):
        pr.runcall(inference.infer, net, root, os.path.join(base, "audio%d" % oracle_phase), "enh", norm=True, oracle_phase=oracle_phase, batch_size=batch)
    dt = time.time() - t0
    print("infer(oracle_phase=%s): %d utterances in %.2f s: %.0f utterances/s (batch %d)" % (oracle_phase, n, dt, n / dt, batch), flush=True)
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(14); print(s.getvalue()[:3500], flush=True)
import shutil; shutil.rmtree(base, ignore_errors=True)

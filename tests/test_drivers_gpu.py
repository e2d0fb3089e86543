"""End-to-end drivers on the GPU: synthetic TFRecord dataset -> train() -> checkpoint -> infer() -> WAVs."""
import os

import numpy as np
import pytest
from scipy.io import wavfile

pytestmark = pytest.mark.gpu

N, T = 3840, 20


def _make_dataset(root, n, seed, embeddings=False):
    import avsi_amd  # noqa: F401
    from avsi_amd import tfrecord_io as tio
    os.makedirs(root, exist_ok=True)
    rng = np.random.default_rng(seed)
    for i in range(n):
        t = np.arange(N)
        wav = np.round(3000 * np.sin(2 * np.pi * (200 + 40 * i) * t / 16000) + rng.normal(0, 300, N)).astype(np.float32)
        mask = np.ones((T, 257), dtype=np.float32)
        s = rng.integers(2, T - 6)
        mask[s:s + 4] = 0
        video = rng.normal(size=(T, 136)).astype(np.float32)
        emb = rng.normal(size=512).astype(np.float32) if embeddings else None
        rec = tio.serialize_sample_fixed(T, 3, wav, video, mask, np.zeros(50), "clip_%03d" % i, embedding=emb)
        tio.write_records(os.path.join(root, "data_%05d.tfrecord" % (i + 1)), [rec])
    np.save(os.path.join(root, "seq_lengths.npy"), np.full(n, T))


@pytest.fixture(scope="module")
def experiment(tmp_path_factory):
    import avsi_amd  # noqa: F401
    from avsi_amd import audio_processing as ap
    import torch
    base = tmp_path_factory.mktemp("exp")
    data = str(base / "tfrecords")
    _make_dataset(os.path.join(data, "training-set"), 12, 0)
    _make_dataset(os.path.join(data, "validation-set"), 4, 1)
    _make_dataset(os.path.join(data, "test-set"), 5, 2)
    # normalisation statistics of the training clips (the audio_preprocessing step)
    from avsi_amd import tfrecord_io as tio
    wavs = []
    for f in sorted(os.listdir(os.path.join(data, "training-set"))):
        if f.endswith(".tfrecord"):
            ctx, _ = tio.decode_sequence_example(next(tio.read_records(os.path.join(data, "training-set", f))))
            wavs.append(ctx['target_audio_wav'])
    spec = ap.frontend(torch.from_numpy(np.stack(wavs)).cuda(), want_spec=True)['spec']
    np.save(str(base / "mean.npy"), spec.mean(dim=(0, 1)).cpu().numpy().astype(np.float64))
    np.save(str(base / "std.npy"), spec.std(dim=(0, 1), unbiased=False).cpu().numpy().astype(np.float64))
    cfg = base / "train.config"
    cfg.write_text("\n".join([
        "### synthetic experiment", "model = av-blstm", "audio_feat_dim = 257", "video_feat_dim = 136",
        "audio_len = %d" % N, "batch_size = 4", "net_dim = [250, 250, 250]", "dropout_rate = 0.0",
        "max_n_epochs = 3", "n_earlystop_epochs = 5", "optimizer_type = adam", "starter_learning_rate = 0.001",
        "lr_decay = 1.0", "lr_updating_steps = 10000", "l2 = 0.0", "root_folder = %s" % data,
        "exp_folder = %s" % (base / "logs" / "av_exp0"), "device = /gpu:0",
        "audio_feat_mean = %s" % (base / "mean.npy"), "audio_feat_std = %s" % (base / "std.npy"), ""]))
    return base, data, str(cfg)


def test_train_writes_reference_directory_contract(experiment, capsys):
    from avsi_amd import training
    base, data, cfg = experiment
    model = training.train(cfg)
    out = capsys.readouterr().out
    exp = base / "logs" / "av_exp0"
    net = exp / "netmodel"
    for f in ("config.txt", "audio_features_mean.npy", "audio_features_std.npy", "sinet.npz"):
        assert (net / f).is_file(), f
    log = (exp / "training_log.txt").read_text().splitlines()
    assert log[0] == "+-- EXPERIMENT NAME - av_exp0 --+"
    assert "## Model type: av-blstm" in log
    assert "Epoch\tLR\tTraining loss\tValidation loss\t[TIME]" in log
    rows = [l for l in log if l[:1].isdigit()]
    assert len(rows) == 3 and rows[0].split("\t")[0] == "1" and "|" in rows[0].split("\t")[2]
    losses = [float(r.split("\t")[2].split("|")[0]) for r in rows]
    assert losses[-1] < losses[0]                         # it learns something in 9 steps
    assert "+---- Done training: epoch limit reached ----+" in out
    assert "Step[      1] Loss[" in out
    assert model.global_step == 9


def test_resume_from_checkpoint_restores_variables(experiment):
    import torch
    from avsi_amd import models, training
    from avsi_amd.config_utils import check_trainconfiguration, load_configfile
    base, data, cfg = experiment
    net = base / "logs" / "av_exp0" / "netmodel"
    config = check_trainconfiguration(load_configfile(str(net / "config.txt")))
    m = training.build_model(config, np.load(str(net / "audio_features_mean.npy")), np.load(str(net / "audio_features_std.npy")))
    before = m.variables.flat.clone()
    m.variables.restore(str(net / "sinet"))
    assert not torch.equal(before, m.variables.flat)
    assert m.variables.global_step in (3, 6, 9)
    assert m.variables.adam_m is not None
    with pytest.raises(ValueError):
        m.variables.restore(str(net / "config.txt"))


@pytest.mark.parametrize("oracle_phase", [True, False])
def test_infer_writes_int16_wavs(experiment, oracle_phase, capsys):
    from avsi_amd import inference
    base, data, cfg = experiment
    net = base / "logs" / "av_exp0" / "netmodel"
    audio_out = base / ("audio_%d" % oracle_phase)
    loss = inference.infer(str(net), os.path.join(data, "test-set"), str(audio_out), "av_exp0", norm=True,
                           oracle_phase=oracle_phase, batch_size=2)
    out = capsys.readouterr().out
    assert "Written 2 enhanced wavs. Total samples written so far 2." in out
    assert "Written 1 enhanced wavs. Total samples written so far 5." in out
    assert "Loss hole:" in out and np.isfinite(loss)
    for i in range(5):
        rate, wav = wavfile.read(str(audio_out / ("clip_%03d" % i) / "enhanced" / "av_exp0.wav"))
        assert rate == 16000 and wav.dtype == np.int16 and wav.shape == (T * 192,)
        assert np.abs(wav).max() > 0


def test_infer_default_phase_is_lws_refined(experiment, monkeypatch):
    """oracle_phase=False (the reference's default, inference.py:141-154): every batch goes through LWS phase
    reconstruction, and what is written is its output."""
    from avsi_amd import inference
    from avsi_amd import lws as lws_mod
    base, data, cfg = experiment
    net = base / "logs" / "av_exp0" / "netmodel"
    seen = []
    orig = lws_mod.lws.refine_enhanced

    def spy(self, enhanced, masks, num_samples=None, **kw):
        out = orig(self, enhanced, masks, num_samples, **kw)
        seen.append((enhanced.cpu().numpy(), out.cpu().numpy()))
        return out
    monkeypatch.setattr(lws_mod.lws, "refine_enhanced", spy)
    audio_out = base / "audio_lws"
    inference.infer(str(net), os.path.join(data, "test-set"), str(audio_out), "p", norm=True, oracle_phase=False, batch_size=5)
    assert len(seen) == 1 and seen[0][0].shape == (5, N)
    before, after = seen[0]
    assert after.shape == before.shape and not np.allclose(before, after)
    _, wav = wavfile.read(str(audio_out / "clip_000" / "enhanced" / "p.wav"))
    assert np.array_equal(wav, after[0][: T * 192].astype(np.int16))


def test_infer_collects_batches_for_one_lws_launch(experiment, monkeypatch):
    """infer() refines the phase of several small batches in ONE LWS launch (AVSI_LWS_GROUP utterances): the files are
    the same, sample for sample, as with one launch per batch, and the per-batch lines are still printed in order."""
    from avsi_amd import inference
    from avsi_amd import lws as lws_mod
    base, data, cfg = experiment
    net = base / "logs" / "av_exp0" / "netmodel"
    calls = []
    orig = lws_mod.lws.refine_enhanced

    def spy(self, enhanced, masks, num_samples=None, **kw):
        calls.append(int(enhanced.shape[0]))
        return orig(self, enhanced, masks, num_samples, **kw)
    monkeypatch.setattr(lws_mod.lws, "refine_enhanced", spy)
    monkeypatch.setenv('AVSI_INFER_COALESCE', '0')         # one model step per reader batch: the LWS grouping alone
    outs = {}
    for group in (1, 4, 128):
        monkeypatch.setenv('AVSI_LWS_GROUP', str(group))
        calls.clear()
        audio_out = base / ("audio_group%d" % group)
        inference.infer(str(net), os.path.join(data, "test-set"), str(audio_out), "g", norm=True, oracle_phase=False, batch_size=2)
        assert calls == {1: [2, 2, 1], 4: [4, 1], 128: [5]}[group]
        outs[group] = [wavfile.read(str(audio_out / ("clip_%03d" % i) / "enhanced" / "g.wav"))[1] for i in range(5)]
    for group in (4, 128):
        for a, b in zip(outs[1], outs[group]):
            assert np.array_equal(a, b)


@pytest.mark.parametrize("oracle_phase", [True, False])
def test_infer_runs_several_reader_batches_in_one_model_step(experiment, monkeypatch, capsys, oracle_phase):
    """infer() coalesces the model step of several reader batches (AVSI_INFER_COALESCE utterances, round 6): utterances are
    independent, so the files equal those of one step per batch (to the last bit of the int16 samples or one LSB: the
    recurrence may run on another kernel family at the larger batch), the per-batch lines are printed as before, and the
    mean over the reader batches' losses is the same number."""
    from avsi_amd import inference, models
    base, data, cfg = experiment
    net = base / "logs" / "av_exp0" / "netmodel"
    steps = []
    orig = models.StackedBLSTMModel.feed

    def spy(self, **kw):
        if kw.get('sequence_lengths') is not None:
            steps.append(len(kw['sequence_lengths']))
        return orig(self, **kw)
    monkeypatch.setattr(models.StackedBLSTMModel, "feed", spy)
    res = {}
    for coalesce in (0, 4, 1024):
        monkeypatch.setenv('AVSI_INFER_COALESCE', str(coalesce))
        steps.clear()
        audio_out = base / ("audio_co%d_%d" % (coalesce, oracle_phase))
        loss = inference.infer(str(net), os.path.join(data, "test-set"), str(audio_out), "c", norm=True,
                               oracle_phase=oracle_phase, batch_size=2)
        out = capsys.readouterr().out
        assert steps == {0: [2, 2, 1], 4: [4, 1], 1024: [5]}[coalesce]
        assert "Written 2 enhanced wavs. Total samples written so far 4." in out
        assert "Written 1 enhanced wavs. Total samples written so far 5." in out
        res[coalesce] = (loss, [wavfile.read(str(audio_out / ("clip_%03d" % i) / "enhanced" / "c.wav"))[1] for i in range(5)])
    for coalesce in (4, 1024):
        assert abs(res[coalesce][0] - res[0][0]) <= 2e-5 * abs(res[0][0])
        for a, b in zip(res[0][1], res[coalesce][1]):
            assert np.abs(a.astype(np.int32) - b.astype(np.int32)).max() <= 1


@pytest.mark.parametrize("model_name,fmt", [("a-blstm-emb", "npz"), ("av-blstm-ssnn", "tf"), ("av-blstm-twosteps", "tf"),
                                            ("av-blstm-twosteps", "npz")])
def test_variant_models_train_and_infer(experiment, tmp_path, model_name, fmt, capsys):
    """The model variants through the same drivers: TFRecords (with the embedding context feature
    for *-emb) -> train() -> checkpoint in either format -> infer() -> WAVs."""
    import torch
    from avsi_amd import inference, training
    base, data0, cfg0 = experiment
    data = str(tmp_path / "tfrecords")
    emb = model_name.endswith('-emb')
    _make_dataset(os.path.join(data, "training-set"), 8, 0, emb)
    _make_dataset(os.path.join(data, "validation-set"), 4, 1, emb)
    _make_dataset(os.path.join(data, "test-set"), 3, 2, emb)
    exp = tmp_path / "logs" / "exp"
    text = open(cfg0).read().replace("model = av-blstm", "model = %s" % model_name)
    text = text.replace("root_folder = %s" % data0, "root_folder = %s" % data)
    text = text.replace("exp_folder = %s" % (base / "logs" / "av_exp0"), "exp_folder = %s" % exp)
    text = text.replace("max_n_epochs = 3", "max_n_epochs = 2") + "integration_layer = 1\n"
    cfg = tmp_path / "v.config"
    cfg.write_text(text)
    model = training.train(str(cfg), checkpoint_format=fmt)
    assert model.global_step == 4
    net = exp / "netmodel"
    if fmt == 'tf':
        from avsi_amd import tf_checkpoint as tc
        names = [n for n, _, _ in tc.list_variables(str(net / "sinet"))]
        if model_name == 'av-blstm-ssnn':
            assert 'av-blstm-ssnn/speaker_embedding/weights_1' in names
            assert any('/blstm_2/cudnn_lstm/stack_bidirectional_rnn/cell_1/' in n for n in names)
            assert any('/blstm_1/cudnn_lstm/stack_bidirectional_rnn/cell_0/' in n for n in names)
        else:
            assert any(n.startswith('v-blstm/cudnn_lstm/') for n in names)
            assert any(n.startswith('av-blstm-twosteps/cudnn_lstm/') and n.endswith('/Adam_1') for n in names)
    else:
        assert (net / "sinet.npz").is_file()
        assert (net / "sinet.vnet.npz").is_file() == (model_name == 'av-blstm-twosteps')
    audio_out = tmp_path / "audio"
    loss = inference.infer(str(net), os.path.join(data, "test-set"), str(audio_out), "exp", norm=True,
                           oracle_phase=False, batch_size=2)
    assert np.isfinite(loss)
    for i in range(3):
        rate, wav = wavfile.read(str(audio_out / ("clip_%03d" % i) / "enhanced" / "exp.wav"))
        assert rate == 16000 and wav.dtype == np.int16 and wav.shape == (T * 192,) and np.abs(wav).max() > 0
    # the restored model is the trained one
    from avsi_amd.config_utils import check_trainconfiguration, load_configfile
    config = check_trainconfiguration(load_configfile(str(net / "config.txt")))
    m2 = training.build_model(config, np.load(str(net / "audio_features_mean.npy")),
                              np.load(str(net / "audio_features_std.npy")), is_training=False)
    m2.variables.restore(str(net / "sinet"))
    assert m2.variables.global_step in (2, 4)
    if model_name == 'av-blstm-twosteps':
        assert torch.equal(m2.video_variables.flat, model.video_variables.flat)


def test_ctc_multitask_model_trains_and_infers(experiment, tmp_path, capsys):
    """The multi-task model of the shipped blstm_ctc.config through the drivers (reference
    training_ctc.py): three-part loss and PER columns in the console and the log, model selection on the
    inpainting loss, TF checkpoint with both heads, inference without labels."""
    from avsi_amd import inference, training
    from avsi_amd import tf_checkpoint as tc
    base, data, cfg0 = experiment
    exp = tmp_path / "logs" / "ctc_exp"
    text = open(cfg0).read().replace("model = av-blstm", "model = v-blstm-ssnn-ctc")
    text = text.replace("exp_folder = %s" % (base / "logs" / "av_exp0"), "exp_folder = %s" % exp)
    text = text.replace("max_n_epochs = 3", "max_n_epochs = 2") + "num_asr_labels = 33\nctc_loss = 0.001\n"
    cfg = tmp_path / "ctc.config"
    cfg.write_text(text)
    model = training.train(str(cfg), checkpoint_format='tf')
    out = capsys.readouterr().out
    assert model.global_step == 6 and model.num_classes == 34
    assert "## CTC-loss coefficient: 0.001000" in out
    step = [l for l in out.splitlines() if l.startswith("Step[      1] Loss[")][0]
    assert step.count("|") == 2 and "PER[" in step
    assert "; PER: " in out and "Validation loss: " in out
    log = (exp / "training_log.txt").read_text().splitlines()
    assert "Epoch\tLR\tTraining loss\tTraining PER \tValidation loss\tValidation PER[TIME]" in log
    rows = [l.split("\t") for l in log if l[:1].isdigit()]
    assert len(rows) == 2 and len(rows[0]) == 7 and rows[0][2].count("|") == 2 and rows[0][4].count("|") == 2
    total, ipt, ctc = (float(v) for v in rows[0][2].split("|"))
    # (the reference's running means floor-divide value * frames by the feature size, so the three
    # columns are only approximately consistent with each other)
    assert ctc > 0 and ipt > 0 and abs(total - (ipt + 0.001 * ctc)) < 0.1 * total
    names = [n for n, _, _ in tc.list_variables(str(exp / "netmodel" / "sinet"))]
    for want in ("v-blstm-ssnn-ctc/inpainting/weights", "v-blstm-ssnn-ctc/asr/weights", "v-blstm-ssnn-ctc/asr/biases",
                 "v-blstm-ssnn-ctc/speaker_embedding/weights_3", "v-blstm-ssnn-ctc/asr/weights/Adam"):
        assert want in names, want
    audio_out = tmp_path / "audio"
    loss = inference.infer(str(exp / "netmodel"), os.path.join(data, "test-set"), str(audio_out), "ctc", norm=True,
                           oracle_phase=False, batch_size=2)
    assert np.isfinite(loss)
    rate, wav = wavfile.read(str(audio_out / "clip_000" / "enhanced" / "ctc.wav"))
    assert rate == 16000 and wav.dtype == np.int16 and wav.shape == (T * 192,)


@pytest.mark.parametrize("batch_size,macro_records,batches", [(4, 128, 3), (5, 10, 3), (5, 0, 3), (1, 5, 12)])
def test_reader_uploads_batches_from_its_prefetch_thread(experiment, monkeypatch, batch_size, macro_records, batches):
    """get_iterator(device=...): the bulky fields arrive as device tensors equal to the numpy ones, ordered by
    an event instead of a host synchronisation; small fields stay on the host.  Small batches are read several at a time
    (AVSI_READER_MACRO records per read, round 6) and handed out as views of one upload: 12 files as 3 x 4 in one read,
    as 5 + 5 | 2 (a short last batch, alone in the second read), one batch per read, and 12 single records in reads of 5."""
    import torch
    monkeypatch.setenv('AVSI_READER_MACRO', str(macro_records))
    from avsi_amd.dataset_reader import DataManager
    from avsi_amd.training import unpack_batch
    base, data, cfg = experiment
    files = sorted(os.path.join(data, "training-set", f) for f in os.listdir(os.path.join(data, "training-set"))
                   if f.endswith(".tfrecord"))
    dm = DataManager(num_audio_samples=N, audio_feat_size=257, video_feat_size=136)
    _, it = dm.get_iterator(dm.get_dataset(files, shuffle=False), batch_size=batch_size, n_epochs=1, device='cuda')
    assert it.macro == max(1, macro_records // batch_size)
    _, ref = dm.get_iterator(dm.get_dataset(files, shuffle=False), batch_size=batch_size, n_epochs=1)
    n = 0
    for b, r in zip(it, ref):
        n += 1
        assert sorted(b.device_arrays) == [2, 5, 6]           # audio, video, mask
        feed, paths = unpack_batch(b, False)
        assert feed['target_sources'].is_cuda and feed['masks'].is_cuda and isinstance(feed['sequence_lengths'], np.ndarray)
        torch.cuda.current_stream().synchronize()
        np.testing.assert_array_equal(feed['target_sources'].cpu().numpy(), r[2])
        np.testing.assert_array_equal(feed['video_features'].cpu().numpy(), r[5])
        np.testing.assert_array_equal(feed['masks'].cpu().numpy(), r[6])
        # the host views of the uploaded fields pointed into two page-locked arenas the reader recycles: the tuple holds
        # None there, consumers use the device copies; the small fields are copies
        assert b[2] is None and b[5] is None and b[6] is None
        np.testing.assert_array_equal(b[0], r[0])
        assert list(paths) == list(r[3])
    assert n == batches


def _launch_dp_training(cfg, ranks, extra_env=None, timeout=1200):
    import socket
    import subprocess
    import sys
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, AVSI_DIST_BACKEND='gloo', HSA_ENABLE_IPC_MODE_LEGACY='0', AVSI_COOP_CUS=str(256 // ranks // 8 * 8))
    env.update(extra_env or {})
    here = os.path.dirname(os.path.abspath(__file__))
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(ranks), '--master-addr',
           '127.0.0.1', '--master-port', str(port), os.path.join(here, 'dp_train_worker.py'), str(cfg)]
    return subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout)


def _rank_lines(stdout, ranks):
    import re
    found = {int(m.group(1)): (int(m.group(2)), int(m.group(3)), m.group(4))
             for m in re.finditer(r'RANK (\d+) STEPS (\d+) FALLBACKS (\d+) VARS ([0-9a-f]+)', stdout)}
    assert sorted(found) == list(range(ranks)), stdout[-2000:]
    return found


@pytest.mark.parametrize("ranks,batch_size,steps", [(2, 4, 2), (4, 2, 2), (3, 1, 8)])
def test_data_parallel_training_through_the_driver(experiment, tmp_path, ranks, batch_size, steps):
    """train() under torch.distributed.run with 2, 4 and 3 ranks (sharing this box's GPU: gloo process group; the box
    allows at most 6 processes on its card -- this one, the launcher and 4 ranks -- so world 8 itself is rehearsed on the
    CPU: tests/test_parallel_gloo.py, test_tfrecord_io.py::test_even_rounds_with_eight_readers): batches are dealt in whole rounds, every rank takes the
    same number of steps, gradients are all-reduced inside train_op, every rank ends with the same bits, rank 0 alone
    writes the log and the checkpoints.
    12 training samples: 2 ranks x batches of 4 = 3 batches = one whole round + one left over (dropped): one step per
    epoch; 4 ranks x batches of 2 = 6 batches = one round + two left over: one step per epoch -- and the 2 validation
    batches leave ranks 2 and 3 without one; 3 ranks x batches of 1 = four rounds: four steps per epoch."""
    base, data, cfg0 = experiment
    exp = tmp_path / "logs" / "dp_exp"
    text = open(cfg0).read().replace("exp_folder = %s" % (base / "logs" / "av_exp0"), "exp_folder = %s" % exp)
    text = text.replace("max_n_epochs = 3", "max_n_epochs = 2").replace("batch_size = 4", "batch_size = %d" % batch_size)
    cfg = tmp_path / "dp.config"
    cfg.write_text(text)
    run = _launch_dp_training(cfg, ranks)
    assert run.returncode == 0, run.stderr[-3000:]
    found = _rank_lines(run.stdout, ranks)
    assert all(v[0] == steps and v[1] == 0 for v in found.values()), found
    assert len({v[2] for v in found.values()}) == 1, found             # the same variables, bit for bit, on every rank
    log = (exp / "training_log.txt").read_text().splitlines()
    assert len([l for l in log if l[:1].isdigit()]) == 2 and log[0] == "+-- EXPERIMENT NAME - dp_exp --+"
    assert (exp / "netmodel" / "sinet.npz").is_file()
    assert run.stdout.count('+---- Done training: epoch limit reached ----+') == 1      # rank 0 only


def test_cooperative_timeout_on_one_rank_falls_back_on_every_rank(experiment, tmp_path):
    """Rank 1 parks 16 CUs of the shared GPU from its second batch on: 32-way cooperative launches give up their bounded
    wait, the guard words summed inside the last gradient bucket void that step's update on BOTH ranks, both fall back to
    the neighbour-tolerant cooperative kernels at the same step, repeat the skipped batches and finish the run with
    identical variables and the full number of steps."""
    base, data, cfg0 = experiment
    exp = tmp_path / "logs" / "dp_park"
    text = open(cfg0).read().replace("exp_folder = %s" % (base / "logs" / "av_exp0"), "exp_folder = %s" % exp)
    text = text.replace("max_n_epochs = 3", "max_n_epochs = 2").replace("batch_size = 4", "batch_size = 2")
    cfg = tmp_path / "dp_park.config"
    cfg.write_text(text)
    run = _launch_dp_training(cfg, 2, {'AVSI_TEST_PARK_RANK': '1', 'AVSI_COOP_CUS': '256'})
    assert run.returncode == 0, run.stderr[-3000:]
    found = _rank_lines(run.stdout, 2)
    # 12 samples in batches of 2 over 2 ranks: 3 rounds per epoch, 2 epochs
    assert all(v[0] == 6 and v[1] >= 1 for v in found.values()), (found, run.stderr[-2000:])
    assert found[0][1] == found[1][1] and found[0][2] == found[1][2]
    assert run.stderr.count('falling back to the cooperative kernels that tolerate neighbours') == 2


def test_cooperative_timeout_of_a_peer_is_repeated_by_a_rank_that_is_already_batch_stationary(experiment, tmp_path):
    """The ranks' fall-back levels may differ (validate() lets a rank fall back alone).  Rank 1 starts on the
    batch-stationary kernels (level 2), rank 0 parks 16 CUs from its second batch on and times out: the summed guard word
    voids the step on both; rank 1 must rewind and repeat it with rank 0 -- not raise and strand rank 0 in the next
    all-reduce -- and both finish every step with identical variables."""
    base, data, cfg0 = experiment
    exp = tmp_path / "logs" / "dp_levels"
    text = open(cfg0).read().replace("exp_folder = %s" % (base / "logs" / "av_exp0"), "exp_folder = %s" % exp)
    text = text.replace("max_n_epochs = 3", "max_n_epochs = 2").replace("batch_size = 4", "batch_size = 2")
    cfg = tmp_path / "dp_levels.config"
    cfg.write_text(text)
    run = _launch_dp_training(cfg, 2, {'AVSI_TEST_PARK_RANK': '0', 'AVSI_TEST_LEVEL2_RANK': '1', 'AVSI_COOP_CUS': '256'})
    assert run.returncode == 0, run.stderr[-3000:]
    found = _rank_lines(run.stdout, 2)
    assert all(v[0] == 6 for v in found.values()), (found, run.stderr[-2000:])
    assert found[0][1] >= 1 and found[1][1] == 0, found          # rank 1 had nowhere to fall: it only repeated
    assert found[0][2] == found[1][2]


def test_non_finite_loss_on_one_rank_stops_every_rank(experiment, tmp_path):
    """The NaN / Inf abort of the trainer (training_emb.py:244-249, exit code 1) under data parallelism: the verdict
    travels in the last gradient all-reduce bucket (model.nonfinite_flag), so the rank whose loss is fine leaves at the
    same step as the rank whose loss is NaN -- nobody is left waiting in the next collective."""
    base, data, cfg0 = experiment
    exp = tmp_path / "logs" / "dp_nan"
    text = open(cfg0).read().replace("exp_folder = %s" % (base / "logs" / "av_exp0"), "exp_folder = %s" % exp)
    cfg = tmp_path / "dp_nan.config"
    cfg.write_text(text)
    run = _launch_dp_training(cfg, 2, {'AVSI_TEST_NAN_RANK': '1'})
    assert run.returncode != 0
    assert 'GOT INSTABILITY: loss is NaN. Leaving...' in run.stdout
    assert 'GOT INSTABILITY on another rank: loss is not finite there. Leaving...' in run.stdout
    assert (tmp_path / "exit_rank0").read_text() == "1" and (tmp_path / "exit_rank1").read_text() == "1"


def test_training_survives_two_fall_backs_in_a_row_and_restores_the_polls(experiment, tmp_path, monkeypatch, capsys):
    """A step whose guard reports a cooperative timeout, and whose repetition one level down reports one AGAIN (level 1
    kernels can still time out), ends on the batch-stationary kernels: train() goes on, counts every batch once, and
    leaves ops.COOP_POLL_RAISES as it found it -- also when the loop is left by the NaN abort's sys.exit.
    (The guard words are faked on the host here: the device did apply the updates, so only the counts are checked.)"""
    from avsi_amd import ops, training
    base, data, cfg0 = experiment
    exp = tmp_path / "logs" / "twice"
    text = open(cfg0).read().replace("exp_folder = %s" % (base / "logs" / "av_exp0"), "exp_folder = %s" % exp)
    cfg = tmp_path / "twice.config"
    cfg.write_text(text.replace("max_n_epochs = 3", "max_n_epochs = 1"))
    ops.coop_fall_back_reset()
    plain_get = training._LateScalars.get
    state = {'n': 0}

    def faulty(handle):
        vals = plain_get(handle)
        state['n'] += 1
        if state['n'] in (2, 3):            # the second step, and its first repetition
            vals[-1] = 1.0
        return vals
    monkeypatch.setattr(training._LateScalars, 'get', staticmethod(faulty))
    try:
        model = training.train(str(cfg))
        assert model.coop_fallbacks == 2 and ops.coop_level() == 2
        assert model.global_step == 3                      # 12 clips in batches of 4, one epoch: no batch lost, none twice
        assert ops.COOP_POLL_RAISES is True
        err = capsys.readouterr().err
        assert 'the cooperative kernels that tolerate neighbours' in err and 'the batch-stationary recurrent kernels' in err
        # a third timeout at level 2 is not a residency matter any more: it surfaces
        state['n'] = -100
        monkeypatch.setattr(training._LateScalars, 'get', staticmethod(lambda h: plain_get(h)[:-1] + [1.0]))
        with pytest.raises(ops.CoopTimeout):
            training.train(str(cfg))
        assert ops.COOP_POLL_RAISES is True
    finally:
        ops.coop_fall_back_reset()

"""Inference driver on MI355X: TFRecords in, inpainted ``enhanced/<prefix>.wav`` files out.

``infer`` keeps the signature, directory contract and console output of the reference
(``av_speech_inpainting/inference.py:20-170``): the model directory holds ``config.txt``,
``audio_features_{mean,std}.npy`` and the ``sinet`` checkpoint; each utterance is written to
``<audio_path>/<sample_path>/enhanced/<out_file_prefix>.wav`` as 16 kHz int16 truncated to
``seq_len * 192`` samples.

Phase: ``oracle_phase=True`` uses the target phase everywhere; otherwise the phase of the masked
target STFT (zero inside gaps), whose gap frames are then refined by LWS phase reconstruction exactly
as the reference does with the ``lws`` package (inference.py:119,141-154) -- here ``avsi_amd.lws``,
the gfx950 implementation of the published algorithm (see that module: unpinned against the package).
"""
import os
import queue
import sys
import threading
from glob import glob

import numpy as np
from scipy.io import wavfile

from . import ops, parallel
from . import lws as lws_mod
from .config_utils import check_trainconfiguration, load_configfile
from .dataset_reader import DataManager, OutOfRangeError
from .training import EMBEDDING_SIZE, build_model, unpack_batch, uses_embeddings


class _WavWriter(object):
    """Writes the enhanced utterances of a finished batch: ``<audio_path>/<sample>/enhanced/<prefix>.wav``, 16 kHz int16,
    ``seq_len * 192`` samples, as the reference's loop does.  In line by default (80 us per file; the reader thread, not
    this, bounds the driver); ``AVSI_WAV_THREADS=n`` hands the files to n worker threads instead -- worth it on slow
    storage only: on a local disk four threads were slower (2.4 k against 2.7 k utterances/s, they compete with the
    reader thread for the interpreter)."""

    def __init__(self, audio_path, prefix, threads):
        self.audio_path, self.prefix = audio_path, prefix
        self.jobs = queue.Queue(maxsize=4 * max(1, threads))
        self.errors = []
        self.workers = [threading.Thread(target=self._run, daemon=True) for _ in range(max(0, threads))]
        for w in self.workers:
            w.start()

    def _write(self, wav, sample_dir, seq_len):
        out_dir = os.path.join(self.audio_path, sample_dir.decode(), 'enhanced')
        os.makedirs(out_dir, exist_ok=True)
        wavfile.write(os.path.join(out_dir, self.prefix + '.wav'), 16000, wav[: int(seq_len) * 192].astype(np.int16))

    def _run(self):
        while True:
            job = self.jobs.get()
            if job is None:
                return
            try:
                self._write(*job)
            except Exception as e:      # surfaces in close(): a failed write must not pass silently
                self.errors.append(e)

    def submit(self, wavs, sample_dirs, seq_lens):
        for job in zip(wavs, sample_dirs, seq_lens):
            if self.workers:
                self.jobs.put(job)
            else:
                self._write(*job)

    def close(self):
        for _ in self.workers:
            self.jobs.put(None)
        for w in self.workers:
            w.join()
        if self.errors:
            raise self.errors[0]


def infer(model_path, data_path_test, audio_path, out_file_prefix, norm=True, oracle_phase=False, batch_size=1):
    config = check_trainconfiguration(load_configfile(os.path.join(model_path, 'config.txt')))
    rank, world = parallel.init()

    dm = DataManager(num_audio_samples=config['audio_len'], audio_feat_size=config['audio_feat_dim'],
                     video_feat_size=config['video_feat_dim'], buffer_size=4000, mode='fixed',
                     embedding_size=EMBEDDING_SIZE if uses_embeddings(config) else None)
    test_files = sorted(glob(os.path.join(data_path_test, '*.tfrecord')))
    import torch
    _, test_it = dm.get_iterator(dm.get_dataset(test_files, shuffle=False), batch_size=batch_size, n_epochs=1,
                                 drop_remainder=False, shard=(rank, world),
                                 device=torch.device('cuda', torch.cuda.current_device()))

    if norm:
        audio_feat_mean = np.load(os.path.join(model_path, 'audio_features_mean.npy'))
        audio_feat_std = np.load(os.path.join(model_path, 'audio_features_std.npy'))
    else:
        audio_feat_mean = np.zeros(config['audio_feat_dim'])
        audio_feat_std = np.ones(config['audio_feat_dim'])

    print('Building speech inpainting inference model:')
    model = build_model(config, audio_feat_mean, audio_feat_std, is_training=False)
    print('Model building done.')
    print('done.\n')
    print('Restore weigths:')
    try:
        model.variables.restore(os.path.join(model_path, 'sinet'))
    except ValueError as e:
        print(str(e))
        sys.exit(2)
    print('done.\n')

    # LWS module initialization (reference inference.py:119)
    lws_processor = lws_mod.lws(384, 192, fftsize=512, mode='speech')

    total_samples = 0
    loss_list = []
    writer = _WavWriter(audio_path, out_file_prefix, int(os.environ.get('AVSI_WAV_THREADS', '0')))
    # The LWS kernel is a pipeline of ~100 sweeps: a batch of 32 takes it 16 ms, four such batches together 27 ms
    # (DESIGN 4.3d).  Batches are therefore collected until LWS_GROUP utterances wait for their phase, refined in one
    # launch and written in the order they came; the per-batch lines below are printed when their files are queued.
    lws_group = int(os.environ.get('AVSI_LWS_GROUP', '128'))
    pending = []                    # (enhanced, masks, paths, lengths) of batches whose phase is still to be refined

    def written(paths, lengths):
        nonlocal total_samples
        total_samples += len(lengths)
        print('Written {:d} enhanced wavs. Total samples written so far {:d}.'.format(len(lengths), total_samples))

    def flush():
        if not pending:
            return
        if len(pending) == 1:
            wavs, masks = pending[0][0], pending[0][1]
        else:
            wavs = torch.cat([p[0] for p in pending])
            masks = torch.cat([p[1] for p in pending])
        # Reconstruct phase with LWS algorithm (reference inference.py:141-154), all collected batches on the device
        out = lws_processor.refine_enhanced(wavs, masks, num_samples=wavs.shape[1]).cpu().numpy()
        ops.coop_check()            # the read-back synchronised: a cooperative kernel that gave up its bounded wait (its
        at = 0                      # outputs are invalid) raises HERE, before any file of these batches is written
        for _, _, paths, lengths in pending:
            writer.submit(out[at:at + len(lengths)], paths, lengths)
            at += len(lengths)
            written(paths, lengths)
        pending.clear()

    print('Starting inference on dataset: {:s}'.format(data_path_test))
    while True:
        try:
            feed, test_sample_path = unpack_batch(test_it.get_next(), uses_embeddings(config))
            test_length = feed['sequence_lengths']
        except OutOfRangeError:
            flush()
            print('done.')
            break
        model.feed(**feed)
        enhanced = model.enhanced_sources_oracle_phase if oracle_phase else model.enhanced_sources
        loss = model.loss                       # read back once, after the loop: no host round trip per batch
        loss = loss.detach().clone() if hasattr(loss, 'detach') else float(loss)
        ops.coop_poll()
        loss_list.append(loss)
        if oracle_phase:
            enhanced_host = enhanced.cpu().numpy()
            ops.coop_check()                    # as in flush(): checked after the synchronising read-back, before the files
            writer.submit(enhanced_host, test_sample_path, test_length)
            written(test_sample_path, test_length)
            continue
        masks = model.masks
        if pending and (pending[0][0].shape[1:] != enhanced.shape[1:] or pending[0][1].shape[1:] != masks.shape[1:]):
            flush()                             # a batch of another length: it cannot share the launch
        if len(test_length) >= lws_group and not pending:
            pending.append((enhanced, masks, test_sample_path, test_length))        # alone: no copy needed
        else:
            pending.append((enhanced.clone(), masks.clone(), test_sample_path, test_length))   # the model reuses its buffers
        if sum(len(p[3]) for p in pending) >= lws_group:
            flush()
    writer.close()          # every file is on disk (or its error raised) before the summary line
    ops.coop_check()
    loss_list = [float(x) for x in loss_list]

    # np.mean over the batches of ALL ranks (reference inference.py:170), whatever share of them each rank had
    tot, cnt = parallel.all_reduce_sum_scalars([float(np.sum(loss_list)), float(len(loss_list))])
    mean_loss = tot / cnt if cnt else 0.0
    print('Loss hole: {:.5}'.format(mean_loss))
    return mean_loss

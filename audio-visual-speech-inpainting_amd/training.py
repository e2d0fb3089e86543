"""Training driver of the inpainting models on MI355X.

``train(config_file)`` follows the reference trainer for the plain a/v/av-BLSTM models
(``av_speech_inpainting/training_emb.py:23-383``, the internally consistent copy of
``training.py``) and, for the multi-task ``*-ctc`` models, ``training_ctc.py:23-425`` (the
trainer the reference CLI binds to ``training``; its extra loss terms, PER columns and its choice of
the inpainting loss for model selection are kept): same config keys, same directory contract
(``<exp_folder>/netmodel/{config.txt, audio_features_mean.npy, audio_features_std.npy, sinet*,
ckpt*}``, ``<exp_folder>/training_log.txt``), same console / log line formats, NaN/Inf abort with
exit code 1, bad checkpoint with exit code 2, best-on-validation checkpoint ``sinet``, periodic
``ckpt`` every 1000 steps, early stopping.  TensorBoard summaries are not written (tensorboard is
not a dependency of this path).

Launched under ``torch.distributed.run`` it trains data-parallel: every rank reads the same file
list, whole batches are dealt round-robin to ranks, gradients are all-reduced (RCCL) inside
``model.train_op``, rank 0 writes logs and checkpoints.
"""
import os
import random
import shutil
import sys
from glob import glob
from time import time

import numpy as np

from . import models as net
from . import ops, parallel
from .config_utils import check_trainconfiguration, load_configfile
from .dataset_reader import DataManager, OutOfRangeError

_INPUT_OF = {'a-blstm': 'a', 'v-blstm': 'v', 'av-blstm': 'av'}
EMBEDDING_SIZE = 512           # training_emb.py:71


def is_ctc(config):
    """Multi-task models with the CTC head (training_ctc.py:110-127)."""
    return str(config['model']).endswith('-ctc')


def uses_embeddings(config):
    """Models fed with the external speaker embedding stored in the TFRecords (training_emb.py:102-110)."""
    return str(config['model']).endswith('-emb')


def build_model(config, mean, std, variables=None, is_training=True):
    """Model selection of the drivers (reference training_emb.py:78-117, inference.py:63-85)."""
    from . import model_variants as mv
    kind = str(config['model'])
    if kind == 'unet':
        model = net.UNetFConvModel(None, None, None, mean, std, 0.0, config, is_training=is_training,
                                   variables=variables)
    elif kind == 'av-blstm-twosteps':
        model = mv.StackedBLSTM2StepsModel(None, None, None, mean, std, 0.0, config, None, is_training=is_training,
                                           variables=variables)
    elif kind in _INPUT_OF:
        model = net.StackedBLSTMModel(None, None, None, mean, std, 0.0, config, input=_INPUT_OF[kind],
                                      variables=variables, is_training=is_training)
    elif kind[:-5] in _INPUT_OF and kind.endswith('-ssnn'):
        model = mv.StackedBLSTMSSNNModel(None, None, None, mean, std, 0.0, config, input=_INPUT_OF[kind[:-5]],
                                         variables=variables, is_training=is_training)
    elif kind[:-4] in _INPUT_OF and kind.endswith('-emb'):
        model = mv.StackedBLSTMEmbeddingModel(None, None, None, mean, std, 0.0, config, input=_INPUT_OF[kind[:-4]],
                                              variables=variables, is_training=is_training)
    elif kind.endswith('-ssnn-ctc') and kind[:-9] in _INPUT_OF:
        model = mv.StackedBLSTMSSNNCTCLossModel(None, None, None, None, None, mean, std, 0.0, config,
                                                input=_INPUT_OF[kind[:-9]], variables=variables, is_training=is_training)
    elif kind.endswith('-ctc') and kind[:-4] in _INPUT_OF:
        # the reference class cannot be built; see model_variants.StackedBLSTMCTCLossModel
        model = mv.StackedBLSTMCTCLossModel(None, None, None, None, None, mean, std, 0.0, config,
                                            input=_INPUT_OF[kind[:-4]], variables=variables, is_training=is_training)
    else:
        print('Model selection must be "a-blstm", "v-blstm", "av-blstm" (optionally with "-ssnn", "-emb", "-ctc" or '
              '"-ssnn-ctc"), '
              '"av-blstm-twosteps" or "unet" (got "{:s}"). Closing...'.format(kind))
        sys.exit(1)
    model.build_graph(var_scope=kind)
    return model


def unpack_batch(batch, with_embeddings, with_labels=False):
    """Reader tuple -> feed kwargs (+ sample paths).  Reference: training_emb.py:238-249,
    training_ctc.py:260-275 (labels and their lengths for the multi-task models)."""
    if with_embeddings:
        length, lab_len, audio, emb, paths, labels, video, mask = batch
    else:
        length, lab_len, audio, paths, labels, video, mask = batch
        emb = None
    if getattr(batch, 'device_arrays', None):
        # uploaded by the reader's prefetch thread (dataset_reader.Batch): use the device copies, ordered behind
        # their upload on the current stream
        n = len(batch)
        dev = lambda i, host: batch.to_device(i) if i in batch.device_arrays else host
        audio, video, mask = dev(2, audio), dev(n - 2, video), dev(n - 1, mask)
        if with_embeddings:
            emb = dev(3, emb)
    feed = dict(sequence_lengths=length, target_sources=audio, video_features=video, masks=mask)
    if with_embeddings:
        feed['embeddings'] = emb
    if with_labels:
        feed['labels'], feed['labels_lengths'] = labels, lab_len
    return feed, paths


class _LateScalars(object):
    """Device scalars of one step, read ONE STEP LATE without waiting for the step in between.

    ``float(tensor)`` is a device-to-host copy on the current stream: issued after the next step has been enqueued it
    waits for THAT step too, the host then starts launching the step after it on an idle GPU, and every step pays its
    launch time in the open (7.9 ms per step of 32 utterances against 6.2 ms of kernels).  Here the values of step k are
    gathered into one small tensor and copied to a page-locked buffer behind step k's own kernels, with an event; step
    k + 1 is enqueued, and only then the host waits for that EVENT -- which the GPU passed before it started step k + 1."""

    def __init__(self, n=3):
        self.slots = []
        self.n = n
        self.at = 0

    def push(self, values):
        import torch
        dev = [i for i, v in enumerate(values) if isinstance(v, torch.Tensor) and v.is_cuda]
        out = {'values': list(values), 'dev': dev, 'event': None, 'host': None}
        if dev:
            if len(self.slots) < self.n:
                self.slots.append(torch.empty(16, dtype=torch.float64).pin_memory())
            host = self.slots[self.at % self.n]
            self.at += 1
            packed = torch.stack([values[i].reshape(-1)[0].to(torch.float64) for i in dev])
            host[:len(dev)].copy_(packed, non_blocking=True)
            out['event'] = torch.cuda.Event()
            out['event'].record()
            out['host'] = host
        return out

    @staticmethod
    def get(handle):
        vals = list(handle['values'])
        if handle['event'] is not None:
            handle['event'].synchronize()
            for j, i in enumerate(handle['dev']):
                vals[i] = float(handle['host'][j])
        return [float(v) for v in vals]


def gap_elements(batch):
    """Number of zero elements of the batch's mask (the weight of the running loss means, training_emb.py:252).
    Counted on the device behind the reader's upload when the batch came through it (dataset_reader.Batch.gap_count, a
    device scalar that `book` resolves one step late with the losses): on the training thread the scan of 2 M mask
    elements cost more host time per step than launching the step."""
    n = batch.gap_count() if hasattr(batch, 'gap_count') else None
    return n if n is not None else int(np.count_nonzero(batch[-1] == 0))


def train(config_file, checkpoint_format=None):
    """
    Train the speech inpainting model.

    checkpoint_format: 'npz' (default) or 'tf' -- TensorFlow tensor bundles under the reference's
    variable names, readable by tf.train.Saver (also selectable with AVSI_CHECKPOINT_FORMAT).
    Restoring (``model_ckp``) accepts either format.
    """
    polls = ops.COOP_POLL_RAISES
    try:
        return _train(config_file, checkpoint_format)
    finally:
        # whatever way the loop is left (return, sys.exit of the NaN abort, an exception): the host-side polls of the
        # cooperative kernels raise again for whoever uses this process next
        ops.COOP_POLL_RAISES = polls


def _train(config_file, checkpoint_format):
    checkpoint_format = checkpoint_format or os.environ.get('AVSI_CHECKPOINT_FORMAT', 'npz')
    if checkpoint_format not in ('npz', 'tf'):
        raise ValueError("checkpoint_format must be 'npz' or 'tf', got %r" % (checkpoint_format,))
    config = check_trainconfiguration(load_configfile(config_file))
    rank, world = parallel.init()
    chief = rank == 0

    data_path = config['root_folder']
    data_path_train = os.path.join(data_path, 'training-set')
    data_path_val = os.path.join(data_path, 'validation-set')
    exp_path = config['exp_folder']
    exp_name = os.path.basename(exp_path)
    checkpoints_dir = os.path.join(exp_path, 'netmodel')
    log_path = os.path.join(exp_path, 'training_log.txt')
    feat_dim = config['audio_feat_dim']

    def manager():
        return DataManager(num_audio_samples=config['audio_len'], audio_feat_size=feat_dim,
                           video_feat_size=config['video_feat_dim'], buffer_size=4000, mode='fixed',
                           embedding_size=EMBEDDING_SIZE if uses_embeddings(config) else None)
    import torch
    # This process launches GPU work and shuffles small host arrays: with torch's default intra-op pool (one thread per
    # core) every tiny host-side tensor op wakes the whole pool, whose spin-waiting competes with the launching thread
    # and the reader thread for the cores the box grants (bench.py's training entry: 8.1 -> 7.4 ms per step).
    # AVSI_HOST_THREADS overrides.
    torch.set_num_threads(max(1, int(os.environ.get('AVSI_HOST_THREADS', min(torch.get_num_threads(), 4)))))
    device = torch.device('cuda', torch.cuda.current_device())       # the reader uploads batches from its prefetch thread
    train_files = sorted(glob(os.path.join(data_path_train, '*.tfrecord')))
    # data parallel: the same order on every rank.  AVSI_SHUFFLE_SEED makes a single-process run repeatable too.
    shuffle_seed = os.environ.get('AVSI_SHUFFLE_SEED')
    shuffle_seed = int(shuffle_seed) if shuffle_seed is not None else (0 if world > 1 else None)
    random.Random(shuffle_seed).shuffle(train_files)
    train_dm, val_dm = manager(), manager()
    _, train_it = train_dm.get_iterator(train_dm.get_dataset(train_files, shuffle=True,
                                                             seed=None if shuffle_seed is None else 1234 + shuffle_seed),
                                        batch_size=config['batch_size'], n_epochs=1, shard=(rank, world), device=device,
                                        even_rounds=True, count_gaps=True)
    val_files = sorted(glob(os.path.join(data_path_val, '*.tfrecord')))
    _, val_it = val_dm.get_iterator(val_dm.get_dataset(val_files, shuffle=False), batch_size=config['batch_size'],
                                    n_epochs=1, shard=(rank, world), device=device, count_gaps=True)

    ctc = is_ctc(config)
    audio_feat_mean = np.load(config['audio_feat_mean'])
    audio_feat_std = np.load(config['audio_feat_std'])
    model = build_model(config, audio_feat_mean, audio_feat_std)
    print('Model building done.')

    def save_checkpoint(prefix):
        if checkpoint_format == 'tf':
            return model.variables.save_tf(prefix, config['model'])
        return model.variables.save(prefix)

    if chief:
        os.makedirs(checkpoints_dir, exist_ok=True)
        dest_config = os.path.join(checkpoints_dir, 'config.txt')
        if os.path.abspath(dest_config) != os.path.abspath(config_file):
            shutil.copy(config_file, dest_config)
        shutil.copy(config['audio_feat_mean'], os.path.join(checkpoints_dir, 'audio_features_mean.npy'))
        shutil.copy(config['audio_feat_std'], os.path.join(checkpoints_dir, 'audio_features_std.npy'))
    log = open(log_path, 'a') if chief else open(os.devnull, 'w')

    train_size = len(np.load(os.path.join(data_path_train, 'seq_lengths.npy')))
    val_size = len(np.load(os.path.join(data_path_val, 'seq_lengths.npy')))
    # steps of ONE rank per epoch: global_step counts per-rank steps, so resuming recovers the epoch from it
    n_steps_epoch = int(train_size / (config['batch_size'] * world))
    n_steps = n_steps_epoch * config['max_n_epochs']

    header = [
        '+-- EXPERIMENT NAME - {:s} --+'.format(exp_name),
        '## Model type: {:s}'.format(config['model']),
        '## Network dimensions: {:s}'.format(str(config['net_dim'])),
        '## Optimizer: {:s}'.format(config['optimizer_type']),
        '## Starter learning rate: {:.6f}'.format(config['starter_learning_rate']),
        '## Learning rate update steps: {:d}'.format(config['lr_updating_steps']),
        '## Learning rate decay: {:.6f}'.format(config['lr_decay']),
    ] + (['## CTC-loss coefficient: {:.6f}'.format(config['ctc_loss'])] if ctc else []) + [
        '## L2 regularization coefficient: {:.6f}'.format(config['l2']),
        '## Dropout rate (no dropout if 0): {:.6f}'.format(config['dropout_rate']),
        '## Training dataset: {:s}'.format(data_path_train),
        '## Training size: {:d}'.format(train_size),
        '## Validation dataset: {:s}'.format(data_path_val),
        '## Validation size: {:d}'.format(val_size),
        '## Batch size: {:d}'.format(config['batch_size']),
        '## Approximated number of steps per epoch: {:d}'.format(n_steps_epoch),
        '## Number of training epochs: {:d}'.format(config['max_n_epochs']),
    ]
    if config.get('model_ckp_vnet') and hasattr(model, 'video_variables'):      # training_emb.py:162-168
        try:
            model.video_variables.restore(config['model_ckp_vnet'])
            print('Visual model variables restored.')
        except ValueError:
            print('{:s} is not a valid checkpoint. Closing...'.format(config['model_ckp_vnet']))
            sys.exit(2)
    if config['model_ckp']:
        try:
            model.variables.restore(config['model_ckp'])
            print('Model variables restored.')
        except ValueError:
            print('{:s} is not a valid checkpoint. Closing...'.format(config['model_ckp']))
            sys.exit(2)
    else:
        log.write('\n'.join(header) + '\n')
        log.write('## Approximated total number of steps: {:d}\n'.format(n_steps))
        log.write('\nEpoch\tLR\tTraining loss\tTraining PER \tValidation loss\tValidation PER[TIME]\n' if ctc else
                  '\nEpoch\tLR\tTraining loss\tValidation loss\t[TIME]\n')
    if chief:
        print('')
        print('\n'.join(header))
        print('')

    # The steps are launched on a HIGH-PRIORITY stream.  HIP multiplexes streams onto a few hardware queues per priority
    # level, and which of the model's side streams (weight-gradient GEMMs beside the BPTT of the layers below) happens to
    # share a queue with the launch stream decides what overlaps: with the legacy default stream the same step took 6.3 to
    # 7.4 ms depending on how many streams the process had created before (tools/train_step_time.py).  A queue of another
    # priority level is shared with none of them, and the cooperative recurrent grids get their CUs before the side grids.
    prev_stream = None
    if os.environ.get('AVSI_TRAIN_STREAM', 'high') == 'high':
        prev_stream = torch.cuda.current_stream(device)
        launch_stream = torch.cuda.Stream(device=device, priority=-1)
        launch_stream.wait_stream(prev_stream)
        torch.cuda.set_stream(launch_stream)

    tot_step = model.global_step
    epoch_counter = int(tot_step / n_steps_epoch) if n_steps_epoch else 0
    best_val_checkpoint = (0, 0)
    best_val_loss = -1.0
    cneg_epochs = 0
    train_start_time = time()
    lr = model.learning_rate
    # running means weighted by the number of gap frames: [loss, loss_func] for the plain models
    # (training_emb.py:252-253), [loss, loss_hole, ctc_loss, PER] for the multi-task ones (training_ctc.py:293-306)
    n_avg = 4 if ctc else 2
    train_avg = [float('nan')] * n_avg
    val_avg = [float('nan')] * n_avg
    epoch_duration = 0.0

    def fetch(training):
        """The reference's fetch list for one batch, still on the device (the CTC diagnostics are host work)."""
        if ctc:
            return [float(model.loss), float(model.loss_hole), float(model.ctc_loss), float(np.mean(model.per))]
        if training:
            return [model.loss, model.loss_func]
        return [model.loss_func] * 2

    late = _LateScalars()
    coop_fallbacks = 0
    # inside the loop the step guard decides about a cooperative-kernel timeout, on every rank at the same step; a
    # host-side poll that raised on one rank alone would leave its peers in the next collective
    ops.COOP_POLL_RAISES = False
    # AVSI_TRAIN_TIMING=1: host time of the phases of a training iteration, printed at the end (diagnostics)
    timing = [0.0] * 4 if os.environ.get('AVSI_TRAIN_TIMING') else None

    def resolve(vals):
        """Device scalars -> floats, synchronously (validation loop).  The training loop reads its scalars one step
        late through `late` instead (see _LateScalars); the NaN / Inf abort therefore fires one step after the
        offending batch."""
        vals = [float(x) for x in vals]
        ops.coop_check()                # the float() above synchronised; a cooperative-kernel timeout raises CoopTimeout
        return vals

    def validate(feed):
        """The reference's validation fetch for one batch (training_emb.py:312-327).  A cooperative-kernel timeout
        here costs the batch a repetition one level down (first the cooperative kernels that tolerate neighbours, then
        the batch-stationary ones, which cannot time out: the third attempt always returns).  No collective runs inside
        this loop, so a rank falls back alone and never leaves its peers waiting in the all-reduce of the validation
        means behind it; the levels of the ranks may then differ -- they choose kernels, not results."""
        nonlocal coop_fallbacks
        while True:
            model.set_dropout_rate(0.0)                             # training_emb.py:314,326
            model.feed(**feed)
            try:
                return resolve(fetch(False))
            except ops.CoopTimeout:
                if ops.coop_level() >= 2:
                    raise                       # no cooperative kernel is left to have timed out: not a residency matter
                ops.coop_fall_back(device)
                coop_fallbacks += 1

    def accumulate(avg, vals, nframe_sum, frames, first):
        if first:
            return list(vals), frames // feat_dim
        prev = nframe_sum
        nframe_sum += frames // feat_dim
        # reference: (avg * prev + value * count // dim) / total, floor division included
        return [(a * prev + v * frames // feat_dim) / nframe_sum for a, v in zip(avg, vals)], nframe_sum

    def launch(item):
        """Enqueue one training step (reference fetch list [loss, loss_func, ..., train_op], training_emb.py:244-251) and
        the asynchronous copy of its scalars: losses, gap elements of the batch, the two step-guard words."""
        model.set_dropout_rate(config['dropout_rate'])          # training_emb.py:251
        model.feed(**item['feed'])
        vals, item['lr'] = fetch(True), model.learning_rate
        model.train_op
        guard = model.step_guard
        item['vals'] = late.push(list(vals) + [gap_elements(item['batch']), guard[0], guard[1]])

    for _ in range(config['max_n_epochs']):
        epoch_counter += 1
        n_step = 0
        epoch_start_time = time()
        train_it.initializer()
        if chief:
            print('-> Epoch {:d}'.format(epoch_counter))
        nframe_sum = 0
        inflight = []             # steps enqueued but not yet booked, oldest first (at most two)

        def book(item):
            """Bookkeeping of one finished training step (reference training_emb.py:244-262)."""
            nonlocal train_avg, nframe_sum
            n_step, tot_step, lr = item['n_step'], item['tot_step'], item['lr']
            # [losses ..., gap elements, guard: non-finite word of all ranks, cooperative timeouts of all ranks]
            vals = _LateScalars.get(item['vals'])
            timeouts, nonfinite = vals.pop(), vals.pop()
            if timeouts != 0.0:
                # the update of this step (and of every step enqueued behind it: the status word is sticky) was
                # skipped on the device, on every rank alike: recoverable, see `recover`
                raise ops.CoopTimeout(ops._COOP_MSG)
            frames = int(vals.pop())
            # The verdict came with the gradients: the last all-reduce bucket of the step carries 'my loss is not
            # finite' of every rank (model.step_guard), the fused Adam obeyed it on the device -- the variables
            # are those of the last good step -- and it is read here, one step late like the loss: no collective and
            # no host wait of its own.  Every rank leaves at the SAME step (a rank that exits alone leaves its peers
            # waiting in the next gradient all-reduce).
            if not np.isfinite(nonfinite) and np.isfinite(vals[0]):
                print('GOT INSTABILITY on another rank: loss is not finite there. Leaving...')
                sys.exit(1)
            if np.isnan(vals[0]):
                print('GOT INSTABILITY: loss is NaN. Leaving...')
                sys.exit(1)
            if np.isinf(vals[0]):
                print('GOT INSTABILITY: loss is inf. Leaving...')
                sys.exit(1)
            train_avg, nframe_sum = accumulate(train_avg, vals, nframe_sum, frames, n_step == 1)
            if chief and (n_step % 200 == 0 or n_step == 1):
                if ctc:
                    print('Step[{:7d}] Loss[{:3.5f}|{:3.5f}|{:3.5f}] PER[{:.5f}] LR[{:.6f}] Epoch training time[{:.2f}]'.format(
                        tot_step, train_avg[0], train_avg[1], train_avg[2], train_avg[3], lr, time() - epoch_start_time))
                else:
                    print('Step[{:7d}] Loss[{:3.5f}|{:3.5f}] LR[{:.6f}] Epoch training time[{:.2f}]'.format(
                        tot_step, train_avg[0], train_avg[1], lr, time() - epoch_start_time))
            if chief and n_step % 1000 == 0:
                # booked before the next step is enqueued (see `settle`): the variables are those after step n_step
                # exactly, as in the reference (training_emb.py:266-268)
                print('Model checkpoint saved in file %s' % save_checkpoint(os.path.join(checkpoints_dir, 'ckpt')))

        def recover(items):
            """A cooperative recurrent launch timed out (another resident of the GPU -- e.g. RCCL's kernels -- left its
            workgroups no room): every step from the first void one on was skipped on the device.  Fall back to the
            batch-stationary kernels IN THIS PROCESS, take the skipped steps' counts back and repeat their batches.
            All ranks read the same summed guard words, so all of them arrive here at the same step -- but their LEVELS may
            differ (`validate` lets a rank fall back alone): a rank that is on the batch-stationary kernels already
            rewinds and repeats with its peers without falling further, because the timeout it read may be a peer's.
            Every attempt lowers every rank that can still fall by one level, so from the second attempt of one
            recovery on ALL ranks run batch-stationary kernels; a timeout after that is not a residency matter and is
            raised -- by every rank at the same attempt (they count attempts on the same summed word)."""
            nonlocal coop_fallbacks
            counted = len(items)                # steps whose optimiser count is ahead of the variables
            attempts = 0
            while True:
                if ops.coop_level() < 2:
                    coop_fallbacks += 1
                ops.coop_fall_back(device)      # (at level 2 it only waits for the streams and clears the status words)
                attempts += 1
                for _ in range(counted):
                    model.variables.rewind_step()
                try:
                    # one at a time: level 1 can still time out, and a void step takes every step enqueued behind it along
                    while items:
                        launch(items[0])
                        book(items[0])
                        items.pop(0)
                    return
                except ops.CoopTimeout:
                    if attempts >= 2:
                        raise                   # every rank was at level 2 in this attempt
                    counted = 1                 # the step just launched: counted, not applied, still first in `items`

        def settle(keep):
            """Book the steps in flight, oldest first, until at most `keep` are left."""
            while len(inflight) > keep:
                try:
                    book(inflight[0])
                except ops.CoopTimeout:
                    items = list(inflight)
                    del inflight[:]
                    recover(items)
                    return
                inflight.pop(0)

        while True:
            if timing is not None:
                t_a = time()
            try:
                batch = train_it.get_next()
                feed, _ = unpack_batch(batch, uses_embeddings(config), ctc)
            except OutOfRangeError:
                settle(0)
                if chief:
                    if ctc:
                        print('Completed epoch {:d} at step {:d} --> Training loss: {:3.5f} - {:3.5f} - {:3.5f}; PER: {:3.5f}'
                              .format(epoch_counter, tot_step, *train_avg))
                    else:
                        print('Completed epoch {:d} at step {:d} --> Training loss: {:3.5f} - {:3.5f}'.format(
                            epoch_counter, tot_step, *train_avg))
                epoch_duration = time() - epoch_start_time
                if chief:
                    print('Epoch training time (seconds) = {:.6f}'.format(epoch_duration))
                break
            n_step += 1
            tot_step += 1
            if timing is not None:
                t_b = time()
            item = dict(batch=batch, feed=feed, n_step=n_step, tot_step=tot_step)
            launch(item)
            inflight.append(item)
            if timing is not None:
                t_e = time()
            # one step stays in flight while the next is enqueued -- except in front of a periodic checkpoint, which
            # must hold the variables after ITS step and no later one
            settle(0 if n_step % 1000 == 0 else 1)
            if timing is not None:
                t_f = time()
                for i, dt_ in enumerate((t_b - t_a, t_e - t_b, t_f - t_e)):
                    timing[i] += dt_
                timing[3] += 1
        if chief:
            print('Start validation set evaluation...')
        model.is_training = False          # validation: no BPTT reserve, no gradient stream
        val_it.initializer()
        n_step = 0
        nframe_sum = 0
        while True:
            try:
                batch = val_it.get_next()
                feed, _ = unpack_batch(batch, uses_embeddings(config), ctc)
            except OutOfRangeError:
                break
            n_step += 1
            val_avg, nframe_sum = accumulate(val_avg, validate(feed), nframe_sum, int(gap_elements(batch)), n_step == 1)
            if chief and (n_step % 200 == 0 or n_step == 1):
                print('Step[{:7d}] Loss[{:3.5f}]'.format(n_step, val_avg[1]))
        model.is_training = True
        if world > 1:
            # combine the ranks' frame-weighted means (a rank may have seen no validation batch at all)
            w = float(nframe_sum) if n_step else 0.0
            sums = parallel.all_reduce_sum_scalars([a * w if n_step else 0.0 for a in val_avg] + [w])
            val_avg = [v / sums[-1] if sums[-1] > 0 else float('nan') for v in sums[:-1]]
        val_avg_loss = val_avg[1]          # model selection: loss_func (plain) / the inpainting loss (training_ctc.py:383-392)
        if chief:
            print('done.')
            if ctc:
                print('Validation loss: {:3.5f}; PER: {:3.5f}. Best loss so far {:2.5f} [Epoch {:d} (step {:d})]'.format(
                    val_avg_loss, val_avg[3], best_val_loss, best_val_checkpoint[0], best_val_checkpoint[1]))
            else:
                print('Validation loss: {:3.5f}. Best loss so far {:2.5f} [Epoch {:d} (step {:d})]'.format(
                    val_avg_loss, best_val_loss, best_val_checkpoint[0], best_val_checkpoint[1]))
        if best_val_checkpoint == (0, 0) or val_avg_loss < best_val_loss:
            if chief:
                print('Model saved in file %s' % save_checkpoint(os.path.join(checkpoints_dir, 'sinet')))
            best_val_checkpoint = (epoch_counter, tot_step)
            best_val_loss = val_avg_loss
            cneg_epochs = 0
        else:
            cneg_epochs += 1
        if chief:
            print('')
        if ctc:
            log.write('{:d}\t{:.6f}\t{:.6f}|{:.6f}|{:.6f}\t{:.6f}\t{:.6f}|{:.6f}|{:.6f}\t{:.6f}\t[{:.2f}]\n'.format(
                epoch_counter, lr, train_avg[0], train_avg[1], train_avg[2], train_avg[3], val_avg[0], val_avg[1], val_avg[2],
                val_avg[3], epoch_duration))
        else:
            log.write('{:d}\t{:.6f}\t{:.6f}|{:.6f}\t{:.6f}\t[{:.2f}]\n'.format(
                epoch_counter, lr, train_avg[0], train_avg[1], val_avg_loss, epoch_duration))
        log.flush()
        if cneg_epochs >= config['n_earlystop_epochs']:
            break

    log.close()
    if timing is not None and timing[3]:
        print('host ms per training iteration: next batch %.2f, launches of the step + its scalars %.2f, wait for the step '
              'before %.2f' % tuple(1e3 * t / timing[3] for t in timing[:3]), file=sys.stderr)
    model.coop_fallbacks = coop_fallbacks
    if prev_stream is not None:
        torch.cuda.current_stream(device).synchronize()
        torch.cuda.set_stream(prev_stream)
    if chief:
        if cneg_epochs >= config['n_earlystop_epochs']:
            print('+---- Done training: early stopped ----+')
        else:
            print('+---- Done training: epoch limit reached ----+')
        print('Total training time: {:.2f} s'.format(time() - train_start_time))
        print('{:d} epochs, {:d} steps.'.format(epoch_counter, tot_step))
        print('Best validation checkpoint: {:d} ({:d}) - Loss: {:.5f}'.format(
            best_val_checkpoint[0], best_val_checkpoint[1], best_val_loss))
    return model

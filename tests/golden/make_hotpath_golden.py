#!/usr/bin/env python3
"""Generate tests/golden/hotpath_golden.npz: seeded inputs of the hot path and the outputs of the
float64 CPU oracle for them (front end, 3 x BLSTM-250 forward, losses, gradient summaries, three
TF-Adam steps, waveform reconstruction).

The reference's own implementation (TensorFlow 1.x) cannot run in the build container
(SURVEY 8(c)), so these vectors come from the oracle restatement -- "parity unpinned" -- and serve
as a regression anchor: tests/test_golden.py checks the oracle against them on CPU and the HIP path
against them on the GPU.  Weights are not stored (4 M floats): they are regenerated from the seed
with oracle.blstm.init_params.
    python tests/golden/make_hotpath_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import blstm as O          # noqa: E402
from oracle import frontend as OF      # noqa: E402

PARAM_SEED, BIAS_SEED = 2024, 2025


def rand_biases(params, seed):
    rng = np.random.default_rng(seed)
    for layer in params['layers']:
        for d in ('fw', 'bw'):
            layer[d]['bias'] = rng.normal(0, 0.1, size=layer[d]['bias'].shape).astype(np.float32)
    params['proj']['biases'] = rng.normal(0, 0.1, size=params['proj']['biases'].shape).astype(np.float32)
    return params


def inputs():
    rng = np.random.default_rng(7)
    B, N = 3, 2304                      # 12 frames
    T = N // 192
    wav = np.clip(np.round(rng.normal(0, 3000, size=(B, N))), -32768, 32767).astype(np.float32)
    masks = np.ones((B, T, 257), dtype=np.float32)
    for b in range(B):
        s = rng.integers(1, T - 4)
        masks[b, s:s + 3] = 0
    video = rng.normal(size=(B, T, 136)).astype(np.float32)
    spec = OF.get_spectrogram(OF.get_stft(wav, window_size=24, step_size=12), log=True)
    mean, std = OF.feature_stats(list(spec))
    seq = np.array([T, T, T - 2])
    return wav, masks, video, mean.astype(np.float32), std.astype(np.float32), seq


def main():
    wav, masks, video, mean, std, seq = inputs()
    out = dict(wav=wav, masks=masks, video=video, mean=mean, std=std, seq_len=seq,
               param_seed=PARAM_SEED, bias_seed=BIAS_SEED)
    for kind, D in (('a', 257), ('av', 393)):
        p = rand_biases(O.init_params(PARAM_SEED, D), BIAS_SEED)
        fwd = O.model_forward(wav, masks, mean, std, seq, p, video=video, input_type=kind, keep=True)
        g = O.model_backward(fwd, masks.astype(np.float64), seq)
        out[kind + '_prediction'] = fwd['prediction'].astype(np.float32)
        out[kind + '_losses'] = np.array([fwd['loss_func'], fwd['loss_hole'], fwd['loss_valid']])
        out[kind + '_grad_l2'] = np.array([np.sqrt((v.astype(np.float64) ** 2).sum()) for _, v in O.flatten_params(g)])
        out[kind + '_grad_proj_bias'] = g['proj']['biases'].astype(np.float32)
        out[kind + '_logmel'] = OF.logmel_of_prediction(fwd['prediction'], mean, std).astype(np.float32)
        if kind == 'a':
            out['target_spec_norm'] = fwd['target_spec_norm'].astype(np.float32)
            out['enhanced_masked'] = O.enhanced_sources(fwd['prediction'], mean, std, fwd['target_stft'], masks,
                                                        num_samples=wav.shape[1]).astype(np.float32)
            out['enhanced_oracle'] = O.enhanced_sources(fwd['prediction'], mean, std, fwd['target_stft'], None,
                                                        num_samples=wav.shape[1]).astype(np.float32)
            # three TF-Adam steps: loss trajectory
            p64 = O.cast_params(p, np.float64)
            flat = [v for _, v in O.flatten_params(p64)]
            ms, vs = [np.zeros_like(v) for v in flat], [np.zeros_like(v) for v in flat]
            traj = []
            for step in (1, 2, 3):
                f = O.model_forward(wav, masks, mean, std, seq, p64, keep=True)
                gg = O.model_backward(f, masks.astype(np.float64), seq)
                traj.append(f['loss'])
                for (_, gv), pv, mv, vv in zip(O.flatten_params(gg), flat, ms, vs):
                    O.adam_tf_step(pv, gv, mv, vv, step, lr=1e-3)
            out['a_adam_losses'] = np.array(traj)
    lm = OF.get_log_mel_spectrogram(OF.get_spectrogram(OF.get_stft(wav, window_size=24, step_size=12), power=2))
    out['target_logmel'] = lm.astype(np.float32)
    np.savez_compressed(os.path.join(HERE, 'hotpath_golden.npz'), **out)
    print('wrote hotpath_golden.npz', {k: np.asarray(v).shape for k, v in out.items()})


if __name__ == '__main__':
    main()

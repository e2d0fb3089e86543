"""Command line of the MI355X speech-inpainting path.

Sub-commands, flags, defaults and dispatch follow the reference CLI
(``av_speech_inpainting/speech_inpainting_main.py:18-257``).  The sub-commands that run the hot
path and the data preparation around it are implemented (``dataset_generator``,
``tfrecords_generator``, ``audio_preprocessing``, ``masking``, ``training``, ``inference``); the
reference's landmark extraction, ASR and evaluation sub-commands are accepted by the parser
(same flags) but exit with a message: they are outside the scope of this package.
``training`` serves every model name of the reference's bound trainer (training_ctc.py:78-131): the
plain, ``-ssnn``, ``-emb``, two-step and U-Net models follow training_emb.py, the ``-ctc`` ones training_ctc.py
(the reference's own ``training`` only works for the latter).
"""
import argparse
import sys


def _flag(p, *names, **kw):
    p.add_argument(*names, **kw)


def build_parser():
    parser = argparse.ArgumentParser(description="Audio-visual speech inpainting system (MI355X hot path). "
                                     "Run '<subcommand> --help' for the options of a sub-command.")
    sub = parser.add_subparsers(dest='subparser_name')
    on = dict(action='store_const', const=True, default=False)

    p = sub.add_parser('dataset_generator', description='Generate masks dataset. Files are saved in <dest_dir>.')
    _flag(p, '-ca', '--clean_audio_dir', required=True)
    _flag(p, '-bs', '--speaker_ids', nargs='+', type=int, required=True)
    _flag(p, '-d', '--dest_dir', required=True)
    _flag(p, '-num', '--num_samples', type=int, required=True)
    _flag(p, '-al', '--audio_length', type=int, default=1024)
    _flag(p, '-i', '--num_max_intr', type=int, default=1)
    _flag(p, '-cm', '--mask_coverage_mean', type=float, default=0.3)
    _flag(p, '-cs', '--mask_coverage_std', type=float, default=0.1)
    _flag(p, '-e', '--ext', default='wav')

    p = sub.add_parser('audio_preprocessing',
                       description='Mean and standard deviation of audio features over <audio_dir>/<sample>/<file_prefix>.<ext>; '
                                   'results in <audio_dir>/<out_prefix>_{mean,std}.npy.')
    _flag(p, '-a', '--audio_dir', required=True, help='directory with one sub-directory per audio sample')
    _flag(p, '-p', '--file_prefix', required=True, help='file name (without extension) of the audio inside each sample directory')
    _flag(p, '-o', '--out_prefix', required=True, help='prefix of the output statistics files')
    _flag(p, '-t', '--type', default='spec', choices=['spec', 'fbanks', 'mfcc'], help='feature kind (default: spec)')
    _flag(p, '-sr', '--sample_rate', type=int, default=16000, help='target sample rate in Hz (default: 16000)')
    _flag(p, '-fs', '--fft_size', type=int, default=512, help='FFT size (default: 512)')
    _flag(p, '-ws', '--window_size', type=int, default=25, help='STFT window in ms (default: 25)')
    _flag(p, '-ss', '--step_size', type=int, default=10, help='STFT hop in ms (default: 10)')
    _flag(p, '-pe', '--preemph', type=float, default=0, help='pre-emphasis coefficient (default: 0 = off)')
    _flag(p, '-nm', '--num_mel_bins', type=int, default=80, help='mel filters (default: 80)')
    _flag(p, '-nmf', '--num_mfcc', type=int, default=13, help='MFCCs kept (default: 13)')
    _flag(p, '-d', '--delta', type=int, default=0, help='derivative orders appended (default: 0)')
    _flag(p, '-am', '--apply_mask', help='mask the features (mask.npy) before the statistics', **on)
    _flag(p, '-s', '--save_feat', help='also save the features of every sample as NPY', **on)
    _flag(p, '-e', '--ext', default='wav', help='audio file extension')

    p = sub.add_parser('video_preprocessing', description='Face-landmark extraction (not part of this package).')
    _flag(p, '-data', '--data_dir', required=True)
    _flag(p, '-s', '--speaker_ids', nargs='+', type=int, required=True)
    _flag(p, '-v', '--video_dir', required=True)
    _flag(p, '-d', '--dest_dir', required=True)
    _flag(p, '-sp', '--shape_predictor', required=True)
    _flag(p, '-e', '--ext', required=True, default='mpg')

    p = sub.add_parser('tfrecords_generator', description='TFRecord creation from the sample folders of dataset_generator.')
    _flag(p, '--embeddings', help='also store <sample>/vgg_embeddings/target.npy (the *-emb models)', **on)
    _flag(p, '-m', '--mode', default='fixed', choices=['fixed', 'var'])
    _flag(p, '-a', '--dataset_dir', required=True)
    _flag(p, '-d', '--dest_dir', required=True)
    _flag(p, '-df', '--dict_file', required=True)

    p = sub.add_parser('tfrecords_grouping', description='TFRecord grouping (not part of this package).')
    _flag(p, '-i', '--input_dir', required=True)
    _flag(p, '-o', '--output_dir', required=True)
    _flag(p, '-gs', '--group_size', type=int, default=16)
    _flag(p, '-d', '--del_input_dir', **on)

    p = sub.add_parser('masking', description='Write the gapped (masked) wavs of a TFRecord set and report the hole loss.')
    _flag(p, '-d', '--data_dir', required=True, help='directory with the TFRecords')
    _flag(p, '-ad', '--audio_dir', required=True, help='directory with one sub-directory per audio sample')
    _flag(p, '-m', '--mode', default='fixed', choices=['fixed', 'var'])
    _flag(p, '-af', '--audio_feat_dim', type=int, default=257)
    _flag(p, '-vf', '--video_feat_dim', type=int, default=136)
    _flag(p, '-ns', '--num_audio_samples', type=int, default=48000)
    _flag(p, '-op', '--oracle_phase', **on)
    _flag(p, '-bs', '--batch_size', type=int, default=0)

    p = sub.add_parser('training', description='Train a speech inpainting model.')
    _flag(p, '--config', required=True, type=str, help='configuration file')
    p = sub.add_parser('training_asr', description='Train an ASR model (not part of this package).')
    _flag(p, '--config', required=True, type=str)

    p = sub.add_parser('inference_model_generation', description='Re-save a model for inference (not needed here).')
    _flag(p, '--config', required=True, type=str, default="")
    _flag(p, '--model', type=str, choices=['enh', 'asr', 'enhasr'], default='enh')
    _flag(p, '--input_model', required=True, type=str)
    _flag(p, '--output_model', required=True, type=str)

    p = sub.add_parser('inference', description='Inference with a trained speech inpainting model.')
    _flag(p, '-d', '--data_dir', required=True, help='directory with the TFRecords')
    _flag(p, '-ad', '--audio_dir', required=True, help='output root: <audio_dir>/<sample>/enhanced/<out_file_prefix>.wav')
    _flag(p, '-ef', '--out_file_prefix', required=True, help='file name of the enhanced wavs')
    _flag(p, '-m', '--model_path', required=True, help='netmodel directory (config.txt, statistics, sinet checkpoint)')
    _flag(p, '-n', '--norm', help='apply the stored feature normalisation', **on)
    _flag(p, '-bs', '--batch_size', type=int, default=0)
    _flag(p, '-op', '--oracle_phase', help='use the target phase for the inverse STFT', **on)

    for name in ('inference_asr', 'inference_siasr'):
        p = sub.add_parser(name, description='ASR inference (not part of this package).')
        _flag(p, '-d', '--data_dir', required=True)
        _flag(p, '-ad', '--audio_dir', required=True)
        _flag(p, '-ef', '--out_file_prefix', required=True)
        if name == 'inference_asr':
            _flag(p, '-m', '--model_path', required=True)
            _flag(p, '-am', '--apply_mask', **on)
        else:
            _flag(p, '-ms', '--model_path_si', required=True)
            _flag(p, '-mr', '--model_path_asr', required=True)
            _flag(p, '-op', '--oracle_phase', **on)
        _flag(p, '-n', '--norm', **on)
        _flag(p, '-bs', '--batch_size', type=int, default=0)
        _flag(p, '-df', '--dict_file', required=True)

    p = sub.add_parser('evaluation', description='Speech-enhancement metrics (not part of this package).')
    _flag(p, '-ed', '--eval_audio_dir', required=True)
    _flag(p, '-ef', '--enhanced_file', required=True)
    _flag(p, '-o', '--out_file', required=True)
    _flag(p, '-me', '--masked_eval', **on)
    _flag(p, '--pesq_path', required=True)
    _flag(p, '--pesq_mode', required=True, choices=['nb', 'wb'])
    _flag(p, '-fs', '--fft_size', type=int, default=512)
    _flag(p, '-ws', '--window_size', type=int, default=25)
    _flag(p, '-ss', '--step_size', type=int, default=10)
    return parser


OUT_OF_SCOPE = ('video_preprocessing', 'tfrecords_grouping', 'training_asr',
                'inference_model_generation', 'inference_asr', 'inference_siasr', 'evaluation')


def main(argv=None):
    args = build_parser().parse_args(argv)
    cmd = args.subparser_name
    if cmd == 'dataset_generator':
        from .dataset_generator import create_syn_dataset
        create_syn_dataset(args.clean_audio_dir, args.dest_dir, args.speaker_ids, args.num_samples, args.audio_length,
                           args.num_max_intr, args.mask_coverage_mean, args.mask_coverage_std, args.ext)
    elif cmd == 'tfrecords_generator':
        from .tfrecord_utils import create_dataset
        create_dataset(args.dataset_dir, args.dest_dir, args.dict_file, args.mode, args.embeddings)
    elif cmd == 'audio_preprocessing':
        from .audio_feat_preprocessing import compute_mean_std_features
        compute_mean_std_features(args.audio_dir, args.file_prefix, args.out_prefix, args.type, args.sample_rate,
                                  args.fft_size, args.window_size, args.step_size, args.preemph, args.num_mel_bins,
                                  args.num_mfcc, args.delta, args.apply_mask, args.save_feat, args.ext)
    elif cmd == 'masking':
        from .masking import mask_app
        mask_app(args.data_dir, args.audio_dir, args.mode, args.oracle_phase, args.audio_feat_dim, args.video_feat_dim,
                 args.num_audio_samples, args.batch_size)
    elif cmd == 'training':
        from .training import train
        train(args.config)
    elif cmd == 'inference':
        from .inference import infer
        infer(args.model_path, args.data_dir, args.audio_dir, args.out_file_prefix, args.norm, args.oracle_phase,
              max(1, args.batch_size))
    elif cmd in OUT_OF_SCOPE:
        print("Sub-command '{:s}' belongs to the reference's offline data preparation / ASR / evaluation tooling and "
              "is not part of the MI355X hot-path package. Closing...".format(cmd))
        sys.exit(1)
    else:
        print('Bad subcommand name. Closing...')
        sys.exit(1)


if __name__ == '__main__':
    main()

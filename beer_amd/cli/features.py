"""`beer features extract | archive` (beer/cli/subcommands/features/).

`extract` reads the list of WAV files (or shell commands ending with `|`),
sends the signals to the GPU in batches and runs the whole front-end
(`beer_amd.features.extract`) for a batch in a handful of kernel launches;
one `.npy` per utterance is written as the reference does.
"""

import argparse
import glob
import io
import os
import subprocess
import sys
from zipfile import ZipFile

import numpy as np
import yaml

from .. import features as fea

# samples per GPU batch: 2^25 int16 samples = 35 minutes of 16 kHz audio
BATCH_SAMPLES = 1 << 25


class _ShowDefaults(argparse.Action):
    def __init__(self, option_strings, dest, **kwargs):
        super().__init__(option_strings, dest, nargs=0, **kwargs)

    def __call__(self, parser, namespace, values, option_string=None):
        print(yaml.dump(dict(fea.FEACONF), default_flow_style=False), end='')
        parser.exit()


def _read_wav(spec, logger):
    from scipy.io.wavfile import read
    if spec.endswith('|'):
        logger.debug(f'reading command: {spec[:-1]}')
        proc = subprocess.run(spec[:-1], shell=True, stdout=subprocess.PIPE)
        return read(io.BytesIO(proc.stdout))
    logger.debug(f'reading file: {spec}')
    return read(spec)


class extract:
    'extract speech features from a list of wav files'

    @staticmethod
    def setup(parser):
        parser.add_argument('--show-default-conf', action=_ShowDefaults,
                            help='show the default configuration and exit')
        parser.add_argument('feaconf', help='configuration file of the features')
        parser.add_argument('wav_list', help='list of WAV files or "-" for stdin')
        parser.add_argument('outdir', help='output directory')

    @staticmethod
    def main(args, logger):
        with open(args.feaconf, 'r') as fid:
            new_conf = yaml.safe_load(fid) or {}
        for key in new_conf:
            if key not in fea.FEACONF:
                logger.error(f'Unknown setting "{key}"')
                sys.exit(1)
        conf = dict(fea.FEACONF)
        conf.update(new_conf)
        lines = sys.stdin if args.wav_list == '-' else open(args.wav_list, 'r')
        pending, n_pending, counts = [], 0, 0

        def flush():
            nonlocal pending, n_pending, counts
            if not pending:
                return
            feats = fea.extract([sig for _, sig in pending], conf)
            for (uttid, _), mat in zip(pending, feats):
                np.save(os.path.join(args.outdir, uttid), mat)
            counts += len(pending)
            pending, n_pending = [], 0

        for line in lines:
            tokens = line.strip().split()
            if not tokens:
                continue
            uttid, spec = tokens[0], ' '.join(tokens[1:])
            srate, signal = _read_wav(spec, logger)
            if srate != conf['srate']:
                logger.error(f'Sampling rate ({conf["srate"]}) does not match the one of '
                             f'the given file ({srate}).')
                sys.exit(1)
            pending.append((uttid, signal))
            n_pending += len(signal)
            if n_pending >= BATCH_SAMPLES:
                flush()
        flush()
        if lines is not sys.stdin:
            lines.close()
        logger.info(f'extracted features for {counts} file(s)')


class archive:
    'create an archive from a features directory'

    @staticmethod
    def setup(parser):
        parser.add_argument('-e', '--extension', default='npy',
                            help='extension of the features file (default: npy)')
        parser.add_argument('feadir', help='features directory')
        parser.add_argument('out', help='output zip archived')

    @staticmethod
    def main(args, logger):
        counts = 0
        with ZipFile(args.out, 'w') as f:
            for path in sorted(glob.glob(os.path.join(args.feadir, '*' + args.extension))):
                name = os.path.basename(path).replace('.' + args.extension, '')
                f.write(path, arcname=name)
                counts += 1
        logger.info(f'created archive from {counts} features files')


COMMANDS = [extract, archive]

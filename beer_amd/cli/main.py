"""`beer <cmd> <subcmd>` command line (beer/cli/beer:20-43): same grammar,
global options and RNG seeding; `python -m beer_amd.cli` or `bin/beer`."""

import argparse
import logging
import pickle
import random

import numpy as np
import torch

from . import features as fea_cmds
from . import hmm as hmm_cmds
from .dataset import Dataset


class dataset_create:
    'compile a data set with the given features'

    @staticmethod
    def setup(parser):
        parser.add_argument('datadir', help='data directory')
        parser.add_argument('features', help='features archive (npz format)')
        parser.add_argument('out', help='output compiled dataset')

    @staticmethod
    def main(args, logger):
        import os
        dataset = Dataset.from_archive(os.path.abspath(args.features))
        with open(args.out, 'wb') as f:
            pickle.dump(dataset, f)
        logger.info(f'created dataset with {len(dataset)} utterances '
                    f'(total frame count: {dataset.size})')


GROUPS = {
    'dataset': ('dataset management', {'create': dataset_create}),
    'features': ('features related command', {c.__name__: c for c in fea_cmds.COMMANDS}),
    'hmm': ('Hidden Markov Model (HMM)',
            {c.__name__: c for c in hmm_cmds.COMMANDS}),
}


def build_parser():
    parser = argparse.ArgumentParser(prog='beer', description='BEER -- the Bayesian spEEch '
                                     'Recognizer (MI355X-native VB hot path)')
    parser.add_argument('-d', '--debug', action='store_true', help='show debug messages')
    parser.add_argument('-s', '--seed', type=int, default=-1, help='seed the RNG')
    sub = parser.add_subparsers(title='possible commands', metavar='<cmd>')
    sub.required = True
    for gname, (doc, cmds) in GROUPS.items():
        gparser = sub.add_parser(gname, help=doc)
        gsub = gparser.add_subparsers(title='possible commands', metavar='<cmd>')
        gsub.required = True
        for cname, cmd in cmds.items():
            cparser = gsub.add_parser(cname, help=cmd.__doc__)
            cmd.setup(cparser)
            cparser.set_defaults(func=cmd.main)
    return parser


def main(argv=None):
    logging.basicConfig(format='%(levelname)s: %(message)s')
    logger = logging.getLogger()
    logger.setLevel(logging.INFO)
    args = build_parser().parse_args(argv)
    if args.seed >= 0:
        torch.manual_seed(args.seed)
        np.random.seed(args.seed)
        random.seed(args.seed)
    if args.debug:
        logger.setLevel(logging.DEBUG)
    args.func(args, logger)


if __name__ == '__main__':
    main()

"""The model of the reference's own recipes, in memory.

recipes/aud/conf/hmm.yml:3-5, 31-33 (and recipes/timit_v2/conf/hmm.yml): a `non-speech-unit`
group -- 5 emitting states in an ergodic block, 10 diagonal Gaussians per state -- and a
`speech-unit` group -- 3 left-to-right states, 4 diagonal Gaussians per state --, one MixtureSet
per group joined in a JointModelSet (mkphones.py:100-113), a phone loop that starts and ends in
the non-speech group (monophone.sh:75) under a Dirichlet prior over the units.  Features:
recipes/*/conf/mfcc.yml, 13 x 3 = 39 dimensions (42 with energy as bench config 5 extracts them).

`phone_loop()` walks the steps of monophone.sh:64-82 (`beer hmm mkphones / mkphoneloopgraph /
mkdecodegraph / mkphoneloop`) with the command line's own builders (beer_amd/cli/hmm.py: build_units,
loop_graph, decode_graph, phone_loop),
without the pickles in between.  The configuration below is the recipe's, as numbers.
"""

import torch

import beer_amd as beer
from beer_amd.cli import hmm as cli

_T3 = [(0, 1, 1.), (1, 1, .75), (1, 2, .25), (2, 2, .75), (2, 3, .25), (3, 3, .75), (3, 4, .25)]
# the non-speech unit: state 1 enters, 2-4 are fully connected among themselves, 5 leaves
_T5 = [(0, 1, 1.)] + [(1, e, .25) for e in (1, 2, 3, 4)] + \
    [(s, e, .25) for s in (2, 3, 4) for e in (2, 3, 4, 5)] + [(5, 5, .75), (5, 6, .25)]


def _topology(arcs):
    return [{'start_id': s, 'end_id': e, 'trans_prob': w} for s, e, w in arcs]


def hmm_conf(n_normal_non_speech=10, n_normal_speech=4, cov_type='diagonal'):
    'recipes/aud/conf/hmm.yml as the list of dicts yaml.load gives mkphones.'
    common = {'prior_strength': 1., 'noise_std': .1, 'cov_type': cov_type, 'shared_cov': False}
    return [dict(common, group_name='non-speech-unit', n_normal_per_state=n_normal_non_speech,
                 topology=_topology(_T5)),
            dict(common, group_name='speech-unit', n_normal_per_state=n_normal_speech,
                 topology=_topology(_T3))]


def phone_loop(n_speech_units, mean, var, conf=None, n_non_speech_units=1, weights_prior='dirichlet',
               noise_std=None, seed=0):
    """(phone-loop model on the CPU in float32, units {name: Graph}).  Unit names: 'sil', 'sil2',
    ... for the non-speech group, then 0 .. n_speech_units - 1.  `noise_std` overrides the
    recipe's 0.1 (tests spread the initial means wider to get informative posteriors)."""
    conf = conf or hmm_conf()
    groups = {g['group_name']: (dict(g, noise_std=noise_std) if noise_std is not None else g)
              for g in conf}
    names = {'non-speech-unit': ['sil' + (str(i + 1) if i else '') for i in range(n_non_speech_units)],
             'speech-unit': list(range(n_speech_units))}
    torch.manual_seed(seed)
    units, emissions = cli.build_units(groups, names, mean, var)                         # mkphones
    loop = cli.loop_graph(list(units), edge_units=names['non-speech-unit'])              # mkphoneloopgraph
    graph, start_pdf, end_pdf = cli.decode_graph(loop, units)                            # mkdecodegraph
    ploop = cli.phone_loop(graph, start_pdf, end_pdf, emissions, weights_prior)          # mkphoneloop
    return ploop.float(), units

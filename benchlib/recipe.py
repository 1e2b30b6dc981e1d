"""The model of the reference's own recipes, in memory.

recipes/aud/conf/hmm.yml:3-5, 31-33 (and recipes/timit_v2/conf/hmm.yml): a `non-speech-unit`
group -- 5 emitting states in an ergodic block, 10 diagonal Gaussians per state -- and a
`speech-unit` group -- 3 left-to-right states, 4 diagonal Gaussians per state --, one MixtureSet
per group joined in a JointModelSet (mkphones.py:100-113), a phone loop that starts and ends in
the non-speech group (monophone.sh:75) under a Dirichlet prior over the units.  Features:
recipes/*/conf/mfcc.yml, 13 x 3 = 39 dimensions (42 with energy as bench config 5 extracts them).

`phone_loop()` walks the steps of monophone.sh:64-82 (`beer hmm mkphones / mkphoneloopgraph /
mkdecodegraph / mkphoneloop`) with the command line's own building blocks (beer_amd/cli/hmm.py),
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
    # -- mkphones (mkphones.py:95-113)
    start_pdf_id, pdfs, units = 0, [], {}
    for group, gconf in groups.items():
        tot = 0
        for name in names[group]:
            graph, start_pdf_id = cli.create_unit_graph(gconf['topology'], start_pdf_id)
            units[name] = graph
            tot += cli.count_emitting_state(graph)
        pdfs.append(cli.create_pdfs(mean, var, tot, gconf))
    emissions = beer.JointModelSet(pdfs)
    # -- mkphoneloopgraph --start-end-group non-speech-unit (mkphoneloopgraph.py)
    graph = beer.graph.Graph()
    graph.start_state, graph.end_state = graph.add_state(), graph.add_state()
    pivot = graph.add_state()
    unit2state = {name: graph.add_state() for name in units}
    for name in names['non-speech-unit']:
        graph.add_arc(graph.start_state, unit2state[name])
    for name in names['non-speech-unit']:
        graph.add_arc(unit2state[name], graph.end_state)
    for name in units:
        graph.add_arc(pivot, unit2state[name])
        graph.add_arc(unit2state[name], pivot)
    graph.normalize()
    # -- mkdecodegraph (mkdecodegraph.py)
    for name, hmm in units.items():
        graph.replace_state(unit2state[name], hmm)
    graph.normalize()
    start_pdf = {n: cli._single_pdf(h, h.find_next_pdf_ids(h.start_state)) for n, h in units.items()}
    end_pdf = {n: cli._single_pdf(h, h.find_previous_pdf_ids(h.end_state)) for n, h in units.items()}
    # -- mkphoneloop (mkphoneloop.py)
    size = len(start_pdf)
    if weights_prior == 'dirichlet':
        cat = beer.Categorical.create(torch.ones(size) / size, prior_strength=size / 2)
    elif weights_prior == 'dirichlet_process':
        cat = beer.SBCategorical.create(truncation=size, prior_strength=size / 2)
    else:
        cat = beer.SBCategoricalHyperPrior.create(truncation=size, prior_strength=size / 2,
                                                  hyper_prior_strength=1.)
    ploop = beer.PhoneLoop.create(graph.compile(), start_pdf, end_pdf, emissions, cat)
    return ploop.float(), units

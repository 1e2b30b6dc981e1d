"""`beer hmm <cmd>`: the callers either side of the hot path
(beer/cli/subcommands/hmm/*.py).  Same positional arguments, options, pickled
artefacts and log lines; `accumulate` / `decode` process the whole list of
utterances as ONE ragged batch on the GPU instead of a Python loop."""

import os
import pickle
import sys

import numpy as np
import torch
import yaml

import beer_amd as beer
from . import compat

# ---------------------------------------------------------------------------
# helpers


def _load(path):
    with open(path, 'rb') as f:
        return compat.load(f)


def _dump(obj, path):
    with open(path, 'wb') as f:
        pickle.dump(obj, f)


def _device():
    return torch.device('cuda', torch.cuda.current_device())


def _utt_ids(args_utts, dataset):
    if args_utts:
        stream = sys.stdin if args_utts == '-' else open(args_utts)
        return [line.strip().split()[0] for line in stream if line.strip()]
    return [utt.id for utt in dataset.utterances(random_order=False)]


def _cpu_elbo(elbo):
    'ELBO object with every tensor on the host (portable pickle).'
    # parameters are named by their uuid (models.parameters.ParameterRef): the file
    # holds neither device tensors nor model objects, `update` reads it without a GPU
    from ..models.parameters import ParameterRef
    acc = {ParameterRef(p.uuid): s.cpu() for p, s in elbo._acc_stats.items()}
    value = elbo.value.cpu() if isinstance(elbo.value, torch.Tensor) else elbo.value
    mbsize = elbo._minibatchsize
    if isinstance(mbsize, torch.Tensor):                  # after an RCCL all-reduce
        mbsize = int(round(float(mbsize)))
    refs = [ParameterRef(p.uuid) for p in elbo._model_parameters]
    return beer.EvidenceLowerBoundInstance(value, acc, refs, mbsize, elbo._datasize)


class UnitTopology:
    """A unit's HMM topology as conf/hmm.yml writes it (a list of {start_id, end_id, trans_prob})
    held as arrays: the lowest and the highest state id are the non-emitting entry and exit, the
    ids between them emit, in order (mkphones.py:12-43).  `graph(first_pdf)` instantiates it."""

    def __init__(self, arcs_conf):
        arcs = sorted({(int(a['start_id']), int(a['end_id']), float(a['trans_prob'])) for a in arcs_conf})
        self.src, self.dst = (np.asarray([a[i] for a in arcs], dtype=np.int64) for i in (0, 1))
        self.weight = np.asarray([a[2] for a in arcs], dtype=np.float64)
        ids = np.unique(np.concatenate([self.src, self.dst]))
        if not np.array_equal(ids, np.arange(len(ids))):
            raise ValueError(f'state ids of a topology must be 0 .. n-1, got {ids.tolist()}')
        self.n_states, self.n_emitting = len(ids), len(ids) - 2

    def graph(self, first_pdf):
        g = beer.graph.Graph()
        last = self.n_states - 1
        for sid in range(self.n_states):
            g.add_state(pdf_id=None if sid in (0, last) else first_pdf + sid - 1)
        g.start_state, g.end_state = 0, last
        for s, e, w in zip(self.src.tolist(), self.dst.tolist(), self.weight.tolist()):
            g.add_arc(s, e, w)
        return g


def build_units(groups_conf, grouped_names, mean, var):
    """`beer hmm mkphones` (mkphones.py:95-113): ({unit name: Graph}, JointModelSet of one
    MixtureSet per group).  `groups_conf`: {group name: its entry of conf/hmm.yml};
    `grouped_names`: {group name: [unit names]}, pdf ids running through the groups in order."""
    units, sets, next_pdf = {}, [], 0
    for group, names in grouped_names.items():
        conf = groups_conf[group]
        topo = UnitTopology(conf['topology'])
        for name in names:
            units[name] = topo.graph(next_pdf)
            next_pdf += topo.n_emitting
        n_states = topo.n_emitting * len(names)
        normals = beer.NormalSet.create(
            mean=mean, cov=var, size=n_states * conf['n_normal_per_state'],
            prior_strength=conf['prior_strength'], noise_std=conf['noise_std'],
            cov_type=conf['cov_type'], shared_cov=conf['shared_cov'])
        sets.append(beer.MixtureSet.create(n_states, normals, prior_strength=conf['prior_strength']))
    return units, beer.JointModelSet(sets)


def loop_graph(unit_names, edge_units=None):
    """`beer hmm mkphoneloopgraph` (mkphoneloopgraph.py:28-75): start, end and a pivot state, every
    unit reachable from and returning to the pivot; `edge_units` (the --start-end-group) are the
    units an utterance starts and ends with, default: the pivot itself.  `graph.symbols` maps a
    state to its unit (the reference's names for the three special states)."""
    graph = beer.graph.Graph()
    graph.start_state, graph.end_state = graph.add_state(), graph.add_state()
    pivot = graph.add_state()
    state_of = {name: graph.add_state() for name in unit_names}
    edges = [state_of[u] for u in edge_units] if edge_units else [pivot]
    for arc in [(graph.start_state, e) for e in edges] + [(e, graph.end_state) for e in edges]:
        graph.add_arc(*arc)
    for state in state_of.values():
        graph.add_arc(pivot, state)
        graph.add_arc(state, pivot)
    graph.symbols = {graph.start_state: '\\<s\\>', graph.end_state: '\\</s\\>', pivot: '#1',
                     **{state: name for name, state in state_of.items()}}
    graph.normalize()
    return graph


def _only_pdf(graph, walk):
    states = [s for s, _ in walk]
    if len(states) != 1:
        raise ValueError(f'expected only one emitting state, got: {len(states)}')
    return graph.state_from_id(states[0]).pdf_id


def decode_graph(loop, units):
    """`beer hmm mkdecodegraph` (mkdecodegraph.py:19-55): every unit state of the loop replaced by
    the unit's HMM; (graph, {unit: its first pdf}, {unit: its last pdf})."""
    state_of = {unit: state for state, unit in loop.symbols.items()}
    for unit, hmm in units.items():
        loop.replace_state(state_of[unit], hmm)
    loop.normalize()
    first = {u: _only_pdf(h, h.find_next_pdf_ids(h.start_state)) for u, h in units.items()}
    last = {u: _only_pdf(h, h.find_previous_pdf_ids(h.end_state)) for u, h in units.items()}
    return loop, first, last


PRIORS = ('dirichlet', 'dirichlet_process', 'gamma_dirichlet_process')


def phone_loop(graph, start_pdf, end_pdf, emissions, weights_prior='gamma_dirichlet_process',
               concentration=None):
    '`beer hmm mkphoneloop` (mkphoneloop.py:30-66).'
    size = len(start_pdf)
    conc = concentration if concentration else size / 2
    if weights_prior == 'dirichlet':
        cat = beer.Categorical.create(torch.ones(size) / size, prior_strength=conc)
    elif weights_prior == 'dirichlet_process':
        cat = beer.SBCategorical.create(truncation=size, prior_strength=conc)
    elif weights_prior == 'gamma_dirichlet_process':
        cat = beer.SBCategoricalHyperPrior.create(truncation=size, prior_strength=conc,
                                                  hyper_prior_strength=1.)
    else:
        raise ValueError(f'unknown prior over the weights: {weights_prior!r}')
    return beer.PhoneLoop.create(graph.compile(), start_pdf, end_pdf, emissions, cat)


def phones_of_path(path, start_pdf, per_frame=False):
    """A pdf-id path as unit symbols (decode.py:27-40): a new unit starts wherever the path ENTERS
    the first pdf of a unit from another pdf; `per_frame` repeats the running unit for every frame."""
    path = np.asarray(path, dtype=np.int64).reshape(-1)
    names = list(start_pdf)
    firsts = np.asarray([start_pdf[n] for n in names], dtype=np.int64)
    if path[0] not in firsts:
        raise KeyError(int(path[0]))                  # (as the reference's dictionary look-up)
    enters = np.concatenate([[True], (path[1:] != path[:-1]) & np.isin(path[1:], firsts)])
    which = {int(pdf): n for n, pdf in zip(names, firsts)}
    heads = [which[int(p)] for p in path[enters]]
    if not per_frame:
        return heads
    return [heads[i] for i in np.cumsum(enters) - 1]


# ---------------------------------------------------------------------------
# commands: each is (setup(parser), main(args, logger))

class mkphones:
    'create a set of left-to-right HMM representing "phones"'

    @staticmethod
    def setup(parser):
        group = parser.add_mutually_exclusive_group(required=True)
        group.add_argument('-d', '--dataset', help='dataset for initialization')
        group.add_argument('-D', '--dimension', type=int, help='dimension of the features')
        parser.add_argument('conf', help='configuration file')
        parser.add_argument('units', help='list of units and their group')
        parser.add_argument('out', help='output phone HMMs')

    @staticmethod
    def main(args, logger):
        with open(args.conf) as f:
            conf = yaml.safe_load(f)
        groups = {g['group_name']: g for g in conf}
        grouped = {name: [] for name in groups}
        with open(args.units) as f:
            for line in f:
                name, group = line.strip().split()
                grouped[group].append(name)
        if args.dataset:
            dataset = _load(args.dataset)
            mean, var = dataset.mean, dataset.var
        else:
            mean, var = torch.zeros(args.dimension).float(), torch.ones(args.dimension).float()
        units, emissions = build_units(groups, grouped, mean, var)
        _dump((units, emissions), args.out)
        logger.info(f'created {len(units)} HMMs for a total of {len(emissions)} emitting states')
        logger.info(f'expected features dimension: {len(mean)}')


class mkphoneloopgraph:
    'create a phone-loop graph'

    @staticmethod
    def setup(parser):
        parser.add_argument('-s', '--start-end-group', help='group starting/ending the loop')
        parser.add_argument('units', help='list of units and their group')
        parser.add_argument('out', help='output phone-loop graph')

    @staticmethod
    def main(args, logger):
        with open(args.units) as f:
            units = [tuple(line.strip().split()) for line in f if line.strip()]
        edge_units = [name for name, group in units if group == args.start_end_group] \
            if args.start_end_group else None
        graph = loop_graph([name for name, _ in units], edge_units)
        _dump(graph, args.out)
        logger.info(f'created phone-loop graph. # states: {len(list(graph.states()))} '
                    f'# arcs: {len(list(graph.arcs()))} start/end group: {args.start_end_group}')


class mkdecodegraph:
    'combine a set of HMMs with a phone-loop graph'

    @staticmethod
    def setup(parser):
        parser.add_argument('phoneloop', help='phone loop graph')
        parser.add_argument('hmms', help="phones' hmms")
        parser.add_argument('out', help='output decoding graph')

    @staticmethod
    def main(args, logger):
        graph = _load(args.phoneloop)
        units, _ = _load(args.hmms)
        graph, start_pdf, end_pdf = decode_graph(graph, units)
        _dump((graph, start_pdf, end_pdf), args.out)
        logger.info(f'created decoding graph. # states: {len(list(graph.states()))} '
                    f'# arcs: {len(list(graph.arcs()))} ')


class mkphoneloop:
    'create a phone-loop model'
    PRIORS = PRIORS

    @staticmethod
    def setup(parser):
        parser.add_argument('-c', '--concentration', type=float,
                            help='concentration of the Dirichlet Process')
        parser.add_argument('--weights-prior', default='gamma_dirichlet_process',
                            choices=mkphoneloop.PRIORS, help='prior over the phone weights')
        parser.add_argument('decode_graph', help='decoding graph')
        parser.add_argument('hmms', help="phones' hmm")
        parser.add_argument('out', help='phone loop model')

    @staticmethod
    def main(args, logger):
        graph, start_pdf, end_pdf = _load(args.decode_graph)
        _, emissions = _load(args.hmms)
        size = len(start_pdf)
        ploop = phone_loop(graph, start_pdf, end_pdf, emissions, args.weights_prior, args.concentration)
        _dump(ploop, args.out)
        logger.info(f'successfully created a phone-loop model with {size} phones')


class mkaligraph:
    'create the alignment graphs from transcriptions (stdin)'

    @staticmethod
    def setup(parser):
        parser.add_argument('hmms', help='hmm graph for each unit')
        parser.add_argument('outdir', help='output directory')

    @staticmethod
    def main(args, logger):
        hmm_graphs, _ = _load(args.hmms)
        uttids, seqs = [], []
        for line in sys.stdin:
            tokens = line.strip().split()
            if not tokens:
                continue
            uttid, phones = tokens[0], tokens[1:]
            if not phones:
                logger.error(f'utterance {uttid} has no transcription')
                continue
            uttids.append(uttid)
            seqs.append(phones)
        # every transcription in one native call (beer_aligraphs_compile); the
        # files keep the reference's format: a pickled dense CompiledGraph
        graphs = beer.graph.compile_alignments(seqs, hmm_graphs)
        for uttid, graph in zip(uttids, graphs):
            arr = np.empty(1, dtype=object)
            arr[0] = graph.to_dense()
            np.save(os.path.join(args.outdir, uttid + '.npy'), arr)
        logger.info(f'created alignment graphs for {len(uttids)} utterances')


class phonelist:
    "print the list of phones from a set of phones' HMM"

    @staticmethod
    def setup(parser):
        parser.add_argument('hmms', help="phones' hmms")

    @staticmethod
    def main(args, logger):
        units, _ = _load(args.hmms)
        import re
        natkey = lambda s: [int(t) if t.isdigit() else t for t in re.split(r'(\d+)', s.lower())]
        for key in sorted(units.keys(), key=natkey):
            print(key)


def _shard(model, dataset, uttids, alis, logger):
    'Features and per-utterance graphs of the utterances that exist.'
    feats, graphs, kept = [], [], []
    for uttid in uttids:
        try:
            utt = dataset[uttid]
        except KeyError:
            logger.warning(f'no utterance {uttid} in the dataset')
            continue
        graph = None
        if alis is not None:
            try:
                graph = alis[uttid][0]
            except KeyError:
                logger.warning(f'no alignment graph for utterance "{uttid}"')
        feats.append(utt.features)
        graphs.append(graph)
        kept.append(uttid)
    return feats, graphs, kept


def _load_alis(path):
    '''alis.npz: one `.npy` object array [CompiledGraph] per utterance
    (mkaligraph.py:60-63, accumulate.py:35,50).  Loaded eagerly; archives
    written by the reference are remapped to beer_amd classes.'''
    if not path:
        return None
    return compat.load_npz(path)


class accumulate:
    'Accumulate the ELBO from a list of utterances given from "stdin"'

    @staticmethod
    def setup(parser):
        parser.add_argument('-a', '--alis', help='alignment graphs in a "npz" archive')
        parser.add_argument('-s', '--acoustic-scale', default=1., type=float)
        parser.add_argument('model', help='hmm based model')
        parser.add_argument('dataset', help='training data set')
        parser.add_argument('out', help='output accumulated ELBO')

    @staticmethod
    def main(args, logger):
        model = _load(args.model).to(_device())
        dataset = _load(args.dataset)
        alis = _load_alis(args.alis)
        uttids = [line.strip().split()[0] for line in sys.stdin if line.strip()]
        feats, graphs, kept = _shard(model, dataset, uttids, alis, logger)
        elbo = beer.evidence_lower_bound(datasize=dataset.size)
        count = len(kept)
        # Utterances with an alignment graph train the emissions only (the phone
        # weights get no counts: phoneloop.py:98-100); those without one go through
        # the free phone loop and DO count phones, as the reference's per-utterance
        # `inference_graph=None` does (accumulate.py:45-54): two batches, one sum.
        aligned = [i for i, g in enumerate(graphs) if g is not None]
        free = [i for i, g in enumerate(graphs) if g is None]
        if aligned:
            elbo = elbo + beer.accumulate_elbo(
                model, [feats[i] for i in aligned], datasize=dataset.size,
                inference_graphs=[graphs[i] for i in aligned], scale=args.acoustic_scale)
        if free:
            elbo = elbo + beer.accumulate_elbo(model, [feats[i] for i in free],
                                               datasize=dataset.size, scale=args.acoustic_scale)
        _dump((_cpu_elbo(elbo), count), args.out)
        norm = max(count, 1) * dataset.size
        logger.info(f'accumulated ELBO over {count} utterances: {float(elbo) / norm :.3f}.')


class update:
    'Update the parameters of the model given the set of ELBO loaded from stdin'

    @staticmethod
    def setup(parser):
        parser.add_argument('-l', '--learning-rate', default=1., type=float)
        parser.add_argument('-o', '--optim-state', help='optimizer state')
        parser.add_argument('model', help='model to update')
        parser.add_argument('out_model', help='updated model')

    @staticmethod
    def main(args, logger):
        model = _load(args.model)
        optim = beer.VBConjugateOptimizer(model.conjugate_bayesian_parameters(keepgroups=True),
                                          lrate=args.learning_rate)
        if args.optim_state and os.path.isfile(args.optim_state):
            optim.load_state_dict(torch.load(args.optim_state))
        optim.init_step()
        elbo, nutts = None, 0
        for line in sys.stdin:
            if not line.strip():
                continue
            elbo_batch, nutts_batch = _load(line.strip())
            elbo = elbo_batch if elbo is None else elbo + elbo_batch
            nutts += nutts_batch
        elbo.sync(model)
        elbo.backward()
        optim.step()
        _dump(model, args.out_model)
        if args.optim_state:
            torch.save(optim.state_dict(), args.optim_state)
        logger.info(f'accumulated ELBO={float(elbo) / (nutts * elbo._datasize):.3f}')


class train:
    'train a HMM based model on a single machine'

    @staticmethod
    def setup(parser):
        parser.add_argument('-b', '--batch-size', type=int, default=-1,
                            help='utterances per update (-1: all)')
        parser.add_argument('-e', '--epochs', type=int, default=1)
        parser.add_argument('-l', '--lrate', type=float, default=1.)
        parser.add_argument('model', help='hmm based model')
        parser.add_argument('dataset', help='training data set')
        parser.add_argument('out', help='output model')

    @staticmethod
    def main(args, logger):
        # (the reference's `train` refers to an optimizer that no longer exists,
        #  train.py:34; this is the working equivalent)
        model = _load(args.model).to(_device())
        dataset = _load(args.dataset)
        optim = beer.VBConjugateOptimizer(model.mean_field_factorization(), lrate=args.lrate)
        for epoch in range(1, args.epochs + 1):
            utts = list(dataset.utterances())
            bsize = len(utts) if args.batch_size <= 0 else args.batch_size
            for b in range(0, len(utts), bsize):
                batch = utts[b:b + bsize]
                optim.init_step()
                elbo = beer.accumulate_elbo(model, [u.features for u in batch],
                                            datasize=dataset.size)
                elbo.backward()
                optim.step()
                logger.info(f'epoch={epoch} batch={b // bsize + 1} '
                            f'ELBO={float(elbo) / (len(batch) * dataset.size):.3f}')
        _dump(model.cpu(), args.out)
        logger.info(f'finished training after {args.epochs} epochs. '
                    f'KL(q || p) = {float(model.kl_div_posterior_prior()): .3f}')


class decode:
    'print the most likely path of all the utterances of a dataset'

    @staticmethod
    def setup(parser):
        parser.add_argument('-a', '--alis', help='alignment graphs in a "npz" archive')
        parser.add_argument('--per-frame', action='store_true')
        parser.add_argument('-s', '--acoustic-scale', default=1., type=float)
        parser.add_argument('-u', '--utts', help='utterances to decode ("-" for stdin)')
        parser.add_argument('model', help='hmm based model')
        parser.add_argument('dataset', help='data set')

    @staticmethod
    def main(args, logger):
        model = _load(args.model).to(_device())
        dataset = _load(args.dataset)
        alis = _load_alis(args.alis)
        feats, graphs, kept = _shard(model, dataset, _utt_ids(args.utts, dataset), alis, logger)
        use = None if alis is None else [g if g is not None else model.graph for g in graphs]
        paths = beer.decode_batch(model, feats, inference_graphs=use, scale=args.acoustic_scale) \
            if kept else []
        for uttid, path in zip(kept, paths):
            phones = phones_of_path(path.cpu().numpy(), model.start_pdf, args.per_frame)
            print(uttid, ' '.join(phones))
        logger.info(f'successfully decoded {len(kept)} utterances.')


class posteriors:
    'state / phone posteriors of all the utterances of a dataset'
    EPS = 1e-5

    @staticmethod
    def setup(parser):
        parser.add_argument('-S', '--state', action='store_true', help='state level')
        parser.add_argument('-l', '--log', action='store_true', help='log domain')
        parser.add_argument('-s', '--acoustic-scale', default=1., type=float)
        parser.add_argument('-u', '--utts', help='utterances ("-" for stdin)')
        parser.add_argument('model', help='hmm based model')
        parser.add_argument('dataset', help='data set')
        parser.add_argument('outdir', help='output directory')

    @staticmethod
    def main(args, logger):
        model = _load(args.model).to(_device())
        dataset = _load(args.dataset)
        count = 0
        for uttid in _utt_ids(args.utts, dataset):
            try:
                utt = dataset[uttid]
            except KeyError:
                logger.warning(f'no data for utterance {uttid}')
                continue
            posts = model.posteriors(utt.features, scale=args.acoustic_scale).cpu().numpy()
            if not args.state:
                out = np.zeros((len(posts), len(model.start_pdf)))
                for i, unit in enumerate(model.start_pdf):
                    out[:, i] = posts[:, model.start_pdf[unit]:model.end_pdf[unit]].sum(axis=-1)
                posts = out
            if args.log:
                posts = np.log(posteriors.EPS + posts)
            np.save(os.path.join(args.outdir, f'{uttid}.npy'), posts)
            count += 1
        logger.info(f'successfully computed the posteriors for {count} utterances.')


COMMANDS = [accumulate, decode, mkaligraph, mkdecodegraph, mkphoneloop, mkphoneloopgraph,
            mkphones, posteriors, phonelist, train, update]

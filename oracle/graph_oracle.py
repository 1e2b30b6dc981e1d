"""CPU oracle for graph compilation.

TEST INFRASTRUCTURE ONLY.  A plain-Python restatement of Graph.compile
(beer/graph.py:185-240: non-emitting states removed by path following,
156-182; rows renormalised without changing the diagonal, 230-237) and of
create_graph_from_seq (beer/cli/subcommands/hmm/mkaligraph.py:18-39), used to
check the native compiler (`beer_graph_compile`, `beer_aligraphs_compile`).

Parity status: PINNED -- `tests/golden/g12_graph_compile.npz` hold the reference's
compiled tables for the same topologies (tests/test_oracle_golden.py,
tests/test_host.py).

It works on any object with the reference's Graph interface (`_states`,
`arcs(state_id, incoming)`, `start_state`, `end_state`); float32 tables like
the reference's `torch.zeros`.
"""

import numpy as np


def _walk(graph, start_state, init_weight, incoming):
    'graph.py:156-182'
    frontier = [(arc, init_weight) for arc in graph.arcs(start_state, incoming=incoming)]
    visited = {start_state}
    while frontier:
        arc, weight = frontier.pop()
        nxt = arc.start if incoming else arc.end
        if graph._states[nxt].pdf_id is not None:
            yield nxt, weight * arc.weight
        elif nxt not in visited:
            frontier += [(a, arc.weight * weight) for a in graph.arcs(nxt, incoming=incoming)]
            visited.add(nxt)


def compile_graph(graph):
    '''(init_probs [S], final_probs [S], trans_probs [S, S], pdf_id_mapping),
    float32 probabilities (graph.py:185-240).'''
    index, pdf_id_mapping = {}, []
    for state_id, state in graph._states.items():
        if state.pdf_id is not None:
            index[state_id] = len(pdf_id_mapping)
            pdf_id_mapping.append(state.pdf_id)
    n = len(pdf_id_mapping)
    init = np.zeros(n, dtype=np.float32)
    final = np.zeros(n, dtype=np.float32)
    trans = np.zeros((n, n), dtype=np.float32)
    for state_id, weight in _walk(graph, graph.start_state, 1.0, False):
        init[index[state_id]] += np.float32(weight)
    init /= init.sum()
    for state_id, weight in _walk(graph, graph.end_state, 1.0, True):
        final[index[state_id]] += np.float32(weight)
    final /= final.sum()
    for arc in graph.arcs():
        if graph._states[arc.start].pdf_id is None:
            continue
        src = index[arc.start]
        if graph._states[arc.end].pdf_id is None:
            for state_id, weight in _walk(graph, arc.end, arc.weight, False):
                trans[src, index[state_id]] += np.float32(weight)
        else:
            trans[src, index[arc.end]] += np.float32(arc.weight)
    for i in range(n):
        diag = trans[i, i]
        off_diag = trans[i, :].sum() - diag
        if diag > 0. and off_diag > 0:
            trans[i, :] /= off_diag / (1 - diag)
            trans[i, i] = diag
    return init, final, trans, pdf_id_mapping


def alignment_graph(seq, units, graph_cls):
    '''mkaligraph.py:18-39: chain of placeholder states, each replaced by its
    unit HMM, normalised.  `graph_cls` is the Graph class to build with.'''
    graph = graph_cls()
    graph.start_state = graph.add_state()
    last, placeholders = graph.start_state, []
    for _ in seq:
        state = graph.add_state()
        placeholders.append(state)
        graph.add_arc(last, state)
        last = state
    graph.end_state = graph.add_state()
    graph.add_arc(last, graph.end_state)
    for state, unit in zip(placeholders, seq):
        graph.replace_state(state, units[unit])
    graph.normalize()
    return graph

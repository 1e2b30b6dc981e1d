"""Host profile of the alignment-graph stage of config 5 (compile + list + device image)."""
import cProfile, pstats, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench, beer_amd as beer
rng = np.random.RandomState(5)
units, pdf = {}, 0
for p in range(40):
    u = beer.graph.Graph()
    for sid in range(5):
        u.add_state(pdf_id=None if sid in (0, 4) else pdf + sid - 1)
    u.start_state, u.end_state = 0, 4
    for arc in bench.TOPO:
        u.add_arc(*arc)
    units[p] = u
    pdf += 3
seqs = [[int(v) for v in rng.randint(0, 40, 30)] for _ in range(3597)]
torch.zeros(1, device='cuda')
for rep in range(3):
    t0 = time.perf_counter()
    gset = beer.graph.compile_alignments(seqs, units)
    t1 = time.perf_counter()
    graphs = list(gset)
    t2 = time.perf_counter()
    gset.device_image(torch.float32)
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    print(f'compile {1e3 * (t1 - t0):.1f} ms, list {1e3 * (t2 - t1):.1f} ms, device image {1e3 * (t3 - t2):.1f} ms')
pr = cProfile.Profile()
pr.enable()
gset = beer.graph.compile_alignments(seqs, units)
graphs = list(gset)
gset.device_image(torch.float32)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats('tottime').print_stats(14)

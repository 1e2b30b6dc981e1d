"""Time of the fused forward-backward launch alone (config-3 phone loop, 3.33 M frames):
    python tools/probes/fb_time.py [frames] [reps]
Prints ms per launch and a checksum of the posteriors (to compare builds)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench, beer_amd as beer
from beer_amd import hmm_kernels as hk

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 3_333_334
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = torch.device('cuda:0')
lengths = bench.hmm_corpus(frames)
T = sum(lengths)
ploop = bench.make_phone_loop('diagonal', dev)
g = torch.Generator(device=dev).manual_seed(1)
pc = torch.randn(T, 120, generator=g, device=dev) * 3 - 60
batch = hk.HmmBatch([ploop.graph], [0] * len(lengths), lengths, torch.float32)
assert hk.fused_ok(batch)
utt = torch.zeros(len(lengths), dtype=torch.float64, device=dev)
for _ in range(2):
    sr, g0, flow = hk.posteriors_fused(batch, pc, 1., want_counts=True, utt_llh=utt)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(reps):
    sr, g0, flow = hk.posteriors_fused(batch, pc, 1., want_counts=True, utt_llh=utt)
b.record()
torch.cuda.synchronize()
print(f'fb fused: {a.elapsed_time(b) / reps:.3f} ms per launch of {T} frames in {len(lengths)} utterances; '
      f'checksum sr {float(sr.double().sum()):.6f} {float((sr.double() ** 2).sum()):.6f} '
      f'g0 {float(g0.sum()):.9f} flow {float(flow.sum()):.6f}')

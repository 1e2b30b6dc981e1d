"""Config 3's phone loop with 39-dimensional frames (13 MFCCs x 3, not a multiple of four):
one VB iteration over 3.33 M frames with the frame fragment images and without (BEER_FRAME_IMAGE=0)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench, beer_amd as beer
dev = torch.device('cuda:0')
dim = int(sys.argv[1]) if len(sys.argv) > 1 else 39
lens = bench.hmm_corpus(3_333_334)
T = sum(lens)
X = torch.randn(T, dim, device=dev)
for img in ('1', '0'):
    os.environ['BEER_FRAME_IMAGE'] = img
    ploop = bench.make_phone_loop('diagonal', dev, dim=dim)
    optim = beer.VBConjugateOptimizer(ploop.mean_field_factorization(), 1.)
    images, statics = beer.FrameImages(X), beer.ShardStatics()
    def step():
        optim.init_step()
        elbo = beer.accumulate_elbo(ploop, (X, lens), datasize=T, frame_images=images, statics=statics)
        elbo.backward(); optim.step()
        return elbo
    for _ in range(2): e = step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): e = step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    print(f'D = {dim}, frame images {"on " if img == "1" else "off"}: {dt * 1e3:.2f} ms per iteration = {T / dt / 1e6:.1f} M frames/s, '
          f'ELBO/frame {float(e) / (len(lens) * T):.6f}, images built {images.builds}')

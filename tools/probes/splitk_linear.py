"""Weight gradient of a Linear layer over 1 M rows: torch's mm against a batched split over the rows."""
import torch, time
dev = torch.device('cuda:0')
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
T = 1_000_003
for fin, fout in ((40, 128), (128, 40), (64, 128), (128, 64), (128, 128)):
    X = torch.randn(T, fin, device=dev); dY = torch.randn(T, fout, device=dev)
    ref = dY.t().mm(X)
    print(f'in {fin:4d} out {fout:4d}: mm {t(lambda: dY.t().mm(X)):.3f} ms', end='')
    for rows in (1024, 4096, 16384):
        c = T // rows
        def split():
            main = c * rows
            g = torch.bmm(dY[:main].view(c, rows, fout).transpose(1, 2), X[:main].view(c, rows, fin)).sum(0)
            if main < T: g = g + dY[main:].t().mm(X[main:])
            return g
        err = float((split() - ref).abs().max() / ref.abs().max())
        print(f' | rows {rows}: {t(split):.3f} ms (err {err:.1e})', end='')
    print()

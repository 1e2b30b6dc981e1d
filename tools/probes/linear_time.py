"""Forward time of torch.nn.Linear at config 4's network shapes ([1 M, d_in] x [d_in, d_out], float32)
with the BLAS back ends torch offers on ROCm."""
import torch, time
T = 1_000_000
dev = 'cuda'
def bench(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
for lib in ('default', 'cublas', 'cublaslt'):
    if lib != 'default':
        try:
            torch.backends.cuda.preferred_blas_library(lib)
        except Exception as e:
            print(lib, 'unavailable', e); continue
    for din, dout in ((40, 128), (128, 128), (128, 64), (64, 128), (128, 40)):
        x = torch.randn(T, din, device=dev)
        lin = torch.nn.Linear(din, dout).to(dev)
        with torch.no_grad():
            t1 = bench(lambda: lin(x))
            t2 = bench(lambda: x @ lin.weight.t())
            t3 = bench(lambda: torch.addmm(lin.bias, x, lin.weight.t()))
        print(f'{lib:9s} [{T}, {din}] -> {dout}: linear {t1:.3f} ms, matmul {t2:.3f} ms, addmm {t3:.3f} ms '
              f'(streaming floor {(4. * T * (din + dout)) / 5e12 * 1e3:.3f} ms at 5 TB/s)')
print('weight gradients  dW = dY^T X  ([d_out, T] x [T, d_in]):')
for lib in ('cublaslt', 'cublas'):
    torch.backends.cuda.preferred_blas_library(lib)
    for din, dout in ((40, 128), (128, 128), (128, 64), (64, 128), (128, 40)):
        x = torch.randn(T, din, device=dev)
        dy = torch.randn(T, dout, device=dev)
        t1 = bench(lambda: dy.t() @ x)
        lin = torch.nn.Linear(din, dout).to(dev)
        xr = x.clone().requires_grad_(True)
        def fb():
            lin.zero_grad(set_to_none=True)
            lin(xr).backward(dy)
        t2 = bench(fb)
        print(f'{lib:9s} d_in {din} d_out {dout}: dY^T X {t1:.3f} ms; Linear forward + backward {t2:.3f} ms')

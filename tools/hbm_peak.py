"""GPU experiment: what plain streaming kernels reach on this box (torch copy / add / fused multiply-add on 491.5 MB vectors)."""
import time, torch
n = 512 * 24 * 10000
a = torch.randn(n, device="cuda"); b = torch.randn(n, device="cuda"); c = torch.empty_like(a); d = torch.randn(n, device="cuda")
def t(fn, k=20):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(k): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / k
for name, fn, nb in (("copy  (1r 1w)", lambda: c.copy_(a), 8), ("add   (2r 1w)", lambda: torch.add(a, b, out=c), 12),
                     ("addcmul (3r 1w)", lambda: torch.addcmul(a, b, d, out=c), 16), ("fill  (1w)", lambda: c.fill_(1.0), 4),
                     ("sum   (1r)", lambda: a.sum(), 4), ("axpy in place (2r 1w)", lambda: a.add_(b, alpha=0.5), 12)):
    dt = t(fn)
    print(f"{name:24s} {dt*1e6:8.1f} us  {nb*n/dt/1e9:8.1f} GB/s", flush=True)

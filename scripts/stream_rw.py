"""Attainable HBM rates of plain streaming kernels (torch elementwise) at the tensor sizes of the 1x1 epilogues."""
import torch
def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters
for mb in (34, 137, 275, 550):
    n = mb * 1024 * 1024 // 2
    a = torch.randn(n, device="cuda").to(torch.bfloat16); b = torch.randn(n, device="cuda").to(torch.bfloat16); y = torch.empty_like(a)
    t_fill = timeit(lambda: y.zero_())
    t_copy = timeit(lambda: y.copy_(a))
    t_add = timeit(lambda: torch.add(a, b, out=y))
    t_read = timeit(lambda: a.sum())
    print(f"{mb:4d} MB: write {mb/1e3/t_fill*1e3/1e3:5.2f} TB/s | copy(r+w) {2*mb/1e6/t_copy*1e3:5.2f} TB/s | add(2r+w) {3*mb/1e6/t_add*1e3:5.2f} TB/s | read {mb/1e6/t_read*1e3:5.2f} TB/s", flush=True)

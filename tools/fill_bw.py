"""Reference points for the bandwidth-bound kernels: a pure 251 MB write (the gwc volume's size), a copy and a read-sum."""
import torch
x = torch.empty(40 * 48 * 136 * 240, device="cuda")
y = torch.empty_like(x)


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


mb = x.numel() * 4 / 1e6
ms = timeit(lambda: x.fill_(1.0)); print(f"fill  {mb:.0f} MB: {ms*1e3:.1f} us  {mb/ms/1e3:.2f} TB/s")
ms = timeit(lambda: y.copy_(x)); print(f"copy  {2*mb:.0f} MB: {ms*1e3:.1f} us  {2*mb/ms/1e3:.2f} TB/s")
ms = timeit(lambda: x.sum()); print(f"sum   {mb:.0f} MB: {ms*1e3:.1f} us  {mb/ms/1e3:.2f} TB/s")

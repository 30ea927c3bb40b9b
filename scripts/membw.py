"""Reference points for streaming kernels on this GPU: time of torch reductions / copies of a given size."""
import sys
import torch

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
x = torch.randint(0, 1 << 20, (n,), dtype=torch.int32, device="cuda")
y = torch.empty_like(x)
for name, fn in (("sum", lambda: x.sum()), ("copy", lambda: y.copy_(x)), ("gt", lambda: torch.gt(x, 5, out=y.bool() if False else None))):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(50):
        fn()
    b.record()
    torch.cuda.synchronize()
    us = a.elapsed_time(b) * 1000 / 50
    print(f"{name}: {n * 4 / 1e6:.0f} MB in {us:.1f} us = {n * 4 / us / 1e6:.2f} TB/s (read side)")

#!/usr/bin/env python
"""HBM bandwidth probe (context for the HBM-bound layers): torch copy / in-place add on tensors larger
than the 256 MiB Infinity Cache."""
import torch
for mb in (138, 550, 1100):
    n = mb * 1024 * 1024 // 2
    x = torch.empty(n, dtype=torch.bfloat16, device="cuda").normal_()
    y = torch.empty_like(x)
    for name, fn, traffic in (("copy", lambda: y.copy_(x), 2), ("add_", lambda: x.add_(1.0), 2), ("read(sum)", lambda: x.sum(), 1),
                              ("fill", lambda: y.fill_(1.0), 1)):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print(f"{mb:5d} MB {name:10s} {ms * 1e3:8.1f} us  {traffic * n * 2 / ms / 1e9:6.2f} TB/s")

# 2 reads + 1 write (the traffic mix of a residual epilogue), tensors >> Infinity Cache
n = 550 * 1024 * 1024 // 2
a = torch.empty(n, dtype=torch.bfloat16, device="cuda").normal_()
b = torch.empty_like(a).normal_()
c = torch.empty_like(a)
for name, fn, traffic in (("c=a+b (2R+1W)", lambda: torch.add(a, b, out=c), 3), ("c=relu(a) (1R+1W)", lambda: torch.relu(a, out=c) if False else c.copy_(a).relu_(), 3)):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"  550 MB {name:20s} {ms * 1e3:8.1f} us  {traffic * n * 2 / ms / 1e9:6.2f} TB/s")

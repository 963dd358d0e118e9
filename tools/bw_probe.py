#!/usr/bin/env python3
"""What does the box deliver for plain streams?  fill_ (write only), copy_ (read + write) and a 1:7 read:write mix at the sizes of
the one-hot outputs -- the ceiling the write-dominated kernels (one_hot, fused one-hot step) are compared with (profiling aid)."""
import torch
dev = torch.device("cuda", 0)


def t_ms(fn, n=20, warm=5):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for gb in (1.2, 4.3):
    n = int(gb * 1e9)
    a = torch.empty(n, dtype=torch.uint8, device=dev)
    b = torch.empty(n, dtype=torch.uint8, device=dev)
    ai, bi = a.view(torch.int32), b.view(torch.int32)
    ms = t_ms(lambda: ai.fill_(1))
    print(f"{gb} GB fill_ (write only):        {ms:7.3f} ms  {n / ms / 1e6:7.1f} GB/s written")
    ms = t_ms(lambda: bi.copy_(ai))
    print(f"{gb} GB copy_ (read + write):      {ms:7.3f} ms  {2 * n / ms / 1e6:7.1f} GB/s moved")
    del a, b
    torch.cuda.empty_cache()

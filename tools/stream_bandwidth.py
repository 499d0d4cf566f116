"""Streaming copy / fill / read bandwidth of the box through torch (the yardstick for the HBM-bound kernels, profiles/r02_notes.md)."""
import torch, time
for mb in (64, 256, 1024):
    n = mb * 1024 * 1024 // 4
    a = torch.empty(n, device="cuda"); b = torch.empty(n, device="cuda")
    for f, name, bytes_ in ((lambda: b.copy_(a), "copy", 2 * 4 * n), (lambda: a.zero_(), "fill", 4 * n), (lambda: a.sum(), "read", 4 * n)):
        for _ in range(3): f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): f()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        print(f"{mb} MB {name}: {ms*1e3:.1f} us  {bytes_/ms/1e6:.0f} GB/s")

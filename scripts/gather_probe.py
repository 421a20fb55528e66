"""Random 8-byte gather rate of the machine vs. table size (the roofline the hash-grid lookups actually sit under).

A hash-grid corner fetch at a fine level is an 8-byte read from a random 128-byte line.  This probe times a plain
torch gather of N random 8-byte elements out of tables from L2-sized to HBM-sized and prints gathers/s and the
implied line traffic at 64 B and 128 B per miss.
"""
import torch

dev = "cuda"
N = 1 << 26
g = torch.Generator(device=dev); g.manual_seed(1)
for mb in [2, 16, 64, 256, 1024, 8192]:
    n = mb * (1 << 20) // 8
    table = torch.rand(n, device=dev, dtype=torch.float64)
    idx = torch.randint(0, n, (N,), device=dev, generator=g)
    out = torch.empty(N, device=dev, dtype=torch.float64)
    for _ in range(2):
        torch.index_select(table, 0, idx, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        torch.index_select(table, 0, idx, out=out)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 5 * 1e3
    rate = N / us * 1e6
    print(f"table {mb:5d} MB: {us:9.1f} us for {N} gathers -> {rate:.3e} gathers/s "
          f"(= {rate*64/1e12:.2f} TB/s @64B/miss, {rate*128/1e12:.2f} TB/s @128B/miss; idx+out stream {N*16/us/1e6:.2f} TB/s)")
    del table, idx, out

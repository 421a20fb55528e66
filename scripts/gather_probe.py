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

# ---- what does one miss cost: a 64-byte sector or the whole 128-byte line? -----------------------------------------
# Two 8-byte reads per work item, 1 GB table (far beyond L2 + MALL reuse).  (a) the partner sits in the OTHER 64-byte
# half of the same 128-byte line (element index ^ 8), (b) the partner is an independent random element.  If (a) costs
# what ONE gather costs, a miss brings in the full line (traffic = 128 B per miss); if it costs what (b) costs, misses
# are sector granular (64 B) and the fabric traffic of the hash-grid kernels is half of misses x 128 B.
n = 1024 * (1 << 20) // 8
table = torch.rand(n, device=dev, dtype=torch.float64)
M = 1 << 25
ia = torch.randint(0, n, (M,), device=dev, generator=g)
variants = {"single": ia, "pair_same_line_other_half": torch.stack([ia, ia ^ 8], 1).reshape(-1),
            "pair_same_sector": torch.stack([ia, ia ^ 1], 1).reshape(-1),
            "pair_independent": torch.stack([ia, torch.randint(0, n, (M,), device=dev, generator=g)], 1).reshape(-1)}
for name, idx in variants.items():
    out = torch.empty(idx.numel(), device=dev, dtype=torch.float64)
    for _ in range(2):
        torch.index_select(table, 0, idx, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        torch.index_select(table, 0, idx, out=out)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 5 * 1e3
    print(f"granularity {name:28s}: {us:9.1f} us for {M} items ({idx.numel()} reads) -> {M/us*1e6:.3e} items/s")

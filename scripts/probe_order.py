"""Where do the ~15 us of power_sampler_order_kernel go?  Times, on the bench's batch (4096 rays x 128 samples), the bins alone
(nrhip_power_sampler), the ordering pass alone (nrhip_ray_order) and the combined launch, 200 launches each (HIP events).
    python scripts/probe_order.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from neurad_studio_amd import ops  # noqa: E402


def timeit(fn, n=200):
    for _ in range(10):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def main():
    dev = torch.device("cuda:0")
    fs, o, d, area, fars = bench.make_workload(dev, 1234)
    S = bench.N_SAMPLES
    print("bins alone            %.2f us" % timeit(lambda: ops.power_sampler(None, fars, S, lam=-1.0, scaling=0.1, last_edge=20000.0)))
    print("ordering pass alone   %.2f us" % timeit(lambda: ops.ray_order(o, d, bench.STATIC_SCALE)))
    print("one combined launch   %.2f us" % timeit(lambda: ops.power_sampler_ordered(None, fars, S, o, d, bench.STATIC_SCALE, lam=-1.0,
                                                                                  scaling=0.1, last_edge=20000.0)))
    for r in (1024, 2048):
        print("ordering pass alone, %d rays  %.2f us" % (r, timeit(lambda: ops.ray_order(o[:r].contiguous(), d[:r].contiguous(), bench.STATIC_SCALE))))


if __name__ == "__main__":
    main()

"""Time the device patch / lidar-point samplers (csrc/raygen.hip) next to the reference's torch op chain run on the GPU
(restated here with plain torch ops: pixel_samplers.py:100-103,696-726 and :538-583).  python scripts/bench_pixel_samplers.py"""
import json
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neurad_studio_amd.data.pixel_samplers import lidar_point_sample, patch_sample  # noqa: E402


def timeit(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3  # us


def torch_patches(image, image_idx, u, ps, sc):
    n, h, w, _ = image.shape
    K = ps * sc
    c = (u * torch.tensor([n, h - K + 1, w - K + 1], device=u.device)).long()
    c[:, 1:] += K // 2
    off = torch.arange(-(K // 2), K // 2 + K % 2, device=u.device)
    zeros = off.new_zeros((K, K))
    rel = torch.stack((zeros, *torch.meshgrid(off, off, indexing="ij")), dim=-1)[None]
    rgb = c[:, None, None] + rel
    rays = rgb[:, sc // 2::sc, sc // 2::sc].reshape(-1, 3)
    patches = image[rgb[..., 0], rgb[..., 1], rgb[..., 2]]
    rays[:, 0] = image_idx[rays[:, 0]]
    return rays, patches


def torch_lidar(cloud, npl, lidar_idx, rays, perm, draws):
    n = npl.shape[0]
    first = torch.zeros((n,), device=cloud.device, dtype=torch.int64)
    first[1:] = torch.cumsum(npl, 0)[:-1]
    point = torch.floor(draws * npl.view(n, 1)).long()
    scan = torch.arange(n, device=cloud.device).unsqueeze(1).repeat(1, draws.shape[1])
    scan, point, first = scan[perm], point[perm], first.view(n, 1)[perm]
    idx = torch.stack((scan.flatten(), point.flatten()), -1)[:rays]
    pts = cloud[(point + first).flatten()][:rays]
    idx[:, 0] = lidar_idx[idx[:, 0]]
    return idx, pts


def main():
    res = {}
    n, h, w, ps, sc, P = 6, 1280, 1920, 32, 3, 40
    for name, image in (("uint8", torch.randint(0, 256, (n, h, w, 3), device="cuda", dtype=torch.uint8)),
                        ("fp32", torch.rand((n, h, w, 3), device="cuda"))):
        idx = torch.arange(n, device="cuda")
        u = torch.rand((P, 3), device="cuda")
        r0, p0 = torch_patches(image, idx, u, ps, sc)
        r1, _, p1 = patch_sample(image, ps, sc, uniforms=u, image_idx=idx)
        assert torch.equal(r0, r1) and torch.equal(p0, p1)
        res[f"patch_{name}"] = {"hip_us": timeit(lambda: patch_sample(image, ps, sc, uniforms=u, image_idx=idx)),
                                "torch_chain_us": timeit(lambda: torch_patches(image, idx, u, ps, sc)),
                                "rays": P * ps * ps, "patch_bytes": p1.numel() * p1.element_size()}
    n, rays = 300, 16384
    npl = torch.randint(50000, 150000, (n,), device="cuda")
    cloud = torch.randn((int(npl.sum()), 5), device="cuda")
    lidx = torch.arange(n, device="cuda")
    perm = torch.randperm(n, device="cuda")
    draws = torch.rand((n, math.ceil(rays / n)), device="cuda", dtype=torch.float64)
    i0, q0 = torch_lidar(cloud, npl, lidx, rays, perm, draws)
    i1, q1 = lidar_point_sample(cloud, npl, rays, shuffle=perm, draws=draws, lidar_idx=lidx)
    assert torch.equal(i0, i1) and torch.equal(q0, q1)
    res["lidar_points"] = {"hip_us": timeit(lambda: lidar_point_sample(cloud, npl, rays, shuffle=perm, draws=draws, lidar_idx=lidx)),
                           "torch_chain_us": timeit(lambda: torch_lidar(cloud, npl, lidx, rays, perm, draws)), "rays": rays,
                           "cloud_points": int(npl.sum())}
    print(json.dumps(res))


if __name__ == "__main__":
    main()

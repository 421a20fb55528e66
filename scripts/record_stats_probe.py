"""How many records does the radix-partition table gradient send on the c3 training step, per level, and how many could a
wider merge save?  For each proposal round (F = 1, x-pair records) and the field call (F = 4): corner-pair terms, run heads
inside 16-lane rows (what `emit` sends), inside whole 64-lane waves, inside whole rays (consecutive samples, any length), and
DISTINCT (entry pairs) per 4096-sample chunk (what a perfect workgroup-level merge would send).
python scripts/record_stats_probe.py   (on the GPU box)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import bench
from neurad_studio_amd import autograd as ag
from neurad_studio_amd.cameras.rays import RayBundle
from neurad_studio_amd.models.neurad import NeuRADHotPath, NeuRADHotPathConfig

dev = torch.device("cuda:0")
torch.manual_seed(11)
m = NeuRADHotPath(NeuRADHotPathConfig(), static_scale=bench.STATIC_SCALE, num_sensors=7, duration=8.0).to(dev).train()
with torch.no_grad():
    m.field.hashgrid.static_grid.hash_table.mul_(1000.0)
    for p in m.proposal_fields:
        p.hashgrid.static_grid.hash_table.mul_(2000.0)
o, d, area, times, md = bench.joint_batch(dev, 0, bench.C3_CAMERA_RAYS, bench.C3_LIDAR_RAYS)
calls = []
real_round, real_render = ag.ProposalRoundFn.apply, m.field.render_train


def spy_round(table, dec, spec, scale, o_, d_, a_, eu):
    calls.append(("proposal", spec, scale, eu.detach().clone()))
    return real_round(table, dec, spec, scale, o_, d_, a_, eu)


def spy_render(o_, d_, a_, eu, *a, **k):
    calls.append(("field", m.field.hashgrid.static_grid.spec, m.field.hashgrid.static_scale, eu.detach().clone()))
    return real_render(o_, d_, a_, eu, *a, **k)


ag.ProposalRoundFn.apply, m.field.render_train = spy_round, spy_render
rb = RayBundle(origins=o, directions=d, pixel_area=area.clone(), nears=torch.zeros_like(area), fars=None, times=times, metadata=dict(md))
m.get_nff_outputs(rb, calc_lidar_losses=True)
PY, PZ = 2654435761, 805459861


def positions(eu, scale):
    t = (eu[:, :-1] + eu[:, 1:]) * 0.5
    x = (o[:, None, :] + d[:, None, :] * t[..., None]) / scale
    mag = x.abs().amax(-1, keepdim=True)
    x = torch.where(mag < 1, x, (2 - 1 / mag.clamp_min(1)) * (x / mag.clamp_min(1)))
    return (x + 2) / 4


for name, spec, scale, eu in calls:
    R, S = eu.shape[0], eu.shape[1] - 1
    x01 = positions(eu, float(scale)).reshape(-1, 3).double()
    T = 1 << spec.log2_hashmap_size
    lane = torch.arange(R * S, device=dev)
    print(f"== {name}: {R} rays x {S} samples, L={spec.num_levels} F={spec.features_per_level} T=2^{spec.log2_hashmap_size}")
    tot = dict(terms=0, row16=0, wave64=0, ray=0, chunk=0)
    for l in range(spec.num_levels):
        sc = float(spec.scalings[l])
        f = torch.floor(x01 * sc).long()
        c = torch.ceil(x01 * sc).long()
        keys = []
        for yy in (f[:, 1], c[:, 1]):
            for zz in (f[:, 2], c[:, 2]):
                h = (yy * PY) ^ (zz * PZ)
                keys.append((((f[:, 0] ^ h) & (T - 1)) << 24) | ((f[:, 0] ^ c[:, 0]) & 0xffffff))  # pair id: floor entry + xm
        st = dict(terms=0, row16=0, wave64=0, ray=0, chunk=0)
        # transposed mapping, camera rays only: a 16-lane row = 16 neighbouring rays at ONE sample index
        nc = bench.C3_CAMERA_RAYS
        tr = dict(cam_terms=0, cam_row16=0, cam_across16=0, cam_across32=0, lid_row16=0, lid_terms=0)
        for k in keys:
            k2 = k.view(R, S)
            cam = k2[:nc]
            tr["cam_terms"] += cam.numel()
            same_a = torch.zeros_like(cam, dtype=torch.bool)
            same_a[:, 1:] = cam[:, 1:] == cam[:, :-1]
            col = torch.arange(S, device=dev)[None, :]
            tr["cam_row16"] += int((~(same_a & (col % 16 != 0))).sum())
            same_x = torch.zeros_like(cam, dtype=torch.bool)
            same_x[1:] = cam[1:] == cam[:-1]
            rr = torch.arange(nc, device=dev)[:, None]
            tr["cam_across16"] += int((~(same_x & (rr % 16 != 0))).sum())
            tr["cam_across32"] += int((~(same_x & (rr % 32 != 0))).sum())
            lid = k2[nc:]
            same_l = torch.zeros_like(lid, dtype=torch.bool)
            same_l[:, 1:] = lid[:, 1:] == lid[:, :-1]
            tr["lid_terms"] += lid.numel()
            tr["lid_row16"] += int((~(same_l & (col % 16 != 0))).sum())
        for k in keys:
            same = torch.zeros_like(k, dtype=torch.bool)
            same[1:] = k[1:] == k[:-1]
            st["terms"] += k.numel()
            st["row16"] += int((~(same & (lane % 16 != 0))).sum())
            st["wave64"] += int((~(same & (lane % 64 != 0))).sum())
            st["ray"] += int((~(same & (lane % S != 0))).sum())
        # distinct pairs per 4096-sample chunk, all four (y, z) pairs together
        allk = torch.stack(keys, 1)
        n_chunks = (R * S + 4095) // 4096
        pad = n_chunks * 4096 - R * S
        if pad:
            allk = torch.cat([allk, allk[-1:].expand(pad, 4)])
        ck = allk.view(n_chunks, -1)
        srt = ck.sort(dim=1).values
        st["chunk"] = int((srt[:, 1:] != srt[:, :-1]).sum()) + n_chunks
        for kk in tot:
            tot[kk] += st[kk]
        print(f"  level {l} (res {sc:7.0f}): terms {st['terms']/1e6:7.2f} M | row16 {st['row16']/st['terms']:.3f} | wave64 "
              f"{st['wave64']/st['terms']:.3f} | whole ray {st['ray']/st['terms']:.3f} | distinct per 4096-chunk {st['chunk']/st['terms']:.3f}"
              f" || camera rays: along-ray row16 {tr['cam_row16']/tr['cam_terms']:.3f}, ACROSS 16 rays {tr['cam_across16']/tr['cam_terms']:.3f}, "
              f"across 32 {tr['cam_across32']/tr['cam_terms']:.3f}; lidar along-ray {tr['lid_row16']/max(tr['lid_terms'],1):.3f}")
    print(f"  all levels: terms {tot['terms']/1e6:.1f} M | row16 {tot['row16']/tot['terms']:.3f} | wave64 {tot['wave64']/tot['terms']:.3f} | "
          f"whole ray {tot['ray']/tot['terms']:.3f} | distinct per chunk {tot['chunk']/tot['terms']:.3f}")

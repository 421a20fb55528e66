"""What would an exact per-row deduplication of the table-gradient records buy on the c3 step's camera rays?
For every 16-lane row of the coherent walk (16 neighbouring rays of a patch row at one sample index) and every level:
  same-slot   run heads per corner slot, summed over the 8 slots (what `count` / `emit` send across a row today, before the
              in-thread merge of the quad walk)
  block       distinct lattice points touched by the row's 16 samples where the row spans at most 2 cells per axis (its
              samples then lie in a 3 x 3 x 3 lattice block), else same-slot
  distinct    distinct lattice points of the row whatever it spans (the floor of any row-local merge)
python scripts/lattice_block_probe.py   (on the GPU box)"""
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
nc = bench.C3_CAMERA_RAYS


def positions(eu, scale):
    t = (eu[:, :-1] + eu[:, 1:]) * 0.5
    x = (o[:, None, :] + d[:, None, :] * t[..., None]) / scale
    mag = x.abs().amax(-1, keepdim=True)
    x = torch.where(mag < 1, x, (2 - 1 / mag.clamp_min(1)) * (x / mag.clamp_min(1)))
    return (x + 2) / 4


for name, spec, scale, eu in calls:
    R, S = eu.shape[0], eu.shape[1] - 1
    x01 = positions(eu, float(scale))[:nc].double()            # camera rays only: [nc, S, 3]
    tot = dict(terms=0, same=0, block=0, distinct=0, rows=0, fit=0)
    print(f"== {name}: {nc} camera rays x {S} samples, L={spec.num_levels} F={spec.features_per_level}")
    for l in range(spec.num_levels):
        sc = float(spec.scalings[l])
        cell = torch.floor(x01 * sc).long()                     # [nc, S, 3]
        rows = cell.view(nc // 16, 16, S, 3).permute(0, 2, 1, 3).reshape(-1, 16, 3)   # [rows, 16 lanes, 3]
        nrow = rows.shape[0]
        # same-slot run heads: a lane heads a run in a slot iff its cell differs from the left lane's (all 8 slots alike)
        head = torch.ones(nrow, 16, dtype=torch.bool, device=dev)
        head[:, 1:] = (rows[:, 1:] != rows[:, :-1]).any(-1)
        same = head.sum(1) * 8                                   # records per row today (same-slot merge)
        # distinct lattice points of the row
        offs = torch.tensor([[a, b, c] for a in (0, 1) for b in (0, 1) for c in (0, 1)], device=dev)
        pts = (rows[:, :, None, :] + offs[None, None]).reshape(nrow, 128, 3)
        key = (pts[..., 0] << 42) | (pts[..., 1] << 21) | pts[..., 2]
        ks = key.sort(1).values
        distinct = (ks[:, 1:] != ks[:, :-1]).sum(1) + 1
        span = rows.amax(1) - rows.amin(1)                       # [rows, 3]
        fit = (span <= 1).all(-1)
        block = torch.where(fit, distinct, same)
        st = dict(terms=nrow * 128, same=int(same.sum()), block=int(block.sum()), distinct=int(distinct.sum()), rows=nrow,
                  fit=int(fit.sum()))
        for k in tot:
            tot[k] += st[k]
        print(f"  level {l} (res {sc:6.0f}): same-slot {st['same'] / st['terms']:.3f} | 3x3x3 block {st['block'] / st['terms']:.3f} "
              f"(rows that fit {st['fit'] / nrow:.2f}) | row-distinct {st['distinct'] / st['terms']:.3f}")
    print(f"  all levels: same-slot {tot['same'] / tot['terms']:.3f} | block {tot['block'] / tot['terms']:.3f} (fit "
          f"{tot['fit'] / tot['rows']:.2f}) | row-distinct {tot['distinct'] / tot['terms']:.3f}  => block / same-slot = "
          f"{tot['block'] / tot['same']:.3f}")

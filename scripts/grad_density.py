"""How sparse is the table gradient of the c3 training step?  (Round-2 review, item 9: a sparse-aware gradient exchange pays
only if few entries are touched.)  One training step of bench.py's config[3] model on its joint batch, then per table and
level: fraction of rows with a non-zero gradient, fraction of 128 KB slices with any, and the bytes a list exchange
(all-gather of (int32 row, F values) from every rank) would move against the dense reduce-scatter + all-gather."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    from neurad_studio_amd.cameras.rays import RayBundle
    from neurad_studio_amd.model_components.losses import distortion_loss, zipnerf_interlevel_loss
    from neurad_studio_amd.models.neurad import NeuRADHotPath, NeuRADHotPathConfig

    torch.manual_seed(11)
    cfg = NeuRADHotPathConfig()
    m = NeuRADHotPath(cfg, static_scale=bench.STATIC_SCALE, num_sensors=7, duration=8.0).to(dev).train()
    with torch.no_grad():
        m.field.hashgrid.static_grid.hash_table.mul_(1000.0)
        for p in m.proposal_fields:
            p.hashgrid.static_grid.hash_table.mul_(2000.0)
    n_cam, n_lidar = bench.C3_CAMERA_RAYS, bench.C3_LIDAR_RAYS
    o, d, area, times, md = bench.joint_batch(dev, 0, n_cam, n_lidar)
    R = n_cam + n_lidar
    rb = RayBundle(origins=o, directions=d, pixel_area=area.clone(), nears=torch.zeros((R, 1), device=dev), fars=None,
                   times=times, metadata=dict(md))
    out = m.get_nff_outputs(rb, calc_lidar_losses=True)
    loss = (out["features"].square().mean() + out["depth"].mean() * 1e-2
            + 1e-3 * zipnerf_interlevel_loss(out["weights_list"], out["ray_samples_list"])
            + 2e-3 * distortion_loss(out["weights_list"], out["ray_samples_list"]))
    loss.backward()
    res = {"rays": R, "what": "bench.py config[3] model and joint batch, one training step"}
    tables = {"field": m.field.hashgrid.static_grid}
    for i, p in enumerate(m.proposal_fields):
        tables[f"proposal_{i}"] = p.hashgrid.static_grid
    for name, g in tables.items():
        t = g.hash_table
        if t.grad is None:
            res[name] = "no gradient (never evaluated in this step)"
            continue
        L, F = g.num_levels, g.features_per_level
        T = t.shape[0] // L
        nz = (t.grad.reshape(L, T, F) != 0).any(-1)
        rows_per_slice = (128 * 1024) // (F * t.element_size())
        sl = nz.reshape(L, -1, min(rows_per_slice, T)).any(-1)
        frac = nz.float().mean(1).tolist()
        n = int(nz.sum())
        dense = t.numel() * 4
        entry = {"levels": L, "rows_per_level": T, "features": F, "touched_fraction_per_level": [round(x, 4) for x in frac],
                 "touched_rows": n, "touched_fraction": n / (L * T), "touched_128KB_slices_fraction": float(sl.float().mean()),
                 "dense_bytes": dense, "list_bytes_per_rank": n * (4 + 4 * F)}
        for N in (2, 4, 8):
            entry[f"N{N}"] = {"dense_wire_bytes_per_rank": 2 * (N - 1) / N * dense,
                              "list_allgather_wire_bytes_per_rank": (N - 1) * n * (4 + 4 * F)}
        res[name] = entry
    print(json.dumps(res))


if __name__ == "__main__":
    main()

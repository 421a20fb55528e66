"""Golden vectors for device ray generation from the reference's own generators -- build container
only:  python oracle/make_golden_raygen.py  -> tests/golden/raygen.npz

  cameras: Cameras.generate_rays(camera_indices, coords) for 6 PERSPECTIVE cameras with rolling-shutter metadata
           (nerfstudio/cameras/cameras.py:560-968), PandaSet-style (rows) and both horizontal directions;
  lidars : Lidars.generate_rays(lidar_indices, points) (cameras/lidars.py:399-460), ego-compensated and not."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

import ref_import

ref_import.install()
import synth  # noqa: E402
from make_golden import T, save  # noqa: E402
from nerfstudio.cameras.cameras import Cameras, CameraType  # noqa: E402
from nerfstudio.cameras.lidars import Lidars, LidarType  # noqa: E402


def poses(n, seed):
    """[n,3,4] rigid poses: random rotations (QR of a hashed matrix) + translations along a drive"""
    out = []
    for i in range(n):
        q, _ = np.linalg.qr(synth.normal((3, 3), seed + i).astype(np.float64))
        if np.linalg.det(q) < 0:
            q[:, 0] *= -1
        t = np.array([4.0 * i, 0.3 * i, 1.6]) + synth.normal((3,), seed + 100 + i) * 0.1
        out.append(np.concatenate([q, t[:, None]], 1))
    return np.stack(out).astype(np.float32)


def main():
    C = 6
    c2w = poses(C, 10)
    fx = synth.uniform((C, 1), 1800, 2000, 1)
    fy = synth.uniform((C, 1), 1800, 2000, 2)
    cx = synth.uniform((C, 1), 940, 980, 3)
    cy = synth.uniform((C, 1), 520, 560, 4)
    times = synth.uniform((C, 1), 0, 8, 5)
    md = {"rolling_shutter_time": T(synth.uniform((C, 1), 0.01, 0.03, 6)),
          "time_to_center_pixel": T(synth.uniform((C, 1), -0.01, 0.01, 7)), "velocities": T(synth.normal((C, 3), 8) * 5),
          "sensor_idxs": torch.arange(C)[:, None]}
    R = 512
    cam_idx = (synth.uniform((R,), 0, C, 9)).astype(np.int64).clip(0, C - 1)
    coords = np.stack([np.floor(synth.uniform((R,), 0, 1080, 11)) + 0.5, np.floor(synth.uniform((R,), 0, 1920, 12)) + 0.5],
                      -1).astype(np.float32)
    gold = dict(c2w=c2w, fx=fx, fy=fy, cx=cx, cy=cy, cam_times=times, rolling_shutter_time=md["rolling_shutter_time"],
                time_to_center_pixel=md["time_to_center_pixel"], cam_velocities=md["velocities"], cam_idx=cam_idx,
                coords=coords)
    for tag, direction in (("rows", None), ("cols", "Horizontal"), ("cols_rev", "Horizontal_reversed")):
        meta = dict(md)
        cams = Cameras(camera_to_worlds=T(c2w), fx=T(fx), fy=T(fy), cx=T(cx), cy=T(cy), width=1920, height=1080,
                       camera_type=CameraType.PERSPECTIVE, times=T(times), metadata=meta)
        rb = cams.generate_rays(camera_indices=torch.from_numpy(cam_idx)[:, None], coords=T(coords))
        if direction is not None:  # the reference reads rs_direction from the per-ray metadata copy (cameras.py:941-953)
            # a string cannot live in Cameras.metadata (tensors only); emulate by re-running with the branch forced
            import nerfstudio.cameras.cameras as cam_mod

            orig = cam_mod.Cameras._apply_fn_to_dict

            def patched(self, d, fn, _orig=orig, _dir=direction):
                out = _orig(self, d, fn)
                out["rs_direction"] = _dir
                return out

            cam_mod.Cameras._apply_fn_to_dict = patched
            try:
                rb = cams.generate_rays(camera_indices=torch.from_numpy(cam_idx)[:, None], coords=T(coords))
            finally:
                cam_mod.Cameras._apply_fn_to_dict = orig
        gold.update({f"cam_{tag}_origins": rb.origins, f"cam_{tag}_directions": rb.directions,
                     f"cam_{tag}_pixel_area": rb.pixel_area, f"cam_{tag}_times": rb.times,
                     f"cam_{tag}_directions_norm": rb.metadata["directions_norm"]})
    # ---- lidars ----
    Ln = 4
    l2w = poses(Ln, 40)
    ltimes = synth.uniform((Ln, 1), 0, 8, 41)
    lvel = synth.normal((Ln, 3), 42) * 6
    Rl = 600
    lidx = (synth.uniform((Rl,), 0, Ln, 43)).astype(np.int64).clip(0, Ln - 1)
    pts = np.concatenate([synth.normal((Rl, 3), 44) * np.array([30.0, 30.0, 2.0], np.float32), synth.uniform((Rl, 1), 0, 1, 45),
                          synth.uniform((Rl, 1), -0.05, 0.05, 46)], -1).astype(np.float32)
    pts[:5, :3] *= 60.0  # a few beyond the valid-distance threshold
    gold.update(l2w=l2w, lidar_times=ltimes, lidar_velocities=lvel, lidar_idx=lidx, points=pts)
    for tag, ego in (("ego", True), ("noego", False)):
        lid = Lidars(lidar_to_worlds=T(l2w), lidar_type=LidarType.VELODYNE64E, assume_ego_compensated=ego, times=T(ltimes),
                     metadata={"velocities": T(lvel)}, valid_lidar_distance_threshold=1000.0)
        rb = lid.generate_rays(lidar_indices=torch.from_numpy(lidx)[:, None], points=T(pts))
        gold.update({f"lid_{tag}_origins": rb.origins, f"lid_{tag}_directions": rb.directions,
                     f"lid_{tag}_pixel_area": rb.pixel_area, f"lid_{tag}_times": rb.times,
                     f"lid_{tag}_distance": rb.metadata["directions_norm"],
                     f"lid_{tag}_did_return": rb.metadata["did_return"]})
        gold["hdiv"], gold["vdiv"] = lid.horizontal_beam_divergence, lid.vertical_beam_divergence
    save("raygen", **gold)


if __name__ == "__main__":
    main()

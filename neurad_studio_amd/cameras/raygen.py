"""Ray generation on the device: drop-in for ``Cameras.generate_rays(camera_indices, coords)`` and
``Lidars.generate_rays(lidar_indices, points)`` of the reference (nerfstudio/cameras/cameras.py:560-968,
cameras/lidars.py:399-460), one HIP kernel each (csrc/raygen.hip).

``cameras`` / ``lidars`` are the reference's own objects (or anything with the same tensor attributes: camera_to_worlds,
fx, fy, cx, cy, width, height, times, metadata, camera_type, distortion_params | lidar_to_worlds, times, metadata,
horizontal_beam_divergence, vertical_beam_divergence, assume_ego_compensated, valid_lidar_distance_threshold).  The
result is a RayBundle of the class passed as ``bundle_cls`` (default: this package's; pass the reference's to stay
inside nerfstudio types)."""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch
from torch import Tensor

from .. import _lib
from ..ops import _chk, _ptr, _stream
from .rays import RayBundle

PERSPECTIVE = 1  # CameraType.PERSPECTIVE.value (cameras/cameras.py:43-55)


def _f32(t: Tensor) -> Tensor:
    return _chk(t.reshape(t.shape[0], -1).to(torch.float32), "sensor table")


_ELIGIBLE: dict = {}  # id(cameras) -> (key, error message | None)


def _check_cameras(cameras) -> None:
    """PERSPECTIVE, undistorted cameras only.  The two `.any()` checks read the device, i.e. they synchronise the host with
    the stream: they run ONCE per Cameras object (keyed on the identity and in-place version of its two tensors), not once
    per training step."""
    ct, dp = cameras.camera_type, getattr(cameras, "distortion_params", None)
    key = tuple((t.data_ptr(), t._version, tuple(t.shape)) if isinstance(t, Tensor) else None for t in (ct, dp))
    hit = _ELIGIBLE.get(id(cameras))
    if hit is None or hit[0] != key:
        err = None
        if ct is not None and bool((ct != PERSPECTIVE).any()):
            err = "device ray generation covers PERSPECTIVE cameras; use Cameras.generate_rays otherwise"
        elif dp is not None and bool((dp != 0).any()):
            err = "device ray generation covers undistorted cameras; use Cameras.generate_rays otherwise"
        if len(_ELIGIBLE) >= 64:
            _ELIGIBLE.clear()
        hit = _ELIGIBLE[id(cameras)] = (key, err)
    if hit[1] is not None:
        raise NotImplementedError(hit[1])


def camera_rays(cameras, camera_indices: Tensor, coords: Tensor, bundle_cls=RayBundle):
    """camera_indices [R,1] (or [R]) long, coords [R,2] = (y, x) pixel-centre coordinates."""
    _check_cameras(cameras)
    idx = _chk(camera_indices.reshape(-1).long(), "camera_indices", torch.int64)
    xy = _chk(coords.reshape(-1, 2).to(torch.float32), "coords")
    R, dev = idx.shape[0], idx.device
    keep = [_f32(cameras.camera_to_worlds), _f32(cameras.fx), _f32(cameras.fy), _f32(cameras.cx), _f32(cameras.cy)]
    t = _lib.CameraTable()
    t.camera_to_worlds, t.fx, t.fy, t.cx, t.cy = (k.data_ptr() for k in keep)
    times_tab = None if cameras.times is None else _f32(cameras.times)
    t.times = 0 if times_tab is None else times_tab.data_ptr()
    md = cameras.metadata or {}
    rs = all(k in md for k in ("rolling_shutter_time", "time_to_center_pixel", "velocities"))
    if rs:
        direction = md.get("rs_direction")
        t.rolling_shutter = {"Horizontal": 2, "Horizontal_reversed": 3}.get(direction, 1)
        extent = _f32((cameras.height if t.rolling_shutter == 1 else cameras.width).to(torch.float32))
        keep += [_f32(md["rolling_shutter_time"]), _f32(md["time_to_center_pixel"]), _f32(md["velocities"]), extent]
        t.rolling_shutter_time, t.time_to_center_pixel, t.velocities, t.shutter_extent = (k.data_ptr() for k in keep[-4:])
    o = torch.empty((R, 3), device=dev)
    d = torch.empty((R, 3), device=dev)
    area = torch.empty((R, 1), device=dev)
    norm = torch.empty((R, 1), device=dev)
    times = None if times_tab is None else torch.empty((R, 1), device=dev)
    _lib.call("nrhip_camera_rays", C.byref(t), _ptr(idx), _ptr(xy), R, _ptr(o), _ptr(d), _ptr(area), _ptr(norm), _ptr(times),
              _stream())
    skip = ("rolling_shutter_time", "time_to_center_pixel", "rs_direction") if rs else ()
    metadata = {k: v[idx] for k, v in md.items() if isinstance(v, Tensor) and k not in skip}
    metadata["directions_norm"] = norm
    return bundle_cls(origins=o, directions=d, pixel_area=area, camera_indices=idx[:, None], times=times, metadata=metadata,
                      fars=torch.full_like(area, 1_000_000.0))


def lidar_rays(lidars, lidar_indices: Tensor, points: Tensor, bundle_cls=RayBundle):
    """lidar_indices [R,1] (or [R]) long, points [R, >=4]: xyz (lidar frame), intensity, time offset in the sweep."""
    idx = _chk(lidar_indices.reshape(-1).long(), "lidar_indices", torch.int64)
    pts = _chk(points.reshape(idx.shape[0], -1).to(torch.float32), "points")
    R, dev = idx.shape[0], idx.device
    md = lidars.metadata or {}
    keep = [_f32(lidars.lidar_to_worlds), _f32(lidars.horizontal_beam_divergence), _f32(lidars.vertical_beam_divergence)]
    t = _lib.LidarTable()
    t.lidar_to_worlds, t.horizontal_beam_divergence, t.vertical_beam_divergence = (k.data_ptr() for k in keep)
    times_tab = None if lidars.times is None else _f32(lidars.times)
    vel = _f32(md["velocities"]) if "velocities" in md else None
    t.times = 0 if times_tab is None else times_tab.data_ptr()
    t.velocities = 0 if vel is None else vel.data_ptr()
    t.assume_ego_compensated = int(bool(lidars.assume_ego_compensated))
    t.valid_lidar_distance_threshold = float(lidars.valid_lidar_distance_threshold)
    o = torch.empty((R, 3), device=dev)
    d = torch.empty((R, 3), device=dev)
    area = torch.empty((R, 1), device=dev)
    dist = torch.empty((R, 1), device=dev)
    ret = torch.empty((R, 1), device=dev, dtype=torch.uint8)
    times = None if times_tab is None else torch.empty((R, 1), device=dev)
    _lib.call("nrhip_lidar_rays", C.byref(t), _ptr(idx), _ptr(pts), pts.shape[1], R, _ptr(o), _ptr(d), _ptr(area), _ptr(dist),
              _ptr(ret), _ptr(times), _stream())
    metadata = {k: v[idx] for k, v in md.items() if isinstance(v, Tensor)}
    metadata.update(directions_norm=dist, is_lidar=torch.ones((R, 1), dtype=torch.bool, device=dev), did_return=ret.bool())
    return bundle_cls(origins=o, directions=d, pixel_area=area, camera_indices=idx[:, None], times=times, metadata=metadata,
                      fars=torch.full_like(area, 1_000_000.0))

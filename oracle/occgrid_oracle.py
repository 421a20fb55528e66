"""CPU restatement (numpy / python loops) of the occupancy-grid ray march -- TEST INFRASTRUCTURE ONLY.

Row S6 of SURVEY.md §8: VolumetricSampler.forward -> nerfacc OccGridEstimator.sampling
(nerfstudio/model_components/ray_samplers.py:483-566).  nerfacc 0.5.2 is un-vendored, the reference tree holds no
test or golden for it and nothing instantiates VolumetricSampler -> PARITY UNPINNED.  This file states the marching
rule the HIP kernel (csrc/occgrid.hip) implements; the GPU tests check the kernel against it plus invariants."""
import numpy as np

f32 = np.float32


def occgrid_march(aabb, binaries, origins, dirs, step, near_plane=0.0, far_plane=1e10, t_min=None, t_max=None,
                  cone_angle=0.0, t_rand=None, max_candidates=1 << 16):
    res = binaries.shape[0]
    lo, hi = np.asarray(aabb[:3], f32), np.asarray(aabb[3:], f32)
    ri, ts_out, te_out = [], [], []
    for r in range(origins.shape[0]):
        o, d = origins[r].astype(f32), dirs[r].astype(f32)
        near = max(f32(near_plane), f32(t_min[r]) if t_min is not None else f32(near_plane))
        far = min(f32(far_plane), f32(t_max[r]) if t_max is not None else f32(far_plane))
        if t_rand is not None:
            near = f32(near + f32(t_rand[r]) * f32(step))
        tn, tf, ok = f32(near), f32(far), True
        for a in range(3):
            if d[a] == 0:
                ok &= bool(lo[a] <= o[a] <= hi[a])
                continue
            inv = f32(1) / d[a]
            ta, tb = (lo[a] - o[a]) * inv, (hi[a] - o[a]) * inv
            if ta > tb:
                ta, tb = tb, ta
            tn, tf = max(tn, ta), min(tf, tb)
        if not ok or not (tn < tf):
            continue
        k1, t1 = 0, tn
        if cone_angle > 0 and tn * f32(cone_angle) < f32(step):
            k1 = int(np.ceil((f32(step) / f32(cone_angle) - tn) / f32(step)))
            t1 = f32(tn + f32(k1) * f32(step))

        def at(k):
            if cone_angle <= 0 or k < k1:
                return f32(tn + f32(k) * f32(step))
            return f32(t1 * np.power(f32(1) + f32(cone_angle), f32(k - k1), dtype=f32))

        for k in range(max_candidates):
            ts = at(k)
            if not ts < tf:
                break
            te = min(at(k + 1), tf)
            if not te > ts:
                continue
            p = o + d * (f32(0.5) * (ts + te))
            u = (p - lo) / (hi - lo)
            idx = np.clip(np.floor(u * f32(res)).astype(np.int64), 0, res - 1)
            if binaries[idx[0], idx[1], idx[2]]:
                ri.append(r), ts_out.append(ts), te_out.append(te)
    return np.asarray(ri, np.int64), np.asarray(ts_out, f32), np.asarray(te_out, f32)


def packed_visibility_from_alpha(alphas, segments, early_stop_eps, alpha_thre):
    mask = np.zeros(alphas.shape[0], bool)
    for r in range(len(segments) - 1):
        T = 1.0
        for i in range(segments[r], segments[r + 1]):
            mask[i] = (T >= early_stop_eps) and (alphas[i] >= alpha_thre)
            T *= 1.0 - float(alphas[i])
    return mask

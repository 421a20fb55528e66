"""ctypes wrapper of oracle/neurad_oracle_c.c (checker / CPU baseline only; never imported by the product)."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "neurad_oracle_c.c")
LIB = os.path.join(HERE, "_build", "libneurad_oracle.so")
FP = C.POINTER(C.c_float)


class NroField(C.Structure):
    _fields_ = [("L", C.c_int), ("F", C.c_int), ("log2T", C.c_int), ("scalings", FP), ("table", FP),
                ("static_scale", C.c_float), ("H", C.c_int), ("gw0", FP), ("gb0", FP), ("gw1", FP), ("gb1", FP),
                ("fw0", FP), ("fb0", FP), ("fw1", FP), ("fb1", FP), ("fw2", FP), ("fb2", FP), ("use_sdf", C.c_int),
                ("beta", C.c_float)]


def _cpu_tag() -> str:
    """identifies the instruction set of THIS host: the library is built with -march=native, and the copy that travels
    with the repo snapshot to another machine (the GPU box) must not be reused there if the CPUs differ (SIGILL)"""
    import hashlib

    try:
        with open("/proc/cpuinfo") as f:
            flags = next((l for l in f if l.startswith("flags")), "")
    except OSError:
        flags = ""
    return hashlib.sha1(" ".join(sorted(flags.split(":")[-1].split())).encode()).hexdigest()[:16]


def build() -> str:
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    tag_file, tag = LIB + ".cpu", _cpu_tag()
    built_for = open(tag_file).read().strip() if os.path.exists(tag_file) else None
    if not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(SRC) or built_for != tag:
        subprocess.run(["gcc", "-O3", "-march=native", "-fopenmp", "-fPIC", "-shared", "-std=gnu11", SRC, "-o", LIB,
                        "-lm"], check=True)
        with open(tag_file, "w") as f:
            f.write(tag)
    return LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        _lib.nro_num_threads.restype = C.c_int
    return _lib


def _p(a):
    return a.ctypes.data_as(FP)


def render_fwd(p, origins, directions, pixel_area, starts, ends, per_sample=False):
    """p: neurad_oracle.FieldParams (geo 32->H->33, feat 48->H->H->32).  Same outputs as neurad_oracle.render_rays."""
    f32 = np.float32
    keep = {k: np.ascontiguousarray(v, f32) for k, v in dict(
        table=p.grid.table, scal=p.grid.scalings, gw0=p.geo_w[0], gb0=p.geo_b[0], gw1=p.geo_w[1], gb1=p.geo_b[1],
        fw0=p.feat_w[0], fb0=p.feat_b[0], fw1=p.feat_w[1], fb1=p.feat_b[1], fw2=p.feat_w[2], fb2=p.feat_b[2],
        o=origins, d=directions, a=pixel_area, s=starts, e=ends).items()}
    f = NroField()
    f.L, f.F, f.log2T = p.grid.num_levels, p.grid.n_feat, p.grid.log2_hashmap_size
    f.scalings, f.table, f.static_scale = _p(keep["scal"]), _p(keep["table"]), p.static_scale
    f.H = p.geo_w[0].shape[0]
    for n in ("gw0", "gb0", "gw1", "gb1", "fw0", "fb0", "fw1", "fb1", "fw2", "fb2"):
        setattr(f, n, _p(keep[n]))
    f.use_sdf, f.beta = int(p.use_sdf), abs(p.beta) + p.beta_min
    R, S = keep["s"].shape
    feat, depth, acc = np.empty((R, 32), f32), np.empty((R, 1), f32), np.empty((R, 1), f32)
    w = np.empty((R, S), f32)
    feature = np.empty((R, S, 32), f32) if per_sample else None
    head = np.empty((R, S), f32) if per_sample else None
    rc = lib().nro_render_fwd(C.byref(f), C.c_int64(R), S, S, _p(keep["o"]), _p(keep["d"]), _p(keep["a"]),
                              _p(keep["s"]), _p(keep["e"]), _p(feat), _p(depth), _p(acc), _p(w),
                              _p(feature) if per_sample else None, _p(head) if per_sample else None)
    assert rc == 0, "unsupported configuration for the C oracle"
    out = {"features": feat, "depth": depth, "accumulation": acc, "weights": w}
    if per_sample:
        out.update(feature=feature, head=head)
    return out


def _grid_field(grid, static_scale, keep):
    """an NroField of which only the grid members are set (what the proposal chain reads)"""
    f32 = np.float32
    keep.append((np.ascontiguousarray(grid.table, f32), np.ascontiguousarray(grid.scalings, f32)))
    f = NroField()
    f.L, f.F, f.log2T = grid.num_levels, grid.n_feat, grid.log2_hashmap_size
    f.table, f.scalings, f.static_scale = _p(keep[-1][0]), _p(keep[-1][1]), static_scale
    return f


def proposal_sampler(props, origins, directions, pixel_area, nears, fars, num_proposal_samples=(128, 64),
                     num_nerf_samples=32, lam=-1.0, scaling=0.1, sky_distance=20000.0, late_binding_quirk=True,
                     stretch_sky=True):
    """Eval-mode chain; same arguments and outputs (starts, ends, prop_weights, prop_starts, prop_ends) as
    neurad_oracle.proposal_sampler.  props: neurad_oracle.ProposalParams."""
    f32 = np.float32
    n = len(num_proposal_samples)
    keep = []
    used = [props[-1] if late_binding_quirk else props[i] for i in range(n)]
    fields = [_grid_field(p.grid, p.static_scale, keep) for p in used]
    decs = [np.ascontiguousarray(p.decoder_w, f32).reshape(-1) for p in used]
    o, d, a = (np.ascontiguousarray(v, f32) for v in (origins, directions, pixel_area))
    R = o.shape[0]
    nr, fr = (np.ascontiguousarray(np.broadcast_to(np.asarray(v, f32).reshape(-1), (R,))) for v in (nears, fars))
    starts, ends = np.empty((R, num_nerf_samples), f32), np.empty((R, num_nerf_samples), f32)
    pw = [np.empty((R, k), f32) for k in num_proposal_samples]
    pe = [np.empty((R, k + 1), f32) for k in num_proposal_samples]
    FieldPtr = C.POINTER(NroField)
    rc = lib().nro_proposal_sampler((FieldPtr * n)(*[C.pointer(f) for f in fields]), (FP * n)(*[_p(x) for x in decs]), n,
                                    (C.c_int * n)(*num_proposal_samples), num_nerf_samples, C.c_int64(R), _p(o), _p(d),
                                    _p(a.reshape(-1)), _p(nr), _p(fr), C.c_float(lam), C.c_float(scaling),
                                    C.c_float(sky_distance), int(stretch_sky), _p(starts), _p(ends),
                                    (FP * n)(*[_p(x) for x in pw]), (FP * n)(*[_p(x) for x in pe]))
    assert rc == 0, "unsupported configuration for the C oracle"
    return {"starts": starts, "ends": ends, "prop_weights": pw, "prop_starts": [e[:, :-1] for e in pe],
            "prop_ends": [e[:, 1:] for e in pe]}


def num_threads() -> int:
    return lib().nro_num_threads()

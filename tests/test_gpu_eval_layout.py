"""Eval-time layout of the coarse hash-grid levels (csrc/eval_layout.hip): the fused render kernel reading the shadow copies
must produce the SAME BITS as the plain table -- with processing orders, early termination, fp16 storage -- and the eval
table must be rebuilt when the parameter changes."""
import numpy as np
import pytest
import torch

import neurad_oracle as O
from conftest import rel_l2
from test_gpu_parity import TOL, _sample_rays, dev, field_params, host, to_spec

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from neurad_studio_amd import ops as _ops

    return _ops


CFGS = [  # (L, F, lg, min_res, max_res, H, n shadow levels expected)
    (16, 2, 18, 16, 1024, 64, 5),   # BASELINE config 2's grid at T = 2^18: levels 16..48 -> 2^15 / 2^18-row shadows
    (8, 4, 18, 32, 8192, 32, 1),    # NeuRAD defaults, small table: level 0 (res 32 -> 2^18 rows)
    (8, 4, 21, 32, 8192, 64, 2),    # + level 1 (res 70 -> 2^21 rows)
]


@pytest.mark.parametrize("half", [False, True], ids=["fp32", "fp16"])
@pytest.mark.parametrize("cfg", CFGS)
def test_render_with_eval_layout_is_bit_identical(ops, cfg, half, monkeypatch, switches):
    """the eval-table kernel exists with fp32-MFMA products only (csrc/render.hip: the relayout and the pair products do not
    combine) and is dispatched FIRST when a table is handed over; it is held bit for bit to the plain fp32-MFMA kernel
    (NRHIP_MLP_PAIRS=0) -- and shown to be the kernel that ran: the default (pair) kernel's output differs in the last bits"""
    L, F, lg, mn, mx, H, n_shadow = cfg
    switches.set("NRHIP_MLP_PAIRS", "0")
    p = field_params(use_sdf=True, L=L, F=F, lg=lg, H=H, mn=mn, mx=mx, scale=0.5)
    p.beta = 3.0
    fs = to_spec(ops, p, half=half)
    lay, rows, ns = ops.eval_layout_plan(fs.grid, fs.table.dtype)
    assert ns == n_shadow and rows <= fs.grid.table_rows
    R, S = 3000, 48
    o, d, area, s, e, eu = _sample_rays(R, S, seed=9)
    do, dd, da, edges = dev(o), dev(d), dev(area), dev(eu)
    order = ops.ray_order(do, dd, 100.0)
    outs = {}
    for on in (False, True):
        monkeypatch.setattr(ops, "_EVAL_RELAYOUT", on)
        ops.clear_eval_tables()
        outs[on] = [ops.render_fwd(fs, do, dd, da, edges[:, :-1], edges[:, 1:], return_weights=True),
                    ops.render_fwd(fs, do, dd, da, edges[:, :-1], edges[:, 1:], order=order),
                    ops.render_fwd(fs, do, dd, da, edges[:, :-1], edges[:, 1:], early_stop_eps=1e-3)]
    assert len(ops._EVAL_TABLES) == 1  # built once, reused by the three calls
    for a, b in zip(outs[False], outs[True]):
        for x, y in zip(a, b):
            assert torch.equal(x, y)
    switches.unset("NRHIP_MLP_PAIRS")  # pair products are the default of the composited kernels; a handed-over eval table wins
    relay = ops.render_fwd(fs, do, dd, da, edges[:, :-1], edges[:, 1:], return_weights=True)
    monkeypatch.setattr(ops, "_EVAL_RELAYOUT", False)
    pairs = ops.render_fwd(fs, do, dd, da, edges[:, :-1], edges[:, 1:], return_weights=True)
    monkeypatch.setattr(ops, "_EVAL_RELAYOUT", True)
    assert torch.equal(relay[0], outs[True][0][0])  # the eval-table kernel ran, pairs or not
    assert not torch.equal(pairs[0], relay[0]) and rel_l2(host(pairs[0]), host(relay[0])) < 1e-6
    switches.set("NRHIP_MLP_PAIRS", "0")
    # and against the oracle on a slice (fp16: the oracle on the rounded table)
    if half:
        p.grid.table = host(fs.table.float())
    sl = slice(0, 24)
    ref = O.render_rays(p, o[sl], d[sl], area[sl], s[sl], e[sl])
    assert rel_l2(host(outs[True][0][0][sl]), ref["features"]) < TOL
    # the cache follows the parameter: an in-place update bumps its version and the eval table is rebuilt
    before = outs[True][0][0].clone()
    fs.table.mul_(0.5)
    after = ops.render_fwd(fs, do, dd, da, edges[:, :-1], edges[:, 1:])[0]
    monkeypatch.setattr(ops, "_EVAL_RELAYOUT", False)
    plain = ops.render_fwd(fs, do, dd, da, edges[:, :-1], edges[:, 1:])[0]
    assert torch.equal(after, plain) and not torch.equal(after, before)


def test_eval_table_holds_the_hashed_entries_of_the_lattice(ops, monkeypatch):
    """the shadow region of a level, read at ix | iy << s | iz << 2s, is the entry the reference's hash points at"""
    spec = ops.GridSpec(16, 2, 16, 16, 1024)
    table = torch.randn(spec.table_rows, 2, device="cuda")
    monkeypatch.setattr(ops, "_EVAL_RELAYOUT", True)
    et, lay = ops.eval_table(spec, table)
    T = 1 << 16
    for l in range(16):
        my, mz, mask, row0 = (int(lay[4 * l + k]) for k in range(4))
        if my == 2654435761:  # hashed level: a straight copy
            assert torch.equal(et[row0:row0 + T], table[l * T:(l + 1) * T])
            continue
        s = my.bit_length() - 1
        n = int(spec.scalings[l]) + 1  # lattice coordinates 0 .. ceil(scale)
        ix, iy, iz = torch.meshgrid(*[torch.arange(n, device="cuda")] * 3, indexing="ij")
        src = ((ix ^ (iy * 2654435761) ^ (iz * 805459861)) & (T - 1)) + l * T
        dst = row0 + (ix | (iy << s) | (iz << (2 * s)))
        assert torch.equal(et[dst.reshape(-1)], table[src.reshape(-1)])
    ops.clear_eval_tables()

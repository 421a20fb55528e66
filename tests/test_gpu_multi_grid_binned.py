"""The per-actor grids' table gradients without memory-side atomics (nrhip_hashgrid_multi_bwd_binned, ABI 511).

The reference loops over actor ids and lets autograd scatter-add into each actor's HashEncoding
(field_components/neurad_encoding.py:270-295); ``ops.hashgrid_multi_bwd`` produces the same per-grid gradients for all
actors at once -- by fp32 atomics for small batches, by the radix partition over (slot, level, slice) from 2^15 samples on.
Here the partition is held against (a) the atomic multi-grid kernel on the same inputs and (b) the SINGLE-grid entry point run
once per grid over that grid's samples (the path pinned to the oracle in test_gpu_parity.py) -- with grids that have no samples, grid ids outside the range, exactly-zero gradient rows, an fp16 block, a
poisoned row, one slot only, and a shape too large to partition (falls back to the atomics)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _case(n, n_grids, L, F, log2T, seed, absent=(), res=(16, 512)):
    from neurad_studio_amd import ops

    g = torch.Generator(device="cuda")
    g.manual_seed(seed)
    spec = ops.GridSpec(L, F, log2T, res[0], res[1])
    x = torch.rand((n, 3), device="cuda", generator=g)
    # clustered positions too: consecutive samples in one cell (runs of equal entries merge before they are emitted)
    x[: n // 4] = (x[: n // 4 : 16].repeat_interleave(16, 0)[: n // 4] + 1e-4 * torch.rand((n // 4, 3), device="cuda", generator=g)).clamp(0, 1)
    gid = torch.randint(0, n_grids, (n,), device="cuda", generator=g, dtype=torch.int32)
    for a in absent:
        gid[gid == a] = 0
    gid[::97] = -1  # samples outside every box
    gid[5::1013] = n_grids + 3  # (out of range: ignored like -1)
    go = torch.randn((n, L * F), device="cuda", generator=g)
    go[torch.rand(n, device="cuda", generator=g) < 0.3] = 0.0  # silent samples
    return spec, x, gid, go


def _present(gid, n_grids):
    ok = (gid >= 0) & (gid < n_grids)
    return [bool(v) for v in torch.bincount(gid[ok].long(), minlength=n_grids).cpu() > 0]


def _per_grid_reference(spec, x, gid, go, n_grids):
    """the single-grid gradient over each grid's own samples (atomic single-grid kernel: < 2^15 samples per grid)"""
    from neurad_studio_amd import ops

    out = []
    for a in range(n_grids):
        sel = gid == a
        if not bool(sel.any()):
            out.append(None)
            continue
        out.append(ops.hashgrid_bwd(spec, None, x[sel].contiguous(), go[sel].contiguous()))
    return out


def _atomic(spec, x, gid, go, n_grids, present, monkeypatch):
    from neurad_studio_amd import ops

    monkeypatch.setattr(ops, "_FORCE_ATOMIC_SCATTER", True)
    try:
        return ops.hashgrid_multi_bwd(spec, n_grids, gid, x, go, present=present)
    finally:
        monkeypatch.setattr(ops, "_FORCE_ATOMIC_SCATTER", False)


@pytest.mark.parametrize("shape", [(120_000, 32, 4, 4, 17), (200_000, 32, 4, 1, 15), (40_000, 5, 4, 2, 12), (70_000, 1, 2, 4, 16)],
                         ids=["field-actors", "proposal-actors", "small-tables", "one-grid"])
def test_multi_grid_partition_matches_the_atomics_and_the_single_grid_path(shape, monkeypatch):
    from neurad_studio_amd import ops

    n, n_grids, L, F, log2T = shape
    absent = (3, 4) if n_grids > 8 else ()
    spec, x, gid, go = _case(n, n_grids, L, F, log2T, seed=n_grids + F, absent=absent)
    present = _present(gid, n_grids)
    assert not absent or not all(present)
    got = ops.hashgrid_multi_bwd(spec, n_grids, gid, x, go, present=present)
    again = ops.hashgrid_multi_bwd(spec, n_grids, gid, x, go, present=present)
    atomic = _atomic(spec, x, gid, go, n_grids, present, monkeypatch)
    single = _per_grid_reference(spec, x, gid, go, n_grids)
    for a in range(n_grids):
        if not present[a]:
            assert got[a] is None and atomic[a] is None and single[a] is None
            continue
        assert got[a].shape == (spec.table_rows, F) and got[a].dtype == torch.float32 and got[a].is_contiguous()
        assert torch.equal(got[a], again[a])  # integer accumulation: bit-reproducible
        scale = float(single[a].abs().max())
        for other in (atomic[a], single[a]):
            assert float((got[a] - other).abs().max()) <= 2e-6 * scale + 1e-12, a
        # the rows nobody touched are exactly zero (the block is not zero-filled by the caller: the partition writes them)
        assert torch.equal(got[a] == 0, single[a] == 0)


def test_multi_grid_partition_fp16_block_and_requested_subset(monkeypatch):
    """fp16-storage actor grids get their gradient in fp16 from the partition itself (one rounding of the fp32 sum); a grid
    marked not-present receives nothing even though samples name it"""
    from neurad_studio_amd import ops

    n, n_grids, L, F, log2T = 100_000, 12, 4, 4, 15
    spec, x, gid, go = _case(n, n_grids, L, F, log2T, seed=7)
    go *= 1e-2
    present = _present(gid, n_grids)
    present[2] = False  # the caller does not want this grid's gradient
    got16 = ops.hashgrid_multi_bwd(spec, n_grids, gid, x, go, present=present, out_dtype=torch.float16)
    got32 = ops.hashgrid_multi_bwd(spec, n_grids, gid, x, go, present=present)
    assert got16[2] is None and got32[2] is None
    for a in range(n_grids):
        if present[a]:
            assert got16[a].dtype == torch.float16
            assert torch.equal(got16[a], got32[a].half())


def test_multi_grid_partition_poisons_like_an_atomic_add_and_falls_back_when_too_large(monkeypatch):
    from neurad_studio_amd import ops

    n, n_grids, L, F, log2T = 60_000, 4, 2, 2, 14
    spec, x, gid, go = _case(n, n_grids, L, F, log2T, seed=3)
    bad = int(torch.nonzero((gid == 1) & (go.abs().sum(1) > 0))[0])
    go[bad, 1] = float("nan")
    present = _present(gid, n_grids)
    got = ops.hashgrid_multi_bwd(spec, n_grids, gid, x, go, present=present)
    atomic = _atomic(spec, x, gid, go, n_grids, present, monkeypatch)
    for a in range(n_grids):
        assert torch.equal(torch.isnan(got[a]), torch.isnan(atomic[a])), a
        assert (1 <= int(torch.isnan(got[a]).sum()) <= 8) if a == 1 else not bool(torch.isnan(got[a]).any())  # 8 corners, one feature
    # 40 tables of 2^19 entries x 4 features: 40 x 128 slices per level > 2048 columns -> the atomic kernel, same sums
    n, n_grids, L, F, log2T = 40_000, 40, 2, 4, 19
    spec, x, gid, go = _case(n, n_grids, L, F, log2T, seed=5, res=(16, 64))
    import ctypes as C

    g = spec.c_grid(torch.empty((spec.table_rows, F), device="meta"))
    nb = C.c_int64(-1)
    ops.call("nrhip_hashgrid_multi_bwd_binned_workspace", C.byref(g), n_grids, n, C.byref(nb))
    assert nb.value == 0
    present = _present(gid, n_grids)
    got = ops.hashgrid_multi_bwd(spec, n_grids, gid, x, go, present=present)
    single = _per_grid_reference(spec, x, gid, go, n_grids)
    for a in range(n_grids):
        if present[a]:
            assert float((got[a] - single[a]).abs().max()) <= 2e-6 * float(single[a].abs().max()) + 1e-12


@pytest.mark.parametrize("n", [3000, 50_000], ids=["atomics", "partition"])
def test_shared_table_stack_gives_the_per_node_gradients(n):
    """A grid set looked into by two nodes of one step (the proposal field's actor grids: once per sampler round): through
    ag.StackTablesFn + MultiHashGridStackedFn the two gradients meet in one add of the block; every table must receive what
    two MultiHashGridFn nodes give it, and a table neither batch touches must receive None (its Adam state must not move)."""
    from neurad_studio_amd import autograd as ag
    from neurad_studio_amd import ops

    A, L, F, log2T = 6, 4, 1, 15
    spec = ops.GridSpec(L, F, log2T, 16, 256)
    torch.manual_seed(0)
    tables = [torch.nn.Parameter(torch.randn((spec.table_rows, F), device="cuda") * 0.1) for _ in range(A)]
    batches = []
    for k in range(2):
        x = torch.rand((n, 3), device="cuda", requires_grad=True)
        ids = torch.randint(0, A, (n,), device="cuda")
        ids[ids == 4] = 0           # grid 4: in neither batch
        if k == 0:
            ids[ids == 2] = 1       # grid 2: in the second batch only
        batches.append((x, ids, torch.randn((n, L * F), device="cuda")))

    def run(shared):
        for t in tables:
            t.grad = None
        outs = []
        if shared:
            bundle = ag.TableBundle(A)
            stacked = ag.StackTablesFn.apply(bundle, *tables)
        for x, ids, w in batches:
            x.grad = None
            f = ag.MultiHashGridStackedFn.apply(x, ids, spec, stacked, bundle) if shared else ag.MultiHashGridFn.apply(x, ids, spec, *tables)
            outs.append((f * w).sum())
        (outs[0] + outs[1]).backward()
        return [None if t.grad is None else t.grad.clone() for t in tables], [b[0].grad.clone() for b in batches], [float(o) for o in outs]

    g_plain, gx_plain, v_plain = run(False)
    g_shared, gx_shared, v_shared = run(True)
    assert v_plain == v_shared
    assert g_plain[4] is None and g_shared[4] is None
    for a in range(A):
        if a != 4:
            assert float((g_plain[a] - g_shared[a]).abs().max()) <= 2e-6 * float(g_plain[a].abs().max()), a
    for p, s_ in zip(gx_plain, gx_shared):
        assert torch.equal(p, s_)

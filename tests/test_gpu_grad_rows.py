"""csrc/grad_rows.hip (the device side of the level-sparse gradient exchange) against the torch formulation the
gloo / CPU path of parallel/data_parallel.py uses: counts, ordered compaction with padding, zero / add application."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _sparse_grad(L, T, F, fracs, seed):
    g = torch.Generator().manual_seed(seed)
    a = torch.zeros(L, T, F)
    for lvl, frac in enumerate(fracs):
        n = int(round(frac * T))
        rows = torch.randperm(T, generator=g)[:n]
        a[lvl, rows] = torch.randn(n, F, generator=g)
    return a.reshape(L * T, F).cuda()


@pytest.mark.parametrize("F", [1, 2, 4, 8])
@pytest.mark.parametrize("T", [1000, 4096])  # (1000: the last block of a level is ragged)
def test_count_compact_apply_match_torch(F, T):
    from neurad_studio_amd import ops

    L = 5
    grad = _sparse_grad(L, T, F, (0.0, 3 / T, 0.05, 0.4, 1.0), seed=11 * F + T)
    grad[2 * T + 7, 0] = float("nan")  # a poisoned row is a row to send
    if F > 1:
        grad[3 * T + 5] = 0.0
        grad[3 * T + 5, F - 1] = 1e-30  # non-zero in the LAST feature only
    counts, blocks = ops.grad_rows_count(grad, L)
    mask = (grad.view(L, T, F) != 0).any(-1)
    assert counts.tolist() == mask.sum(1).tolist()
    levels = [1, 2, 3]
    caps = [int(counts[l]) + pad for l, pad in zip(levels, (2, 0, 5))]  # the agreed capacity can exceed the own count
    rows, vals = ops.grad_rows_compact(grad, L, blocks, levels, caps, scale=0.5)
    o = 0
    for l, c in zip(levels, caps):
        want = mask[l].nonzero()[:, 0].int()
        n = want.numel()
        assert torch.equal(rows[o:o + n], want) and bool((rows[o + n:o + c] == -1).all())
        got, ref = vals[o:o + n], grad.view(L, T, F)[l][want.long()] * 0.5
        assert torch.equal(torch.nan_to_num(got, nan=7.0), torch.nan_to_num(ref, nan=7.0)) and bool((vals[o + n:o + c] == 0).all())
        o += c
    # apply: zero the own rows, then add two lists (the own one twice, as two ranks with equal gradients would)
    g2 = grad.clone()
    ops.grad_rows_apply(g2, L, levels, caps, rows, None, add=False)
    ref = grad.clone().view(L, T, F)
    for l in levels:
        ref[l][mask[l]] = 0.0
    assert torch.equal(torch.nan_to_num(g2, nan=7.0), torch.nan_to_num(ref.view(L * T, F), nan=7.0))
    ops.grad_rows_apply(g2, L, levels, caps, rows, vals, add=True)
    ops.grad_rows_apply(g2, L, levels, caps, rows, vals, add=True)
    want = grad.clone().view(L, T, F)
    for l in levels:
        want[l] = torch.where(mask[l][:, None], grad.view(L, T, F)[l] * 0.5 + grad.view(L, T, F)[l] * 0.5, want[l])
    assert torch.equal(torch.nan_to_num(g2, nan=7.0), torch.nan_to_num(want.view(L * T, F), nan=7.0))
    # untouched levels are untouched
    assert torch.equal(g2.view(L, T, F)[4], grad.view(L, T, F)[4]) and torch.equal(g2.view(L, T, F)[0], grad.view(L, T, F)[0])


def test_empty_lists_and_argument_checks():
    from neurad_studio_amd import _lib, ops

    grad = torch.zeros(2 * 512, 4, device="cuda")
    counts, blocks = ops.grad_rows_count(grad, 2)
    assert counts.tolist() == [0, 0]
    rows, vals = ops.grad_rows_compact(grad, 2, blocks, [0, 1], [0, 0])
    assert rows.numel() == 0 and vals.shape == (0, 4)
    ops.grad_rows_apply(grad, 2, [0, 1], [0, 0], rows, vals, add=True)  # nothing to do, no launch
    with pytest.raises(_lib.NeuradHipError):
        ops.grad_rows_compact(grad, 2, blocks, [2], [4])  # level out of range
    with pytest.raises(_lib.NeuradHipError):
        ops.grad_rows_count(torch.zeros(2 * 512, 3, device="cuda"), 2)  # features_per_level not in {1, 2, 4, 8}

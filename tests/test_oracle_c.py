"""Pin the C/OpenMP restatement (oracle/neurad_oracle_c.c, the bench's cpu_baseline) against the numpy oracle,
which is itself pinned against the reference's golden vectors."""
import numpy as np
import pytest

import neurad_oracle as O
import oracle_c
import synth
from conftest import rel_l2


def params(L, F, lg, H, mn, mx, use_sdf):
    grid = O.GridParams(synth.hash_table(L * 2**lg, F, seed=51, scale=1.0), L, mn, mx, lg)
    gw, gb, fw, fb = [], [], [], []
    for k, (o, i) in enumerate([(H, 32), (33, H)]):
        w, b = synth.linear(o, i, 200 + 10 * k)
        gw.append(w), gb.append(b)
    for k, (o, i) in enumerate([(H, 48), (H, H), (32, H)]):
        w, b = synth.linear(o, i, 300 + 10 * k)
        fw.append(w), fb.append(b)
    return O.FieldParams(grid, 100.0, gw, gb, fw, fb, beta=3.0, use_sdf=use_sdf)


@pytest.mark.parametrize("cfg", [(16, 2, 12, 64, 16, 1024, True, 24, 128), (8, 4, 11, 32, 32, 8192, False, 17, 33)])
def test_c_oracle_matches_numpy_oracle(cfg):
    L, F, lg, H, mn, mx, use_sdf, R, S = cfg
    p = params(L, F, lg, H, mn, mx, use_sdf)
    o, d, area, _ = synth.rays(R, 5)
    _, eu, _ = O.power_sampler(np.zeros(R), np.full(R, 300.0, np.float32), S)
    s, e = np.ascontiguousarray(eu[:, :-1]), np.ascontiguousarray(eu[:, 1:])
    ref = O.render_rays(p, o, d, area, s, e)
    got = oracle_c.render_fwd(p, o, d, area, s, e, per_sample=True)
    assert rel_l2(got["feature"], ref["feature"]) < 1e-5
    assert rel_l2(got["weights"], ref["weights"]) < 1e-5
    assert rel_l2(got["features"], ref["features"]) < 1e-5
    assert rel_l2(got["accumulation"], ref["accumulation"]) < 1e-5
    assert rel_l2(got["depth"], ref["depth"]) < 1e-5 or np.abs(got["depth"] - ref["depth"]).max() < 1e-5
    assert oracle_c.num_threads() >= 1


@pytest.mark.parametrize("quirk", [True, False])
def test_c_proposal_sampler_chain_matches_numpy_oracle(quirk):
    """S1-S5 + M1 in C (what the full-size config-3 GPU parity test checks against) == the numpy oracle's chain,
    which tests/test_oracle_golden.py pins to the reference's own sampler outputs."""
    R = 96
    # table scale 0.3 -> densities O(1): well conditioned.  (With saturated densities exp(-cumsum) amplifies the 1-ulp
    # powf differences in the bin edges to 1e-4 in the weights, in ANY two implementations.)
    props = [O.ProposalParams(O.GridParams(synth.hash_table(6 * 2**11, 1, seed=70 + i, scale=0.3), 6, 128, 4096, 11), 100.0,
                              synth.normal((1, 6), 80 + i)) for i in range(2)]
    o, d, area, _ = synth.rays(R, 13)
    nears, fars = np.zeros(R, np.float32), np.full(R, 1e9, np.float32)  # far beyond the sky distance: clamped
    ref = O.proposal_sampler(props, o, d, area, nears, fars, late_binding_quirk=quirk)
    got = oracle_c.proposal_sampler(props, o, d, area, nears, fars, late_binding_quirk=quirk)
    for i in range(2):
        assert np.allclose(got["prop_starts"][i], ref.prop_starts[i], rtol=2e-5, atol=1e-5)
        assert rel_l2(got["prop_weights"][i], ref.prop_weights[i]) < 5e-5
    for k, r in (("starts", ref.starts), ("ends", ref.ends)):
        assert np.allclose(got[k], r, rtol=1e-4, atol=1e-5), k
    assert np.all(got["ends"][:, -1] == np.float32(20000.0))

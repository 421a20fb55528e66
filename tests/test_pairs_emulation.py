"""The arithmetic of render.hip:mfma_layer_pairs, emulated in numpy (no GPU): product sums formed from fp16 pairs
x = fp16(x) + fp16(x - fp16(x)) with the kernel's two recentrings (weights x 2^7, activations x 2^6) are as exact as fp32
product sums over the magnitudes a NeRF MLP lives at -- the claim behind making that form the composited kernels' default
(the GPU side is tests/test_gpu_render_variants.py::test_fp16_pair_matrix_products_are_fp32_equivalent)."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts"))
import pairs_emulation as E  # noqa: E402


def _case(x_scale, w_scale, seed):
    rng = np.random.default_rng(seed)
    K, N, M = 64, 64, 2048
    W = (rng.uniform(-1, 1, (N, K)) * w_scale / np.sqrt(K)).astype(np.float32)
    x = np.maximum(rng.normal(0, 1, (K, M)), 0).astype(np.float32) * np.float32(x_scale)
    t = W.astype(np.float64) @ x.astype(np.float64)
    rel = lambda y: float(np.linalg.norm(y - t) / np.linalg.norm(t))  # noqa: E731
    return rel((W @ x).astype(np.float64)), rel(E.pairs(W, x, 7, 6)), rel(E.pairs(W, x, 0, 0))


@pytest.mark.parametrize("x_scale", [1e-2, 0.1, 1.0, 10.0])
@pytest.mark.parametrize("w_scale", [0.03, 0.3, 3.0])
def test_recentred_pairs_are_as_exact_as_the_fp32_product_sum(x_scale, w_scale):
    e32, e_pairs, e_plain = _case(x_scale, w_scale, seed=int(1000 * x_scale) + int(100 * w_scale))
    assert e_pairs < 1.5 * e32 and e_pairs < 2e-7, (e32, e_pairs)
    assert e_plain >= e_pairs  # (what the two recentrings buy: up to 100 x at small weights)


def test_small_activations_degrade_gracefully_and_the_split_is_exact_in_range():
    # an untrained field (table entries of 1e-4): 3e-6, still far inside the 1e-4 parity tolerance
    _, e_pairs, e_plain = _case(1e-4, 0.3, seed=5)
    assert e_pairs < 5e-6 and e_plain > 20 * e_pairs
    # hi + lo reproduces x to 2^-22 for everything the fast path admits (|x| < 65000 after the recentring)
    x = (np.random.default_rng(0).uniform(-1, 1, 100000) * 6.0e4).astype(np.float32)
    h, l = E.split(x)
    assert np.all(np.abs((h + l) - x.astype(np.float64)) <= np.abs(x) * 2.0 ** -22)

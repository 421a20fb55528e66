"""numpy emulation of the fp16-pair product sums of render.hip:mfma_layer_pairs (no GPU needed): relative error of a 64-term
product sum W x against fp64, for the fp32 product sum and for x = fp16(x) + fp16(x - fp16(x)) (round to nearest, fp16
subnormals kept) with the three products hi.hi + hi.lo + lo.hi accumulated exactly, as a function of the operands' magnitudes
and of the two power-of-two recentrings (weights x 2^cw, activations x 2^a).  The kernel uses cw = 7, a = 6.
    python scripts/pairs_emulation.py > profiles/r05_pairs_emulation.txt"""
import numpy as np

rng = np.random.default_rng(0)


def split(x):
    x = x.astype(np.float32)
    h = x.astype(np.float16)
    l = (x - h.astype(np.float32)).astype(np.float32).astype(np.float16)
    return h.astype(np.float64), l.astype(np.float64)


def pairs(W, x, cw, a):
    Wh, Wl = split(W * np.float32(2.0**cw))
    xh, xl = split(x * np.float32(2.0**a))
    return (Wh @ xh + Wh @ xl + Wl @ xh) / 2.0 ** (cw + a)


def main():
    K, N, M = 64, 64, 4096
    cases = ((0, 0), (0, 6), (7, 0), (7, 6))
    print("# rel. L2 error vs fp64 of a %d-term product sum; post-ReLU normal activations x scale, uniform weights" % K)
    print("# activation scale | max |w| | fp32 product sum | pairs with (cw, a) = " + ", ".join(map(str, cases)))
    for xs in (1e-4, 1e-3, 1e-2, 0.1, 1.0, 10.0):
        for ws in (0.03, 0.1, 0.3, 1.0, 3.0):
            W = (rng.uniform(-1, 1, (N, K)) * ws / np.sqrt(K)).astype(np.float32)
            x = np.maximum(rng.normal(0, 1, (K, M)), 0).astype(np.float32) * np.float32(xs)
            t = W.astype(np.float64) @ x.astype(np.float64)
            e32 = np.linalg.norm((W @ x).astype(np.float64) - t) / np.linalg.norm(t)
            errs = [np.linalg.norm(pairs(W, x, cw, a) - t) / np.linalg.norm(t) for cw, a in cases]
            print(f"{xs:8g} | {ws / np.sqrt(K):8.3g} | {e32:.1e} | " + " ".join(f"{e:.1e}" for e in errs))


if __name__ == "__main__":
    main()

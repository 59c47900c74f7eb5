"""numpy simulation behind DESIGN 3.3a / 10: dot products of K = 6400 terms with the magnitudes of the ConvLSTM operands (hidden state x
weights, dz x weights, activations x dz) -- error against fp64 of (a) the fp32 fmaf chain of v_mfma_f32_32x32x2_f32, (b) bf16x3: six bf16
products of the exact three-way split, fp32 accumulation in groups of 16 (the product path), (c) fp16x2: three fp16 products of a two-way split
with per-tensor power-of-two scales (the next step), (d) the same without scales.  usage: python tools/split_sim.py"""
import numpy as np


def bf16(x):
    x = np.asarray(x, np.float32)
    u = x.view(np.uint32).astype(np.uint64)
    return (((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16).astype(np.uint32).view(np.float32)


def f16(x):
    return np.asarray(x, np.float32).astype(np.float16).astype(np.float32)


def pow2scale(x, top=2.0 ** 14):
    return 2.0 ** np.floor(np.log2(top / float(np.abs(x).max())))


def main(K=6400, M=1500, seed=1):
    rng = np.random.default_rng(seed)
    heavy = lambda: (rng.standard_normal((M, K)) * 1e-4 * np.exp(rng.standard_normal((M, K)))).astype(np.float32)  # noqa: E731
    w = lambda: (rng.standard_normal((M, K)) * 0.02).astype(np.float32)  # noqa: E731
    cases = {'h x w': ((np.tanh(rng.standard_normal((M, K)) * 0.5) * rng.random((M, K))).astype(np.float32), w()),
             'dz x w': (heavy(), w()),
             'x x dz (weight gradient)': ((np.abs(rng.standard_normal((M, K))) * 0.7).astype(np.float32), heavy())}

    def chain32(p):
        acc = np.zeros(M, np.float32)
        for k in range(K):
            acc = (acc.astype(np.float64) + p[:, k]).astype(np.float32)
        return acc

    def mfma(terms):
        acc = np.zeros(M, np.float32)
        for (x, y) in terms:
            pp = x.astype(np.float64) * y.astype(np.float64)
            for k0 in range(0, K, 16):
                acc = (acc.astype(np.float64) + pp[:, k0:k0 + 16].sum(1)).astype(np.float32)
        return acc

    for name, (a, b) in cases.items():
        truth = (a.astype(np.float64) * b.astype(np.float64)).sum(1)
        scale = np.abs(a.astype(np.float64) * b).sum(1)
        ah, bh = bf16(a), bf16(b)
        am, bm = bf16(a - ah), bf16(b - bh)
        al, bl = bf16(a - ah - am), bf16(b - bh - bm)
        Sa, Sb = pow2scale(a), pow2scale(b)
        A, B = a * np.float32(Sa), b * np.float32(Sb)
        a1, b1 = f16(A), f16(B)
        a2, b2 = f16(A - a1), f16(B - b1)
        u1, v1 = f16(a), f16(b)
        u2, v2 = f16(a - u1), f16(b - v1)
        rows = (('fp32 fmaf chain', chain32(a.astype(np.float64) * b.astype(np.float64))),
                ('bf16x3, 6 products', mfma([(al, bh), (am, bm), (ah, bl), (am, bh), (ah, bm), (ah, bh)])),
                ('fp16x2 scaled, 3 products', mfma([(a2, b1), (a1, b2), (a1, b1)]) / (Sa * Sb)),
                ('fp16x2 unscaled', mfma([(u2, v1), (u1, v2), (u1, v1)])))
        for nm, v in rows:
            e = np.abs(v - truth) / scale
            print('%-26s %-28s max %.2e  rms %.2e  (of sum |a b|)' % (name, nm, e.max(), np.sqrt((e ** 2).mean())))


if __name__ == '__main__':
    main()

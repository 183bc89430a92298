"""CPU model of the split-operand arithmetic of the BiGRU (csrc/gru_split_kernel.h): which operands lose what, and what a
power-of-two scale per operand class buys.  numpy only; operands are rounded exactly as split_pair<F16> does (hi = RTZ to
f16, lo = RNE(a - hi) to f16, sub-normals kept), products hi*hi + hi*lo + lo*hi are summed in float64 (the fp32 accumulation
of the MFMA is NOT modelled: this isolates the operand error).  Usage: python tools/experiments/split_model.py"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from weights_recipe import make_weights  # noqa: E402


def rtz16(a):
    h = a.astype(np.float16)                       # RNE
    hf = h.astype(np.float64)
    over = np.abs(hf) > np.abs(a)                  # rounded away from zero: step back one ulp
    h = np.where(over, np.nextafter(h, np.float16(0)), h)
    return h


def split16(a, scale=1.0):
    a = np.asarray(a, dtype=np.float32).astype(np.float64) * scale
    hi = rtz16(a)
    lo = (a - hi.astype(np.float64)).astype(np.float16)
    return hi.astype(np.float64), lo.astype(np.float64)


def mm_split(a, w, sa=1.0, sw=1.0, mode="f16"):
    """a [M,K] @ w[N,K]^T with both operands split"""
    if mode == "exact":
        return a.astype(np.float64) @ w.astype(np.float64).T
    ah, al = split16(a, sa)
    wh, wl = split16(w, sw)
    return (ah @ wh.T + ah @ wl.T + al @ wh.T) / (sa * sw)


def sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def gru_dir(x, w_ih, w_hh, b_ih, b_hh, reverse, mm):
    B, T, _ = x.shape
    H = w_hh.shape[1]
    h = np.zeros((B, H))
    out = np.zeros((B, T, H))
    for s in range(T):
        t = T - 1 - s if reverse else s
        gi = mm(x[:, t], w_ih, "x") + b_ih
        gh = mm(h, w_hh, "h") + b_hh
        r = sigmoid(gi[:, :H] + gh[:, :H])
        z = sigmoid(gi[:, H:2 * H] + gh[:, H:2 * H])
        n = np.tanh(gi[:, 2 * H:] + r * gh[:, 2 * H:])
        h = (1 - z) * n + z * h
        out[:, t] = h
    return out


def bigru(x, w, mm1, mm2):
    def layer(x, name, mm):
        f = gru_dir(x, w[name + ".weight_ih_l0"], w[name + ".weight_hh_l0"], w[name + ".bias_ih_l0"], w[name + ".bias_hh_l0"], False, mm)
        b = gru_dir(x, w[name + ".weight_ih_l0_reverse"], w[name + ".weight_hh_l0_reverse"], w[name + ".bias_ih_l0_reverse"],
                    w[name + ".bias_hh_l0_reverse"], True, mm)
        return np.concatenate([f, b], axis=-1)
    h1 = layer(x.astype(np.float64), "lstm", mm1)
    h2 = layer(h1, "lstm_2", mm2)
    return h1, h2


def main():
    z = np.load(os.path.join(ROOT, "tests", "golden", "models_BiGRU_NACGT_range.npz"))
    manifest = [(k, tuple(s)) for k, s in json.loads(str(z["manifest"]))]
    x = z["x"]
    groups = [("realistic", 0, 16), ("d120", 16, 20), ("d600", 20, 24), ("d2500", 24, 28), ("d8000", 28, 32), ("zeros", 32, 36),
              ("onehot", 36, 40), ("tiny", 40, 44), ("shallow", 44, 48), ("gappy", 48, 52)]
    for scale in (1.0, 2.0):
        w = {k: v.astype(np.float64) for k, v in make_weights(manifest, seed=4, scale=scale).items()}
        exact = lambda a, ww, kind: mm_split(a, ww, mode="exact")
        e1, e2 = bigru(x, w, exact, exact)
        variants = {
            "f16 unscaled": (lambda a, ww, kind: mm_split(a, ww),) * 2,
            "f16 h*2^14 w*2^k": (lambda a, ww, kind: mm_split(a, ww, 1.0 if kind == "x" else 2.0 ** 14, 2.0 ** np.floor(np.log2(32768 / np.abs(ww).max()))),
                                 lambda a, ww, kind: mm_split(a, ww, 2.0 ** 14, 2.0 ** np.floor(np.log2(32768 / np.abs(ww).max())))),
        }
        # fp32-operand model for comparison: operands rounded to fp32 (they are), products exact => zero operand error; the fp32
        # kernels' error is all accumulation, which this model leaves out.
        for name, (m1, m2) in variants.items():
            h1, h2 = bigru(x, w, m1, m2)
            print("weights x%.1f  %-18s" % (scale, name), " ".join("%s %.1e/%.1e" % (g, np.abs(h1[a:b] - e1[a:b]).max(), np.abs(h2[a:b] - e2[a:b]).max())
                                                                     for g, a, b in groups))


if __name__ == "__main__":
    main()

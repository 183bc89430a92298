"""Range fixtures: the four reference network classes OUTSIDE the O(1)-activation regime of weights_recipe.py.

Run in the build container only (imports /root/reference):   python tests/golden/gen_range.py
Writes tests/golden/models_<cls>_range.npz = data only:
  x          [N,33,34] fp32 inputs (range_inputs(): realistic rescaled windows from region.json.gz, UNRESCALED deep
             windows up to depth 8 000 - predict.py:604-608 builds one generator with min_rescale_cov=None -, all-zero
             windows, one-hot windows, tiny fractional inputs)
  per set s in SETS ("x1", "x1.5", "x2", "x3": every weight matrix of the recipe times that factor; "sat": recipe
  weights with head_gain 8 so the two-way softmax saturates and 8-decimal probabilities print as 1.00000000):
    logits32_<s> [K,N,2]  the reference module in fp32 (torch CPU, 1 thread) - what the reference computes
    logits64_<s> [K,N,2]  the SAME module after .double() - what the reference's arithmetic means
    probs32_<s>, probs64_<s>  Softmax(dim=1) of those (predict.py:660-684)
The fp64 run is what makes the sweep meaningful: beyond x1.5 two fp32 evaluations of the same module (other
summation order) already differ by more than 1e-4 in a logit, so "equal to the fp32 reference" stops being a
property of an implementation; "as close to the fp64 evaluation as the fp32 reference is" still is.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
sys.path.insert(0, HERE)
from weights_recipe import CVT_CFG, make_weights  # noqa: E402

SETS = (("x1", 1.0, 2.0), ("x1.5", 1.5, 2.0), ("x2", 2.0, 2.0), ("x3", 3.0, 2.0), ("sat", 1.0, 8.0))
CLASSES = (("CvT", 4), ("CvT_Indel", 6), ("BiGRU_NACGT", 4), ("BiGRU_NACGT_Indel", 6))


def pileup_like(rng, depth, n):
    """n windows of count tensors with the channel structure of create_tensor_pileup_calling.py:156-228 at a given
    depth (no rescale): forward / reverse base counts, I / D with their max-allele channels, * and #, LMQ / LBQ
    subsets, the reference base of every four-base group negated to -(group sum)."""
    X = np.zeros((n, 33, 34), dtype=np.float64)
    for i in range(n):
        for p in range(33):
            d = max(0, int(rng.normal(depth, depth * 0.15)))
            ref = int(rng.integers(0, 4))
            fwd = int(rng.binomial(d, 0.5))
            for base, tot in ((0, fwd), (9, d - fwd)):
                err = rng.multinomial(tot, [0.94, 0.015, 0.015, 0.015, 0.005, 0.005, 0.005])
                X[i, p, base + ref] = err[0]
                others = [b for b in range(4) if b != ref]
                for k, b in enumerate(others):
                    X[i, p, base + b] = err[1 + k]
                X[i, p, base + 4] = err[4]; X[i, p, base + 5] = rng.integers(0, err[4] + 1)      # I, I1
                X[i, p, base + 6] = err[5]; X[i, p, base + 7] = rng.integers(0, err[5] + 1)      # D, D1
                X[i, p, base + 8] = err[6]                                                      # * / #
            for g in (18, 22, 26, 30):                                                          # LMQ / LBQ fwd, rev
                src = 0 if g in (18, 26) else 9
                frac = 0.07 if g < 26 else 0.3
                X[i, p, g:g + 4] = rng.binomial(X[i, p, src:src + 4].astype(np.int64), frac)
            if p == 16 and rng.random() < 0.7:                                                  # an alt allele at the centre
                alt = (ref + 1 + int(rng.integers(0, 3))) % 4
                k = int(d * rng.uniform(0.05, 0.6) / 2)
                X[i, p, alt] += k; X[i, p, 9 + alt] += k
            for g in (0, 9, 18, 22, 26, 30):
                X[i, p, g + ref] = -X[i, p, g:g + 4].sum()
    return X.astype(np.float32)


def range_inputs(base_x):
    rng = np.random.default_rng(20260929)
    parts = [base_x[:16]]                                   # realistic, rescaled (|x| <= ~50)
    for depth in (120, 600, 2500, 8000):                    # unrescaled deep windows
        parts.append(pileup_like(rng, depth, 4))
    parts.append(np.zeros((4, 33, 34), dtype=np.float32))   # no pileup rows at all
    one = np.zeros((4, 33, 34), dtype=np.float32)
    one[0, 16, 0] = 1; one[1, 0, 33] = -3; one[2, 32, 17] = 40; one[3, 16, 9] = -8000
    parts.append(one)
    parts.append((pileup_like(rng, 30, 4) * np.float32(1.0 / 1024)))   # tiny fractional inputs (sub-normal lo halves)
    parts.append(pileup_like(rng, 6, 4))                    # shallow
    mixed = pileup_like(rng, 50, 4)
    mixed[:, ::2] = 0                                       # every second position uncovered
    parts.append(mixed)
    return np.concatenate(parts).astype(np.float32)


def main():
    sys.path.insert(0, REF)
    import torch
    import clairs.model as rm
    torch.set_num_threads(1)

    def build(cls, n_out, scale, head_gain):
        if cls.startswith("CvT"):
            kw = dict(num_classes=2, s1_emb_dim=CVT_CFG["emb_dim"][0], s2_emb_dim=CVT_CFG["emb_dim"][1],
                      s3_emb_dim=CVT_CFG["emb_dim"][2], s1_heads=CVT_CFG["heads"][0], s2_heads=CVT_CFG["heads"][1],
                      s3_heads=CVT_CFG["heads"][2], s1_depth=CVT_CFG["depth"][0], s2_depth=CVT_CFG["depth"][1],
                      s3_depth=CVT_CFG["depth"][2], apply_softmax=False, model_type="acgt")
            m = getattr(rm, cls)(**kw)
        else:
            m = getattr(rm, cls)(apply_softmax=False, num_classes=2, model_type="nacgt")
        manifest = [(k, list(v.shape)) for k, v in m.state_dict().items() if not k.endswith("num_batches_tracked")]
        w = make_weights(manifest, seed=n_out, head_gain=head_gain, scale=scale)
        sd = m.state_dict()
        for k, v in w.items():
            sd[k] = torch.from_numpy(v.copy())
        m.load_state_dict(sd)
        return m.eval(), manifest

    for cls, n_out in CLASSES:
        base = np.load(os.path.join(HERE, "models_%s.npz" % cls))["x"]
        x = range_inputs(base)
        out = dict(x=x, n_out=n_out)
        for name, scale, gain in SETS:
            m, manifest = build(cls, n_out, scale, gain)
            sm = torch.nn.Softmax(dim=1)
            with torch.no_grad():
                o32 = m(torch.from_numpy(x))
                p32 = [sm(o) for o in o32]
                m64 = m.double()
                o64 = m64(torch.from_numpy(x).double())
                p64 = [sm(o) for o in o64]
            l32 = np.stack([o.numpy() for o in o32]); l64 = np.stack([o.numpy() for o in o64])
            out["logits32_" + name] = l32.astype(np.float32)
            out["logits64_" + name] = l64.astype(np.float64)
            out["probs32_" + name] = np.stack([p.numpy() for p in p32]).astype(np.float32)
            out["probs64_" + name] = np.stack([p.numpy() for p in p64]).astype(np.float64)
            dl = np.abs(l32 - l64).max(); dp = np.abs(out["probs32_" + name] - out["probs64_" + name]).max()
            print("%-18s %-5s max|logit| %9.3g  ref32 vs ref64: |dlogit| %.3g  |dP| %.3g  P==1 rows %d" % (
                cls, name, np.abs(l64).max(), dl, dp, int((out["probs32_" + name].max(-1) >= 0.999999995).sum())))
        out["manifest"] = json.dumps(manifest)
        out["sets"] = json.dumps([list(s) for s in SETS])
        np.savez_compressed(os.path.join(HERE, "models_%s_range.npz" % cls), **out)


if __name__ == "__main__":
    main()

"""Deterministic synthetic weights for the four network classes.

The reference ships no weights (they are downloaded by its Dockerfile), so parity is pinned on seeded
synthetic weights.  The same recipe is used (a) by gen_golden.py, which loads the arrays into the
reference's own modules to produce golden logits, and (b) by the tests, which hand the identical arrays to
the oracle and to the HIP engine - so no weight blob has to be committed, only the (name, shape) manifests.
"""
import zlib

import numpy as np

# CvT configuration hard-coded by the reference for the indel models and used for its SNV models
# (clairs/predict.py:520-553)
CVT_CFG = dict(emb_dim=(16, 64, 128), heads=(1, 3, 4), depth=(1, 2, 3))


def _rng(name, seed):
    return np.random.default_rng([zlib.crc32(name.encode()) & 0xffffffff, seed])


def make_weights(manifest, seed=0, head_gain=2.0, scale=1.0):
    """manifest: list of (name, shape). Returns dict name -> float32 array.

    `scale` multiplies every weight MATRIX (not biases, norm gains or BN statistics): the range fixtures
    (gen_range.py) use 1.5 / 2 / 3 to leave the O(1)-activation regime the defaults were tuned for, and a
    larger `head_gain` to saturate the two-way softmax.

    Scales are chosen so activations stay O(1) through the depth of both networks and the final logits
    spread enough to exercise every decision branch of call_variants."""
    out = {}
    for name, shape in manifest:
        shape = tuple(int(s) for s in shape)
        r = _rng(name, seed)
        if name.endswith("num_batches_tracked"):
            continue
        if name.endswith("running_var"):
            a = r.uniform(0.5, 1.5, size=shape)
        elif name.endswith("running_mean"):
            a = r.uniform(-0.2, 0.2, size=shape)
        elif name.endswith(".g") or (".net.1.weight" in name):          # LayerNorm gain / BatchNorm weight
            a = r.uniform(0.8, 1.2, size=shape)
        elif name.endswith(".b") or name.endswith("bias") or "bias_" in name:
            a = r.uniform(-0.1, 0.1, size=shape)
        else:
            fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else shape[0]
            if len(shape) == 4 and shape[1] == 1:                         # depth-wise 3x3: 3 live taps
                fan_in = 3
            elif len(shape) == 4 and shape[2] == 3:                       # embedding conv: only the middle row is live
                fan_in = shape[1] * 3
            bound = (3.0 / fan_in) ** 0.5
            if "_fc3" in name:
                bound *= head_gain
            if name.startswith("layer1.0."):                              # raw counts come in at O(10..50)
                bound *= 0.05
            if name.startswith("lstm.weight_ih"):
                bound *= 0.05
            a = r.uniform(-bound, bound, size=shape) * scale
        out[name] = a.astype(np.float32)
    return out

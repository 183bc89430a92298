#!/usr/bin/env python3
"""Max abs error of the HIP networks against the reference's own golden logits (tests/golden/models_*.npz)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")]
import numpy as np  # noqa: E402
import torch  # noqa: E402
from conftest import load_models_npz  # noqa: E402
from weights_recipe import make_weights  # noqa: E402
from clairs_to_amd.nn_shims import from_state_dict  # noqa: E402

for cls in ("CvT", "CvT_Indel", "BiGRU_NACGT", "BiGRU_NACGT_Indel"):
    g = load_models_npz(cls)
    m = from_state_dict(cls, make_weights(g["manifest"], seed=g["n_out"]))
    got = m.logits(torch.from_numpy(g["x"]).cuda()).cpu()
    ref = torch.from_numpy(g["logits"])
    dl = (got - ref).abs().max().item()
    dp = (torch.softmax(got, -1) - torch.softmax(ref, -1)).abs().max().item()
    print("%-18s max|dlogit| = %.3g   max|dP| = %.3g   (bar 1e-4)" % (cls, dl, dp))

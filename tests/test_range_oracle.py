"""The oracle against the range fixtures (tests/golden/gen_range.py): the reference's own modules with the recipe's
weight matrices times 1.5 / 2 / 3, a saturating head, unrescaled depths up to 8 000, all-zero and one-hot windows -
evaluated by the reference in fp32 AND in fp64.  The oracle accumulates in double, so it must follow the fp64 run at
least as closely as the reference's fp32 run does (that is what "pinned" can mean once two fp32 evaluations of the
same module differ by more than the bar)."""
import numpy as np
import pytest

from conftest import load_range_npz, range_errors
from weights_recipe import CVT_CFG, make_weights


@pytest.mark.parametrize("cls", ["CvT", "CvT_Indel", "BiGRU_NACGT", "BiGRU_NACGT_Indel"])
def test_oracle_follows_the_references_fp64_run(oracle_lib, cls):
    g = load_range_npz(cls)
    for name, scale, gain in g["sets"]:
        w = make_weights(g["manifest"], seed=g["n_out"], head_gain=gain, scale=scale)
        if cls.startswith("CvT"):
            got = oracle_lib.cvt_forward(w, dict(CVT_CFG, n_out=g["n_out"]), g["x"])
        else:
            got = oracle_lib.bigru_forward(w, g["n_out"], g["x"])
        assert np.isfinite(got).all(), (cls, name)
        e = range_errors(got, g["z"], name)
        print("%-18s %-5s oracle: |dP| vs ref64 %.2e (ref32: %.2e)  rel logit %.2e (ref32: %.2e)" % (
            cls, name, e["dp64"], e["ref_dp"], e["rel"], e["ref_rel"]))
        # the oracle's outputs are fp32 (rounded once at the end): allow that rounding, nothing else
        assert e["dp64"] <= max(2e-6, 1.5 * e["ref_dp"]), (cls, name, e)
        assert e["rel"] <= max(2e-6, 1.5 * e["ref_rel"]), (cls, name, e)
        if e["ref_dp"] < 2e-5:        # where the fp32 reference is itself reproducible, north_star's bar holds against it
            assert e["dp32"] < 1e-4, (cls, name, e)

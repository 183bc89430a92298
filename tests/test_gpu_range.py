"""The HIP networks OUTSIDE the O(1)-activation regime (tests/golden/gen_range.py; round-4 review, task 1).

Fixtures: the reference's own four modules with the recipe's weight matrices x1 / x1.5 / x2 / x3, one set with a saturating head
(8-decimal probabilities that print as 1.00000000), inputs incl. UNRESCALED windows to depth 8 000, all-zero, one-hot and tiny
fractional windows - each evaluated by the reference in fp32 (what it computes) and in fp64 (what its arithmetic means).

The bar is north_star's: |dP| < 1e-4 on the PROBABILITIES, against the fp32 reference wherever that reference is itself
reproducible (its own fp64 run within 2e-5), and "no further from the fp64 run than 3 x the fp32 reference is" beyond -
two fp32 evaluations of clairs/model.py:239-261 with another summation order already differ by 1.5e-3 in a probability at x3.
Logits are held to a RELATIVE bound (|d| / max(1, max |logit|)).  The same sweep runs on the split-operand side channel
(f16 / bf16 halves), whose f16 form has a range to respect: counts above 2 048 have no exact f16 hi half, 8 000-deep
unrescaled windows and weights x3 push products towards 65 504, 1/1024-scaled inputs make every lo half sub-normal."""
import numpy as np
import pytest

from conftest import load_range_npz, range_errors
from weights_recipe import make_weights

pytestmark = pytest.mark.gpu

CLASSES = ["CvT", "CvT_Indel", "BiGRU_NACGT", "BiGRU_NACGT_Indel"]


@pytest.fixture(scope="module")
def dev():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return torch.device("cuda:0")


def _forward(cls, w, x, dev, split=None):
    """logits [K,B,2] through the C ABI (cto_*_create_ex + cto_model_forward) for one weight set"""
    import torch
    from clairs_to_amd._lib import lib, check
    from clairs_to_amd.nn_shims import from_state_dict
    m = from_state_dict(cls, w)
    m.split_operands = split or "f32"
    h = m._handle()
    xd = torch.from_numpy(x).to(dev)
    K = int(lib.cto_model_n_out(h))
    out = torch.empty((K, x.shape[0], 2), device=dev)
    check(lib.cto_model_forward(h, xd.data_ptr(), x.shape[0], out.data_ptr(), int(torch.cuda.current_stream().cuda_stream)))
    torch.cuda.synchronize()
    return out.cpu().numpy()


# How many times the fp32 reference's own distance from its fp64 run an implementation may sit from that run.  Measured on MI355X
# (DESIGN.md section 6, round 5): the fp32 kernels 0.3-3.2 x (the same with libm in place of every fast form: -DCTO_PRECISE_MATH,
# 0.8-2.6 x - the fast forms are not what the distance is made of), f16 halves 0.3-5 x; each figure is the MAXIMUM over 52 windows x
# 2K probabilities of a chaotic amplification of rounding noise, so the bound leaves room.
# Round 6: the product's fp32 kernels are held to 4 x (their measured worst is 3.3 x: CvT_Indel x2, 1.1e-3 against the fp32 reference's own
# 3.3e-4), the opt-in f16 side channel keeps 6 x.
NOISE = 6.0
NOISE_FP32 = 4.0


def _check(cls, name, e, fp32_path, bad):
    noise = NOISE_FP32 if fp32_path else NOISE
    if not e["dp64"] <= max(1e-4, noise * e["ref_dp"]):
        bad.append((cls, name, "dP vs ref64", e["dp64"], e["ref_dp"]))
    if not e["rel"] <= max(1e-5 if fp32_path else 3e-5, noise * e["ref_rel"]):
        bad.append((cls, name, "rel logit", e["rel"], e["ref_rel"]))
    if e["ref_dp"] < 2e-5 and not e["dp32"] < 1e-4:
        bad.append((cls, name, "dP vs ref32", e["dp32"], e["ref_dp"]))


@pytest.mark.parametrize("cls", CLASSES)
def test_fp32_kernels_over_the_range_sweep(dev, cls):
    g = load_range_npz(cls)
    bad = []
    for name, scale, gain in g["sets"]:
        w = make_weights(g["manifest"], seed=g["n_out"], head_gain=gain, scale=scale)
        got = _forward(cls, w, g["x"], dev)
        assert np.isfinite(got).all(), (cls, name)
        e = range_errors(got, g["z"], name)
        print("RANGE %-18s %-5s f32   |dP| vs ref32 %.2e  vs ref64 %.2e (ref32 itself %.2e)  rel logit %.2e (ref32 itself %.2e)  max|logit| %.3g" % (
            cls, name, e["dp32"], e["dp64"], e["ref_dp"], e["rel"], e["ref_rel"], float(np.abs(g["z"]["logits64_" + name]).max())))
        _check(cls, name, e, True, bad)
    assert not bad, bad


@pytest.mark.parametrize("kind", ["f16", "bf16"])
@pytest.mark.parametrize("cls", CLASSES)
def test_split_operand_kernels_over_the_range_sweep(dev, cls, kind):
    g = load_range_npz(cls)
    bad = []
    for name, scale, gain in g["sets"]:
        w = make_weights(g["manifest"], seed=g["n_out"], head_gain=gain, scale=scale)
        got = _forward(cls, w, g["x"], dev, split=kind)
        assert np.isfinite(got).all(), (cls, name, kind)
        e = range_errors(got, g["z"], name)
        print("RANGE %-18s %-5s %-5s |dP| vs ref32 %.2e  vs ref64 %.2e (ref32 itself %.2e)  rel logit %.2e (ref32 itself %.2e)" % (
            cls, name, kind, e["dp32"], e["dp64"], e["ref_dp"], e["rel"], e["ref_rel"]))
        if kind == "bf16":
            if name in ("x1", "sat") and not e["dp32"] < 1e-4:
                bad.append((cls, name, "dP vs ref32", e["dp32"], e["ref_dp"]))
            # 16-17 significant bits: inside the 1e-4 bar on probabilities where the fixtures' activations are O(1) and NOT beyond
            # (measured 2e-4 .. 2e-2, one flipped saturated probability at x3) - reported, not asserted; the drivers' help text says so
            continue
        _check(cls, name, e, False, bad)
    assert not bad, bad


@pytest.mark.parametrize("K,aff,neg", [(4, "CvT", "BiGRU_NACGT"), (6, "CvT_Indel", "BiGRU_NACGT_Indel")])
def test_saturated_probabilities_through_the_networks_and_the_epilogue(dev, oracle_lib, K, aff, neg):
    """call_variants.py:181-196: a probability that prints as 1.00000000 indexes past the likelihood table in the reference
    (IndexError).  Driven through the NETWORKS here ("sat" and "x3" sets: hundreds of such rows), not through from_probs: the
    device epilogue must clamp the bin, flag the site (decision[:,1] bit 0) and otherwise agree with the oracle's epilogue on the
    device's own 8-decimal probabilities."""
    import torch
    from clairs_to_amd.call_variants import Posterior
    from clairs_to_amd.synth import likelihood_table, lik_and_edges
    ga, gn = load_range_npz(aff), load_range_npz(neg)
    lik, edges = lik_and_edges(likelihood_table(K), K)
    post_dev = Posterior(lik, edges, device=dev)
    n_flag = 0
    for (name, scale, gain) in ga["sets"]:
        if name not in ("sat", "x3"):
            continue
        la = _forward(aff, make_weights(ga["manifest"], seed=K, head_gain=gain, scale=scale), ga["x"], dev)
        ln = _forward(neg, make_weights(gn["manifest"], seed=K, head_gain=gain, scale=scale), gn["x"], dev)
        res = post_dev(torch.from_numpy(la).to(dev), torch.from_numpy(ln).to(dev))
        torch.cuda.synchronize()
        probs = res["probs"].cpu().numpy()
        post = res["post"].cpu().numpy()
        dec = res["decision"].cpu().numpy()
        w_probs, _, _, _ = oracle_lib.posterior(la, ln, lik, edges)      # same logits in, the oracle's softmax
        np.testing.assert_allclose(probs, w_probs, rtol=0, atol=2e-7)
        # the epilogue on the device's OWN 8-decimal probabilities (what predict's text row would carry) equals the oracle's bit for bit
        p8dev = np.round(probs[:, :, 1].astype(np.float64) * 1e8) / 1e8
        w_post, w_dec, _ = oracle_lib.posterior_from_probs(p8dev, lik, edges)
        np.testing.assert_array_equal(post, w_post)
        np.testing.assert_array_equal(dec[:, 0], w_dec[:, 0])
        np.testing.assert_array_equal(dec[:, 1] & 3, w_dec[:, 1] & 3)
        # where the reference would raise: p_x prints as 1.00000000 (bin 10 of 0..9) or p_nx as 0.00000000 (1 - p_nx = 1)
        p8 = np.round(probs.astype(np.float64)[..., 1], 8)                # [B, 2K]
        would_raise = (p8[:, :K] >= 1.0).any(axis=1) | (p8[:, K:] <= 0.0).any(axis=1)
        np.testing.assert_array_equal((dec[:, 1] & 1).astype(bool), would_raise)
        ok = ~((dec[:, 1] & 2).astype(bool))
        assert np.isfinite(post[ok]).all()
        n_flag += int(would_raise.sum())
    assert n_flag > 20          # the sets do drive the edge

"""CPU tests against the wider reference-generated fixtures (tests/golden/gen_golden.py: calls_branches, region2k, pickles):
the oracle and the host-side row assembly on every ALT / AF / GT / FILTER branch of the reference's call_variants, on a
2 000-candidate region through the reference's four commands, and the checkpoint seam on genuine clairs.model pickles."""
import base64
import gzip
import hashlib
import io
import zlib

import numpy as np
import pytest

from conftest import load_json_gz, load_models_npz, load_genuine_pickle
from weights_recipe import make_weights, CVT_CFG


def _rows(text):
    return [r.split("\t") for r in text.split("\n") if r]


# ------------------------------------------------------------------------------------------ Q4 / Q6 branches
@pytest.mark.parametrize("mode", ["snv", "indel"])
def test_vcf_row_branches_match_reference(oracle_lib, mode):
    """every branch of output_vcf_from_probability (call_variants.py:306-415, 67-76) on the reference's own output: insertion
    ALT (forward / '#' anchor), deletion REF, 1/1, AF clamp, first-seen tie-break, demotion, depth-0 paths, LowQual"""
    from clairs_to_amd.call_variants import load_likelihood, vcf_row
    g = load_json_gz("calls_branches.json.gz")[mode]
    K = g["n_out"]
    rows = _rows(g["predict_rows"])
    lik, edges = load_likelihood(np.loadtxt(g["likelihood_table"].split("\n")), K)
    p1 = np.array([[float(f.split()[1]) for f in r[6:6 + 2 * K]] for r in rows], dtype=np.float64)
    post, dec, qual = oracle_lib.posterior_from_probs(p1, lik, edges)
    assert not dec[:, 1].any()
    seen_gt, seen_flt = set(), set()
    for tag, run in g["runs"].items():
        qual_pass = int(tag.split("_")[0][4:])
        show_ref = tag.endswith("1")
        out, msgs = [], []
        for i, r in enumerate(rows):
            row = vcf_row(r[0], r[1], r[2], r[3], eval(r[4]), eval(r[5]), int(dec[i, 0]), float(qual[i]), K, show_ref=show_ref,
                          qual_pass=qual_pass, messages=msgs)
            if row is not None:
                out.append(row)
        assert out == run["rows"], tag
        assert msgs.count("low tumor coverage") == run["low_cov_messages"], tag
        for r in out:
            c = r.split("\t")
            seen_gt.add(c[9].split(":")[0])
            seen_flt.add(c[6])
    assert seen_gt == {"0/0", "0/1", "1/1"} and seen_flt == {"PASS", "LowQual", "RefCall"}
    if mode == "indel":
        alts = {r.split("\t")[4] for run in g["runs"].values() for r in run["rows"]}
        refs = {r.split("\t")[3] for run in g["runs"].values() for r in run["rows"]}
        assert any(len(a) > 1 for a in alts) and any(len(a) > 1 for a in refs)       # insertions and deletions were emitted


# ------------------------------------------------------------------------------------------ 2 000-site region
def tensor_text(ctg, sites, ref, ref_lo, tensor, alts, flags):
    """rows of create_tensor_pileup_calling.py:561-569 from integer tensors"""
    out = []
    for i, pos in enumerate(sites):
        if flags[i]:
            continue
        o = pos - ref_lo
        seq = ref[o - 16:o + 17]
        out.append("%s\t%d\t%s\t%s\t%s\t%s\t%s\n" % (ctg, pos, seq, " ".join("%d" % v for v in tensor[i].ravel()), alts[i],
                                                      "unknown", seq[16]))
    return "".join(out)


@pytest.mark.parametrize("tag", ["aff", "neg"])
def test_region2k_oracle_tensor_text_is_byte_identical(oracle_lib, region2k, tag):
    r, g = region2k, region2k["g"]
    tensor, depth, alts, flags = oracle_lib.create_tensor(r["texts"][tag], r["ref"], r["ref_lo"], r["sites"])
    text = tensor_text("chr1", r["sites"], r["ref"], r["ref_lo"], tensor, alts, flags)
    want = g["tensor"][tag]
    got_crc = [zlib.crc32(x.encode()) & 0xffffffff for x in text.split("\n") if x]
    assert len(got_crc) == len(want["row_crc"]) == 2000
    bad = [i for i, (a, b) in enumerate(zip(got_crc, want["row_crc"])) if a != b]
    assert not bad, "rows differ from the reference's tensor text: %s" % bad[:10]
    assert hashlib.sha256(text.encode()).hexdigest() == want["sha"]
    d = np.array([int(a.split("-")[0]) for a in want["alt_info"]])
    assert (d <= 50).sum() > 200 and (d > 50).sum() > 200          # both sides of the rescale threshold


@pytest.mark.parametrize("mode", ["snv", "indel"])
def test_region2k_vcf_rows_from_reference_probabilities(oracle_lib, region2k, mode):
    from clairs_to_amd.call_variants import load_likelihood, vcf_row
    g = region2k["g"]
    c = g["calls"][mode]
    K = c["n_out"]
    alt_by_pos = dict(zip(g["tensor"]["aff"]["pos"], g["tensor"]["aff"]["alt_info"]))
    lik, edges = load_likelihood(np.loadtxt(c["likelihood_table"].split("\n")), K)
    p1 = np.array([[float(v) for v in row] for row in c["p1"]], dtype=np.float64)
    post, dec, qual = oracle_lib.posterior_from_probs(p1, lik, edges)
    out = []
    for i, pos in enumerate(c["pos"]):
        row = vcf_row("chr1", pos, c["ref"][i], alt_by_pos[pos], eval(c["strand"][i][0]), eval(c["strand"][i][1]), int(dec[i, 0]),
                      float(qual[i]), K, show_ref=True)
        if row is not None:
            out.append(row)
    assert len(c["vcf_show_ref"]) > 1900
    assert out == c["vcf_show_ref"]


def test_region2k_oracle_probabilities_on_a_subset(oracle_lib, region2k):
    """tensor -> rescale -> both networks -> softmax on the first 192 predict rows (the GPU test covers all of them)"""
    from clairs_to_amd.synth import lik_and_edges
    r, g = region2k, region2k["g"]
    c = g["calls"]["snv"]
    n = 192
    pos = c["pos"][:n]
    ta, da, _, fa = oracle_lib.create_tensor(r["texts"]["aff"], r["ref"], r["ref_lo"], pos)
    tn, dn, _, fn = oracle_lib.create_tensor(r["texts"]["neg"], r["ref"], r["ref_lo"], pos)
    assert not fa.any() and not fn.any()
    ga, gn = load_models_npz("CvT"), load_models_npz("BiGRU_NACGT")
    la = oracle_lib.cvt_forward(make_weights(ga["manifest"], seed=4), dict(CVT_CFG, n_out=4), oracle_lib.rescale(ta, da))
    ln = oracle_lib.bigru_forward(make_weights(gn["manifest"], seed=4), 4, oracle_lib.rescale(tn, dn))
    lik, edges = lik_and_edges(np.loadtxt(c["likelihood_table"].split("\n")), 4)
    probs, _, _, _ = oracle_lib.posterior(la, ln, lik, edges)
    want = np.array([[float(v) for v in row] for row in c["p1"][:n]])
    # the reference accumulates in fp32 (torch CPU), the oracle in fp64: their difference is the reference's own rounding
    # noise (measured 2.5e-6 max on these logits), far below the 1e-4 bar the HIP path is held to
    np.testing.assert_allclose(probs[:, :, 1], want, rtol=0, atol=5e-6)
    assert (da > 50).any() and (da <= 50).any() and (dn > 50).any()


# ------------------------------------------------------------------------------------------ genuine pickles
@pytest.mark.parametrize("name,emb,heads,depth", [("CvT", (16, 64, 128), (1, 3, 4), (1, 2, 3)),
                                                  ("CvT_Indel", (16, 64, 128), (1, 3, 4), (1, 2, 3)),
                                                  ("CvT:defaults", (32, 64, 128), (1, 3, 6), (1, 2, 10)),
                                                  ("BiGRU_NACGT", None, None, None), ("BiGRU_NACGT_Indel", None, None, None)])
def test_genuine_reference_pickles_resolve_onto_the_shims(name, emb, heads, depth):
    from clairs_to_amd import nn_shims
    m, g = load_genuine_pickle(name)
    cls = name.split(":")[0]
    assert type(m) is getattr(nn_shims, cls)
    assert list(m.state_dict().keys()) == g["state_keys"]
    for k, v in g["attrs"].items():
        assert getattr(m, k) == v
    if emb is not None:
        cfg = m._cfg()
        assert tuple(cfg.emb_dim) == emb and tuple(cfg.heads) == heads and tuple(cfg.depth) == depth
        assert cfg.n_out == (6 if cls.endswith("Indel") else 4)
    assert sum(p.numel() for p in m.parameters()) == g["n_params"]

"""Pins the CPU oracle (oracle/cto_oracle.c) to the golden fixtures produced by the reference itself
(tests/golden/gen_golden.py).  CPU only."""
import numpy as np
import pytest

from conftest import load_json_gz, load_models_npz, parse_tensor_text
from weights_recipe import make_weights, CVT_CFG


def test_decode_columns_match_reference(oracle_lib):
    cases = load_json_gz("columns.json.gz")
    assert len(cases) >= 200
    for i, c in enumerate(cases):
        tensor, depth, alt = oracle_lib.decode_column(c["bases"], c["bq"], c["mq"], c["ref"], c["chunk_ref"], c["cand"])
        assert tensor == c["tensor"], "column %d tensor" % i
        assert alt == c["alt_info"], "column %d alt_info" % i
        assert depth == int(c["alt_info"].split("-")[0])


@pytest.mark.parametrize("tag,text_key", [("aff", "mpileup_aff"), ("neg", "mpileup_neg")])
def test_create_tensor_matches_reference(oracle_lib, golden_region, tag, text_key):
    g = golden_region
    rows, X, alts = parse_tensor_text(g["tensor_" + tag])
    tensor, depth, alt_list, flags = oracle_lib.create_tensor(g[text_key], g["ref"], g["ref_start"], g["sites"])
    kept = [i for i in range(len(g["sites"])) if not flags[i]]
    assert [int(r[1]) for r in rows] == [g["sites"][i] for i in kept]      # same sites emitted, same order
    assert 0 < len(kept) < len(g["sites"])                                  # the fixture exercises the skip rules
    np.testing.assert_array_equal(tensor[kept], X)
    assert [alt_list[i] for i in kept] == alts
    assert [int(a.split("-")[0]) for a in alts] == depth[kept].tolist()


@pytest.mark.parametrize("cls", ["CvT", "CvT_Indel", "BiGRU_NACGT", "BiGRU_NACGT_Indel"])
def test_model_forward_matches_reference(oracle_lib, cls):
    g = load_models_npz(cls)
    w = make_weights(g["manifest"], seed=g["n_out"])
    if cls.startswith("CvT"):
        out = oracle_lib.cvt_forward(w, dict(CVT_CFG, n_out=g["n_out"]), g["x"])
    else:
        out = oracle_lib.bigru_forward(w, g["n_out"], g["x"])
    assert out.shape == g["logits"].shape
    # tolerance: fp32 reference vs double-accumulating restatement (1e-5 abs on logits)
    np.testing.assert_allclose(out, g["logits"], rtol=0, atol=2e-5)


def _parse_predict_rows(text, n_out):
    rows = [r.split("\t") for r in text.strip().split("\n") if r]
    probs = np.array([[[float(v) for v in f.split()] for f in r[6:6 + 2 * n_out]] for r in rows], dtype=np.float64)
    return rows, probs   # probs [B][2K][2]


@pytest.mark.parametrize("mode,aff_cls,neg_cls", [("snv", "CvT", "BiGRU_NACGT"), ("indel", "CvT_Indel", "BiGRU_NACGT_Indel")])
def test_predict_rows_match_reference(oracle_lib, golden_region, mode, aff_cls, neg_cls):
    """tensor text -> rescale -> both nets -> softmax, against the reference's own predict output
    (probabilities are printed with 8 decimals)."""
    from clairs_to_amd.synth import lik_and_edges
    calls = load_json_gz("calls_%s.json.gz" % mode)
    K = calls["n_out"]
    rows, probs_ref = _parse_predict_rows(calls["predict_rows"], K)
    ra, Xa, alts_a = parse_tensor_text(golden_region["tensor_aff"])
    rn, Xn, alts_n = parse_tensor_text(golden_region["tensor_neg"])
    keep = [i for i, r in enumerate(ra) if r[2][16] in "ACGT"]          # predict.py:219
    assert len(rows) == len(keep)
    da = np.array([int(a.split("-")[0]) for a in alts_a])
    dn = np.array([int(a.split("-")[0]) for a in alts_n])
    xa = oracle_lib.rescale(Xa, da)[keep]
    xn = oracle_lib.rescale(Xn, dn)[keep]
    ga, gn = load_models_npz(aff_cls), load_models_npz(neg_cls)
    la = oracle_lib.cvt_forward(make_weights(ga["manifest"], seed=K), dict(CVT_CFG, n_out=K), xa)
    ln = oracle_lib.bigru_forward(make_weights(gn["manifest"], seed=K), K, xn)
    table = np.loadtxt(calls["likelihood_table"].split("\n"))
    lik, edges = lik_and_edges(table, K)
    probs, post, dec, qual = oracle_lib.posterior(la, ln, lik, edges)
    np.testing.assert_allclose(probs, probs_ref, rtol=0, atol=2e-6)
    # strand counts (predict.py:626-642) are printed as python float lists
    f, r = oracle_lib.strand_counts(Xa[keep])
    for i, row in enumerate(rows):
        assert row[4] == str([float(v) for v in f[i]])
        assert row[5] == str([float(v) for v in r[i]])
        assert row[3] == alts_a[keep[i]]


def test_c_text_writer_equals_python_writer(oracle_lib):
    """the C pack->mpileup-text helper used for cpu_baseline samples must render exactly what synth.mpileup_text does"""
    from clairs_to_amd.synth import SynthChunk, mpileup_text
    chunk = SynthChunk(25, seed=77, p_ins=0.05, p_del=0.05, spacing=20)
    for q in (0, 20, 45):
        assert oracle_lib.synth_mpileup_text(chunk, q).decode() == mpileup_text(chunk, q)
    assert oracle_lib.synth_mpileup_text(chunk, 0, (3, 9)).decode() == mpileup_text(chunk, 0, col_range=(3, 9))
    assert oracle_lib.synth_mpileup_text(chunk, 20, None, 20, False).decode() == mpileup_text(chunk, 20, min_mq=20, with_mq=False)


def test_extract_candidates_match_reference(oracle_lib):
    """the reference's own extract_candidates_calling output (SNV / indel candidate BED files) vs the restatement"""
    g = load_json_gz("extract.json.gz")
    pr = g["params"]
    pos, flags, depth = oracle_lib.extract_candidates(g["mpileup_extract"], g["ref"], g["ref_start"], pr["snv_min_af"],
                                                      pr["indel_min_af"], pr["min_coverage"], pr["alt_base_num"], True)
    assert pos[(flags & 1) != 0].tolist() == g["snv"] and len(g["snv"]) > 10
    assert pos[(flags & 2) != 0].tolist() == g["indel"] and len(g["indel"]) > 0
    assert ((flags & 4) != 0).sum() >= len(g["snv"])


@pytest.mark.parametrize("name", ["ont_hybrid", "ont_genotyping", "ont_hybrid_indel"])
def test_hybrid_rows_match_reference(oracle_lib, name):
    """cli_run.json.gz: the `<ctg>.<chunk>_hybrid_info` files and the candidate lists the reference's extract_candidates_calling wrote for
    run_clairs_to's hybrid / genotyping command lines (and the same with indel candidates selected) vs the restatement - the injected
    candidates of :374-383, the count form and the fraction form (a row that passes the gates) of the rows of :352-354."""
    import clisim
    from clairs_to_amd.synth import mpileup_text
    g = load_json_gz("cli_run.json.gz")
    rec = g["executed"][name]
    ch = clisim.chunk()
    seq, L = clisim.contig(ch)
    select_indel = name.endswith("_indel")
    known = sorted({int(r.split("\t")[1]) for r in g["inputs"]["vcf"].split("\n") if r and r[0] != "#" and r.split("\t")[0] == clisim.CTG})
    text = mpileup_text(ch, 20, ctg=clisim.CTG, min_mq=20, with_mq=False)
    n_rows = n_frac = 0
    for chunk_id in range(3):
        size = L // 3 + 1 if L % 3 else L // 3
        if name == "ont_genotyping":                    # the region comes from the span of split_beds/<ctg> (:249-260), the rows from its intervals
            iv = [tuple(int(v) for v in r.split()[1:3]) for r in rec["split_beds"][clisim.CTG].split("\n") if r]
            b0, b1 = min(a for a, _ in iv), max(b for _, b in iv)
            size = (b1 - b0) // 3 + 1 if (b1 - b0) % 3 else (b1 - b0) // 3
            start = b0 + 1 + size * chunk_id
        else:
            iv = None
            start = size * chunk_id
        lo, hi = max(start - 33, 1), start + size + 33
        rows = [r for r in text.split("\n") if r and lo <= int(r.split("\t")[1]) <= hi and
                (iv is None or any(a < int(r.split("\t")[1]) <= b for a, b in iv))]
        sub = "\n".join(rows) + "\n"
        want = rec["candidates"]["%s.%d_hybrid_info" % (clisim.CTG, chunk_id)]
        got = oracle_lib.hybrid_info_rows(sub, seq, 1, known, clisim.CTG, 0.05, 0.1, 4, 3, select_indel)
        assert got == want
        n_rows += want.count("\n")
        n_frac += sum("." in r.split("\t")[3] for r in want.split("\n") if r)
        # the SNV list with the injected positions
        pos, flags, _ = oracle_lib.extract_candidates(sub, seq, 1, 0.05, 0.1, 4, 3, select_indel)
        hyb = np.isin(pos, known)
        snv = pos[((flags & 1) != 0) | (hyb & ((flags & 4) == 0) & ((flags & 8) != 0))]
        bed = rec["candidates"].get("%s.%d_0_1_snv" % (clisim.CTG, chunk_id), "")
        assert [int(r.split("\t")[2]) - 17 for r in bed.split("\n") if r] == snv.tolist()
    assert n_rows > 50 and n_frac >= 6

"""CPU tests of the host-side C entry points added in round 2: cto_vcf_rows_batch (all VCF rows of a chunk in one call) against
the reference's own rows and against the one-site Python form, and the packed-weights manifest against the state_dict of the
reference's own pickled modules."""
import numpy as np
import pytest

from conftest import load_json_gz


def _pack_strings(strs):
    off = np.zeros(len(strs) + 1, dtype=np.int64)
    raw = []
    for i, s in enumerate(strs):
        b = s.encode()
        raw.append(b)
        off[i + 1] = off[i] + len(b)
    return b"".join(raw), off


def _batch(rows, dec, qual, K, show_ref, qual_pass):
    from clairs_to_amd.call_variants import vcf_rows_batch
    n = len(rows)
    info = np.zeros((n, 12), dtype=np.int32)
    for i, r in enumerate(rows):
        info[i, 4:8] = [int(v) for v in eval(r[4])]
        info[i, 8:12] = [int(v) for v in eval(r[5])]
    alt_buf, alt_off = _pack_strings([r[3] for r in rows])
    centre = "".join(r[2] for r in rows).encode()
    text, cnt = vcf_rows_batch(rows[0][0], [int(r[1]) for r in rows], centre, alt_buf, alt_off, info, dec, qual, K,
                               show_ref=show_ref, qual_pass=qual_pass)
    return [x for x in text.split("\n") if x], cnt


@pytest.mark.parametrize("mode", ["snv", "indel"])
def test_c_rows_match_reference_on_every_branch(oracle_lib, mode):
    from clairs_to_amd.call_variants import load_likelihood
    g = load_json_gz("calls_branches.json.gz")[mode]
    K = g["n_out"]
    rows = [r.split("\t") for r in g["predict_rows"].split("\n") if r]
    lik, edges = load_likelihood(np.loadtxt(g["likelihood_table"].split("\n")), K)
    p1 = np.array([[float(f.split()[1]) for f in r[6:6 + 2 * K]] for r in rows], dtype=np.float64)
    post, dec, qual = oracle_lib.posterior_from_probs(p1, lik, edges)
    for tag, run in g["runs"].items():
        got, cnt = _batch(rows, dec, qual, K, tag.endswith("1"), int(tag.split("_")[0][4:]))
        assert got == run["rows"], tag
        assert cnt["low_coverage"] == run["low_cov_messages"] and cnt["rows"] == len(run["rows"]) and cnt["sites"] == len(rows)


@pytest.mark.parametrize("mode", ["snv", "indel"])
def test_c_rows_match_reference_on_2000_sites(oracle_lib, mode):
    from clairs_to_amd.call_variants import load_likelihood
    g = load_json_gz("region2k.json.gz")
    c = g["calls"][mode]
    K = c["n_out"]
    alt_by_pos = dict(zip(g["tensor"]["aff"]["pos"], g["tensor"]["aff"]["alt_info"]))
    rows = [["chr1", str(p), c["ref"][i], alt_by_pos[p], c["strand"][i][0], c["strand"][i][1]] for i, p in enumerate(c["pos"])]
    lik, edges = load_likelihood(np.loadtxt(c["likelihood_table"].split("\n")), K)
    p1 = np.array([[float(v) for v in row] for row in c["p1"]], dtype=np.float64)
    post, dec, qual = oracle_lib.posterior_from_probs(p1, lik, edges)
    got, cnt = _batch(rows, dec, qual, K, True, 0)
    assert got == c["vcf_show_ref"] and cnt["rows"] == len(got) > 1900


def test_c_rows_equal_python_rows_on_random_alt_info():
    """fuzz: random allele tables (duplicate keys, zero counts, '#' anchors, depth 0, long keys), every arg-max, both modes, all
    option combinations - the C batch and the one-site Python form must agree row for row"""
    from clairs_to_amd.call_variants import vcf_row
    rng = np.random.default_rng(5)
    keys = ["XA", "XC", "XG", "XT", "R", "IAC", "IAGT", "I#G", "I#TTA", "DAC", "DACGT", "DA", "ICAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAT"]
    for K in (4, 6):
        rows, dec, qual = [], [], []
        for i in range(1500):
            nk = int(rng.integers(0, 6))
            ks = [keys[j] for j in rng.integers(0, len(keys), size=nk)]
            toks = " ".join("%s %d" % (k, int(rng.integers(0, 40))) for k in ks)
            depth = int(rng.choice([0, 0, 1, 7, 30, 55]))
            alt = "%d-%s-" % (depth, toks)
            if rng.random() < 0.03:
                alt = "%d-" % depth
            ref = "ACGT"[rng.integers(0, 4)] if rng.random() < 0.97 else "N"
            f, r = rng.integers(0, 30, size=4), rng.integers(0, 30, size=4)
            rows.append(["chr7", str(100 + i), ref, alt, str([float(v) for v in f]), str([float(v) for v in r])])
            dec.append([int(rng.integers(0, K)), 0, 0, 0])
            qual.append(float(np.round(rng.uniform(0, 60), 4)) if rng.random() < 0.9 else 0.0)
        dec, qual = np.array(dec, dtype=np.int32), np.array(qual)
        for show_ref in (False, True):
            for qp in (0, 20, None):
                want, msgs = [], []
                for i, r in enumerate(rows):
                    row = vcf_row(r[0], r[1], r[2], r[3], eval(r[4]), eval(r[5]), int(dec[i, 0]), float(qual[i]), K, show_ref=show_ref,
                                  qual_pass=qp, messages=msgs)
                    if row is not None:
                        want.append(row)
                got, cnt = _batch(rows, dec, qual, K, show_ref, qp)
                assert got == want, (K, show_ref, qp)
                assert cnt["low_coverage"] == len(msgs)


def test_c_rows_skip_flagged_sites_and_reject_garbage():
    from clairs_to_amd.call_variants import vcf_rows_batch
    from clairs_to_amd._lib import CtoError
    info = np.zeros((3, 12), dtype=np.int32)
    info[1, 3] = 1                                        # no tensor for this site
    dec = np.array([[3, 0, 0, 0], [3, 0, 0, 0], [3, 3, 0, 0]], dtype=np.int32)      # site 2: NaN winner flag
    alt_buf, alt_off = _pack_strings(["9-XT 9-", "9-XT 9-", "9-XT 9-"])
    text, cnt = vcf_rows_batch("c", [5, 6, 7], b"AAA", alt_buf, alt_off, info, dec, np.array([30.0, 30.0, 0.0]), 4)
    assert text.count("\n") == 1 and text.startswith("c\t5\t.\tA\tT\t30.0000\tPASS") and cnt == dict(rows=1, sites=2, low_coverage=0, clamped=1)
    bad_buf, bad_off = _pack_strings(["x-XT 9-"])
    with pytest.raises(CtoError):
        vcf_rows_batch("c", [5], b"A", bad_buf, bad_off, info[:1], dec[:1], np.array([1.0]), 4)
    with pytest.raises(CtoError):
        vcf_rows_batch("c", [5], b"A", *_pack_strings(["9-XT 9-"]), info[:1], np.array([[9, 0, 0, 0]], dtype=np.int32), np.array([1.0]), 4)


@pytest.mark.parametrize("name", ["CvT", "CvT_Indel", "CvT:defaults", "BiGRU_NACGT", "BiGRU_NACGT_Indel"])
def test_packed_manifest_is_the_reference_state_dict_order(name):
    """cto_model_manifest (the order cto_*_create_packed and the torch ops' packed_weights use) against the state_dict keys and
    shapes of the REFERENCE's own pickled modules"""
    from conftest import load_genuine_pickle
    from clairs_to_amd._lib import model_manifest
    m, g = load_genuine_pickle(name)
    sd = [(k, v.numel()) for k, v in m.state_dict().items() if v.is_floating_point()]
    assert [k for k, _ in sd] == [k for k in g["state_keys"] if not k.endswith("num_batches_tracked")]
    if name.startswith("CvT"):
        got = model_manifest(0, m._cfg())
    else:
        got = model_manifest(1, None, len(m._heads_out))
    assert got == sd

"""GPU parity tests (run with -m gpu on the MI355X box): the HIP path through the C ABI against
(a) the golden fixtures produced by the reference and (b) the CPU oracle on seeded synthetic inputs."""
import numpy as np
import pytest

from conftest import load_json_gz, load_models_npz, parse_tensor_text
from weights_recipe import make_weights, CVT_CFG

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return torch.device("cuda:0")


def _featurize_text(text, ref, ref_start, sites, min_bq, dev, rescale=50):
    import torch
    from clairs_to_amd.pack import ColumnPack
    from clairs_to_amd.featurize import featurize, alt_infos
    pack = ColumnPack.from_mpileup(text, ref, ref_start)
    dp = pack.to_device(dev)
    feat = featurize(dp, torch.tensor(sites, dtype=torch.int32, device=dev), min_bq, rescale, want_raw=True)
    torch.cuda.synchronize()
    return pack, feat, alt_infos(feat, pack)


def test_featurize_matches_reference_region(dev, golden_region):
    """mpileup text (NEG pass, --min-BQ 0) -> pack -> HIP: both tensors, depths, alt_info and skip rules must equal
    what the reference's create_tensor_pileup_calling wrote for its two passes (bit-exact)."""
    g = golden_region
    pack, feat, alts = _featurize_text(g["mpileup_neg"], g["ref"], g["ref_start"], g["sites"], g["min_bq_aff"], dev)
    info = feat.site_info.cpu().numpy()
    kept = np.nonzero(info[:, 3] == 0)[0]
    for tag, raw, dcol in (("aff", feat.raw_aff, 1), ("neg", feat.raw_neg, 2)):
        rows, X, alt_ref = parse_tensor_text(g["tensor_" + tag])
        assert [int(r[1]) for r in rows] == [g["sites"][i] for i in kept]
        np.testing.assert_array_equal(raw.cpu().numpy()[kept].astype(np.int32), X)
        assert info[kept, dcol].tolist() == [int(a.split("-")[0]) for a in alt_ref]
        if tag == "aff":
            assert [alts[i] for i in kept] == alt_ref
        else:
            from clairs_to_amd.featurize import alt_infos
            alts_neg = alt_infos(feat, pack, info, pass_idx=1)
            assert [alts_neg[i] for i in kept] == alt_ref


@pytest.mark.parametrize("p_ins,p_del,n_sites", [(0.01, 0.02, 300), (0.25, 0.25, 60)])
def test_featurize_matches_oracle_synthetic(dev, oracle_lib, p_ins, p_del, n_sites):
    """Seeded synthetic chunk generated directly as a pack (the bench generator) vs the oracle run on the
    equivalent mpileup text of each pass; includes the rescaled fp32 network inputs and strand counts.
    The second case has so many distinct indel keys per column that the kernel's LDS key table overflows and the
    global-atomic path runs."""
    import torch
    from clairs_to_amd.pack import DevicePack
    from clairs_to_amd.featurize import featurize
    from clairs_to_amd.synth import SynthChunk, mpileup_text
    chunk = SynthChunk(n_sites, seed=123, spacing=40, p_ins=p_ins, p_del=p_del, depth_mean=70.0)
    if p_ins > 0.1:
        assert np.diff(chunk.key_off[::16]).max() > 128
    ref, lo = chunk.ref_window()
    dp = DevicePack(chunk.arrays(), dev)
    feat = featurize(dp, torch.from_numpy(chunk.site_pos).to(dev), 20, 50, want_raw=True)
    torch.cuda.synchronize()
    info = feat.site_info.cpu().numpy()
    for q, raw, x, dcol in ((20, feat.raw_aff, feat.x_aff, 1), (0, feat.raw_neg, feat.x_neg, 2)):
        t, depth, _, flags = oracle_lib.create_tensor(mpileup_text(chunk, min_bq=q), ref, lo, chunk.site_pos)
        np.testing.assert_array_equal(info[:, 3] & 1, flags)
        np.testing.assert_array_equal(raw.cpu().numpy().astype(np.int32), t)
        np.testing.assert_array_equal(info[:, dcol], depth)
        np.testing.assert_array_equal(x.cpu().numpy(), oracle_lib.rescale(t, depth))      # bit-exact fp32
        if q == 20:
            f, r = oracle_lib.strand_counts(t)
            np.testing.assert_array_equal(info[:, 4:8], f)
            np.testing.assert_array_equal(info[:, 8:12], r)


def test_featurize_empty_and_ragged(dev):
    import torch
    from clairs_to_amd.pack import ColumnPack
    from clairs_to_amd.featurize import featurize
    # empty pack, sites without any row
    pack = ColumnPack.from_mpileup("", "ACGT" * 50, 1)
    feat = featurize(pack.to_device(dev), torch.tensor([50, 60], dtype=torch.int32, device=dev), 20, 50, want_raw=True)
    torch.cuda.synchronize()
    assert feat.site_info.cpu().numpy()[:, 3].tolist() == [1, 1]
    assert int(feat.raw_aff.abs().sum()) == 0
    # a single deep column (several 64-entry chunks in one wave) and an empty column
    text = "chr1\t40\tN\t300\t" + "A" * 150 + "c" * 150 + "\t" + "I" * 300 + "\t" + "]" * 300 + "\n" + \
           "chr1\t41\tN\t0\t*\t*\t*\n"
    pack = ColumnPack.from_mpileup(text, "ACGT" * 50, 1)
    feat = featurize(pack.to_device(dev), torch.tensor([40], dtype=torch.int32, device=dev), 20, 0, want_raw=True)
    torch.cuda.synchronize()
    raw = feat.raw_aff.cpu().numpy()[0]
    ref40 = ("ACGT" * 50)[39]          # 'T'
    assert ref40 == "T"
    assert raw[16, 0] == 150 and raw[16, 10] == 150 and raw[16, 3] == -150 and raw[16, 12] == -150
    assert feat.site_info.cpu().numpy()[0, 1] == 300


@pytest.mark.parametrize("cls", ["CvT", "CvT_Indel", "BiGRU_NACGT", "BiGRU_NACGT_Indel"])
def test_models_match_reference_logits(dev, cls):
    """HIP forward vs the logits the reference's own modules produced (fp32 CPU torch); bar 1e-4 (north_star)."""
    import torch
    from clairs_to_amd.nn_shims import from_state_dict
    g = load_models_npz(cls)
    m = from_state_dict(cls, make_weights(g["manifest"], seed=g["n_out"]))
    out = m(torch.from_numpy(g["x"]).to(dev))
    assert isinstance(out, tuple) and len(out) == g["n_out"] and tuple(out[0].shape) == (g["x"].shape[0], 2)
    got = torch.stack(out).cpu().numpy()
    np.testing.assert_allclose(got, g["logits"], rtol=0, atol=1e-4)
    p_got = torch.softmax(torch.stack(out), dim=-1).cpu().numpy()
    p_ref = torch.softmax(torch.from_numpy(g["logits"]), dim=-1).numpy()
    assert np.abs(p_got - p_ref).max() < 1e-4


@pytest.mark.parametrize("cls,B", [("CvT", 700), ("BiGRU_NACGT_Indel", 333)])
def test_models_match_oracle_ragged_batch(dev, oracle_lib, cls, B):
    """Batch sizes that are not multiples of any tile, inputs with realistic count statistics."""
    import torch
    from clairs_to_amd.nn_shims import from_state_dict
    g = load_models_npz(cls)
    w = make_weights(g["manifest"], seed=g["n_out"])
    rng = np.random.default_rng(5)
    x = g["x"][rng.integers(0, g["x"].shape[0], size=B)] * rng.uniform(0.5, 1.5, size=(B, 1, 1)).astype(np.float32)
    m = from_state_dict(cls, w)
    got = torch.stack(m(torch.from_numpy(x).to(dev))).cpu().numpy()
    if cls.startswith("CvT"):
        want = oracle_lib.cvt_forward(w, dict(CVT_CFG, n_out=g["n_out"]), x)
    else:
        want = oracle_lib.bigru_forward(w, g["n_out"], x)
    np.testing.assert_allclose(got, want, rtol=0, atol=1e-4)
    assert m.logits(torch.zeros((0, 33, 34), device=dev)).shape == (g["n_out"], 0, 2)


@pytest.mark.parametrize("fused", ["1", "0"])
def test_cvt_constructor_default_config_matches_oracle(dev, oracle_lib, fused, monkeypatch):
    """clairs/model.py:153-177 defaults (emb 32/64/128, heads 1/3/6, depth 1/2/10) - the shipped SNV pickles may use
    them; exercises the C=32 stage-1 and heads=6 stage-3 instantiations of the fused block kernel, and the unfused
    kernel sequence (CTO_CVT_UNFUSED=1) on the same weights."""
    import torch
    from clairs_to_amd.nn_shims import CvT
    from clairs_to_amd.engine import random_state_dict
    monkeypatch.setenv("CTO_CVT_UNFUSED", "0" if fused == "1" else "1")
    m = CvT(model_type="acgt").eval()
    w = random_state_dict(m, seed=11)
    sd = m.state_dict()
    for k, v in w.items():
        sd[k] = torch.from_numpy(v)
    m.load_state_dict(sd)
    x = load_models_npz("CvT")["x"][:37]
    got = m.logits(torch.from_numpy(x).to(dev)).cpu().numpy()
    want = oracle_lib.cvt_forward(w, dict(emb_dim=(32, 64, 128), heads=(1, 3, 6), depth=(1, 2, 10), n_out=4), x)
    np.testing.assert_allclose(got, want, rtol=0, atol=1e-4)


def test_model_rejects_cpu_input():
    import torch
    from clairs_to_amd.nn_shims import from_state_dict
    g = load_models_npz("BiGRU_NACGT")
    m = from_state_dict("BiGRU_NACGT", make_weights(g["manifest"], seed=4))
    with pytest.raises(RuntimeError):
        m(torch.zeros((1, 33, 34)))


@pytest.mark.parametrize("K", [4, 6])
def test_posterior_matches_oracle(dev, oracle_lib, K):
    import torch
    from clairs_to_amd.call_variants import Posterior
    from clairs_to_amd.synth import likelihood_table, lik_and_edges
    rng = np.random.default_rng(K)
    B = 5000
    aff = rng.normal(0, 3, size=(K, B, 2)).astype(np.float32)
    neg = rng.normal(0, 3, size=(K, B, 2)).astype(np.float32)
    aff[0, :5] = [[-30, 30], [30, -30], [0, 0], [-1.7, 20], [20, -1.7]]        # saturating probabilities
    lik, edges = lik_and_edges(likelihood_table(K), K)
    post = Posterior(lik, edges, dev)
    out = post(torch.from_numpy(aff).to(dev), torch.from_numpy(neg).to(dev))
    probs, p, dec, qual = oracle_lib.posterior(aff, neg, lik, edges)
    np.testing.assert_allclose(out["probs"].cpu().numpy(), probs, rtol=0, atol=2e-7)
    # feed the device its own 8-decimal probabilities through the text-seam entry: must equal the oracle bit for bit
    p8 = np.round(probs[:, :, 1].astype(np.float64) * 1e8) / 1e8
    o2 = post.from_probs(torch.from_numpy(p8).to(dev))
    p_o, d_o, q_o = oracle_lib.posterior_from_probs(p8, lik, edges)
    np.testing.assert_array_equal(o2["post"].cpu().numpy(), p_o)
    # QUAL = round(q, 4) exactly as Python rounds (posterior.hip round4).  The device's log() may differ from the host libm's in
    # its last bit; the sites where that could matter (q * 1e4 within 1e-6 of a ...5 boundary) are flagged by the kernel and
    # re-evaluated with the host libm by finalize_qual: bit-exact, asserted
    from clairs_to_amd.call_variants import finalize_qual
    q_d, dec_d = o2["qual"].cpu().numpy(), o2["decision"].cpu().numpy()
    finalize_qual(dec_d, q_d)
    np.testing.assert_array_equal(q_d, q_o)
    np.testing.assert_array_equal(dec_d, d_o)
    assert d_o[:, 1].sum() > 0          # the clamp (reference IndexError) case is exercised
    # saturated, contradictory heads: p = 0.00000000 from both networks -> 0/0; np.argmax semantics (first NaN wins) and
    # flag bit 1 so that the host formats no row from it
    p_nan = p8[:4].copy()
    p_nan[:, 1] = 0.0
    p_nan[:, K + 1] = 0.0
    p_nan[2:, 0] = 0.0
    p_nan[2:, K] = 0.0                  # rows 2, 3: heads 0 and 1 both NaN -> index 0
    o3 = post.from_probs(torch.from_numpy(p_nan).to(dev))
    _, d3, _ = oracle_lib.posterior_from_probs(p_nan, lik, edges)
    np.testing.assert_array_equal(o3["decision"].cpu().numpy(), d3)
    assert d3[:, 0].tolist() == [1, 1, 0, 0] and (d3[:, 1] == 3).all()


def test_qual_on_rounding_boundaries_is_the_hosts(dev, oracle_lib):
    """QUAL = round(q, 4) of clairs/call_variants.py:79-88 with q sitting ON a 4-decimal rounding boundary: probabilities are tuned
    (bisection on the host, full double resolution through the text-seam entry) until q * 1e4 is within ~1e-9 of a ...5 value, on
    both sides.  The kernel must flag every such site; after the host half (cto_qual_finalize: the host's own libm, the one
    math.log uses) QUAL and decision equal the oracle bit for bit, and cto_vcf_rows_batch prints the same digits unfinalised."""
    import torch
    from math import log, e
    from clairs_to_amd.call_variants import Posterior, finalize_qual, vcf_rows_batch
    from clairs_to_amd.synth import likelihood_table, lik_and_edges
    K = 4
    lik, edges = lik_and_edges(likelihood_table(K), K)
    post = Posterior(lik, edges, dev)
    rng = np.random.default_rng(7)

    def q_of(row):
        p_o, d_o, _ = oracle_lib.posterior_from_probs(row[None, :], lik, edges)
        v = float(p_o[0, d_o[0, 0]])
        return max((-10 * log(e, 10)) * log(((1.0 - v) + 1e-10) / (v + 1e-10)) + 2.0, 0.0)

    rows = []
    while len(rows) < 64:
        row = np.round(rng.uniform(0.02, 0.3, size=2 * K), 8)
        row[0] = 0.9                                   # head 0 wins clearly; tune its AFF probability inside its likelihood bin
        e0 = edges[0]
        b = int(np.searchsorted(e0, 0.9, side="right")) - 1
        lo, hi = max(e0[b], 0.55) + 1e-9, min(e0[b + 1], 0.999) - 1e-9
        r_lo, r_hi = row.copy(), row.copy()
        r_lo[0], r_hi[0] = lo, hi
        q_lo, q_hi = q_of(r_lo), q_of(r_hi)
        if not (q_hi > q_lo + 2e-4):
            continue
        target = (np.floor(rng.uniform(q_lo, q_hi - 1e-4) * 1e4) + 0.5) / 1e4          # a ...5 boundary inside the reachable range
        if not (q_lo < target < q_hi):
            continue
        for _ in range(80):                             # q is monotone in the winning head's probability
            mid = 0.5 * (lo + hi)
            r_lo[0] = mid
            if q_of(r_lo) < target:
                lo = mid
            else:
                hi = mid
        for x in (lo, hi, np.nextafter(lo, 0), np.nextafter(hi, 1)):
            r = row.copy()
            r[0] = x
            rows.append(r)
    p1 = np.ascontiguousarray(np.stack(rows))
    o = post.from_probs(torch.from_numpy(p1).to(dev))
    p_o, d_o, q_o = oracle_lib.posterior_from_probs(p1, lik, edges)
    dec, qual = o["decision"].cpu().numpy(), o["qual"].cpu().numpy()
    assert ((dec[:, 1] & 4) != 0).mean() > 0.9          # the tuned sites are recognised as boundary cases
    assert len(set(np.round(q_o, 4))) > 16 and (np.diff(q_o.reshape(-1, 4), axis=1) != 0).any()    # both sides of boundaries present
    # rows straight from the unfinalised device outputs: cto_vcf_rows_batch applies the host half itself
    n = len(rows)
    alt = ("20-XC 9 R 11-",) * n
    alt_buf = "".join(alt).encode()
    alt_off = np.arange(n + 1, dtype=np.int64) * len(alt[0])
    info = np.zeros((n, 12), dtype=np.int32)
    pos = np.arange(1000, 1000 + n, dtype=np.int64)
    centre = np.frombuffer(b"C" * n, dtype=np.uint8)
    text_raw, _ = vcf_rows_batch("chr1", pos, centre, alt_buf, alt_off, info, dec.copy(), qual.copy(), K, show_ref=True, qual_pass=0)
    assert finalize_qual(dec, qual) == int(((o["decision"].cpu().numpy()[:, 1] & 4) != 0).sum())
    np.testing.assert_array_equal(qual, q_o)
    np.testing.assert_array_equal(dec, d_o)
    text_fin, _ = vcf_rows_batch("chr1", pos, centre, alt_buf, alt_off, info, dec, qual, K, show_ref=True, qual_pass=0)
    assert text_raw == text_fin and text_fin.count("\n") == n
    assert finalize_qual(dec, qual) == 0                # idempotent


@pytest.mark.parametrize("mode", ["snv", "indel"])
def test_vcf_rows_from_gpu_posterior(dev, mode):
    """probability rows of the reference -> GPU posterior -> host row assembly == reference VCF rows."""
    import torch
    from clairs_to_amd.call_variants import Posterior, load_likelihood, vcf_row, finalize_qual
    calls = load_json_gz("calls_%s.json.gz" % mode)
    K = calls["n_out"]
    rows = [r.split("\t") for r in calls["predict_rows"].strip().split("\n") if r]
    lik, edges = load_likelihood(np.loadtxt(calls["likelihood_table"].split("\n")), K)
    p1 = np.array([[float(f.split()[1]) for f in r[6:6 + 2 * K]] for r in rows], dtype=np.float64)
    o = Posterior(lik, edges, dev).from_probs(torch.from_numpy(p1).to(dev))
    dec, qual = o["decision"].cpu().numpy(), o["qual"].cpu().numpy()
    finalize_qual(dec, qual)
    for show_ref in (False, True):
        out = [vcf_row(r[0], r[1], r[2], r[3], eval(r[4]), eval(r[5]), int(dec[i, 0]), float(qual[i]), K, show_ref=show_ref)
               for i, r in enumerate(rows)]
        assert [x for x in out if x is not None] == calls["vcf"]["show_ref" if show_ref else "default"]


def test_extract_candidates_match_reference_and_oracle(dev, oracle_lib):
    """GPU gates on the column pack vs (a) the candidate files the reference wrote, (b) the oracle on a bigger chunk,
    including a chunk whose merged-allele table overflows LDS."""
    import torch
    from clairs_to_amd.pack import ColumnPack, DevicePack
    from clairs_to_amd.extract_candidates_calling import extract_candidates, candidate_positions
    from clairs_to_amd.synth import SynthChunk
    g = load_json_gz("extract.json.gz")
    pr = g["params"]
    pack = ColumnPack.from_mpileup(g["mpileup_neg"], g["ref"], g["ref_start"])
    dp = pack.to_device(dev)
    flags, depth = extract_candidates(dp, pr["min_bq"], pr["min_mq"], pr["snv_min_af"], pr["indel_min_af"], pr["min_coverage"],
                                      pr["alt_base_num"], True)
    assert candidate_positions(dp, flags, 1).cpu().tolist() == g["snv"]
    assert candidate_positions(dp, flags, 2).cpu().tolist() == g["indel"]
    for kw, n in ((dict(p_mismatch=0.03, p_ins=0.02, p_del=0.03, depth_mean=30.0), 400),
                  (dict(p_ins=0.25, p_del=0.25, depth_mean=60.0), 60)):
        chunk = SynthChunk(n, seed=5, spacing=25, n_rate=0.02, **kw)
        ref, lo = chunk.ref_window()
        dp = DevicePack(chunk.arrays(), dev)
        flags, depth = extract_candidates(dp, 20, 20, 0.05, 0.05, 4, 3, True)
        text6 = oracle_lib.synth_mpileup_text(chunk, 20, None, 20, False)
        pos, fl, dep = oracle_lib.extract_candidates(text6, ref, lo, 0.05, 0.05, 4, 3, True)
        got_f, got_d = flags.cpu().numpy(), depth.cpu().numpy()
        cols = np.searchsorted(chunk.col_pos, pos)                 # rows exist only where some read passes the MQ gate
        np.testing.assert_array_equal(got_f[cols], fl)
        np.testing.assert_array_equal(got_d[cols], dep)
        mask = np.ones(chunk.col_pos.size, bool)
        mask[cols] = False
        assert not got_f[mask].any()
        assert (fl & 1).sum() > 0


def test_full_chunk_properties(dev):
    """Size-independent properties at the bench's full chunk size (4096 sites, ~6.5 M read-bases):
    (1) batch invariance - a site's outputs do not depend on which other sites share its launch (bit-exact),
    (2) the NEG pass sees a superset of the AFF pass's read-bases, so |NEG count| >= |AFF count| everywhere,
    (3) min_bq = 0 makes the two passes identical (the Illumina `ln -sf` of run_clairs_to:1248-1252),
    (4) the epilogue entered at the text seam with its own 8-decimal probabilities reproduces itself."""
    import torch
    from clairs_to_amd.engine import Engine, synthetic_models
    from clairs_to_amd.featurize import featurize
    from clairs_to_amd.synth import SynthChunk, likelihood_table, lik_and_edges
    ch = SynthChunk(4096, seed=20260928)
    models = synthetic_models(4)
    lik, edges = lik_and_edges(likelihood_table(4), 4)
    eng = Engine(models["aff"], models["neg"], lik, edges, min_bq=20, device=dev)
    dp = eng.upload(ch.arrays())
    sp = torch.from_numpy(ch.site_pos).to(dev)
    full = eng.run_device(dp, sp, want_raw=True)
    torch.cuda.synchronize()
    # (1) odd-sized sub-batches
    cuts = [0, 1, 34, 1000, 2049, 4096]
    for a, b in zip(cuts[:-1], cuts[1:]):
        part = eng.run_device(dp, sp[a:b])
        for k in ("aff_logits", "neg_logits"):
            assert torch.equal(part[k], full[k][:, a:b]), (k, a, b)
        for k in ("probs", "post", "decision", "qual"):
            assert torch.equal(part[k], full[k][a:b]), (k, a, b)
    # (2)
    f = full["features"]
    assert bool((f.raw_neg.abs() >= f.raw_aff.abs()).all())
    info = f.site_info
    assert bool((info[:, 2] >= info[:, 1]).all()) and int(info[:, 3].sum()) == 0
    # (3)
    f0 = featurize(dp, sp, 0, 50, want_raw=True)
    assert torch.equal(f0.raw_aff, f0.raw_neg) and torch.equal(f0.x_aff, f0.x_neg)
    assert torch.equal(f0.raw_neg, f.raw_neg)
    # (4)
    p8 = np.round(full["probs"][:, :, 1].cpu().numpy().astype(np.float64) * 1e8) / 1e8    # IEEE division on the host
    again = eng.posterior.from_probs(torch.from_numpy(np.ascontiguousarray(p8)).to(dev))
    assert torch.equal(again["post"], full["post"]) and torch.equal(again["decision"], full["decision"])
    assert torch.equal(again["qual"], full["qual"])


@pytest.mark.parametrize("platform", ["ont", "ilmn", "hifi"])
@pytest.mark.parametrize("K", [4, 6])
def test_platform_configs_end_to_end(dev, oracle_lib, platform, K):
    """BASELINE.json configs 2/4/5 as parity cases (SURVEY.md 8d generator presets): ONT 50x (min_bq 20), Illumina 50x
    (min_bq 0: NEG tensor == AFF tensor, three-valued BQ) and HiFi 75x (most sites rescaled by 50/depth, BQ up to 93), each with
    the SNV (K=4) and the indel (K=6) model pair: whole engine vs the oracle run on the equivalent mpileup text of each pass."""
    import torch
    import oracle
    from clairs_to_amd.engine import Engine, synthetic_models
    from clairs_to_amd.synth import SynthChunk, PLATFORMS, mpileup_text, likelihood_table, lik_and_edges
    min_bq = PLATFORMS[platform]["min_bq"]
    chunk = SynthChunk.for_platform(platform, 160, spacing=60)
    models = synthetic_models(K)
    lik, edges = lik_and_edges(likelihood_table(K), K)
    eng = Engine(models["aff"], models["neg"], lik, edges, min_bq=min_bq, device=dev)
    res = eng.run_chunk(chunk.arrays(), chunk.site_pos, want_raw=True)
    torch.cuda.synchronize()
    ref, lo = chunk.ref_window()
    ta, da, _, _ = oracle.create_tensor(mpileup_text(chunk, min_bq), ref, lo, chunk.site_pos)
    tn, dn, _, _ = oracle.create_tensor(mpileup_text(chunk, 0), ref, lo, chunk.site_pos)
    f = res["features"]
    np.testing.assert_array_equal(f.raw_aff.cpu().numpy().astype(np.int32).reshape(ta.shape), ta)
    np.testing.assert_array_equal(f.raw_neg.cpu().numpy().astype(np.int32).reshape(tn.shape), tn)
    info = f.site_info.cpu().numpy()
    assert info[:, 1].tolist() == da.tolist() and info[:, 2].tolist() == dn.tolist()
    if platform == "hifi":
        assert (da > 50).mean() > 0.9                     # the rescale branch is the common case here
    xa, xn = oracle.rescale(ta, da), oracle.rescale(tn, dn)
    np.testing.assert_array_equal(f.x_aff.cpu().numpy().reshape(xa.shape), xa)
    np.testing.assert_array_equal(f.x_neg.cpu().numpy().reshape(xn.shape), xn)
    la = oracle.cvt_forward(models["aff_weights"], dict(CVT_CFG, n_out=K), xa)
    ln = oracle.bigru_forward(models["neg_weights"], K, xn)
    probs, post, dec, qual = oracle.posterior(la, ln, lik, edges)
    assert np.abs(res["probs"].cpu().numpy() - probs).max() < 1e-4          # north_star tolerance
    # the epilogue itself is exact: the oracle fed with the device's own 8-decimal probabilities reproduces it bit for bit
    p8 = np.round(res["probs"][:, :, 1].cpu().numpy().astype(np.float64) * 1e8) / 1e8
    post2, dec2, qual2 = oracle.posterior_from_probs(p8, lik, edges)
    np.testing.assert_array_equal(res["post"].cpu().numpy(), post2)
    from clairs_to_amd.call_variants import finalize_qual
    dec_d, qual_d = res["decision"].cpu().numpy(), res["qual"].cpu().numpy()
    finalize_qual(dec_d, qual_d)                      # host half of QUAL (boundary sites re-evaluated with the host libm)
    np.testing.assert_array_equal(dec_d, dec2)
    np.testing.assert_array_equal(qual_d, qual2)


@pytest.mark.parametrize("platform", ["ilmn", "hifi"])
@pytest.mark.parametrize("K", [4, 6])
def test_full_size_platform_configs(dev, oracle_lib, platform, K):
    """BASELINE configs[3] (Illumina 50x) and configs[4] (HiFi 75x) at the bench's FULL chunk size, 4096 sites, with the SNV (K=4)
    and the indel (K=6) model pair: (a) the oracle on a 512-site sample of the same chunk - integer tensors, depths and rescaled
    inputs bit-exact, probabilities within 1e-4, epilogue exact on the device's own 8-decimal probabilities; (b) size-independent
    properties over all 4096 sites: batch invariance (bit-exact), NEG pass a superset of the AFF pass, the platform's min_bq = 0
    making both passes one (Illumina: NEG network fed the AFF tensor, run_clairs_to:1248-1252), the rescale of predict.py:179-207
    recomputed in numpy from the device's own integer tensors and depths, the epilogue reproducing itself at the text seam."""
    import torch
    import oracle
    from clairs_to_amd.call_variants import finalize_qual
    from clairs_to_amd.engine import Engine, synthetic_models
    from clairs_to_amd.synth import SynthChunk, PLATFORMS, likelihood_table, lik_and_edges
    min_bq = PLATFORMS[platform]["min_bq"]
    chunk = SynthChunk.for_platform(platform, 4096)
    models = synthetic_models(K)
    lik, edges = lik_and_edges(likelihood_table(K), K)
    eng = Engine(models["aff"], models["neg"], lik, edges, min_bq=min_bq, device=dev, neg_reads_aff=(platform == "ilmn"))
    dp = eng.upload(chunk.arrays())
    sp = torch.from_numpy(chunk.site_pos).to(dev)
    full = eng.run_device(dp, sp, want_raw=True)
    torch.cuda.synchronize()
    f = full["features"]
    raw_a, raw_n = f.raw_aff.cpu().numpy().astype(np.int32), f.raw_neg.cpu().numpy().astype(np.int32)
    info = f.site_info.cpu().numpy()
    # ---- (a) oracle on the first 512 sites ----
    n_s = 512
    sites = chunk.site_pos[:n_s]
    ref, lo = chunk.ref_window()
    c1 = int(np.searchsorted(chunk.col_pos, int(sites[-1]) + 17, side="right"))
    ta, da, _, _ = oracle.create_tensor(oracle.synth_mpileup_text(chunk, min_bq, (0, c1)), ref, lo, sites)
    tn, dn, _, _ = oracle.create_tensor(oracle.synth_mpileup_text(chunk, 0, (0, c1)), ref, lo, sites)
    np.testing.assert_array_equal(raw_a[:n_s].reshape(ta.shape), ta)
    np.testing.assert_array_equal(raw_n[:n_s].reshape(tn.shape), tn)
    assert info[:n_s, 1].tolist() == da.tolist() and info[:n_s, 2].tolist() == dn.tolist()
    xa, xn = oracle.rescale(ta, da), oracle.rescale(tn, dn)
    np.testing.assert_array_equal(f.x_aff[:n_s].cpu().numpy().reshape(xa.shape), xa)
    np.testing.assert_array_equal(f.x_neg[:n_s].cpu().numpy().reshape(xn.shape), xn)
    la = oracle.cvt_forward(models["aff_weights"], dict(CVT_CFG, n_out=K), xa)
    ln = oracle.bigru_forward(models["neg_weights"], K, xn)
    probs, _, _, _ = oracle.posterior(la, ln, lik, edges)
    got = full["probs"].cpu().numpy()
    assert np.abs(got[:n_s] - probs).max() < 1e-4                      # north_star tolerance
    p8 = np.round(got[:, :, 1].astype(np.float64) * 1e8) / 1e8         # all 4096: epilogue exact on the device's own probabilities
    post2, dec2, qual2 = oracle.posterior_from_probs(p8, lik, edges)
    dec_d, qual_d = full["decision"].cpu().numpy(), full["qual"].cpu().numpy()
    finalize_qual(dec_d, qual_d)
    np.testing.assert_array_equal(full["post"].cpu().numpy(), post2)
    np.testing.assert_array_equal(dec_d, dec2)
    np.testing.assert_array_equal(qual_d, qual2)
    assert len(set(dec2[:, 0].tolist())) > 1
    # ---- (b) properties over all 4096 sites ----
    for a, b in ((0, 1), (1, 35), (35, 1000), (1000, 2049), (2049, 4096)):
        part = eng.run_device(dp, sp[a:b])
        for k in ("aff_logits", "neg_logits"):
            assert torch.equal(part[k], full[k][:, a:b]), (k, a, b)
        for k in ("probs", "post", "decision", "qual"):
            assert torch.equal(part[k], full[k][a:b]), (k, a, b)
    assert (np.abs(raw_n) >= np.abs(raw_a)).all() and (info[:, 2] >= info[:, 1]).all() and int(info[:, 3].sum()) == 0
    assert min_bq == 0 and np.array_equal(raw_a, raw_n) and torch.equal(f.x_aff, f.x_neg)      # both presets: one pass
    depth = info[:, 1].astype(np.float64)
    if platform == "hifi":
        assert (depth > 50).mean() > 0.9                                # the rescale branch is the common case
    scale = np.where(depth > 50, 50.0 / np.maximum(depth, 1.0), 1.0)
    want_x = np.where((depth > 50)[:, None, None], raw_a.astype(np.float64) * scale[:, None, None], raw_a.astype(np.float64)).astype(np.float32)
    np.testing.assert_array_equal(f.x_aff.cpu().numpy(), want_x)
    again = eng.posterior.from_probs(torch.from_numpy(np.ascontiguousarray(p8)).to(dev))
    assert torch.equal(again["post"], full["post"]) and torch.equal(again["decision"], full["decision"])
    assert torch.equal(again["qual"], full["qual"])


def test_c_abi_error_codes(dev):
    """the C ABI reports errors by code + message, never by crashing"""
    import ctypes as C
    import torch
    from clairs_to_amd._lib import lib, CvtCfg, c_vp
    w = c_vp(lib.cto_weights_new())
    out = c_vp()
    assert lib.cto_bigru_create(w, 4, C.byref(out)) == -5 and b"missing" in lib.cto_last_error()       # CTO_EMISSING
    assert lib.cto_bigru_create(w, 5, C.byref(out)) == -1                                                # CTO_EINVAL
    cfg = CvtCfg()
    cfg.emb_dim[:] = [16, 64, 256]
    cfg.heads[:] = [1, 3, 4]
    cfg.depth[:] = [1, 2, 3]
    cfg.n_out = 4
    assert lib.cto_cvt_create(w, C.byref(cfg), C.byref(out)) == -4                                       # CTO_EUNSUPPORTED
    lib.cto_weights_free(w)
    assert lib.cto_posterior(None, None, 4, 1, None, None, None, None, None, None, None) == -1


def test_run_stream_matches_run_device(dev):
    """Host buffers in / host results out with uploads on a copy stream: same numbers as the resident path, chunk order kept,
    pageable and pinned inputs alike."""
    import torch
    from clairs_to_amd.engine import Engine, synthetic_models
    from clairs_to_amd.pack import pin_arrays
    from clairs_to_amd.synth import SynthChunk, likelihood_table, lik_and_edges
    models = synthetic_models(4)
    lik, edges = lik_and_edges(likelihood_table(4), 4)
    eng = Engine(models["aff"], models["neg"], lik, edges, min_bq=20, device=dev)
    chunks = [SynthChunk(n, seed=40 + i) for i, n in enumerate((300, 17, 512, 64, 129))]
    want = []
    for ch in chunks:
        r = eng.run_chunk(ch.arrays(), ch.site_pos)
        want.append({k: r[k].cpu().numpy() for k in ("probs", "post", "decision", "qual")})
    for pinned in (False, True):
        feed = ((pin_arrays(ch.arrays()) if pinned else ch.arrays(), ch.site_pos) for ch in chunks)
        got = list(eng.run_stream(feed, depth=2))
        assert len(got) == len(want)
        for g, w in zip(got, want):
            for k in w:
                np.testing.assert_array_equal(g[k], w[k])


@pytest.mark.parametrize("env", [{}, {"CTO_CVT_NO_EMBED_FUSE": "1"}, {"CTO_CVT_NO_HEAD_FUSE": "1"},
                                 {"CTO_CVT_NO_EMBED_FUSE": "1", "CTO_CVT_NO_HEAD_FUSE": "1"}, {"CTO_CVT_UNFUSED": "1"}])
@pytest.mark.parametrize("cls,K", [("CvT", 4), ("CvT_Indel", 6)])
def test_cvt_fusion_levels_on_poisoned_lds(dev, oracle_lib, monkeypatch, env, cls, K):
    """Every fusion level of the CvT (embedding in the first block, classifier in the last block, fused blocks only, fully
    unfused) against the oracle, each forward preceded by a launch that fills every CU's LDS with NaN patterns: padding that is
    only ever multiplied by zero weights must still be initialised."""
    import torch
    import oracle
    from clairs_to_amd import nn_shims
    from clairs_to_amd._lib import lib, check, current_stream_ptr
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    g = load_models_npz(cls)
    w = make_weights(g["manifest"], seed=3)
    m = nn_shims.from_state_dict(cls, w).to(dev)
    rng = np.random.default_rng(5)
    for B in (1, 16, 37, 300):
        x = (rng.integers(-60, 60, size=(B, 33, 34)) * rng.random((B, 1, 1))).astype(np.float32)
        check(lib.cto_debug_poison_lds(current_stream_ptr()))
        got = m.logits(torch.from_numpy(x).to(dev)).cpu().numpy()
        ref = oracle.cvt_forward(w, dict(CVT_CFG, n_out=K), x)
        assert np.isfinite(got).all() and np.abs(got - ref).max() < 1e-4, (env, B)


@pytest.mark.parametrize("cls,K", [("BiGRU_NACGT", 4), ("BiGRU_NACGT_Indel", 6)])
def test_bigru_on_poisoned_lds(dev, oracle_lib, cls, K):
    import torch
    import oracle
    from clairs_to_amd import nn_shims
    from clairs_to_amd._lib import lib, check, current_stream_ptr
    g = load_models_npz(cls)
    w = make_weights(g["manifest"], seed=3)
    m = nn_shims.from_state_dict(cls, w).to(dev)
    rng = np.random.default_rng(6)
    for B in (1, 33, 200):
        x = (rng.integers(-60, 60, size=(B, 33, 34)) * rng.random((B, 1, 1))).astype(np.float32)
        check(lib.cto_debug_poison_lds(current_stream_ptr()))
        got = m.logits(torch.from_numpy(x).to(dev)).cpu().numpy()
        ref = oracle.bigru_forward(w, K, x)
        assert np.isfinite(got).all() and np.abs(got - ref).max() < 1e-4, B


def test_featurize_matches_reference_edge_columns(dev):
    """Every golden column of tests/golden/columns.json.gz (the reference's decode_pileup_bases on hand-made and random
    columns: 59/60-base deletions, 60/61-base insertions, `*+` / `#+`, N reference, `^x`, `$`, MQ 19/20, BQ 29/30 ...)
    through tokeniser -> pack -> HIP kernels: the 34 channels, the depth and the alt_info string must be the reference's."""
    import torch
    from clairs_to_amd.pack import ColumnPack
    from clairs_to_amd.featurize import featurize, alt_infos
    cases = load_json_gz("columns.json.gz")
    checked = skipped = 0
    for i, c in enumerate(cases):
        n_tok = sum(1 for ch in _strip_indels(c["bases"]) if ch in "ACGTNacgtn*#")
        if not (n_tok == len(c["bq"]) == len(c["mq"])):
            skipped += 1                   # malformed rows: documented deviation
            continue
        text = "chr1\t1000\tN\t%d\t%s\t%s\t%s\n" % (n_tok, c["bases"], c["bq"], c["mq"])
        pack = ColumnPack.from_mpileup(text, c["chunk_ref"], 1000)
        # the fixture draws the reference base and the 60-base reference slice independently (decode_pileup_bases takes both):
        # the slice goes in as the reference sequence (deletion keys), the base is patched into the pack (channel negation)
        pack.numpy()["col_ref"][0] = "ACGT".index(c["ref"])
        feat = featurize(pack.to_device(dev), torch.tensor([1000], dtype=torch.int32, device=dev), 0, 0, want_raw=True, want_x=False)
        info = feat.site_info.cpu().numpy()
        got = feat.raw_neg.cpu().numpy()[0, 16].astype(int).tolist()
        assert got == c["tensor"], "column %d: %r" % (i, c["bases"])
        assert feat.raw_aff.cpu().numpy()[0, 16].astype(int).tolist() == c["tensor"]        # min_bq 0: both passes agree
        assert int(info[0, 2]) == int(c["alt_info"].split("-")[0]) if c["alt_info"] else True
        if c["cand"] and c["alt_info"]:
            assert alt_infos(feat, pack, info, pass_idx=1)[0] == c["alt_info"], "column %d alt_info" % i
        checked += 1
    assert checked >= 200 and skipped <= len(cases) // 5


def _strip_indels(bases):
    """mpileup base string without the `+n<seq>` / `-n<seq>` runs and without the character after `^`."""
    out, i = [], 0
    while i < len(bases):
        ch = bases[i]
        if ch in "+-":
            j = i + 1
            n = 0
            while j < len(bases) and bases[j].isdigit():
                n = n * 10 + int(bases[j])
                j += 1
            i = j + n
        elif ch == "^":
            i += 2
        else:
            out.append(ch)
            i += 1
    return "".join(out)


@pytest.mark.parametrize("B", [5000, 7500])
def test_batch_invariance_across_tile_mixes(dev, B):
    """Batches beyond one round of workgroups: the recurrent kernels run 32-site tiles for whole rounds and 16- or 32-site
    tiles for the remainder (csrc/gru.hip), the CvT runs 8/16-site tiles with a ragged last workgroup.  A site's logits must
    not depend on which launch or tile it lands in: bit-equal to the same sites run in other groupings."""
    import torch
    from clairs_to_amd.engine import synthetic_models
    models = synthetic_models(4)
    rng = np.random.default_rng(B)
    x = torch.from_numpy((rng.integers(-50, 50, size=(B, 33, 34)) * rng.random((B, 1, 1))).astype(np.float32)).to(dev)
    for key in ("aff", "neg"):
        m = models[key].to(dev)
        full = m.logits(x)
        cuts = [0, 17, 4096, 4113, B]
        parts = torch.cat([m.logits(x[a:b].contiguous()) for a, b in zip(cuts[:-1], cuts[1:])], dim=1)
        assert torch.equal(full, parts), key
        assert torch.isfinite(full).all()


@pytest.mark.parametrize("case", ["sparse", "dense_keys", "overlapping", "deep", "gaps"])
def test_one_kernel_featurisation_equals_the_two_stage_path(dev, case):
    """cto_featurize_sites (a workgroup per candidate, column histograms in LDS) against cto_featurize_columns + cto_gather_windows
    (every column's vector through HBM): tensors, rescaled inputs, strand counts, candidate column vectors, first-seen order and
    the key counts of the candidate columns, bit for bit.  `overlapping`: candidates 3 bases apart, so 30 of a window's 33 columns
    are shared with its neighbours; `dense_keys`: more distinct indel keys in a window than one LDS sweep holds (several sweeps);
    `deep`: 600x columns; `gaps`: sites at the edges of the pack and between columns (windows partly or wholly without columns)."""
    import torch
    from clairs_to_amd.pack import DevicePack
    from clairs_to_amd.featurize import featurize
    from clairs_to_amd.synth import SynthChunk
    kw = dict(sparse=dict(n=500, spacing=40, p_ins=0.01, p_del=0.02, depth_mean=60.0),
              dense_keys=dict(n=80, spacing=40, p_ins=0.3, p_del=0.3, depth_mean=90.0),
              overlapping=dict(n=400, spacing=40, p_ins=0.03, p_del=0.03, depth_mean=50.0),
              deep=dict(n=40, spacing=40, p_ins=0.02, p_del=0.02, depth_mean=600.0),
              gaps=dict(n=200, spacing=40, p_ins=0.02, p_del=0.02, depth_mean=40.0))[case]
    chunk = SynthChunk(kw["n"], seed=77, spacing=kw["spacing"], p_ins=kw["p_ins"], p_del=kw["p_del"], depth_mean=kw["depth_mean"])
    dp = DevicePack(chunk.arrays(), dev)
    sites = chunk.site_pos.astype(np.int64)
    if case == "overlapping":
        sites = np.unique(np.concatenate([sites + d for d in range(-15, 16, 3)]))
    elif case == "gaps":
        sites = np.unique(np.concatenate([sites - 30, sites + 17, sites + 20, [1, 5, 17, int(chunk.col_pos[-1]) + 1, int(chunk.col_pos[-1]) + 40]]))
        sites = sites[sites > 0]
    if case == "dense_keys":
        ko = chunk.key_off
        assert (ko[33:] - ko[:-33]).max() > 256, "a window with more keys than one sweep"
    sp = torch.from_numpy(sites.astype(np.int32)).to(dev)
    for min_bq, rescale in ((20, 50), (0, 0)):
        two = featurize(dp, sp, min_bq, rescale, want_raw=True, fused=False)
        one = featurize(dp, sp, min_bq, rescale, want_raw=True, fused=True)
        torch.cuda.synchronize()
        for name in ("x_aff", "x_neg", "raw_aff", "raw_neg", "site_info", "sitefirst"):
            assert torch.equal(getattr(one, name), getattr(two, name)), name
        info = two.site_info.cpu().numpy()
        has = info[:, 0] >= 0
        centre = torch.from_numpy(np.where(has, info[:, 0], 0)).long().to(dev)
        want_cv = two.colvec.index_select(0, centre) * torch.from_numpy(has).to(dev)[:, None]
        assert torch.equal(one.site_colvec, want_cv.to(torch.int16))
        ko = chunk.key_off.astype(np.int64)
        keys = np.concatenate([np.arange(ko[c], ko[c + 1]) for c in info[has, 0]] + [np.zeros(0, dtype=np.int64)]).astype(np.int64)
        if keys.size:
            kt = torch.from_numpy(keys).to(dev)
            assert torch.equal(one.keycnt.index_select(0, kt), two.keycnt.index_select(0, kt))
            assert torch.equal(one.keyfirst.index_select(0, kt), two.keyfirst.index_select(0, kt))

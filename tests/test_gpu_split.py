"""The split-operand experiment (csrc/gru_split_kernel.h; CTO_GRU_SPLIT=f16|bf16 at model creation): both BiGRU recurrences and the
fused fc1 with every operand written as hi + lo (16-bit floats) and three f16 / bf16 MFMA passes per product.  It is a side channel - the default path
and the bench's `value` stay on the fp32 kernels - but it is held to the same oracle and the same 1e-4 tolerance as they are
(clairs/model.py:412-417, 442-448 is what all of them restate)."""
import os

import numpy as np
import pytest

from weights_recipe import CVT_CFG

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return torch.device("cuda:0")


def _engine(K, dev, split, min_bq=20, cvt_split=None, cvt_cfg=None):
    """an engine over fresh module objects (a module creates its C-ABI handle once; the switches are read then)"""
    from clairs_to_amd.engine import Engine, synthetic_models
    from clairs_to_amd.synth import likelihood_table, lik_and_edges
    models = synthetic_models(K, cvt_cfg=cvt_cfg)
    lik, edges = lik_and_edges(likelihood_table(K), K)
    if split:
        os.environ["CTO_GRU_SPLIT"] = split
    if cvt_split:
        os.environ["CTO_CVT_SPLIT"] = cvt_split
    try:
        eng = Engine(models["aff"], models["neg"], lik, edges, min_bq=min_bq, device=dev)
    finally:
        os.environ.pop("CTO_GRU_SPLIT", None)
        os.environ.pop("CTO_CVT_SPLIT", None)
    return eng, models, lik, edges


def _aff_logits(eng, x_aff, B, K, dev):
    import torch
    from clairs_to_amd._lib import lib, check
    out = torch.empty((K, B, 2), device=dev)
    check(lib.cto_model_forward(eng.h_aff, x_aff.data_ptr(), B, out.data_ptr(), int(torch.cuda.current_stream().cuda_stream)))
    torch.cuda.synchronize()
    return out.cpu().numpy()


def _neg_logits(eng, x_neg, B, K, dev):
    import torch
    from clairs_to_amd._lib import lib, check
    out = torch.empty((K, B, 2), device=dev)
    check(lib.cto_model_forward(eng.h_neg, x_neg.data_ptr(), B, out.data_ptr(), int(torch.cuda.current_stream().cuda_stream)))
    torch.cuda.synchronize()
    return out.cpu().numpy()


@pytest.mark.parametrize("kind,tol_logit", [("f16", 5e-5), ("bf16", 2e-4)])
@pytest.mark.parametrize("K", [4, 6])
def test_split_kernel_against_the_oracle_and_the_f32_kernel(dev, oracle_lib, kind, tol_logit, K):
    """(a) 160 sites against the oracle: NEG logits within tol_logit (the fp32 kernel sits at ~4e-6, f16 at ~6e-6, bf16 at ~2e-5),
    probabilities within the path's 1e-4, every decision equal; (b) a full 4096-site chunk and ragged batches (1, 15, 16, 17, 33,
    1000: whole and partial 32- and 16-site tiles, both launch ranges) against the fp32 kernel on the same inputs."""
    import torch
    import oracle
    from clairs_to_amd.featurize import featurize
    from clairs_to_amd.synth import SynthChunk, mpileup_text
    e32, models, lik, edges = _engine(K, dev, None)
    esp, _, _, _ = _engine(K, dev, kind)
    small = SynthChunk(160, seed=5)
    ref, lo = small.ref_window()
    ta, da, _, _ = oracle.create_tensor(mpileup_text(small, 20), ref, lo, small.site_pos)
    tn, dn, _, _ = oracle.create_tensor(mpileup_text(small, 0), ref, lo, small.site_pos)
    la = oracle.cvt_forward(models["aff_weights"], dict(CVT_CFG, n_out=K), oracle.rescale(ta, da))
    ln = oracle.bigru_forward(models["neg_weights"], K, oracle.rescale(tn, dn))
    probs, post, dec, qual = oracle.posterior(la, ln, lik, edges)
    got = esp.run_chunk(small.arrays(), small.site_pos)
    torch.cuda.synchronize()
    assert float(np.abs(got["probs"].cpu().numpy() - probs).max()) < 1e-4
    np.testing.assert_array_equal(got["decision"].cpu().numpy()[:, 0], np.asarray(dec)[:, 0])        # arg-max: 0..K-1, unmasked
    np.testing.assert_array_equal(got["decision"].cpu().numpy()[:, 1] & 3, np.asarray(dec)[:, 1] & 3)
    dp = esp.upload(small.arrays())
    feat = featurize(dp, torch.from_numpy(small.site_pos).to(dev), 20, 50)
    lg = _neg_logits(esp, feat.x_neg, 160, K, dev)
    assert np.isfinite(lg).all()
    assert float(np.abs(lg - np.asarray(ln).reshape(lg.shape)).max()) < tol_logit
    # ---- (b) full chunk and ragged batches against the fp32 kernel ----
    big = SynthChunk(4096, seed=2)
    dpb = e32.upload(big.arrays())
    fb = featurize(dpb, torch.from_numpy(big.site_pos).to(dev), 20, 50)
    full32 = _neg_logits(e32, fb.x_neg, 4096, K, dev)
    fullsp = _neg_logits(esp, fb.x_neg, 4096, K, dev)
    assert np.isfinite(fullsp).all()
    assert float(np.abs(fullsp - full32).max()) < tol_logit
    for B in (1, 15, 16, 17, 33, 1000):
        part = _neg_logits(esp, fb.x_neg, B, K, dev)
        np.testing.assert_array_equal(part, fullsp[:, :B])          # a site's result does not depend on the batch around it
    # the fp32 engine created in the same process is untouched by the switch
    np.testing.assert_array_equal(_neg_logits(e32, fb.x_neg, 4096, K, dev), full32)


@pytest.mark.parametrize("kind,tol_logit", [("f16", 5e-5), ("bf16", 5e-4)])
@pytest.mark.parametrize("K,default_cfg", [(4, False), (6, False), (4, True)])
def test_split_cvt_blocks_against_the_oracle_and_the_f32_kernels(dev, oracle_lib, kind, tol_logit, K, default_cfg):
    """CTO_CVT_SPLIT: the block GEMMs of the 64- and 128-channel stages on split operands (the predict.py configuration and the
    constructor-default one, whose ten stage-3 blocks run as three launches): AFF logits against the oracle on 160 sites and
    against the fp32 kernels on a full chunk and ragged batches; probabilities within the path's 1e-4, decisions equal.
    The 13-block configuration gets four times the logit tolerance (measured: 1e-4 for f16 over 4096 sites - operands below
    ~0.1 have a subnormal f16 lo half, i.e. an absolute 3e-8 instead of a relative 2^-22)."""
    if default_cfg:
        tol_logit *= 4
    import torch
    import oracle
    from clairs_to_amd.engine import CVT_CONSTRUCTOR_CFG
    from clairs_to_amd.featurize import featurize
    from clairs_to_amd.synth import SynthChunk, mpileup_text
    cfg = CVT_CONSTRUCTOR_CFG if default_cfg else None
    ocfg = dict(emb_dim=(32, 64, 128), heads=(1, 3, 6), depth=(1, 2, 10), n_out=K) if default_cfg else dict(CVT_CFG, n_out=K)
    e32, models, lik, edges = _engine(K, dev, None, cvt_cfg=cfg)
    esp, _, _, _ = _engine(K, dev, kind, cvt_split=kind, cvt_cfg=cfg)
    small = SynthChunk(160, seed=5)
    ref, lo = small.ref_window()
    ta, da, _, _ = oracle.create_tensor(mpileup_text(small, 20), ref, lo, small.site_pos)
    tn, dn, _, _ = oracle.create_tensor(mpileup_text(small, 0), ref, lo, small.site_pos)
    la = oracle.cvt_forward(models["aff_weights"], ocfg, oracle.rescale(ta, da))
    ln = oracle.bigru_forward(models["neg_weights"], K, oracle.rescale(tn, dn))
    probs, post, dec, qual = oracle.posterior(la, ln, lik, edges)
    got = esp.run_chunk(small.arrays(), small.site_pos)
    torch.cuda.synchronize()
    assert float(np.abs(got["probs"].cpu().numpy() - probs).max()) < 1e-4
    np.testing.assert_array_equal(got["decision"].cpu().numpy()[:, 0], np.asarray(dec)[:, 0])        # arg-max: 0..K-1, unmasked
    np.testing.assert_array_equal(got["decision"].cpu().numpy()[:, 1] & 3, np.asarray(dec)[:, 1] & 3)
    dp = esp.upload(small.arrays())
    feat = featurize(dp, torch.from_numpy(small.site_pos).to(dev), 20, 50)
    lg = _aff_logits(esp, feat.x_aff, 160, K, dev)
    assert np.isfinite(lg).all()
    assert float(np.abs(lg - np.asarray(la).reshape(lg.shape)).max()) < tol_logit
    big = SynthChunk(4096, seed=2)
    dpb = e32.upload(big.arrays())
    fb = featurize(dpb, torch.from_numpy(big.site_pos).to(dev), 20, 50)
    full32 = _aff_logits(e32, fb.x_aff, 4096, K, dev)
    fullsp = _aff_logits(esp, fb.x_aff, 4096, K, dev)
    assert np.isfinite(fullsp).all()
    assert float(np.abs(fullsp - full32).max()) < tol_logit
    for B in (1, 15, 17, 1000):
        np.testing.assert_array_equal(_aff_logits(esp, fb.x_aff, B, K, dev), fullsp[:, :B])
    np.testing.assert_array_equal(_aff_logits(e32, fb.x_aff, 4096, K, dev), full32)


def test_split_switch_rejects_unknown_kinds(dev):
    from clairs_to_amd._lib import CtoError
    with pytest.raises(CtoError):
        _engine(4, dev, "fp8")
    with pytest.raises(CtoError):
        _engine(4, dev, None, cvt_split="fp8")


def test_native_pipeline_with_split_operands_calls_the_same_variants(tmp_path):
    """The experimental switch through the product pipeline (module.split_operands -> cto_run_chunks): same records as the fp32
    run - position, alleles, genotype, filter - and QUAL within 0.01 (a probability moves by ~1e-5, QUAL = -10 log10(1 - p))."""
    from clairs_to_amd.call_chunks import run_pipeline_native
    from clairs_to_amd.e2e import chunk_namespaces
    from clairs_to_amd.engine import Engine, synthetic_models
    from clairs_to_amd.synth import likelihood_table, lik_and_edges
    from clairs_to_amd.synth_run import make_text_run
    run = make_text_run(str(tmp_path / "run"), n_chunks=2, sites_per_chunk=2048, distinct=2)
    lik, edges = lik_and_edges(likelihood_table(4), 4)
    rows = {}
    for name, kind in (("f32", None), ("f16", "f16")):
        models = synthetic_models(4, seed=0)
        for m in (models["aff"], models["neg"]):
            m.split_operands = kind
        eng = Engine(models["aff"], models["neg"], lik, edges, min_bq=20, device="cuda:0")
        os.makedirs(tmp_path / name)
        n = run_pipeline_native(eng, chunk_namespaces(run, str(tmp_path / name)), producers=2, writers=2, verbose=False)
        assert n > 500
        rows[name] = []
        for fn in sorted(os.listdir(tmp_path / name)):
            rows[name] += [ln.rstrip("\n").split("\t") for ln in open(tmp_path / name / fn) if not ln.startswith("#")]
    assert len(rows["f32"]) == len(rows["f16"])
    worst = 0.0
    for a, b in zip(rows["f32"], rows["f16"]):
        assert a[:5] == b[:5] and a[6] == b[6], (a, b)                       # CHROM POS ID REF ALT, FILTER
        assert a[9].split(":")[0] == b[9].split(":")[0], (a, b)              # GT
        worst = max(worst, abs(float(a[5]) - float(b[5])))
    assert worst < 0.01, worst


@pytest.mark.parametrize("default_cfg", [False, True])
def test_split_error_report_against_the_oracle(dev, oracle_lib, default_cfg):
    """Where each arithmetic stands against the oracle on the same 160 sites (printed with -s; profiles/round4_split_mfma_oracle.txt):
    max |d logit| of both networks and max |dP|, for the fp32 kernels, f16 halves and bf16 halves.  Asserted: every kind inside
    the path's 1e-4 on probabilities, and f16 no further from the oracle than twice the fp32 kernels are (plus 1e-5)."""
    import torch
    import oracle
    from clairs_to_amd.engine import CVT_CONSTRUCTOR_CFG
    from clairs_to_amd.featurize import featurize
    from clairs_to_amd.synth import SynthChunk, mpileup_text
    K = 4
    cfg = CVT_CONSTRUCTOR_CFG if default_cfg else None
    ocfg = dict(emb_dim=(32, 64, 128), heads=(1, 3, 6), depth=(1, 2, 10), n_out=K) if default_cfg else dict(CVT_CFG, n_out=K)
    small = SynthChunk(160, seed=5)
    ref, lo = small.ref_window()
    ta, da, _, _ = oracle.create_tensor(mpileup_text(small, 20), ref, lo, small.site_pos)
    tn, dn, _, _ = oracle.create_tensor(mpileup_text(small, 0), ref, lo, small.site_pos)
    rep = {}
    for kind in (None, "f16", "bf16"):
        eng, models, lik, edges = _engine(K, dev, kind, cvt_split=kind, cvt_cfg=cfg)
        if kind is None:
            la = oracle.cvt_forward(models["aff_weights"], ocfg, oracle.rescale(ta, da))
            ln = oracle.bigru_forward(models["neg_weights"], K, oracle.rescale(tn, dn))
            probs, post, dec, qual = oracle.posterior(la, ln, lik, edges)
        got = eng.run_chunk(small.arrays(), small.site_pos)
        torch.cuda.synchronize()
        dp = eng.upload(small.arrays())
        feat = featurize(dp, torch.from_numpy(small.site_pos).to(dev), 20, 50)
        ga, gn = _aff_logits(eng, feat.x_aff, 160, K, dev), _neg_logits(eng, feat.x_neg, 160, K, dev)
        rep[kind or "f32"] = {"aff_dlogit": float(np.abs(ga - np.asarray(la).reshape(ga.shape)).max()),
                              "neg_dlogit": float(np.abs(gn - np.asarray(ln).reshape(gn.shape)).max()),
                              "dP": float(np.abs(got["probs"].cpu().numpy() - probs).max())}
    print("\nvs oracle, 160 sites, %s CvT:" % ("constructor-default (13 blocks)" if default_cfg else "predict.py (6 blocks)"))
    for k, v in rep.items():
        print("  %-5s max|d logit| AFF %.3g  NEG %.3g   max|dP| %.3g" % (k, v["aff_dlogit"], v["neg_dlogit"], v["dP"]))
    for k, v in rep.items():
        assert v["dP"] < 1e-4, (k, v)
    for net in ("aff_dlogit", "neg_dlogit"):
        assert rep["f16"][net] <= 2 * rep["f32"][net] + 1e-5, rep


@pytest.mark.parametrize("platform", ["ont", "ilmn", "hifi"])
@pytest.mark.parametrize("K", [4, 6])
def test_split_f16_at_full_size_on_every_config_leg(dev, oracle_lib, platform, K):
    """The f16 side channel (the validated one: power-of-two operand scales, tests/test_gpu_range.py) on every single-GPU workload the
    bench line carries as a config leg - ONT 50x, Illumina 50x, HiFi 75x with the SNV (K = 4) and indel (K = 6) model pairs - at the
    full 4096-site chunk: probabilities within 1e-4 of the fp32 kernels' on all sites and of the oracle's on a 256-site sample, every
    decision and every 4-decimal QUAL that is not on a rounding boundary equal, batch invariance."""
    import torch
    import oracle
    from clairs_to_amd.engine import Engine, synthetic_models
    from clairs_to_amd.synth import SynthChunk, PLATFORMS, likelihood_table, lik_and_edges
    min_bq = PLATFORMS[platform]["min_bq"]
    chunk = SynthChunk.for_platform(platform, 4096)
    lik, edges = lik_and_edges(likelihood_table(K), K)
    engs = {}
    for kind in ("f32", "f16"):
        models = synthetic_models(K)
        models["aff"].split_operands = kind
        models["neg"].split_operands = kind
        engs[kind] = Engine(models["aff"], models["neg"], lik, edges, min_bq=min_bq, device=dev, neg_reads_aff=(platform == "ilmn"))
    dp = engs["f32"].upload(chunk.arrays())
    sp = torch.from_numpy(chunk.site_pos).to(dev)
    r32 = engs["f32"].run_device(dp, sp)
    r16 = engs["f16"].run_device(dp, sp)
    torch.cuda.synchronize()
    p32, p16 = r32["probs"].cpu().numpy(), r16["probs"].cpu().numpy()
    assert np.isfinite(p16).all()
    d = float(np.abs(p16 - p32).max())
    print("split f16 %s K=%d: max |dP| vs fp32 kernels %.2e over 4096 sites" % (platform, K, d))
    assert d < 1e-4
    d32, d16 = r32["decision"].cpu().numpy(), r16["decision"].cpu().numpy()
    same = (d32[:, 0] == d16[:, 0]) & ((d32[:, 1] & 3) == (d16[:, 1] & 3))
    # a decision may only differ where the two winning posteriors are within the probabilities' own distance of each other
    post = np.sort(r32["post"].cpu().numpy(), axis=1)
    assert (same | (post[:, -1] - post[:, -2] < 1e-3)).all() and same.mean() > 0.999
    # oracle on a sample
    n_s = 256
    ref, lo = chunk.ref_window()
    sites = chunk.site_pos[:n_s]
    c1 = int(np.searchsorted(chunk.col_pos, int(sites[-1]) + 17, side="right"))
    ta, da, _, _ = oracle.create_tensor(oracle.synth_mpileup_text(chunk, min_bq, (0, c1)), ref, lo, sites)
    tn, dn, _, _ = oracle.create_tensor(oracle.synth_mpileup_text(chunk, 0, (0, c1)), ref, lo, sites)
    models = synthetic_models(K)
    la = oracle.cvt_forward(models["aff_weights"], dict(CVT_CFG, n_out=K), oracle.rescale(ta, da))
    ln = oracle.bigru_forward(models["neg_weights"], K, oracle.rescale(ta if platform == "ilmn" else tn, da if platform == "ilmn" else dn))
    probs, _, _, _ = oracle.posterior(la, ln, lik, edges)
    assert float(np.abs(p16[:n_s] - probs).max()) < 1e-4
    for a, b in ((0, 17), (17, 1000), (1000, 4096)):
        part = engs["f16"].run_device(dp, sp[a:b])
        assert torch.equal(part["probs"], r16["probs"][a:b])

"""GPU tests against the wider reference-generated fixtures: the 2 000-candidate region through the drop-in sub-modules and the
one-invocation driver, the call_variants branch rows through the device epilogue, and genuine clairs.model pickles through the
HIP networks."""
import gzip
import hashlib
import os
import zlib

import numpy as np
import pytest

from conftest import load_json_gz, load_models_npz, load_genuine_pickle
from weights_recipe import make_weights

pytestmark = pytest.mark.gpu


def _write_inputs(tmp_path, r):
    ref, lo = r["ref"], r["ref_lo"]
    full = "A" * (lo - 1) + ref
    fa = tmp_path / "ref.fa"
    fa.write_text(">chr1\n" + "\n".join(full[i:i + 60] for i in range(0, len(full), 60)) + "\n")
    (tmp_path / "ref.fa.fai").write_text("chr1\t%d\t6\t60\t61\n" % len(full))
    bed = tmp_path / "cand.bed"
    bed.write_text("".join("chr1\t%d\t%d\n" % (x - 17, x + 17) for x in r["sites"]))
    mp = tmp_path / "mp.txt"
    mp.write_text(r["texts"]["neg"])
    return str(fa), str(bed), str(mp)


def _genuine_checkpoints(tmp_path, aff_cls, neg_cls, K):
    """the reference's own pickles (parameter values zeroed in the fixture) refilled with the weights_recipe values and
    saved again - the file a user of the reference would hand over"""
    import torch
    paths = {}
    for key, cls in (("model_acgt", aff_cls), ("model_nacgt", neg_cls)):
        m, _ = load_genuine_pickle(cls)
        w = make_weights(load_models_npz(cls)["manifest"], seed=K)
        sd = m.state_dict()
        for k, v in w.items():
            sd[k] = torch.from_numpy(v.copy()).reshape(sd[k].shape)
        m.load_state_dict(sd)
        from clairs_to_amd import nn_shims
        saved = {}
        try:      # pickle under the reference's qualified names, as the fixture's stream does
            for c in vars(nn_shims).values():
                if isinstance(c, type) and c.__module__ == nn_shims.__name__:
                    saved[c] = c.__module__
                    c.__module__ = "clairs.model"
            paths[key] = str(tmp_path / (key + ".pkl"))
            torch.save({key: m}, paths[key])
        finally:
            for c, mod in saved.items():
                c.__module__ = mod
    return paths


def test_region2k_create_tensor_text_is_byte_identical(tmp_path, region2k):
    from argparse import Namespace
    from clairs_to_amd.create_tensor_pileup_calling import create_tensor
    g = region2k["g"]
    fa, bed, mp = _write_inputs(tmp_path, region2k)
    aff, neg = str(tmp_path / "aff.gz"), str(tmp_path / "neg.gz")
    create_tensor(Namespace(candidates_bed_regions=bed, ctg_name="chr1", ref_fn=fa, mpileup_fn=mp, samtools="samtools",
                            tumor_bam_fn=None, max_depth=None, max_indel_length=None, min_bq=g["min_bq_aff"], tensor_can_fn=aff,
                            tensor_can_fn_neg=neg, platform="ont"))
    for fn, tag in ((aff, "aff"), (neg, "neg")):
        text = gzip.open(fn, "rt").read()
        crc = [zlib.crc32(x.encode()) & 0xffffffff for x in text.split("\n") if x]
        want = g["tensor"][tag]
        assert len(crc) == len(want["row_crc"]) == 2000
        bad = [i for i, (a, b) in enumerate(zip(crc, want["row_crc"])) if a != b]
        assert not bad, "%s rows differ from the reference's tensor text: %s" % (tag, bad[:10])
        assert hashlib.sha256(text.encode()).hexdigest() == want["sha"]


@pytest.mark.parametrize("mode,aff_cls,neg_cls", [("snv", "CvT", "BiGRU_NACGT"), ("indel", "CvT_Indel", "BiGRU_NACGT_Indel")])
def test_region2k_pileup_call_matches_reference(tmp_path, region2k, mode, aff_cls, neg_cls):
    """BED + mpileup text + genuine-pickle checkpoints -> VCF in one invocation, against what the reference wrote through its
    four commands on the same 2 000 candidates: probabilities within 1e-4 (north_star), every VCF field identical except
    QUAL / GQ, which may move with the probabilities' last digits."""
    from argparse import Namespace
    from clairs_to_amd.pileup_call import pileup_call
    g = region2k["g"]
    c = g["calls"][mode]
    K = c["n_out"]
    fa, bed, mp = _write_inputs(tmp_path, region2k)
    paths = _genuine_checkpoints(tmp_path, aff_cls, neg_cls, K)
    lik = str(tmp_path / "lik.txt")
    open(lik, "w").write(c["likelihood_table"])
    vcf, pred = str(tmp_path / "one.vcf"), str(tmp_path / "pred.gz")
    n = pileup_call(Namespace(platform="ont", tumor_bam_fn=None, mpileup_fn=mp, ref_fn=fa, ctg_name="chr1", samtools="samtools",
                              min_bq=g["min_bq_aff"], max_depth=None, max_indel_length=None, candidates_bed_regions=bed,
                              chkpnt_fn_acgt=paths["model_acgt"], chkpnt_fn_nacgt=paths["model_nacgt"], min_rescale_cov=50,
                              disable_indel_calling=(mode == "snv"), likelihood_matrix_data=lik, call_fn=vcf, predict_fn=pred,
                              sample_name="SAMPLE", show_ref=True, qual=0, pileup=True))
    rows = [r.split("\t") for r in gzip.open(pred, "rt").read().split("\n") if r]
    assert [int(r[1]) for r in rows] == c["pos"] and [r[2] for r in rows] == c["ref"]
    assert [[r[4], r[5]] for r in rows] == c["strand"]
    alt_by_pos = dict(zip(g["tensor"]["aff"]["pos"], g["tensor"]["aff"]["alt_info"]))
    assert [r[3] for r in rows] == [alt_by_pos[p] for p in c["pos"]]
    p1 = np.array([[float(f.split()[1]) for f in r[6:6 + 2 * K]] for r in rows])
    want = np.array([[float(v) for v in row] for row in c["p1"]])
    assert np.abs(p1 - want).max() < 1e-4                 # north_star tolerance, 1 989 sites x 2K heads
    got = [r for r in open(vcf).read().split("\n") if r and not r.startswith("#")]
    assert n == len(got)
    # a site whose two best posteriors are within the probability tolerance may flip its arg-max; none may differ otherwise
    want_by_pos = {r.split("\t")[1]: r.split("\t") for r in c["vcf_show_ref"]}
    n_same = n_flip = 0
    for row in got:
        a = row.split("\t")
        b = want_by_pos.get(a[1])
        if b is not None and a[:5] == b[:5] and a[6:9] == b[6:9]:
            fa_, fb_ = a[9].split(":"), b[9].split(":")
            assert fa_[0] == fb_[0] and fa_[2:] == fb_[2:]
            assert abs(float(a[5]) - float(b[5])) < 0.05 and abs(int(fa_[1]) - int(fb_[1])) <= 1
            n_same += 1
        else:
            n_flip += 1
    assert n_same >= len(c["vcf_show_ref"]) - 2 and n_flip <= 2, (n_same, n_flip, len(c["vcf_show_ref"]))


@pytest.mark.parametrize("mode", ["snv", "indel"])
def test_region2k_call_variants_on_reference_probabilities(tmp_path, region2k, mode):
    """the reference's own 8-decimal probability rows through the device epilogue + host rows: byte-identical records"""
    from argparse import Namespace
    from clairs_to_amd.call_variants import call_variants_from_probability
    g = region2k["g"]
    c = g["calls"][mode]
    K = c["n_out"]
    alt_by_pos = dict(zip(g["tensor"]["aff"]["pos"], g["tensor"]["aff"]["alt_info"]))
    pred = str(tmp_path / "ref_pred.gz")
    with gzip.open(pred, "wt") as f:
        for i, pos in enumerate(c["pos"]):
            fields = ["chr1", str(pos), c["ref"][i], alt_by_pos[pos], c["strand"][i][0], c["strand"][i][1]]
            fields += ["%0.8f %s" % (1.0 - float(p), p) for p in c["p1"][i]]
            f.write("\t".join(fields) + ("\t\n" if K == 4 else "\n"))
    lik = str(tmp_path / "lik.txt")
    open(lik, "w").write(c["likelihood_table"])
    vcf = str(tmp_path / "out.vcf")
    call_variants_from_probability(Namespace(call_fn=vcf, predict_fn=pred, likelihood_matrix_data=lik, ctg_name="chr1",
                                             sample_name="SAMPLE", qual=0, show_ref=True, disable_indel_calling=(mode == "snv"),
                                             pileup=True, platform="ont"))
    rows = [r for r in open(vcf).read().split("\n") if r and not r.startswith("#")]
    assert rows == c["vcf_show_ref"]


@pytest.mark.parametrize("mode", ["snv", "indel"])
def test_call_variants_branches_on_device(tmp_path, mode):
    """every ALT / AF / GT / FILTER branch (tests/golden calls_branches) through the CLI mirror: device posterior + host rows"""
    from argparse import Namespace
    from clairs_to_amd.call_variants import call_variants_from_probability
    g = load_json_gz("calls_branches.json.gz")[mode]
    pred = str(tmp_path / "pred.gz")
    with gzip.open(pred, "wt") as f:
        f.write(g["predict_rows"])
    lik = str(tmp_path / "lik.txt")
    open(lik, "w").write(g["likelihood_table"])
    for tag, run in g["runs"].items():
        vcf = str(tmp_path / (tag + ".vcf"))
        call_variants_from_probability(Namespace(call_fn=vcf, predict_fn=pred, likelihood_matrix_data=lik, ctg_name="chr1",
                                                 sample_name="SAMPLE", qual=int(tag.split("_")[0][4:]), show_ref=tag.endswith("1"),
                                                 disable_indel_calling=(mode == "snv"), pileup=True, platform="ont"))
        rows = [r for r in open(vcf).read().split("\n") if r and not r.startswith("#")] if os.path.exists(vcf) else []
        assert rows == run["rows"], tag


@pytest.mark.parametrize("cls", ["CvT", "CvT_Indel", "BiGRU_NACGT", "BiGRU_NACGT_Indel"])
def test_genuine_reference_pickles_run_on_the_hip_networks(cls):
    """a pickle written by the reference's own classes -> torch.load -> shim -> HIP forward = the reference's logits"""
    import torch
    m, _ = load_genuine_pickle(cls)
    g = load_models_npz(cls)
    sd = m.state_dict()
    for k, v in make_weights(g["manifest"], seed=g["n_out"]).items():
        sd[k] = torch.from_numpy(v.copy()).reshape(sd[k].shape)
    m.load_state_dict(sd)
    outs = m.eval()(torch.from_numpy(g["x"]).cuda())
    assert isinstance(outs, tuple) and len(outs) == g["n_out"]
    got = np.stack([o.cpu().numpy() for o in outs])
    np.testing.assert_allclose(got, g["logits"], rtol=0, atol=1e-4)


def test_genuine_default_config_pickle_runs():
    """constructor-default CvT (32/64/128, heads 1/3/6, depth 1/2/10) as pickled by the reference: geometry derived from the
    tensors, forward runs and is batch-invariant"""
    import torch
    m, g = load_genuine_pickle("CvT:defaults")
    sd = m.state_dict()
    manifest = [(k, tuple(v.shape)) for k, v in sd.items() if not k.endswith("num_batches_tracked")]
    for k, v in make_weights(manifest, seed=4).items():
        sd[k] = torch.from_numpy(v.copy()).reshape(sd[k].shape)
    m.load_state_dict(sd)
    x = torch.from_numpy(load_models_npz("CvT")["x"]).cuda()
    a = torch.stack(m.eval()(x))
    b = torch.stack(m(x[:7]))
    assert torch.isfinite(a).all() and torch.equal(a[:, :7], b)

"""SURVEY.md 8f #4 - the long-read post-calling filters: clairs_to_amd.haplotype_filtering (C evaluation of the read-level
rules, csrc/hapfilter.cpp) against the output VCF the REFERENCE's src/haplotype_filtering.py wrote on the same inputs
(tests/golden/hapfilter.json.gz: pileup VCF + phased germline VCF + nine-column mpileup text of a simulated haplotagged BAM,
SNV pass and indel pass).  Host code: no GPU needed."""
import os
from argparse import Namespace

import pytest

from conftest import load_json_gz


def _run(tmp_path, g, mode, **over):
    from clairs_to_amd.haplotype_filtering import haplotype_filter
    ref = g["ref"]
    (tmp_path / "ref.fa").write_text(">chr1\n" + ref + "\n")
    (tmp_path / "ref.fa.fai").write_text("chr1\t%d\t6\t%d\t%d\n" % (len(ref), len(ref), len(ref) + 1))
    (tmp_path / "germline.vcf").write_text(g["germline_vcf"])
    m = g["modes"][mode]
    (tmp_path / "pileup.vcf").write_text(m["pileup_vcf"])
    (tmp_path / "mp.txt").write_text(m["mpileup"])
    out = tmp_path / ("out_%s.vcf" % mode)
    a = Namespace(tumor_bam_fn="unused.bam", ref_fn=str(tmp_path / "ref.fa"), ctg_name="chr1", pileup_vcf_fn=str(tmp_path / "pileup.vcf"),
                  output_vcf_fn=str(out), germline_vcf_fn=str(tmp_path / "germline.vcf"), output_dir=str(tmp_path / "work"), threads=4,
                  input_filter_tag=None, show_ref=False, samtools="samtools", mpileup_fn=str(tmp_path / "mp.txt"),
                  apply_haplotype_filtering=True, min_mq=20, min_bq=0, min_alt_coverage=2, is_indel=(mode == "indel"), test_pos=None,
                  flanking=100, haplotype_chunk_max_sites=7, haplotype_chunk_max_span=5000000, disable_read_start_end_filtering=False)
    for k, v in over.items():
        setattr(a, k, v)
    res = haplotype_filter(a)
    return out.read_text(), res


@pytest.mark.parametrize("mode", ["snv", "indel"])
def test_filtered_vcf_is_byte_identical_to_the_reference(tmp_path, mode):
    g = load_json_gz("hapfilter.json.gz")
    got, res = _run(tmp_path, g, mode)
    want = g["modes"][mode]["out_vcf"]
    gl, wl = got.split("\n"), want.split("\n")
    bad = [(a, b) for a, b in zip(gl, wl) if a != b]
    assert not bad, "first differing record:\n%s\n%s" % bad[0]
    assert got == want
    assert len(res) >= (20 if mode == "snv" else 5)


def test_every_filter_is_exercised_by_the_fixture():
    g = load_json_gz("hapfilter.json.gz")
    tags = set()
    for m in g["modes"].values():
        for r in m["out_vcf"].split("\n"):
            if r and not r.startswith("#"):
                c = r.split("\t")
                tags.update(c[6].split(";"))
                tags.update(x.split("=")[0] for x in c[7].split(";"))
    assert {"PASS", "LowQual", "LowAltBQ", "LowAltMQ", "ReadStartEnd", "VariantCluster", "NoAncestry", "MultiHap", "StrandBias",
            "LowSeqEntropy", "H", "SB"} <= tags


@pytest.mark.parametrize("mode", ["snv", "indel"])
def test_result_does_not_depend_on_the_job_cut(tmp_path, mode):
    """every call sees only its own +-flanking window: one job, many small jobs and threads give the same file"""
    g = load_json_gz("hapfilter.json.gz")
    a, _ = _run(tmp_path, g, mode, haplotype_chunk_max_sites=200)
    b, _ = _run(tmp_path, g, mode, haplotype_chunk_max_sites=1, threads=8)
    c, _ = _run(tmp_path, g, mode, haplotype_chunk_max_sites=3, haplotype_chunk_max_span=700)
    assert a == b == c == g["modes"][mode]["out_vcf"]


def test_fisher_matches_scipy():
    from scipy.stats import fisher_exact
    from clairs_to_amd.haplotype_filtering import fisher_exact_two_sided
    import numpy as np
    rng = np.random.default_rng(2)
    same = n = 0
    for _ in range(300):
        a, b, c, d = (int(v) for v in rng.integers(0, 60, size=4))
        if a + b + c + d == 0 or a == b == c == d:
            continue
        p = fisher_exact_two_sided(a, b, c, d)
        q = fisher_exact([[a, b], [c, d]])[1]
        # the reference compares the running table probability with the observed one in floating point, without a tolerance:
        # a table that is EXACTLY as likely as the observed one (symmetric margins) can land one ulp above it and be left out,
        # so the value is the textbook two-sided p or that minus the tied tables - never more
        assert p <= q * (1 + 1e-9) + 1e-12, (a, b, c, d, p, q)
        n += 1
        same += abs(p - q) < 1e-9 * max(1.0, q) + 1e-12
    assert same > 0.9 * n


def test_passthrough_and_disabled_stage(tmp_path):
    g = load_json_gz("hapfilter.json.gz")
    out, _ = _run(tmp_path, g, "snv", disable_read_start_end_filtering=True)
    assert "ReadStartEnd" not in out
    # a non-PASS input record goes through untouched
    src = [r for r in g["modes"]["snv"]["pileup_vcf"].split("\n") if "\tLowQual\t" in r]
    assert src and all(r in out.split("\n") for r in src)
    from clairs_to_amd.haplotype_filtering import haplotype_filter
    a = Namespace(apply_haplotype_filtering=False, pileup_vcf_fn=str(tmp_path / "pileup.vcf"), output_vcf_fn=str(tmp_path / "link.vcf"),
                  output_dir=str(tmp_path / "w2"), ctg_name="chr1", flanking=100, min_alt_coverage=2)
    haplotype_filter(a)
    assert os.path.islink(tmp_path / "link.vcf")


def test_wide_fixture_both_reference_modes(tmp_path):
    """tests/golden/hapfilter_wide.json.gz: the reference on 16 simulated contigs (> 500 calls, SNV + indel pass) in its chunk mode
    AND in its default per-call mode (GNU parallel, one mpileup per call; src/haplotype_filtering.py:1038, 1140-1168) - the two
    wrote the same files, and so does this build (which always works job-wise)."""
    import sys
    from argparse import Namespace
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import hapsim
    import gen_hapfilter_wide as gw
    from clairs_to_amd.haplotype_filtering import haplotype_filter
    g = load_json_gz("hapfilter_wide.json.gz")
    calls = tags = 0
    seen = set()
    for rec in g["contigs"]:
        ctg = rec["name"]
        sim = hapsim.simulate(seed=rec["seed"])
        assert gw.digest(sim, ctg) == rec["inputs_sha256"], "the simulator changed: regenerate tests/golden/hapfilter_wide.json.gz"
        d = tmp_path / ("c%d" % rec["seed"])
        d.mkdir()
        ref = sim["ref"]
        (d / "ref.fa").write_text(">%s\n%s\n" % (ctg, ref))
        (d / "ref.fa.fai").write_text("%s\t%d\t%d\t%d\t%d\n" % (ctg, len(ref), len(ctg) + 2, len(ref), len(ref) + 1))
        (d / "germline.vcf").write_text(gw.germline_vcf(sim, ctg))
        for mode in ("snv", "indel"):
            assert rec[mode]["same_in_both_modes"]
            vcf_text, mp_text = gw.inputs_for(sim, ctg, mode)
            (d / ("pileup_%s.vcf" % mode)).write_text(vcf_text)
            (d / ("mp_%s.txt" % mode)).write_text(mp_text)
            out = d / ("out_%s.vcf" % mode)
            haplotype_filter(Namespace(
                tumor_bam_fn="unused.bam", ref_fn=str(d / "ref.fa"), ctg_name=ctg, pileup_vcf_fn=str(d / ("pileup_%s.vcf" % mode)),
                output_vcf_fn=str(out), germline_vcf_fn=str(d / "germline.vcf"), output_dir=str(d / ("work_" + mode)), threads=4,
                input_filter_tag=None, show_ref=False, samtools="samtools", mpileup_fn=str(d / ("mp_%s.txt" % mode)),
                apply_haplotype_filtering=True, min_mq=20, min_bq=0, min_alt_coverage=2, is_indel=(mode == "indel"), test_pos=None,
                flanking=100, haplotype_chunk_max_sites=200, haplotype_chunk_max_span=5000000, disable_read_start_end_filtering=False))
            got = out.read_text()
            assert got == rec[mode]["out_vcf"], (ctg, mode)
            for r in got.split("\n"):
                if r and r[0] != "#":
                    calls += 1
                    seen.update(r.split("\t")[6].split(";"))
    assert len(g["contigs"]) >= 16 and calls >= 500
    assert {"PASS", "LowAltBQ", "LowAltMQ", "ReadStartEnd", "VariantCluster", "NoAncestry", "MultiHap", "StrandBias", "LowSeqEntropy"} <= seen

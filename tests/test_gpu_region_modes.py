"""extract_candidates_calling's modes inside the run: REGION jobs of cto_run_chunks (`call_chunks --region_list`, rows `ctg i/n`) with the
confident BED (--bed_fn), --call_indels_only_in_these_regions (+ --bed_fn_source), --hybrid_mode_vcf_fn / --genotyping_mode_vcf_fn must extract,
in HBM, exactly the candidates the REFERENCE's extract_candidates_calling wrote for the same command-line set-ups (tests/golden/cli_run.json.gz:
its `<ctg>.<chunk>_0_1_snv|_indel` chunk files and `<ctg>.<chunk>_hybrid_info`, byte for byte), and their VCF records must be those of the
two-step run (the reference's chunk files through `call_chunks --chunk_list`).  src/extract_candidates_calling.py:225-270, 302, 347-383, 437-446,
490-497; the file-seam form of the same set-ups is tests/test_gpu_cli_argv.py."""
import os

import numpy as np
import pytest

from conftest import load_json_gz
import clisim
from test_gpu_cli import _pickle_models
from test_gpu_cli_argv import Work, golden  # noqa: F401  (fixture)

pytestmark = pytest.mark.gpu

SETUPS = [("ont", 4), ("ont", 6), ("ont_bed", 4), ("ont_bed", 6), ("ont_indel_bed", 6), ("ont_hybrid", 4), ("ont_genotyping", 4), ("ont_hybrid_indel", 6),
          ("ilmn", 4), ("ilmn", 6), ("hifi", 4), ("hifi", 6)]


def _opt(argv, k):
    v = argv[argv.index(k) + 1] if k in argv else None
    return None if v in (None, "None") else v


@pytest.mark.parametrize("name,K", SETUPS)
def test_region_jobs_extract_the_references_candidates_in_every_mode(tmp_path, golden, name, K):  # noqa: F811
    from clairs_to_amd.call_chunks import main as call_chunks
    from clairs_to_amd.synth import likelihood_table
    rec = golden["executed"][name]
    argv0 = rec["step1_argv"][0][1]
    select_indel = _opt(argv0, "--select_indel_candidates") == "True"
    if K == 6 and not select_indel:
        pytest.skip("this set-up ran without --select_indel_candidates")
    n_chunks = int(_opt(argv0, "--chunk_num"))
    sfx = "snv" if K == 4 else "indel"
    with Work(tmp_path, name, rec, golden) as wk:
        paths = _pickle_models(tmp_path, "CvT" if K == 4 else "CvT_Indel", "BiGRU_NACGT" if K == 4 else "BiGRU_NACGT_Indel", K)
        lik = tmp_path / "lik.txt"
        np.savetxt(lik, likelihood_table(K, seed=11), fmt="%.17g")
        (tmp_path / "REGIONS").write_text("".join("%s %d/%d\n" % (clisim.CTG, i + 1, n_chunks) for i in range(n_chunks)))
        common = ["--platform", _opt(argv0, "--platform") or "ont", "--tumor_bam_fn", wk.inputs["bam"], "--ref_fn", wk.inputs["ref"], "--bam_reader", "samtools", "--samtools", "samtools",
                  "--chkpnt_fn_acgt", paths["model_acgt"], "--chkpnt_fn_nacgt", paths["model_nacgt"], "--disable_indel_calling", str(K == 4),
                  "--likelihood_matrix_data", str(lik), "--show_ref"]
        modes = ["--snv_min_af", _opt(argv0, "--snv_min_af"), "--indel_min_af", _opt(argv0, "--indel_min_af"), "--min_coverage", _opt(argv0, "--min_coverage")]
        for k in ("--bed_fn", "--bed_fn_source", "--call_indels_only_in_these_regions", "--hybrid_mode_vcf_fn", "--genotyping_mode_vcf_fn"):
            v = _opt(argv0, k)
            if v is not None:
                modes += [k, wk.real(v)]
        cand, out_r = tmp_path / "cand", tmp_path / "out_r"
        call_chunks(["--region_list", str(tmp_path / "REGIONS"), "--output_dir", str(out_r), "--candidates_dir", str(cand)] + common + modes)
        # (a) the candidate lists and the hybrid_info rows are the reference's files
        want_chunks, n_cand = [], 0
        # the region files are named by their bounds: order them by start
        by_start = sorted((f for f in os.listdir(cand) if f.endswith("." + sfx)), key=lambda f: int(f.split("_")[1]))
        assert len(by_start) == n_chunks
        for i, f in enumerate(by_start):
            want = rec["candidates"].get("%s.%d_0_1_%s" % (clisim.CTG, i, sfx), "")
            got = open(cand / f).read()
            assert got == want, (name, K, i)
            n_cand += len(want.split("\n")) - 1
            want_chunks.append(want)
            if "hybrid" in name or "genotyping" in name:
                info = f[:-len("." + sfx)] + "_hybrid_info"
                assert open(cand / info).read() == rec["candidates"]["%s.%d_hybrid_info" % (clisim.CTG, i)], (name, K, i)
        assert n_cand > (20 if K == 4 else 2)
        # (b) the records of each region's VCF are those of the BED-driven job on the reference's chunk file
        files = []
        for i, text in enumerate(want_chunks):
            if not text:
                continue
            fn = tmp_path / ("%s.%d_0_1_%s" % (clisim.CTG, i, sfx))
            fn.write_text(text)
            files.append((i, str(fn)))
        (tmp_path / "CANDIDATES_FILES").write_text("".join(f + "\n" for _, f in files))
        out_b = tmp_path / "out_b"
        call_chunks(["--chunk_list", str(tmp_path / "CANDIDATES_FILES"), "--output_dir", str(out_b)] + common)
        recs = lambda fn: [ln for ln in open(fn).read().split("\n") if ln and not ln.startswith("#")] if os.path.exists(fn) else []
        n_rec = 0
        for i, fn in files:
            region_vcf = out_r / ("p_" + by_start[i][:-len("." + sfx)] + ".vcf")
            a, b = recs(region_vcf), recs(out_b / ("p_" + os.path.basename(fn) + ".vcf"))
            assert a == b, (name, K, i)
            n_rec += len(a)
        assert n_rec > 0

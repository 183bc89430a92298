"""The reference's command lines, run: `python -m clairs_to_amd <sub-module> <the argv run_clairs_to built>` against what the reference's own
sub-modules wrote with the same argv on the same inputs (tests/golden/cli_run.json.gz, made by gen_cli.py in the build container).

STEP 1 (extract_candidates_calling + concat_files) in eight set-ups - default, --bed_fn, --call_indels_only_in_these_regions,
--hybrid_mode_vcf_fn, --genotyping_mode_vcf_fn, the hybrid list with indel candidates, and the Illumina and HiFi platforms' gates (--min_bq 0,
--indel_min_af 0.05) - must leave the candidates folder the reference left, file for file and byte for byte: the BED
chunk files, the list files, bed/<ctg>_<chunk>.bed, <ctg>.<chunk>_hybrid_info.  STEP 2 / STEP 6 (create_tensor_pileup_calling x 2, predict,
call_variants per chunk file) must write the reference's tensor text (SHA-256), its probability rows (non-probability fields equal,
probabilities within 1e-4: north_star's tolerance) and its p_<chunk>.vcf: header byte for byte, records field for field with QUAL / GQ free to
move in the last digit - and, from the reference's OWN probability files, the whole VCF byte for byte.  One whole run (no phasing, no tagging
database: fifteen commands, all of them sub-modules mirrored here) goes down to the final snv.vcf / indel.vcf.

`samtools` is clisim.py's stand-in on both sides (neither box has samtools)."""
import gzip
import hashlib
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, load_json_gz
import clisim
from test_gpu_cli import _pickle_models

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def golden():
    g = load_json_gz("cli_run.json.gz")
    assert g["chunk_kw"] == clisim.CHUNK_KW
    from clairs_to_amd.synth import mpileup_text
    assert hashlib.sha256(mpileup_text(clisim.chunk(), 0, ctg=clisim.CTG).encode()).hexdigest() == g["pileup_sha256"]
    return g


class Work:
    """the scratch tree of one run: <t>/in (inputs), <t>/runs/<name> = @W@, the stand-in `samtools` first on PATH"""

    def __init__(self, tmp_path, name, rec, g):
        self.t = str(tmp_path)
        self.w = os.path.join(self.t, "runs", name)
        self.tmp = rec.get("tmp", "tmp")                                 # tmp_<sample name> when the run names one
        self.inputs = clisim.write_inputs(os.path.join(self.t, "in"))
        for k, text in g["inputs"].items():
            assert open(self.inputs[k]).read() == text, k               # the generator's inputs, regenerated
        clisim.write_shims(os.path.join(self.t, "bin"))
        self.old_path = os.environ["PATH"]
        for sub in ("candidates", "predict", "vcf_output", "pileup_tensor_can_affirmative", "pileup_tensor_can_negational", "split_beds", "split_indel_beds"):
            os.makedirs(os.path.join(self.w, self.tmp, sub), exist_ok=True)
        for k, text in rec["work_files"].items():
            open(os.path.join(self.w, self.tmp, k), "w").write(self.real(text))
        for d in ("split_beds", "split_indel_beds"):
            for f, text in rec.get(d, {}).items():
                open(os.path.join(self.w, self.tmp, d, f), "w").write(text)

    def __enter__(self):
        os.environ["PATH"] = os.path.join(self.t, "bin") + ":" + self.old_path
        return self

    def __exit__(self, *a):
        os.environ["PATH"] = self.old_path

    def real(self, s):
        return s.replace("@W@", self.w).replace("@T@", self.t)

    def run(self, sub, argv):
        from clairs_to_amd.__main__ import dispatch
        dispatch(sub, [self.real(t) for t in argv])

    def files(self, sub):
        out = {}
        d = os.path.join(self.w, self.tmp, sub)
        for base, _, names in os.walk(d):
            for f in names:
                out[os.path.relpath(os.path.join(base, f), d)] = open(os.path.join(base, f)).read().replace(self.w, "@W@").replace(self.t, "@T@")
        return out


def same_candidates(got, want):
    assert sorted(got) == sorted(want)
    for f in want:
        if f in ("SNV_CANDIDATES_FILES", "INDEL_CANDIDATES_FILES"):           # concat_files lists in directory order
            assert sorted(got[f].split("\n")) == sorted(want[f].split("\n")), f
        else:
            assert got[f] == want[f], f


@pytest.mark.parametrize("name", ["ont", "ont_bed", "ont_indel_bed", "ont_hybrid", "ont_genotyping", "ont_hybrid_indel", "ilmn", "hifi"])
def test_step1_writes_the_references_candidates_folder(tmp_path, golden, name):
    rec = golden["executed"][name]
    with Work(tmp_path, name, rec, golden) as wk:
        for sub, argv in rec["step1_argv"]:
            wk.run(sub, argv)
        got = wk.files("candidates")
    same_candidates(got, rec["candidates"])
    if "hybrid" in name or "genotyping" in name:
        info = [f for f in got if f.endswith("_hybrid_info")]
        assert len(info) == 3 and sum(len(got[f].split("\n")) for f in info) > 40


def test_python_m_entry_point_takes_the_argv(tmp_path, golden):
    """the same through the real entry point, one invocation: `python -m clairs_to_amd extract_candidates_calling <argv>`"""
    rec = golden["executed"]["ont_hybrid"]
    with Work(tmp_path, "ont_hybrid", rec, golden) as wk:
        sub, argv = rec["step1_argv"][0]
        p = subprocess.run([sys.executable, "-m", "clairs_to_amd", sub] + [wk.real(t) for t in argv], cwd=ROOT,
                           env=dict(os.environ, PYTHONPATH=ROOT), stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
        assert p.returncode == 0, p.stdout.decode()[-2000:]
        got = wk.files("candidates")
    chunk = argv[argv.index("--chunk_id") + 1]
    mine = {f: t for f, t in rec["candidates"].items() if (".%d_" % (int(chunk) - 1)) in f or f.endswith("_%d" % (int(chunk) - 1)) or
            f == "bed/chr20_%d.bed" % (int(chunk) - 1)}
    assert len(mine) >= 4
    for f, text in mine.items():
        assert got[f] == text, f


def vcf_parts(text):
    rows = text.split("\n")
    return "".join(r + "\n" for r in rows if r.startswith("#")), [r for r in rows if r and not r.startswith("#")]


def write_models(wk, tmp_path):
    """<t>/models: the weights recipe pickled as the reference's releases are, and the likelihood tables gen_cli.make_models wrote"""
    models = os.path.join(wk.t, "models")
    os.makedirs(models, exist_ok=True)
    for mode, K, aff_cls, neg_cls in (("snv", 4, "CvT", "BiGRU_NACGT"), ("indel", 6, "CvT_Indel", "BiGRU_NACGT_Indel")):
        d = tmp_path / ("pk_" + mode)
        d.mkdir()
        paths = _pickle_models(d, aff_cls, neg_cls, K)
        os.replace(paths["model_acgt"], os.path.join(models, "aff_%s.pkl" % mode))
        os.replace(paths["model_nacgt"], os.path.join(models, "neg_%s.pkl" % mode))
        from clairs_to_amd.synth import likelihood_table
        np.savetxt(os.path.join(models, "lik_%s.txt" % mode), likelihood_table(K, seed=7 + K), fmt="%.17g")


def same_vcf_but_the_last_digit(got, want, f):
    """header byte for byte; records field for field with QUAL / GQ free to move with the 8th decimal of the fp32 probabilities"""
    head, rows = vcf_parts(want)
    ghead, grows = vcf_parts(got)
    assert ghead == head, f
    assert len(grows) == len(rows), f
    for a, b in zip(grows, rows):
        a, b = a.split("\t"), b.split("\t")
        assert a[:5] == b[:5] and a[6:9] == b[6:9], (f, a, b)
        assert abs(float(a[5]) - float(b[5])) < 0.02, (f, a, b)
        fa_, fb_ = a[9].split(":"), b[9].split(":")
        assert fa_[0] == fb_[0] and fa_[2:] == fb_[2:] and abs(int(fa_[1]) - int(fb_[1])) <= 1, (f, a, b)
    return len(rows)


@pytest.mark.parametrize("name", ["ont", "ilmn", "hifi"])
def test_step2_and_step6_with_the_references_argv(tmp_path, golden, name):
    """ont, and the Illumina and HiFi platforms: Illumina creates the affirmative tensors only and `ln -sf`s them into the negational folder
    (run_clairs_to:1248-1252: a shell command of the orchestrator, run here as one) - predict then reads the same file through both paths"""
    rec = golden["executed"][name]
    if "step2_argv" not in rec:
        pytest.skip("the fixture holds STEP 1 only for this set-up")
    with Work(tmp_path, name, rec, golden) as wk:
        for sub, argv in rec["step1_argv"]:
            wk.run(sub, argv)
        same_candidates(wk.files("candidates"), rec["candidates"])
        write_models(wk, tmp_path)
        n = {}
        for sub, argv in rec["step2_argv"]:
            if sub == "sh":
                subprocess.run(wk.real(argv[0]), shell=True, check=True)
            else:
                wk.run(sub, argv)
            n[sub] = n.get(sub, 0) + 1
        if name == "ilmn":
            assert n == {"concat_files": 1, "create_tensor_pileup_calling": 6, "sh": 2, "predict": 6, "call_variants": 6}
        else:
            assert n == {"concat_files": 1, "create_tensor_pileup_calling": 12, "predict": 6, "call_variants": 6}
        # tensor text: byte-identical (the reference's gzip stream is not reproducible; its content is)
        for f, sha in rec["tensor_sha256"].items():
            assert hashlib.sha256(gzip.open(os.path.join(wk.w, wk.tmp, f), "rb").read()).hexdigest() == sha, f
        # probability rows
        for f, text in rec["predict"].items():
            K = 4 if f.endswith("_snv") else 6
            got = [r.split("\t") for r in gzip.open(os.path.join(wk.w, wk.tmp, "predict", f), "rt").read().split("\n") if r]
            want = [r.split("\t") for r in text.split("\n") if r]
            assert len(got) == len(want) > (10 if K == 4 else 2)
            worst = 0.0
            for a, b in zip(got, want):
                assert a[:6] == b[:6] and len(a) == len(b)
                pa = np.array([[float(v) for v in x.split()] for x in a[6:6 + 2 * K]])
                pb = np.array([[float(v) for v in x.split()] for x in b[6:6 + 2 * K]])
                worst = max(worst, float(np.abs(pa - pb).max()))
            assert worst < 1e-4, (f, worst)                                  # north_star's tolerance, fp32 probabilities
        # p_<chunk>.vcf through the whole chain
        got = wk.files("vcf_output")
        assert sorted(got) == sorted(rec["vcf_output"])
        for f, text in rec["vcf_output"].items():
            assert same_vcf_but_the_last_digit(got[f], text, f) > 0
        # call_variants on the reference's own probability files: the whole VCF, byte for byte, with and without --show_ref
        for f, text in rec["predict"].items():
            with gzip.open(os.path.join(wk.w, wk.tmp, "predict", f), "wt") as out:
                out.write(text)
        for sub, argv in rec["step2_argv"]:
            if sub != "call_variants":
                continue
            wk.run(sub, argv)
            if "vcf_output_show_ref" in rec:
                a2 = list(argv)
                i = a2.index("--call_fn") + 1
                a2[i] = a2[i].replace("vcf_output", "vcf_output_show_ref")
                wk.run(sub, a2 + ["--show_ref"])
        assert wk.files("vcf_output") == rec["vcf_output"]
        if "vcf_output_show_ref" in rec:
            assert wk.files("vcf_output_show_ref") == rec["vcf_output_show_ref"]


WHOLE = ["ont_whole", "ont_whole_knobs", "ont_hac_whole", "hifi_whole"]


@pytest.mark.parametrize("name", WHOLE)
def test_the_whole_run_without_phasing(tmp_path, golden, name):
    """`run_clairs_to --disable_intermediate_phasing --disable_nonsomatic_tagging`: every command of that run is a sub-module of the hot path
    or of its tail, so the reference executed all fifteen on the simulated pileup (gen_cli.py: STEP 1, STEP 2, sort_vcf, `ln -sf`,
    postprocess_vcf, STEP 6, sort_vcf, `ln -sf`, postprocess_vcf) - the same command lines through `python -m clairs_to_amd`'s dispatcher
    must leave <output>/snv.vcf and <output>/indel.vcf as the reference left them.  `ont_whole_knobs`: the same run with --print_ref_calls
    (RefCall rows through sort_vcf and postprocess_vcf), a confident BED, --qual 12, other AF / coverage gates, --max_indel_length 50 and a
    sample name (work folder tmp_T1, outputs snv_T1.vcf / indel_T1.vcf; the merged VCFs keep SAMPLE: sort_vcf is not told the name).
    `ont_hac_whole` / `hifi_whole`: the run under two other platform tables (--min_bq 15 in extraction and in the affirmative tensors; HiFi's gates)."""
    rec = golden["executed"][name]
    with Work(tmp_path, name, rec, golden) as wk:
        for sub, argv in rec["step1_argv"]:
            wk.run(sub, argv)
        same_candidates(wk.files("candidates"), rec["candidates"])
        write_models(wk, tmp_path)

        def tail(after_predict=None):
            n = {}
            for sub, argv in rec["whole_argv"]:
                if sub == "sh":
                    subprocess.run(wk.real(argv[0]), shell=True, check=True)
                else:
                    wk.run(sub, argv)
                    if sub == "predict" and after_predict:
                        after_predict(wk.real(argv[argv.index("--predict_fn") + 1]))
                n[sub] = n.get(sub, 0) + 1
            final = {f: open(os.path.join(wk.w, f)).read().replace(wk.w, "@W@").replace(wk.t, "@T@") for f in sorted(os.listdir(wk.w))
                     if f.endswith(".vcf")}
            return n, final

        n, final = tail()
        assert n == {"concat_files": 1, "create_tensor_pileup_calling": 12, "predict": 6, "call_variants": 6, "sort_vcf": 2, "sh": 2,
                     "postprocess_vcf": 2}
        assert sorted(final) == sorted(rec["final"]) == (["indel_T1.vcf", "snv_T1.vcf"] if name == "ont_whole_knobs" else ["indel.vcf", "snv.vcf"])
        got = wk.files("vcf_output")
        assert sorted(got) == sorted(rec["vcf_output"])
        for f, text in rec["vcf_output"].items():
            same_vcf_but_the_last_digit(got[f], text, f)
        for f, text in rec["final"].items():
            assert same_vcf_but_the_last_digit(final[f], text, f) > (10 if f.startswith("snv") else 1)
            assert "\tPASS\t" in text and ("\tRefCall\t" in text) == (name == "ont_whole_knobs" and f.startswith("snv"))

        # the same tail on the reference's own probability files: every file of the run, byte for byte
        def swap(path):
            with gzip.open(path, "wt") as out:
                out.write(rec["predict"][os.path.basename(path)])
        _, final = tail(after_predict=swap)
        assert wk.files("vcf_output") == rec["vcf_output"]
        assert final == rec["final"]


@pytest.mark.parametrize("name", WHOLE)
def test_the_whole_run_as_one_invocation_per_model(tmp_path, golden, name):
    """The same runs the way this package is meant to be driven: `call_chunks --region_list` (rows `ctg i/n`), once for the SNV models and
    once for the indel models - extraction, tensor creation, both networks, the epilogue, the chunk VCFs, sort_vcf and postprocess_vcf in
    one process each, no candidates folder, no tensor or probability files - must leave the final snv.vcf / indel.vcf the REFERENCE left
    after its fifteen commands (records with QUAL / GQ free in the last digit: these probabilities never pass through the 6-decimal
    text of the probability files)."""
    from clairs_to_amd.call_chunks import main as call_chunks
    rec = golden["executed"][name]
    opt = lambda argv, k: argv[argv.index(k) + 1]
    a1 = rec["step1_argv"][0][1]
    n_chunks = int(opt(a1, "--chunk_num"))
    first = lambda sub: [a for s, a in rec["whole_argv"] if s == sub][0]
    post = {("snv" if "True" == opt(a, "--disable_indel_calling") else "indel"): a for s, a in rec["whole_argv"] if s == "postprocess_vcf"}
    with Work(tmp_path, name, rec, golden) as wk:
        write_models(wk, tmp_path)
        regions = tmp_path / "REGIONS"
        regions.write_text("".join("%s %d/%d\n" % (clisim.CTG, i + 1, n_chunks) for i in range(n_chunks)))
        modes = ["--snv_min_af", opt(a1, "--snv_min_af"), "--indel_min_af", opt(a1, "--indel_min_af"), "--min_coverage", opt(a1, "--min_coverage")]
        for k in ("--bed_fn", "--bed_fn_source", "--call_indels_only_in_these_regions"):
            if k in a1 and opt(a1, k) != "None":
                modes += [k, wk.real(opt(a1, k))]
        if "--show_ref" in first("call_variants"):
            modes += ["--show_ref"]
        modes += ["--min_bq", opt(first("create_tensor_pileup_calling"), "--min_bq"), "--extract_min_bq", opt(a1, "--min_bq")]      # 20 / 15 (HAC) / HiFi's
        if "--max_indel_length" in first("create_tensor_pileup_calling"):
            modes += ["--max_indel_length", opt(first("create_tensor_pileup_calling"), "--max_indel_length")]
        sample = opt(post["snv"], "--sample_name")
        for mode in ("snv", "indel"):
            pa = post[mode]
            out = tmp_path / ("chunks_" + mode)
            call_chunks(["--region_list", str(regions), "--output_dir", str(out), "--platform", opt(a1, "--platform"),
                         "--tumor_bam_fn", wk.inputs["bam"], "--ref_fn", wk.inputs["ref"], "--bam_reader", "samtools", "--samtools", "samtools",
                         "--chkpnt_fn_acgt", os.path.join(wk.t, "models", "aff_%s.pkl" % mode),
                         "--chkpnt_fn_nacgt", os.path.join(wk.t, "models", "neg_%s.pkl" % mode),
                         "--likelihood_matrix_data", os.path.join(wk.t, "models", "lik_%s.txt" % mode),
                         "--disable_indel_calling", str(mode == "snv")] + modes +
                        ["--merged_vcf_fn", os.path.join(wk.w, wk.tmp, "vcf_output", mode + "_pileup.vcf"), "--final_vcf_fn", wk.real(opt(pa, "--output_fn")),
                         "--postprocess_qual", opt(pa, "--qual"), "--postprocess_qual_cutoff_phaseable_region", opt(pa, "--qual_cutoff_phaseable_region"),
                         "--postprocess_qual_cutoff_unphaseable_region", opt(pa, "--qual_cutoff_unphaseable_region"), "--postprocess_af", opt(pa, "--af"),
                         "--sample_name", sample, "--cmdline", wk.real(opt(pa, "--cmdline"))])
        merged = wk.files("vcf_output")
        for mode in ("snv", "indel"):
            # the reference's sort_vcf is not told the sample name (run_clairs_to:1309-1315): its merged VCF says SAMPLE, this one the name given
            want = rec["vcf_output"][mode + "_pileup.vcf"].replace("\tFORMAT\tSAMPLE\n", "\tFORMAT\t%s\n" % sample)
            same_vcf_but_the_last_digit(merged[mode + "_pileup.vcf"], want, mode)
            fn = opt(post[mode], "--output_fn")
            got = open(wk.real(fn)).read().replace(wk.w, "@W@").replace(wk.t, "@T@")
            assert same_vcf_but_the_last_digit(got, rec["final"][os.path.basename(fn)], mode) > (10 if mode == "snv" else 1)
        assert not os.listdir(os.path.join(wk.w, wk.tmp, "candidates")) and not os.listdir(os.path.join(wk.w, wk.tmp, "predict"))

#!/usr/bin/env python3
"""The reference's own command lines, captured and executed (build container only: /root/reference must exist).

  argv.json.gz     * `parsers`: the option table of every reference sub-module this package mirrors (option strings, action, nargs, type,
                     default), read from the ArgumentParser each sub-module's main() builds;
                   * `runs`: what `run_clairs_to --dry_run` prints for ont / ilmn / hifi x SNV-only / SNV+indel x --print_ref_calls x --bed_fn /
                     --call_indels_only_in_these_regions / --genotyping_mode_vcf_fn / --hybrid_mode_vcf_fn / --debug ...: per shell command the
                     `clairs_to.py <sub-module> ...` invocations as argv lists, GNU parallel's replacement strings ({1} {2} {3} {1/} {1/.}) still in
                     place, scratch paths replaced by @W@ (the work directory) / @REF@ (the reference checkout).
  cli_run.json.gz  those command lines EXECUTED with the reference's sub-modules on the simulated run of clisim.py (GNU parallel emulated: one
                   invocation per row of the `::::` file): STEP 1 in eight set-ups (default, --bed_fn, --call_indels_only_in_these_regions,
                   --hybrid_mode_vcf_fn, --genotyping_mode_vcf_fn, hybrid + indel candidates; the Illumina and HiFi gates) -> every file of the
                   candidates folder; for ont / ilmn / hifi STEP 2 (SNV) and STEP 6 (indel): the probability files and the whole p_<chunk>.vcf
                   files, header included, for ont also with --print_ref_calls; `ont_whole` (--disable_intermediate_phasing
                   --disable_nonsomatic_tagging): all fifteen commands of the run, in order, down to the final snv.vcf / indel.vcf.

Only data is stored (argv lists, option tables, file contents the reference wrote).  Usage: python tests/golden/gen_cli.py"""
import argparse
import gzip
import importlib
import json
import os
import shlex
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import numpy as np  # noqa: E402
import clisim  # noqa: E402

MIRRORED = {"extract_candidates_calling": "src.extract_candidates_calling", "create_tensor_pileup_calling": "src.create_tensor_pileup_calling",
            "predict": "clairs.predict", "call_variants": "clairs.call_variants", "sort_vcf": "src.sort_vcf", "postprocess_vcf": "src.postprocess_vcf",
            "haplotype_filtering": "src.haplotype_filtering", "realign_reads": "src.realign_reads", "realign_variants": "src.realign_variants",
            "concat_files": "src.concat_files"}


def dump_json_gz(name, obj):
    raw = json.dumps(obj, separators=(",", ":"), sort_keys=True).encode()
    with open(os.path.join(HERE, name), "wb") as f:
        with gzip.GzipFile(fileobj=f, mode="wb", mtime=0) as g:
            g.write(raw)
    print("wrote", name, len(raw), "bytes raw")


# ------------------------------------------------------------------------------------------ option tables
PARSER_PROBE = r'''
import argparse, importlib, json, sys
sys.path.insert(0, %r)
class Stop(Exception): pass
cap = {}
def grab(self, *a, **k):
    cap["p"] = self
    raise Stop()
argparse.ArgumentParser.parse_args = grab
argparse.ArgumentParser.parse_known_args = grab
mod = importlib.import_module(sys.argv[1])
sys.argv = [sys.argv[1], "--probe"]      # some mains print the help and exit when given nothing
try:
    mod.main()
except Stop:
    pass
rows = []
for a in cap["p"]._actions:
    if not a.option_strings or isinstance(a, argparse._HelpAction):
        continue
    d = a.default
    rows.append(dict(options=a.option_strings, action=type(a).__name__, nargs=a.nargs, type=getattr(a.type, "__name__", None),
                     default=d if isinstance(d, (int, float, str, bool, type(None))) else repr(d), required=bool(a.required)))
print(json.dumps(rows))
'''


def gen_parsers(tmp):
    """The realign modules load their two ctypes libraries at import (src/realign_reads.py:56-65): a scratch conda prefix holds oracle/_ref's build of
    the reference's realigner and this package's debruijn_graph.so under the names it looks for, exactly as gen_realign.py does."""
    mods = os.path.join(tmp, "conda", "bin", "preprocess", "realign")
    os.makedirs(mods, exist_ok=True)
    if not os.path.exists(os.path.join(tmp, "conda", "bin", "python")):
        os.symlink(sys.executable, os.path.join(tmp, "conda", "bin", "python"))
    shutil.copy(os.path.join(ROOT, "oracle", "_ref", "librealigner_ref.so"), os.path.join(mods, "realigner"))
    shutil.copy(os.path.join(ROOT, "clairs_to_amd", "realign", "debruijn_graph.so"), os.path.join(mods, "debruijn_graph"))
    env = dict(os.environ, PATH=os.path.join(tmp, "conda", "bin") + ":" + os.environ["PATH"],
               LD_LIBRARY_PATH=os.path.join(ROOT, "clairs_to_amd") + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
    out = {}
    for name, mod in MIRRORED.items():
        p = subprocess.run([sys.executable, "-c", PARSER_PROBE % REF, mod], stdout=subprocess.PIPE, env=env, check=True, cwd=tmp)
        out[name] = json.loads(p.stdout.decode().strip().split("\n")[-1])
        print("parser", name, len(out[name]), "options")
    return out


# ------------------------------------------------------------------------------------------ dry runs
def fake_conda(tmp):
    """what check_args insists on finding under --conda_prefix (run_clairs_to:821-838 and the longphase probe): empty stand-ins"""
    c = os.path.join(tmp, "conda")
    os.makedirs(os.path.join(c, "bin", "clairs-to_databases"), exist_ok=True)
    for f in ("gnomad.r2.1.af-ge-0.001.sites.vcf.gz", "dbsnp.b138.non-somatic.sites.vcf.gz", "1000g-pon.sites.vcf.gz",
              "CoLoRSdb.GRCh38.v1.1.0.deepvariant.glnexus.af-ge-0.001.vcf.gz"):
        open(os.path.join(c, "bin", "clairs-to_databases", f), "w").close()
    clisim.write_shims(os.path.join(c, "bin"))
    return c


def dry_run(tmp, name, platform, flags, conda, inputs, models):
    w = os.path.join(tmp, "runs", name)
    cmd = [sys.executable, os.path.join(REF, "run_clairs_to"), "-T", inputs["bam"], "-R", inputs["ref"], "-o", w, "-t", "4", "-p", platform,
           "--conda_prefix", conda, "--chunk_size", str(clisim.CHUNK_SIZE), "--dry_run",
           "--snv_pileup_affirmative_model_path", models["snv_aff"], "--snv_pileup_negational_model_path", models["snv_neg"],
           "--indel_pileup_affirmative_model_path", models["indel_aff"], "--indel_pileup_negational_model_path", models["indel_neg"],
           "--snv_likelihood_matrix_data", models["snv_lik"], "--indel_likelihood_matrix_data", models["indel_lik"]] + flags
    env = dict(os.environ, PATH=os.path.join(conda, "bin") + ":" + os.environ["PATH"])
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=env, cwd=tmp)
    log = p.stdout.decode()
    assert p.returncode == 0, log[-3000:]
    lines = log.split("\n")
    commands = [lines[i + 1] for i, l in enumerate(lines) if l.startswith("[INFO] RUN THE FOLLOWING COMMAND:")]
    assert commands, log[-3000:]
    return w, commands


STOP = {"::::", ")", "&&", "2>&1", "|", "("}


def invocations(command):
    """[(sub-module, argv, name of the `::::` file or None)] of one shell command of the dry run"""
    toks = shlex.split(command)
    out = []
    i = 0
    while i < len(toks):
        if toks[i] == os.path.join(REF, "clairs_to.py"):
            sub = toks[i + 1]
            j = i + 2
            while j < len(toks) and toks[j] not in STOP:
                j += 1
            source = toks[j + 1] if j < len(toks) and toks[j] == "::::" else None
            out.append((sub, toks[i + 2:j], source))
            i = j
        else:
            i += 1
    return out


def norm(tok, tmp, w):
    return tok.replace(w, "@W@").replace(tmp, "@T@").replace(REF, "@REF@")


def substitute(argv, fields):
    """GNU parallel's replacement strings for one input row: {n} the n-th column, {1/} basename, {1/.} basename without its last extension"""
    out = []
    for t in argv:
        for n, v in enumerate(fields, 1):
            t = t.replace("{%d}" % n, v)
        base = os.path.basename(fields[0])
        t = t.replace("{1/.}", base.rsplit(".", 1)[0] if "." in base else base).replace("{1/}", base)
        out.append(t)
    return out


# ------------------------------------------------------------------------------------------ executing them with the reference
def run_ref(sub, argv, env, cwd):
    p = subprocess.run([sys.executable, os.path.join(REF, "clairs_to.py"), sub] + argv, env=env, cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    assert p.returncode == 0, (sub, argv, p.stdout.decode()[-3000:])
    return p.stdout.decode()


def run_step(commands, wanted, w, env, only_first_rows=None):
    """Execute the `wanted` sub-module invocations of the dry run's commands, in order; the `::::` file gives one invocation per row."""
    ran = []
    for command in commands:
        for sub, argv, source in invocations(command):
            if sub not in wanted:
                continue
            if source is None:
                run_ref(sub, argv, env, w)
                ran.append((sub, argv))
                continue
            rows = [r for r in open(source).read().split("\n") if r.strip()]
            for r in sorted(rows):
                fields = r.split(" ") if os.path.basename(source) == "CHUNK_LIST" else [r]
                a = substitute(argv, fields)
                run_ref(sub, a, env, w)
                ran.append((sub, a))
    return ran


def folder_files(d, w, tmp):
    out = {}
    for base, _, files in os.walk(d):
        for f in files:
            p = os.path.join(base, f)
            out[os.path.relpath(p, d)] = norm(open(p).read(), tmp, w)
    return out


def make_models(tmp):
    """reference modules with the weights recipe, torch.save'd as its releases are; the likelihood tables of synth.likelihood_table"""
    sys.path.insert(0, REF)
    import torch
    from gen_golden import build_reference_model
    from clairs_to_amd.synth import likelihood_table
    d = os.path.join(tmp, "models")
    os.makedirs(d, exist_ok=True)
    m = {}
    for mode, n_out, aff_cls, neg_cls in (("snv", 4, "CvT", "BiGRU_NACGT"), ("indel", 6, "CvT_Indel", "BiGRU_NACGT_Indel")):
        ma, _ = build_reference_model(aff_cls, n_out)
        mn, _ = build_reference_model(neg_cls, n_out)
        m[mode + "_aff"], m[mode + "_neg"] = os.path.join(d, "aff_%s.pkl" % mode), os.path.join(d, "neg_%s.pkl" % mode)
        torch.save({"model_acgt": ma}, m[mode + "_aff"])
        torch.save({"model_nacgt": mn}, m[mode + "_neg"])
        m[mode + "_lik"] = os.path.join(d, "lik_%s.txt" % mode)
        np.savetxt(m[mode + "_lik"], likelihood_table(n_out, seed=7 + n_out), fmt="%.17g")
    return m


DRY_MATRIX = [
    # name, platform, extra run_clairs_to flags
    ("ont", "ont_r10_dorado_sup_5khz", []),
    ("ont_snv_only", "ont_r10_dorado_sup_5khz", ["--disable_indel_calling"]),
    ("ont_ref_calls", "ont_r10_dorado_sup_5khz", ["--print_ref_calls"]),
    ("ont_bed", "ont_r10_dorado_sup_5khz", ["--bed_fn", "@bed@"]),
    ("ont_indel_bed", "ont_r10_dorado_sup_5khz", ["--call_indels_only_in_these_regions", "@indel_bed@"]),
    ("ont_hybrid", "ont_r10_dorado_sup_5khz", ["--hybrid_mode_vcf_fn", "@vcf@"]),
    ("ont_genotyping", "ont_r10_dorado_sup_5khz", ["--genotyping_mode_vcf_fn", "@vcf@"]),
    ("ont_debug_nophase", "ont_r10_dorado_sup_5khz", ["--debug", "True", "--disable_intermediate_phasing", "--disable_nonsomatic_tagging", "--qual", "12",
                                                      "--snv_min_af", "0.08", "--indel_min_af", "0.12", "--min_coverage", "6", "--sample_name", "T1",
                                                      "--bam_mplp_set_maxcnt", "4000", "--max_indel_length", "50"]),
    ("ont_debug", "ont_r10_dorado_sup_5khz", ["--debug", "True", "--haplotype_input_filter_tag", "PASS"]),
    ("ont_hac", "ont_r10_dorado_hac_4khz", ["--ctg_name", "chr20"]),
    ("ilmn", "ilmn", []),
    ("ilmn_ref_calls_norealign", "ilmn", ["--print_ref_calls", "--enable_realignment", "False", "--enable_postfilter", "False"]),
    ("hifi", "hifi_revio", []),
    ("hifi_use_gpu", "hifi_revio", ["--use_gpu", "--region", "chr20:400-3000"]),
    # every command of this one is a sub-module of the hot path or its tail (no phasing, no tagging database): executed from end to end
    ("ont_whole", "ont_r10_dorado_sup_5khz", ["--disable_intermediate_phasing", "--disable_nonsomatic_tagging"]),
    # the same with the knobs turned: RefCall rows through sort_vcf / postprocess_vcf, a confident BED, other gates, a sample name (-> tmp_T1, snv_T1.vcf)
    ("ont_whole_knobs", "ont_r10_dorado_sup_5khz", ["--disable_intermediate_phasing", "--disable_nonsomatic_tagging", "--print_ref_calls", "--bed_fn", "@bed@",
                                                    "--qual", "12", "--snv_min_af", "0.08", "--indel_min_af", "0.12", "--min_coverage", "6", "--sample_name", "T1",
                                                    "--max_indel_length", "50"]),
    # and under two other platform tables: --min_bq 15 (the 4 kHz HAC model's gate, in extraction and in the affirmative tensors), HiFi's gates
    ("ont_hac_whole", "ont_r10_dorado_hac_4khz", ["--disable_intermediate_phasing", "--disable_nonsomatic_tagging"]),
    ("hifi_whole", "hifi_revio", ["--disable_intermediate_phasing", "--disable_nonsomatic_tagging"]),
]
EXEC = {"ont": ("extract_candidates_calling", "concat_files"), "ont_bed": ("extract_candidates_calling",), "ont_indel_bed": ("extract_candidates_calling",),
        "ont_hybrid": ("extract_candidates_calling",), "ont_genotyping": ("extract_candidates_calling",),
        # STEP 1 under the other platforms' gates (--min_bq, --indel_min_af of run_clairs_to's platform tables) on the same simulated pileup
        "ilmn": ("extract_candidates_calling", "concat_files"), "hifi": ("extract_candidates_calling", "concat_files"),
        "ont_whole": ("extract_candidates_calling", "concat_files"), "ont_whole_knobs": ("extract_candidates_calling", "concat_files"),
        "ont_hac_whole": ("extract_candidates_calling", "concat_files"), "hifi_whole": ("extract_candidates_calling", "concat_files")}
TAIL = ("concat_files", "create_tensor_pileup_calling", "predict", "call_variants", "sort_vcf", "postprocess_vcf")


def main():
    assert os.path.isdir(REF)
    tmp = tempfile.mkdtemp(prefix="gen_cli_")
    try:
        parsers = gen_parsers(tmp)
        conda = fake_conda(tmp)
        inputs = clisim.write_inputs(os.path.join(tmp, "in"))
        models = make_models(tmp)
        env = dict(os.environ, PATH=os.path.join(conda, "bin") + ":" + os.environ["PATH"], PYTHONPATH=REF)
        runs, executed = [], {}
        for name, platform, flags in DRY_MATRIX:
            flags = [inputs.get(f.strip("@"), f) if f.startswith("@") else f for f in flags]
            w, commands = dry_run(tmp, name, platform, flags, conda, inputs, models)
            inv = []
            for ci, command in enumerate(commands):
                for sub, argv, source in invocations(command):
                    inv.append(dict(command=ci, submodule=sub, argv=[norm(t, tmp, w) for t in argv], source=os.path.basename(source) if source else None))
            runs.append(dict(name=name, platform=platform, flags=[norm(f, tmp, w) for f in flags], invocations=inv))
            print("dry run", name, len(commands), "commands,", len(inv), "sub-module invocations")
            if name not in EXEC:
                continue
            tmp_name = "tmp" if os.path.isdir(os.path.join(w, "tmp")) else [d for d in sorted(os.listdir(w)) if d.startswith("tmp")][0]
            wt = os.path.join(w, tmp_name)                                      # tmp_<sample name> when one is given
            rec = dict(work_files={k: norm(open(os.path.join(wt, k)).read(), tmp, w) for k in ("CHUNK_LIST", "CONTIGS")})
            for sd in ("split_beds", "split_indel_beds"):
                if os.path.isdir(os.path.join(wt, sd)):
                    rec[sd] = folder_files(os.path.join(wt, sd), w, tmp)
            ran = run_step(commands[:1], EXEC[name], w, env)
            rec["step1_argv"] = [[s, [norm(t, tmp, w) for t in a]] for s, a in ran]
            rec["candidates"] = folder_files(os.path.join(wt, "candidates"), w, tmp)
            print("  step 1:", len(ran), "invocations,", len(rec["candidates"]), "files")
            if name in ("ont", "ilmn", "hifi"):
                # STEP 2 (create_tensor x2 - Illumina: once, then `ln -sf` of the affirmative tensors into the negational folder,
                # run_clairs_to:1248-1252 - predict, call_variants) and STEP 6 (the concat + the same commands on the indel lists)
                sel = [c for c in commands if "extract_candidates_calling" not in c and
                       ("SNV_CANDIDATES_FILES" in c or "INDEL_CANDIDATES_FILES" in c or (c.startswith("ln -sf") and "pileup_tensor_can" in c))]
                ran2 = []
                for c in sel:
                    if c.startswith("ln -sf"):
                        subprocess.run(c, shell=True, check=True)              # a plain shell command of the orchestrator: the glob is the shell's
                        ran2.append(("sh", [c]))
                    else:
                        ran2 += run_step([c], ("concat_files", "create_tensor_pileup_calling", "predict", "call_variants"), w, env)
                rec["step2_argv"] = [[s, [norm(t, tmp, w) for t in a]] for s, a in ran2]
                rec["predict"] = {f: gzip.open(os.path.join(wt, "predict", f), "rt").read() for f in sorted(os.listdir(os.path.join(wt, "predict")))}
                rec["vcf_output"] = folder_files(os.path.join(wt, "vcf_output"), w, tmp)
                import hashlib
                rec["tensor_sha256"] = {}
                for sd in ("pileup_tensor_can_affirmative", "pileup_tensor_can_negational"):
                    for f in sorted(os.listdir(os.path.join(wt, sd))):
                        rec["tensor_sha256"][sd + "/" + f] = hashlib.sha256(gzip.open(os.path.join(wt, sd, f), "rb").read()).hexdigest()
                print("  step 2/6:", len(ran2), "invocations,", len(rec["vcf_output"]), "VCFs")
                # the same call_variants commands with --print_ref_calls's --show_ref, on the same probability files
                rec["vcf_output_show_ref"] = {}
                for s, a in ran2:
                    if s == "call_variants" and name == "ont":
                        a2 = list(a)
                        out = a2[a2.index("--call_fn") + 1]
                        a2[a2.index("--call_fn") + 1] = out.replace("vcf_output", "vcf_output_show_ref")
                        run_ref(s, a2 + ["--show_ref"], env, w)
                if name == "ont":
                    rec["vcf_output_show_ref"] = folder_files(os.path.join(wt, "vcf_output_show_ref"), w, tmp)
                else:
                    del rec["vcf_output_show_ref"]
            if tmp_name != "tmp":
                rec["tmp"] = tmp_name
            if "whole" in name:
                # the whole run: the dry run's commands 1.. in order - STEP 2, sort_vcf, `ln -sf` (STEP 3 without the databases,
                # run_clairs_to:1356-1360), postprocess_vcf, STEP 6, sort_vcf, `ln -sf`, postprocess_vcf - down to <output>/snv.vcf and indel.vcf
                if os.path.exists(os.path.join(wt, "CMD")):
                    rec["work_files"]["CMD"] = norm(open(os.path.join(wt, "CMD")).read(), tmp, w)
                ran2 = []
                for c in commands[1:]:
                    if "clairs_to.py" not in c:
                        subprocess.run(c, shell=True, check=True)
                        ran2.append(("sh", [c]))
                    else:
                        got = run_step([c], TAIL, w, env)
                        assert got and len(got) >= len(invocations(c)), c
                        ran2 += got
                rec["whole_argv"] = [[s, [norm(t, tmp, w) for t in a]] for s, a in ran2]
                rec["predict"] = {f: gzip.open(os.path.join(wt, "predict", f), "rt").read() for f in sorted(os.listdir(os.path.join(wt, "predict")))}
                rec["vcf_output"] = folder_files(os.path.join(wt, "vcf_output"), w, tmp)
                rec["final"] = {f: norm(open(os.path.join(w, f)).read(), tmp, w) for f in sorted(os.listdir(w)) if f.endswith(".vcf")}
                print("  whole run:", len(ran2), "commands,", {f: t.count("\n") for f, t in rec["final"].items()})
            executed[name] = rec
            if name == "ont_hybrid":
                # not a run_clairs_to set-up (hybrid mode switches indel calling off there): the same command lines with indel candidates selected
                # and an indel BED, so that the per-allele form of the hybrid rows and the indel injection (:379-383) are pinned as well
                w2 = os.path.join(tmp, "runs", "ont_hybrid_indel")
                os.makedirs(os.path.join(w2, "tmp", "split_indel_beds"))
                open(os.path.join(w2, "tmp", "split_indel_beds", clisim.CTG), "w").write("%s 100 2000\n%s 3900 5200" % (clisim.CTG, clisim.CTG))
                ran2 = []
                for sub, argv in ran:
                    a2 = [t.replace(w, w2) for t in argv] + ["--select_indel_candidates", "True", "--call_indels_only_in_these_regions",
                                                            os.path.join(w2, "tmp", "split_indel_beds", clisim.CTG)]
                    run_ref(sub, a2, env, w2)
                    ran2.append((sub, a2))
                executed["ont_hybrid_indel"] = dict(work_files=dict(rec["work_files"]), split_indel_beds=folder_files(os.path.join(w2, "tmp", "split_indel_beds"), w2, tmp),
                                                    step1_argv=[[s_, [norm(t, tmp, w2) for t in a]] for s_, a in ran2],
                                                    candidates=folder_files(os.path.join(w2, "tmp", "candidates"), w2, tmp))
                print("  + ont_hybrid_indel:", len(executed["ont_hybrid_indel"]["candidates"]), "files")
        ch = clisim.chunk()
        import hashlib
        from clairs_to_amd.synth import mpileup_text
        dump_json_gz("argv.json.gz", dict(parsers=parsers, runs=runs))
        dump_json_gz("cli_run.json.gz", dict(executed=executed, chunk_kw=clisim.CHUNK_KW,
                                             pileup_sha256=hashlib.sha256(mpileup_text(ch, 0, ctg=clisim.CTG).encode()).hexdigest(),
                                             inputs={k: open(v).read() for k, v in inputs.items() if k != "bam"}))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""A wider pin for the long-read haplotype filter (SURVEY.md 8f #4a): the REFERENCE's src/haplotype_filtering.py, run unmodified
from /root/reference, on 16 simulated contigs (hapsim.simulate with 16 seeds: >= 500 calls, SNV and indel pass) in BOTH of its modes -

  chunk mode      --haplotype_filtering_chunk_mode True (one in-process mpileup per <= 200 calls), and
  per-call mode   the DEFAULT (src/haplotype_filtering.py:1038, 1267): GNU `parallel` starts one `clairs_to.py haplotype_filtering
                  --pos P ...` per call (:804-832), each with its own `samtools mpileup` of pos +- flanking.

Neither samtools nor GNU parallel exists here: `samtools` is the mpileup / faidx shim gen_golden.py uses for hapfilter.json.gz, and
`parallel` is a ten-line stand-in that does what `parallel -C ' ' -j N cmd {1}..{8} :::: file` does (one command per line of the
file, columns split on single blanks).  Inputs are regenerated from the seeds by the test and checked by SHA-256; the output VCFs
are stored.

  hapfilter_wide.json.gz  {"contigs": [{"name", "seed", "inputs_sha256", "snv": {"out_vcf", "same_in_both_modes"}, "indel": {...}}]}

Usage: python tests/golden/gen_hapfilter_wide.py      (from the repo root; build container only)
"""
import gzip
import hashlib
import json
import os
import stat
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
sys.path.insert(0, HERE)
import hapsim  # noqa: E402

SEEDS = list(range(101, 117))
NAMES = ["chr%d" % (i + 1) for i in range(14)] + ["chrX", "contig_16.alt"]

SHIM_SAMTOOLS = r'''#!/usr/bin/env python3
import os, sys
a = sys.argv[1:]
if a[0] == "faidx":
    ref = open(os.environ["FAKE_REF"]).read().strip()
    ctg, rng = a[2].rsplit(":", 1)
    lo, hi = [int(x) for x in rng.split("-")]
    sub = ref[lo - 1:hi]
    sys.stdout.write(">%s\n" % a[2])
    for i in range(0, len(sub), 60):
        sys.stdout.write(sub[i:i + 60] + "\n")
elif a[0] == "mpileup":
    ctg, rng = a[a.index("-r") + 1].rsplit(":", 1)
    lo, hi = [int(x) for x in rng.split("-")]
    bed = None
    if "-l" in a:
        bed = [tuple(int(v) for v in r.split("\t")[1:3]) for r in open(a[a.index("-l") + 1]) if r.strip()]
    for row in open(os.environ["FAKE_MPILEUP_HAP"]):
        p = int(row.split("\t", 2)[1])
        if lo <= p <= hi and (bed is None or any(b < p <= e for b, e in bed)):
            sys.stdout.write(row)
else:
    sys.exit(1)
'''

SHIM_PARALLEL = r'''#!/usr/bin/env python3
# what `parallel -C ' ' -j N <command with {1}..{n}> :::: FILE` does, one job at a time: a command per line, columns split on ' '
import subprocess, sys
a = sys.argv[1:]
assert a[0] == "-C" and a[2] == "-j"
sep = a[1]
cmd = a[4:a.index("::::")]
rc = 0
for line in open(a[a.index("::::") + 1]):
    cols = line.rstrip("\n").split(sep)
    argv = []
    for tok in cmd:
        for i, c in enumerate(cols):
            tok = tok.replace("{%d}" % (i + 1), c)
        argv.append(tok)
    r = subprocess.run(argv, stdout=subprocess.PIPE, universal_newlines=True)
    sys.stdout.write(r.stdout)
    rc = rc or r.returncode
sys.exit(rc)
'''

HEAD = ("##fileformat=VCFv4.2\n##FILTER=<ID=PASS,Description=\"All filters passed\">\n"
        "##FORMAT=<ID=TU,Number=1,Type=Integer,Description=\"Count of T in the tumor BAM\">\n"
        "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tSAMPLE\n")


def inputs_for(sim, ctg, mode, flank=100):
    """(pileup VCF text, nine-column mpileup text) of one pass - the same construction as gen_golden.gen_hapfilter"""
    calls = sim["snv_calls"] if mode == "snv" else sim["indel_calls"]
    positions = sorted({p for c in calls for p in range(max(1, c[0] - flank), c[0] + flank + 1)})
    text = hapsim.pileup_rows(sim, positions, ctg=ctg)
    by_pos = {int(r.split("\t", 2)[1]): r.split("\t") for r in text.split("\n") if r}
    rows = []
    for i, (p, rb, ab) in enumerate(calls):
        cols = by_pos[p]
        toks = cols[7].count(",") + 1
        n_alt = sum(1 for c in cols[4].upper() if c == ab) if (len(rb) == 1 and len(ab) == 1) else cols[4].count("+") + cols[4].count("-")
        af = min(1.0, n_alt / float(toks))
        flt = "PASS" if i % 11 != 10 else "LowQual"
        rows.append("%s\t%d\t.\t%s\t%s\t%.4f\t%s\tFAU=1;FCU=2;FGU=3;FTU=4;RAU=5;RCU=6;RGU=7;RTU=8\tGT:GQ:DP:AF:AD:AU:CU:GU:TU\t0/1:%d:%d:%.4f:%d,%d:1:2:3:4\n"
                    % (ctg, p, rb, ab, 12.5 + i, flt, 12 + i, toks, af, toks - n_alt, n_alt))
    return HEAD + "".join(rows), text


def germline_vcf(sim, ctg):
    return HEAD + "".join("%s\t%d\t.\t%s\t%s\t30.0\tPASS\t.\tGT:GQ\t%s:30\n" % ((ctg,) + g) for g in sim["germline"])


def digest(sim, ctg):
    h = hashlib.sha256()
    h.update(sim["ref"].encode())
    h.update(germline_vcf(sim, ctg).encode())
    for mode in ("snv", "indel"):
        v, t = inputs_for(sim, ctg, mode)
        h.update(v.encode())
        h.update(t.encode())
    return h.hexdigest()


def main():
    out = {"contigs": []}
    n_calls = 0
    with tempfile.TemporaryDirectory() as tmp:
        for seed, ctg in zip(SEEDS, NAMES):
            sim = hapsim.simulate(seed=seed)
            d = os.path.join(tmp, "c%d" % seed)
            os.makedirs(d)
            ref = sim["ref"]
            open(os.path.join(d, "ref.fa"), "w").write(">%s\n%s\n" % (ctg, ref))
            open(os.path.join(d, "ref.fa.fai"), "w").write("%s\t%d\t%d\t%d\t%d\n" % (ctg, len(ref), len(ctg) + 2, len(ref), len(ref) + 1))
            open(os.path.join(d, "ref.txt"), "w").write(ref)
            for name, text in (("samtools", SHIM_SAMTOOLS), ("parallel", SHIM_PARALLEL)):
                fn = os.path.join(d, name)
                open(fn, "w").write(text)
                os.chmod(fn, os.stat(fn).st_mode | stat.S_IEXEC)
            open(os.path.join(d, "fake.bam"), "w").write("")
            germ = os.path.join(d, "germline.vcf")
            open(germ, "w").write(germline_vcf(sim, ctg))
            rec = {"name": ctg, "seed": seed, "inputs_sha256": digest(sim, ctg)}
            for mode in ("snv", "indel"):
                vcf_text, mp_text = inputs_for(sim, ctg, mode)
                pile_vcf, mp = os.path.join(d, "pileup_%s.vcf" % mode), os.path.join(d, "mp_%s.txt" % mode)
                open(pile_vcf, "w").write(vcf_text)
                open(mp, "w").write(mp_text)
                got = {}
                for how in ("chunk", "percall"):
                    out_vcf = os.path.join(d, "out_%s_%s.vcf" % (mode, how))
                    cmd = [sys.executable, os.path.join(REF, "clairs_to.py"), "haplotype_filtering", "--tumor_bam_fn", os.path.join(d, "fake.bam"),
                           "--ref_fn", os.path.join(d, "ref.fa"), "--ctg_name", ctg, "--pileup_vcf_fn", pile_vcf, "--germline_vcf_fn", germ,
                           "--output_vcf_fn", out_vcf, "--output_dir", os.path.join(d, "work_%s_%s" % (mode, how)), "--samtools", os.path.join(d, "samtools"),
                           "--threads", "1", "--parallel", os.path.join(d, "parallel"), "--pypy3", sys.executable]
                    if how == "chunk":
                        cmd += ["--haplotype_filtering_chunk_mode", "True"]
                    if mode == "indel":
                        cmd.append("--is_indel")
                    res = subprocess.run(cmd, cwd=d, env=dict(os.environ, PYTHONPATH=REF, FAKE_REF=os.path.join(d, "ref.txt"), FAKE_MPILEUP_HAP=mp,
                                                              PYTHONHASHSEED="0"), capture_output=True, text=True)
                    assert res.returncode == 0, res.stderr[-2000:]
                    got[how] = open(out_vcf).read()
                rec[mode] = {"out_vcf": got["percall"], "same_in_both_modes": got["chunk"] == got["percall"]}
                if got["chunk"] != got["percall"]:
                    rec[mode]["out_vcf_chunk_mode"] = got["chunk"]
                n_calls += sum(1 for r in vcf_text.split("\n") if r and r[0] != "#")
            out["contigs"].append(rec)
            print(ctg, "seed", seed, "snv same:", rec["snv"]["same_in_both_modes"], "indel same:", rec["indel"]["same_in_both_modes"], flush=True)
    raw = json.dumps(out, separators=(",", ":")).encode()
    with open(os.path.join(HERE, "hapfilter_wide.json.gz"), "wb") as f:
        with gzip.GzipFile(fileobj=f, mode="wb", mtime=0) as g:
            g.write(raw)
    print("wrote hapfilter_wide.json.gz: %d contigs, %d calls, %d bytes raw" % (len(out["contigs"]), n_calls, len(raw)))


if __name__ == "__main__":
    main()

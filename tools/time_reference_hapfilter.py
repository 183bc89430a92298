#!/usr/bin/env python3
"""The genuine reference's long-read haplotype filter, timed in the BUILD container (it cannot travel to the GPU box): src/haplotype_filtering.py
run unmodified from /root/reference through `clairs_to.py haplotype_filtering`, chunk mode (one in-process mpileup per <= 200 calls - its
fastest form here) and the default per-call mode (one process + one mpileup per call), `--threads 1` and `--threads N`, on the simulated
contigs of tests/golden/hapsim.py (the generator of hapfilter_wide.json.gz and of bench.py's `hapfilter` leg), with the same `samtools` /
`parallel` stand-ins gen_hapfilter_wide.py uses (mpileup text pre-made: BAM decoding is NOT in these figures).
Writes profiles/reference_hapfilter_timing.json; bench.py quotes it next to its own calls/s.  Usage: python tools/time_reference_hapfilter.py"""
import json
import os
import stat
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import gen_hapfilter_wide as gw  # noqa: E402
import hapsim  # noqa: E402

REF = "/root/reference"


def main():
    n_contigs = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    cpus = len(os.sched_getaffinity(0))
    runs = {}
    with tempfile.TemporaryDirectory() as tmp:
        jobs = []
        for k in range(n_contigs):
            seed, ctg = 1000 + k, "chr%d" % (k + 1)
            sim = hapsim.simulate(seed=seed)
            d = os.path.join(tmp, "c%d" % seed)
            os.makedirs(d)
            ref = sim["ref"]
            open(os.path.join(d, "ref.fa"), "w").write(">%s\n%s\n" % (ctg, ref))
            open(os.path.join(d, "ref.fa.fai"), "w").write("%s\t%d\t%d\t%d\t%d\n" % (ctg, len(ref), len(ctg) + 2, len(ref), len(ref) + 1))
            open(os.path.join(d, "ref.txt"), "w").write(ref)
            for name, text in (("samtools", gw.SHIM_SAMTOOLS), ("parallel", gw.SHIM_PARALLEL)):
                fn = os.path.join(d, name)
                open(fn, "w").write(text)
                os.chmod(fn, os.stat(fn).st_mode | stat.S_IEXEC)
            open(os.path.join(d, "fake.bam"), "w").write("")
            open(os.path.join(d, "germline.vcf"), "w").write(gw.germline_vcf(sim, ctg))
            for mode in ("snv", "indel"):
                v, t = gw.inputs_for(sim, ctg, mode)
                open(os.path.join(d, "pileup_%s.vcf" % mode), "w").write(v)
                open(os.path.join(d, "mp_%s.txt" % mode), "w").write(t)
                jobs.append((d, ctg, mode, sum(1 for r in v.split("\n") if r and r[0] != "#" and "\tPASS\t" in r)))
        for how, threads in (("chunk", 1), ("chunk", cpus), ("percall", 1)):
            per_mode = {}
            for mode in ("snv", "indel"):
                t_sum, calls = 0.0, 0
                for d, ctg, m, n in jobs:
                    if m != mode:
                        continue
                    out_vcf = os.path.join(d, "out_%s_%s_%d.vcf" % (mode, how, threads))
                    cmd = [sys.executable, os.path.join(REF, "clairs_to.py"), "haplotype_filtering", "--tumor_bam_fn", os.path.join(d, "fake.bam"),
                           "--ref_fn", os.path.join(d, "ref.fa"), "--ctg_name", ctg, "--pileup_vcf_fn", os.path.join(d, "pileup_%s.vcf" % mode),
                           "--germline_vcf_fn", os.path.join(d, "germline.vcf"), "--output_vcf_fn", out_vcf,
                           "--output_dir", os.path.join(d, "work_%s_%s_%d" % (mode, how, threads)), "--samtools", os.path.join(d, "samtools"),
                           "--threads", str(threads), "--parallel", os.path.join(d, "parallel"), "--pypy3", sys.executable]
                    if how == "chunk":
                        cmd += ["--haplotype_filtering_chunk_mode", "True"]
                    if mode == "indel":
                        cmd.append("--is_indel")
                    env = dict(os.environ, PYTHONPATH=REF, FAKE_REF=os.path.join(d, "ref.txt"), FAKE_MPILEUP_HAP=os.path.join(d, "mp_%s.txt" % mode), PYTHONHASHSEED="0")
                    t0 = time.perf_counter()
                    res = subprocess.run(cmd, cwd=d, env=env, capture_output=True, text=True)
                    t_sum += time.perf_counter() - t0
                    assert res.returncode == 0, res.stderr[-2000:]
                    calls += n
                per_mode[mode] = {"calls": calls, "seconds": round(t_sum, 3), "calls_per_s": round(calls / t_sum, 2), "ms_per_call": round(t_sum / calls * 1e3, 2)}
            runs["%s_mode_threads_%d" % (how, threads)] = per_mode
            print(how, threads, per_mode, flush=True)
    out = {"what": "HKU-BAL/ClairS-TO v0.4.4 src/haplotype_filtering.py run from /root/reference (CPython; `samtools mpileup` replaced by a stand-in that prints pre-made "
                   "nine-column text: BAM decoding excluded; one process start per contig job included, as in a real run)",
           "host_cpus": cpus, "contigs": n_contigs, "runs": runs,
           "note": "PASS calls of the pileup VCF are what the filter evaluates; per-call mode starts one interpreter per call through the `parallel` stand-in (serially: an upper bound of its cost)"}
    with open(os.path.join(ROOT, "profiles", "reference_hapfilter_timing.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("wrote profiles/reference_hapfilter_timing.json")


if __name__ == "__main__":
    main()

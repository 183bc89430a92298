#!/usr/bin/env python3
"""Golden vectors for the Illumina realigner (SURVEY.md 8f #4b) from the REFERENCE's own native code.

`make -C oracle ref` compiles /root/reference/src/realign/{realigner.cpp,ssw_cpp.cpp,ssw.c} into oracle/_ref/librealigner_ref.so
(plain g++, outputs only); this script drives its C ABI `realign_reads` (src/realign/realigner.cpp:860-865) on synthetic windows
(tests/realignutil.gen_window: fast-pass hits, <= 2 mismatches, Smith-Waterman fallbacks with substitutions / indels / clipped
ends, repeats and two-letter sequences for tie-breaking, N bases, 1..18 haplotypes, reads of 20..250 bases) and stores what it
returned.  Inputs are regenerated from the seed by the test and checked by SHA-256; outputs (positions, CIGARs) are stored.

  realign.json.gz   {"seed", "n_windows", "inputs_sha256", "windows": [[[pos - ref_start, cigar], ...], ...]}

`flow` target: the reference's Python around that native code, run unmodified from /root/reference -
  clairs_to.py realign_variants  (src/realign_variants.py: per low-QUAL PASS call `samtools mpileup`, a child `clairs_to.py
  realign_reads --pos P` piped into `samtools mpileup -`, the demotion rule, the output VCF) and, per position,
  clairs_to.py realign_reads --pos P  on its own (src/realign_reads.py: the realigned SAM text)
on the simulated short-read data of realignsim.py.  `samtools` is realignsim.SHIM (neither box has samtools); the reference finds
its two ctypes modules through its own fall-back look-up (src/realign_reads.py:56-65: <dirname(dirname(`which python`))>/bin/
preprocess/realign/{realigner,debruijn_graph}), which a scratch directory on PATH points at oracle/_ref/librealigner_ref.so (the
reference's realigner, compiled from its sources) and at clairs_to_amd/realign/debruijn_graph.so - the BUILD's consensus behind the
reference's `get_consensus` ABI, because the reference's own needs Boost.Graph and cannot be built here.  So in this fixture the
consensus strings are an INPUT (recorded per window), everything else is the reference's output.

  realign_flow.json.gz  {"seed", "inputs_sha256", "vcf": output VCF text, "vcf_indel": the same with --is_indel, "positions": {pos: {"sha256", "n_rows", "moved": [[name, pos,
                         cigar], ...]}}, "consensus": [[window reference, [haplotypes]], ...] in call order}

Usage: make -C oracle ref && python tests/golden/gen_realign.py [windows|flow]     (from the repo root; build container only)
"""
import gzip
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import realignutil as ru  # noqa: E402

SEED, N_WINDOWS = 20260928, 640


def window_digest(h, w):
    h.update(json.dumps([w["seqs"], w["positions"], w["cigars"], w["reference"], w["haplotypes"], w["ref_start"], w["ref_prefix"],
                         w["ref_suffix"]], separators=(",", ":")).encode())


def main():
    assert ru.ref_lib() is not None, "run `make -C oracle ref` first"
    rng = np.random.default_rng(SEED)
    h = hashlib.sha256()
    out = []
    moved = reads = 0
    for _ in range(N_WINDOWS):
        w = ru.gen_window(rng)
        window_digest(h, w)
        pos, cig = ru.ref_realign(w)
        out.append([[p - w["ref_start"], c] for p, c in zip(pos, cig)])
        reads += len(cig)
        moved += sum(1 for c, c0 in zip(cig, w["cigars"]) if c != c0)
    raw = json.dumps({"seed": SEED, "n_windows": N_WINDOWS, "inputs_sha256": h.hexdigest(), "windows": out}, separators=(",", ":")).encode()
    with open(os.path.join(HERE, "realign.json.gz"), "wb") as f:
        with gzip.GzipFile(fileobj=f, mode="wb", mtime=0) as g:
            g.write(raw)
    print("wrote realign.json.gz: %d windows, %d reads, %d realigned, %d bytes raw" % (N_WINDOWS, reads, moved, len(raw)))


def flow_inputs_digest(paths):
    h = hashlib.sha256()
    for k in ("ref", "vcf"):
        h.update(open(paths[k], "rb").read())
    h.update(open(paths["bam"] + ".sam", "rb").read())
    return h.hexdigest()


def gen_flow():
    import shutil
    import subprocess
    import tempfile
    sys.path.insert(0, HERE)
    import realignsim
    REF = "/root/reference"
    assert ru.ref_lib() is not None and os.path.isdir(REF)
    sim = realignsim.simulate()
    tmp = tempfile.mkdtemp(prefix="realign_flow_")
    try:
        paths = realignsim.write_inputs(sim, tmp)
        mods = os.path.join(tmp, "conda", "bin", "preprocess", "realign")
        os.makedirs(mods)
        os.symlink(sys.executable, os.path.join(tmp, "conda", "bin", "python"))
        shutil.copy(ru.REF_SO, os.path.join(mods, "realigner"))
        shutil.copy(os.path.join(ROOT, "clairs_to_amd", "realign", "debruijn_graph.so"), os.path.join(mods, "debruijn_graph"))
        env = dict(os.environ, PATH=os.path.join(tmp, "conda", "bin") + ":" + os.environ["PATH"], REALIGNSIM_TESTS=os.path.join(ROOT, "tests"),
                   LD_LIBRARY_PATH=os.path.join(ROOT, "clairs_to_amd") + ":" + os.environ.get("LD_LIBRARY_PATH", ""),
                   CTO_DBG_LOG=os.path.join(tmp, "dbg.log"))
        out_vcf = os.path.join(tmp, "out", "realigned.vcf")
        subprocess.run([sys.executable, os.path.join(REF, "clairs_to.py"), "realign_variants", "--bam_fn", paths["bam"], "--ref_fn", paths["ref"],
                        "--ctg_name", realignsim.CTG, "--pileup_vcf_fn", paths["vcf"], "--output_vcf_fn", out_vcf, "--samtools", paths["samtools"],
                        "--python", sys.executable, "--threads", "8"], check=True, env=env, cwd=tmp)
        vcf = open(out_vcf).read()
        # the indel pass (--is_indel: indel records are kept; a deletion's ALT is its bare anchor base, so the reference judges such a record by
        # how many reads show that letter at the anchor - its behaviour, reproduced as it is; an insertion's ALT never equals an allele string)
        out_vcf_i = os.path.join(tmp, "out", "realigned_indel.vcf")
        subprocess.run([sys.executable, os.path.join(REF, "clairs_to.py"), "realign_variants", "--bam_fn", paths["bam"], "--ref_fn", paths["ref"],
                        "--ctg_name", realignsim.CTG, "--pileup_vcf_fn", paths["vcf"], "--output_vcf_fn", out_vcf_i, "--samtools", paths["samtools"],
                        "--python", sys.executable, "--threads", "8", "--is_indel"], check=True, env=env, cwd=tmp)
        vcf_indel = open(out_vcf_i).read()
        todo = []
        for row in open(paths["vcf"]):
            c = row.split("\t")
            if row[0] != "#" and c[6] == "PASS" and float(c[5]) < 8 and len(c[3]) == 1 and len(c[4]) == 1:
                todo.append(int(c[1]))
        positions, consensus = {}, []
        orig = {}
        for r in realignsim.parse_sam(realignsim.sam_text(sim)):
            orig[(r["name"] + "_" + str(int(bool(r["flag"] & 16))), r["flag"])] = (r["pos"], "".join("%d%s" % (n, o) for o, n in r["cigar"]))
        for pos in todo:
            log = os.path.join(tmp, "dbg_%d.log" % pos)
            p = subprocess.run([sys.executable, os.path.join(REF, "clairs_to.py"), "realign_reads", "--pos", str(pos), "--ctg_name", realignsim.CTG,
                                "--bam_fn", paths["bam"], "--ref_fn", paths["ref"], "--samtools", paths["samtools"]], check=True,
                               env=dict(env, CTO_DBG_LOG=log), cwd=tmp, stdout=subprocess.PIPE, universal_newlines=True)
            rows = [r for r in p.stdout.split("\n") if r and r[0] != "@"]
            moved = []
            for r in rows:
                c = r.split("\t")
                if orig[(c[0], int(c[1]))] != (int(c[3]) - 1, c[5]):
                    moved.append([c[0], int(c[3]), c[5]])
            positions[str(pos)] = {"sha256": hashlib.sha256(p.stdout.encode()).hexdigest(), "n_rows": len(rows), "moved": moved}
            if os.path.exists(log):
                for ln in open(log):
                    ref_w, haps = ln.rstrip("\n").split("\t")
                    consensus.append([pos, ref_w, [h for h in haps.split(",") if h]])
        # the same entry point over a region longer than its 5 000-base chunk (--realign_flanking_window 3000: 8 200 bases of reads), which
        # takes the chunk-boundary path of src/realign_reads.py:283-286, 618-634 that `realign_variants` (window 100) never reaches
        wide = {}
        inner = [q for q in todo if q > 4200]            # the read region must not start below 1
        for pos in (inner[0], inner[len(inner) // 2], inner[-1]):
            p = subprocess.run([sys.executable, os.path.join(REF, "clairs_to.py"), "realign_reads", "--pos", str(pos), "--ctg_name", realignsim.CTG,
                                "--bam_fn", paths["bam"], "--ref_fn", paths["ref"], "--samtools", paths["samtools"], "--realign_flanking_window", "3000"],
                               check=True, env=env, cwd=tmp, stdout=subprocess.PIPE, universal_newlines=True)
            rows = [r for r in p.stdout.split("\n") if r and r[0] != "@"]
            wide[str(pos)] = {"sha256": hashlib.sha256(p.stdout.encode()).hexdigest(), "n_rows": len(rows),
                              "n_moved": sum(1 for r in rows if orig[(r.split("\t")[0], int(r.split("\t")[1]))] != (int(r.split("\t")[3]) - 1, r.split("\t")[5]))}
        obj = {"seed": 20260929, "inputs_sha256": flow_inputs_digest(paths), "vcf": vcf, "wide_window": wide, "vcf_indel": vcf_indel, "positions": positions, "consensus": consensus}
        raw = json.dumps(obj, separators=(",", ":")).encode()
        with open(os.path.join(HERE, "realign_flow.json.gz"), "wb") as f:
            with gzip.GzipFile(fileobj=f, mode="wb", mtime=0) as g:
                g.write(raw)
        print("wrote realign_flow.json.gz: %d calls re-examined, %d demoted, %d reads moved in total, %d consensus calls, %d bytes raw" % (
            len(todo), vcf.count("LowQual;Realignment"), sum(len(v["moved"]) for v in positions.values()), len(consensus), len(raw)))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    if which in ("all", "windows"):
        main()
    if which in ("all", "flow"):
        gen_flow()

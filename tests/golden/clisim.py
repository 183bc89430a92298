"""Synthetic inputs of a whole `run_clairs_to` STEP 1 / STEP 2 / STEP 6 run for the command-line fixtures (`gen_cli.py`, `tests/test_cli_argv.py`,
`tests/test_gpu_cli_argv.py`): one contig `chr20` with an ONT-like pileup (a `SynthChunk`, regenerated from its seed on either side), the
`samtools` stand-in both the reference and this package are pointed at (neither box has samtools; it prints what `samtools mpileup` prints for the
simulated BAM, honouring `-r`, `-l`, `--min-MQ`, `--min-BQ`, `--output-MQ`), the confident BED, the indel BED and the hybrid / genotyping VCF.
Input synthesis for the tests and for gen_cli.py only; the product never imports this."""
import os
import stat
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))

CTG, CTG2 = "chr20", "chr21"          # chr21 is in the .fai but has no reads: run_clairs_to drops it after `samtools idxstats`
CHUNK_SIZE = 3200                     # run_clairs_to --chunk_size: two chunks on chr20
CHUNK_KW = dict(n_sites=150, seed=20260929, start=150, spacing=40, depth_mean=14.0, p_mismatch=0.03, p_ins=0.03, p_del=0.04, n_rate=0.02)


def chunk():
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    from clairs_to_amd.synth import SynthChunk
    return SynthChunk(**CHUNK_KW)


def contig(ch):
    """(sequence of chr20 from position 1, its length): the chunk's reference window, 'A' before it."""
    ref, lo = ch.ref_window()
    seq = "A" * (lo - 1) + ref
    return seq, len(seq)


def bed_rows(path):
    out = []
    for row in open(path):
        c = row.split()
        if len(c) >= 3 and row[0] != "#":
            out.append((c[0], int(c[1]), int(c[2])))
    return out


def samtools_main(argv):
    """The stand-in: `--version`, `idxstats`, `faidx <fa> ctg:s-e ...`, `mpileup ...` (rows of the simulated pileup: a position p is printed when it
    lies in -r's [s, e] and, with -l, in a BED interval  start < p <= end  - htslib's bed_overlap(pos-1, pos))."""
    ch = chunk()
    seq, L = contig(ch)
    out = sys.stdout
    if argv[0] == "--version":
        out.write("samtools 1.17\nUsing htslib 1.17\n")
    elif argv[0] == "idxstats":
        out.write("%s\t%d\t%d\t0\n%s\t500\t0\t0\n*\t0\t0\t0\n" % (CTG, L, ch.col_off[-1] // 100, CTG2))
    elif argv[0] == "faidx":
        for reg in argv[2:]:
            name, _, rng = reg.partition(":")
            s, e = (int(v) for v in rng.split("-")) if rng else (1, L)
            e = min(e, L)
            sub = seq[s - 1:e] if name == CTG else ""
            out.write(">%s\n" % reg)
            for i in range(0, len(sub), 60):
                out.write(sub[i:i + 60] + "\n")
    elif argv[0] == "mpileup":
        from clairs_to_amd.synth import mpileup_text
        opt = lambda k, d=None: argv[argv.index(k) + 1] if k in argv else d
        name, _, rng = opt("-r").partition(":")
        if name != CTG:
            return 0
        s, e = (int(v) for v in rng.split("-")) if rng else (1, L)
        pos = ch.col_pos.astype(np.int64)
        c0, c1 = int(np.searchsorted(pos, s)), int(np.searchsorted(pos, e, side="right"))
        text = mpileup_text(ch, min_bq=int(opt("--min-BQ", 13)), ctg=CTG, col_range=(c0, c1), min_mq=int(opt("--min-MQ", 0)),
                            with_mq="--output-MQ" in argv)
        bed = opt("-l")
        if bed is not None:
            iv = [(a, b) for c, a, b in bed_rows(bed) if c == CTG]
            keep = []
            for row in text.split("\n"):
                if row:
                    p = int(row.split("\t", 2)[1])
                    if any(a < p <= b for a, b in iv):
                        keep.append(row)
            text = "\n".join(keep) + ("\n" if keep else "")
        out.write(text if text.strip() else "")
    else:
        return 1
    return 0


SHIM = "#!/bin/sh\nexec %s %s samtools \"$@\"\n"


def write_shims(bin_dir, python=sys.executable):
    """bin_dir/samtools (this module's stand-in) plus version-only stand-ins of the tools run_clairs_to probes before a dry run."""
    os.makedirs(bin_dir, exist_ok=True)
    progs = {
        "samtools": SHIM % (python, os.path.abspath(__file__)),
        "parallel": "#!/bin/sh\necho 'GNU parallel 20230722'\n",
        "pypy3": "#!/bin/sh\nif [ \"$1\" = \"--version\" ] || [ $# -eq 0 ]; then echo 'Python 3.9.18 (x, Jan 01 2024)'; echo '[PyPy 7.3.15 with GCC 10.2.1]'; "
                 "else exec %s \"$@\"; fi\n" % python,
        "whatshap": "#!/bin/sh\necho 2.0\n",
        "longphase": "#!/bin/sh\necho 'longphase 1.7'\n",
    }
    for name, body in progs.items():
        p = os.path.join(bin_dir, name)
        with open(p, "w") as f:
            f.write(body)
        os.chmod(p, os.stat(p).st_mode | stat.S_IEXEC | stat.S_IXGRP | stat.S_IXOTH)


def write_inputs(w):
    """ref.fa(.fai), t.bam(.bai) placeholders, conf.bed, indel.bed, hybrid.vcf under `w`; -> dict of paths."""
    ch = chunk()
    seq, L = contig(ch)
    os.makedirs(w, exist_ok=True)
    p = {k: os.path.join(w, v) for k, v in dict(ref="ref.fa", bam="t.bam", bed="conf.bed", indel_bed="indel.bed", vcf="known.vcf").items()}
    with open(p["ref"], "w") as f:
        f.write(">%s\n" % CTG + "\n".join(seq[i:i + 60] for i in range(0, L, 60)) + "\n>%s\n" % CTG2 + "ACGT" * 125 + "\n")
    off2 = len(">%s\n" % CTG) + L + (L + 59) // 60 + len(">%s\n" % CTG2)
    with open(p["ref"] + ".fai", "w") as f:
        f.write("%s\t%d\t%d\t60\t61\n%s\t500\t%d\t500\t501\n" % (CTG, L, len(CTG) + 2, CTG2, off2))
    for k in ("bam",):
        open(p[k], "w").close()
        open(p[k] + ".bai", "w").close()
    with open(p["bed"], "w") as f:           # confident regions: three stretches, a zero-length row, another contig, a comment
        f.write("#track\n%s\t300\t1200\n%s\t2500\t4100\n%s\t4600\t4600\n%s\t5000\t5600\n%s\t10\t90\n" % (CTG, CTG, CTG, CTG, CTG2))
    with open(p["indel_bed"], "w") as f:
        f.write("%s\t100\t2000\n%s\t3900\t5200\n" % (CTG, CTG))
    rng = np.random.default_rng(77)
    pos = ch.col_pos.astype(np.int64)
    # 60 random columns (most fail the AF gates), every 10th planted site (most pass them: their rows are printed as fractions), and three
    # positions without a pileup row
    picks = sorted(set(int(v) for v in rng.choice(pos[(pos > 60)], size=60, replace=False)) | set(int(v) for v in ch.site_pos[::10]) |
                   {int(pos[-1]) + 40, 48, 20})
    with open(p["vcf"], "w") as f:
        f.write("##fileformat=VCFv4.2\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tSAMPLE\n")
        for i, x in enumerate(picks):
            rb = seq[x - 1] if seq[x - 1] in "ACGT" else "A"
            ab = "ACGT"[("ACGT".index(rb) + 1 + i % 3) % 4]
            if i % 7 == 3:
                rb, ab = rb + "GT", rb                    # a deletion record
            elif i % 7 == 5:
                ab = rb + "CA"                            # an insertion record
            elif i % 11 == 4:
                ab = ab + "," + rb + "T"                  # two ALT alleles: the first decides
            gt = ("0/1", "1/1", "0/0", "./.")[i % 4]
            f.write("%s\t%d\t.\t%s\t%s\t30\tPASS\t.\tGT\t%s\n" % (CTG, x, rb, ab, gt))
        f.write("%s\t50\t.\tA\tC\t30\tPASS\t.\tGT\t0/1\n" % CTG2)
    return p


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "samtools":
        sys.exit(samtools_main(sys.argv[2:]))

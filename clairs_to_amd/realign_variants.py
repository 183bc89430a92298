"""Drop-in counterpart of `clairs_to.py realign_variants` (reference: src/realign_variants.py; STEP 4-1 / 8-1 of run_clairs_to for
Illumina input, run_clairs_to:1450-1481, :1705-1736; SURVEY.md 8f #4b): every PASS call below the platform's QUAL bar is looked at
again after a local realignment of the reads around it, and demoted to `LowQual;Realignment` (QUAL 0.0000) when the realigned
pileup supports the alternative allele with fewer reads AND a smaller fraction than the original pileup did.

Same inputs, options and output VCF as the reference.  What differs is how the work is done: the reference starts, per call, a
`samtools mpileup`, a second Python interpreter running `realign_reads`, and a second `samtools mpileup` fed by it, under a process
pool; here the realignment runs in-process (realign_reads.realign_region over the library's consensus and realigner) on a thread
pool, and only the two samtools commands remain subprocesses (the text they exchange is the reference's).
"""
import os
import subprocess
import sys
from argparse import ArgumentParser, SUPPRESS
from collections import Counter
from concurrent.futures import ThreadPoolExecutor
from io import StringIO

from .haplotype_filtering import read_vcf, header_up_to_last_format, str2bool
from . import realign_reads as rr

QUAL_BAR = 8            # shared/param.py:47  qual_dict['ilmn']
EXCL_FLAGS = 2316       # realign_variants.py:76, :99


def column_alleles(bases):
    """get_base_list (realign_variants.py:31-56): the read entries of an mpileup base string, upper-cased with their indel
    attached ("A", "T+2AC", "*", "#"); `$` and the character after `^` are skipped, everything unknown is skipped"""
    out, i, n = [], 0, len(bases)
    while i < n:
        ch = bases[i]
        if ch == "+" or ch == "-":
            i += 1
            ln = 0
            while bases[i].isdigit():
                ln = ln * 10 + int(bases[i])
                i += 1
            out[-1] = out[-1] + ch + bases[i:i + ln]
            i += ln - 1
        elif ch in "ACGTNacgtn#*":
            out.append(ch)
        elif ch == "^":
            i += 1
        i += 1
    return [a.upper() for a in out]


def decide(raw_bases, realigned_bases, alt):
    """realign_variants.py:112-123 -> (passes, (raw support, raw depth, realigned support, realigned depth))"""
    raw, new = column_alleles(raw_bases), column_alleles(realigned_bases)
    rs, ns = Counter(raw)[alt], Counter(new)[alt]
    fails = rs / float(len(raw)) > ns / len(new) and ns < rs
    return not fails, (rs, len(raw), ns, len(new))


def _mpileup(args, source, region_or_none, stdin_text=None, reverse_del=False):
    cmd = [args.samtools, "mpileup", source] + (["--reverse-del"] if reverse_del else []) + \
          ["--min-MQ", str(args.min_mq), "--min-BQ", str(args.min_bq), "--excl-flags", str(EXCL_FLAGS)] + (["-r", region_or_none] if region_or_none else [])
    return subprocess.run(cmd, input=stdin_text, stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True).stdout


def word_match(line, word):
    """`grep -w`: the number as a whole word somewhere in the line"""
    import re
    return re.search(r"(?<![A-Za-z0-9_])%s(?![A-Za-z0-9_])" % re.escape(word), line) is not None


def evaluate_call(args, rec):
    """extract_base (realign_variants.py:59-123) for one call -> (ctg, pos, passes, counts)"""
    ctg, pos = (args.ctg_name if args.ctg_name is not None else rec["ctg"]), rec["pos"]
    try:
        qual = float(rec["qual"]) if rec["qual"] is not None else None
    except ValueError:
        qual = None
    if qual is not None and qual >= QUAL_BAR:
        return ctg, pos, True, (-1, -1, -1, -1)
    cols = _mpileup(args, args.bam_fn, "{}:{}-{}".format(ctg, pos, pos)).rstrip().split("\t")
    if len(cols) < 4:
        return ctg, pos, True, (-1, -1, -1, -1)
    # the reference's inner command: realign_reads --pos P | samtools mpileup - --reverse-del ... | grep -w P
    sam = StringIO()
    inner = ArgumentParser()
    ns = inner.parse_args([])
    ns.pos, ns.ctg_name, ns.bam_fn, ns.ref_fn, ns.samtools = pos, ctg, args.bam_fn, args.ref_fn, args.samtools
    ns.min_mq, ns.min_coverage, ns.realign_flanking_window, ns.max_distance = 20, 2.0, 100, 50        # realign_reads' own defaults (:690-711)
    rr.reads_realignment(ns, out=sam)
    text = _mpileup(args, "-", None, stdin_text=sam.getvalue(), reverse_del=True)
    hits = [ln for ln in text.split("\n") if word_match(ln, str(pos))]
    new_cols = "\n".join(hits).rstrip().split("\t")
    if len(new_cols) < 4:
        return ctg, pos, True, (-1, -1, -1, -1)
    ok, counts = decide(cols[4], new_cols[4], rec["alt"])
    return ctg, pos, ok, counts


def realign_variants(args):
    if not args.enable_realignment:
        if os.path.lexists(args.output_vcf_fn):
            os.remove(args.output_vcf_fn)
        os.symlink(args.pileup_vcf_fn, args.output_vcf_fn)
        return
    header, calls = read_vcf(args.pileup_vcf_fn, args.ctg_name, show_ref=args.show_ref, discard_indel=not args.is_indel)
    todo = [r for r in calls.values() if r["filter"] == "PASS"]
    threads = max(1, int(args.threads * 4 / 5))
    failed, done = set(), 0
    with ThreadPoolExecutor(max_workers=threads) as ex:          # samtools children and the C calls run outside the GIL
        for ctg, pos, ok, _ in ex.map(lambda r: evaluate_call(args, r), todo):
            if not ok:
                failed.add((ctg, pos))
            done += 1
            if done % 1000 == 0:
                print("[INFO] Processing in {}, total processed positions: {}".format(ctg, done), flush=True)
    out_header = header_up_to_last_format(header)
    fai = args.ref_fn + ".fai" if os.path.exists(args.ref_fn + ".fai") else ".".join(args.ref_fn.split(".")[:-1]) + ".fai"
    names = None if args.ctg_name is None else args.ctg_name.split(",")
    for row in open(fai):
        c = row.strip().split("\t")
        if names is None or c[0] in names:
            out_header += "##contig=<ID=%s,length=%s>\n" % (c[0], c[1])
    out_header += "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tSAMPLE\n"
    os.makedirs(os.path.dirname(os.path.abspath(args.output_vcf_fn)), exist_ok=True)
    with open(args.output_vcf_fn, "w") as out:
        out.write(out_header)
        for key, rec in calls.items():                         # file order, as the reference's dict keeps it
            ctg = args.ctg_name if args.ctg_name is not None else rec["ctg"]
            cols = rec["row"].rstrip().split("\t")
            if (ctg, rec["pos"]) in failed:
                cols[5], cols[6] = "0.0000", "LowQual;Realignment"
            out.write("\t".join(cols) + "\n")
    print("[INFO] Total input calls: {}, filtered by realignment: {}".format(len(todo), len(failed)), flush=True)
    return failed


def main():
    p = ArgumentParser(description="Reads realignment workflow for all input variants")
    p.add_argument("--bam_fn", type=str, default=None)
    p.add_argument("--ref_fn", type=str, default="ref.fa")
    p.add_argument("--ctg_name", type=str, default=None)
    p.add_argument("--pileup_vcf_fn", type=str, default=None)
    p.add_argument("--output_dir", type=str, default=None)
    p.add_argument("--output_vcf_fn", type=str, default=None)
    p.add_argument("--samtools", type=str, default="samtools")
    p.add_argument("--threads", type=int, default=1)
    p.add_argument("--python", type=str, default="python3", help="accepted for compatibility: the realignment runs in-process")
    p.add_argument("--show_ref", action="store_true")
    p.add_argument("--min_mq", type=int, default=20)             # shared/param.py:17
    p.add_argument("--min_bq", type=int, default=0)              # shared/param.py:19
    p.add_argument("--enable_realignment", type=str2bool, default=True)
    p.add_argument("--qual", type=float, default=None, help="accepted for compatibility (unused by the reference too)")
    p.add_argument("--pos", type=int, default=None, help=SUPPRESS)
    p.add_argument("--is_indel", action="store_true", help=SUPPRESS)
    if len(sys.argv[1:]) == 0:
        p.print_help()
        sys.exit(1)
    realign_variants(p.parse_args())


if __name__ == "__main__":
    main()

"""Drop-in counterpart of `clairs_to.py realign_variants` (reference: src/realign_variants.py; STEP 4-1 / 8-1 of run_clairs_to for
Illumina input, run_clairs_to:1450-1481, :1705-1736; SURVEY.md 8f #4b): every PASS call below the platform's QUAL bar is looked at
again after a local realignment of the reads around it, and demoted to `LowQual;Realignment` (QUAL 0.0000) when the realigned
pileup supports the alternative allele with fewer reads AND a smaller fraction than the original pileup did.

Same inputs, options and output VCF as the reference.  What differs is how the work is done: the reference starts, per call, a
`samtools mpileup`, a second Python interpreter running `realign_reads`, and a second `samtools mpileup` fed by it, under a process
pool; here the realignment runs in-process (realign_reads.realign_region over the library's consensus and realigner) on a thread
pool, and only the two samtools commands remain subprocesses (the text they exchange is the reference's).

Threading model: `--threads` workers (4/5 of it, as in the reference).  The samtools children and the C calls (consensus, realigner,
read-evidence scan) run outside the GIL; the row parsing and window bookkeeping between them is Python and holds it, so a thread
pool levels off at a few cores' worth of work.  `--pool process` runs the same calls in worker PROCESSES (spawned, one interpreter
each - what the reference's ProcessPoolExecutor does) and scales with the cores; the output VCF is the same either way.  The
samtools children's stderr goes to this process's stderr (the reference captures it per call).
"""
import os
import subprocess
import sys
from argparse import ArgumentParser, SUPPRESS
from collections import Counter
from concurrent.futures import ProcessPoolExecutor, ThreadPoolExecutor
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
    """realign_variants.py:112-123 -> (passes, (raw support, raw depth, realigned support, realigned depth)); the columns as
    mpileup base strings or as allele lists"""
    raw = column_alleles(raw_bases) if isinstance(raw_bases, str) else raw_bases
    new = column_alleles(realigned_bases) if isinstance(realigned_bases, str) else realigned_bases
    rs, ns = Counter(raw)[alt], Counter(new)[alt]
    fails = rs / float(len(raw)) > ns / len(new) and ns < rs
    return not fails, (rs, len(raw), ns, len(new))


# ---- the two pileup columns without samtools (--bam_reader native; PARITY UNPINNED, rules of csrc/bam.cpp's header) ----
def bam_column_alleles(args, ctg, pos):
    """the raw column of `samtools mpileup bam --min-MQ --min-BQ --excl-flags 2316 -r ctg:pos-pos` as upper-cased alleles"""
    from .fasta import read_region
    from .pack import ColumnPack
    ref_lo = max(1, pos - 100)
    ref = read_region(args.ref_fn, ctg, ref_lo, pos + 100)
    pack = ColumnPack.from_bam(args.bam_fn, ctg, pos, pos, ref, ref_lo, excl_flags=EXCL_FLAGS, min_mq=args.min_mq)
    a = pack.numpy()
    if pack.n_cols == 0:
        return None
    out = []
    k0 = int(a["key_off"][0])
    for e in a["entries"][int(a["col_off"][0]):int(a["col_off"][1])].tolist():
        if ((e >> 6) & 127) < args.min_bq:
            continue
        code, kind = e & 15, (e >> 4) & 3
        tok = "ACGTACGT*#NN"[code]
        if kind:                          # an indel rides on this base: the allele is never a bare letter (only that matters below)
            key = pack.key_string(k0 + (e >> 21))
            tok += ("+" + key[2:]) if key[0] == "I" else ("-" + "N" * (len(key) - 2))
        out.append(tok)
    return out


def sam_column_alleles(sam_text, pos, min_mq, min_bq):
    """the column at `pos` (1-based) of `samtools mpileup - --reverse-del --min-MQ --min-BQ --excl-flags 2316` over SAM rows, as
    upper-cased alleles: excluded flags, unmapped, orphans (paired without proper-pair), MAPQ, then per read the base / deletion
    placeholder at pos with the insertion or deletion that follows its aligned run, and the read-pair overlap rule between rows of
    the same name (csrc/bam.cpp: soften_overlap) before the BQ gate"""
    p0 = pos - 1
    reads = []
    for row in sam_text.split("\n"):
        if not row or row[0] == "@":
            continue
        c = row.split("\t")
        flag, start, mq, cigar = int(c[1]), int(c[3]) - 1, int(c[4]), c[5]
        if (flag & EXCL_FLAGS) or (flag & 4) or mq < min_mq or cigar == "*" or c[9] == "*" or ((flag & 1) and not (flag & 2)):
            continue
        ops = list(rr._cigar_ops(cigar))
        rlen = sum(n for op, n in ops if op in "MDN=X")
        if sum(n for op, n in ops if op in "MIS=X") != len(c[9]) or rlen == 0 or not (start <= p0 < start + rlen):
            continue
        qual = [0] * len(c[9]) if c[10] == "*" else [min(ord(ch) - 33, 93) for ch in c[10]]
        reads.append(dict(name=c[0], flag=flag, start=start, ops=ops, seq=c[9], qual=qual, end=start + rlen))
    # read-pair overlaps: the mate that entered first keeps min(200, qa + qb) when the bases agree, the better base 0.8 x its quality
    # when they differ; the other base -> 0 (only the two bases AT pos matter here)
    def locate(r):
        rp, qp = r["start"], 0
        for i, (op, n) in enumerate(r["ops"]):
            if op in "M=X":
                if rp <= p0 < rp + n:
                    return ("B", qp + (p0 - rp), i, p0 == rp + n - 1)
                rp += n; qp += n
            elif op == "D":
                if rp <= p0 < rp + n:
                    return ("D", qp, i, False)
                rp += n
            elif op == "N":
                if rp <= p0 < rp + n:
                    return None
                rp += n
            elif op in "IS":
                qp += n
        return None
    first = {}
    for r in reads:
        r["loc"] = locate(r)
        if (r["flag"] & 1) and r["loc"] and r["loc"][0] == "B":
            m = first.get(r["name"])
            if m is not None and m["loc"] and m["loc"][0] == "B":
                qa, qb = m["loc"][1], r["loc"][1]
                if m["seq"][qa] == r["seq"][qb]:
                    m["qual"][qa], r["qual"][qb] = min(200, m["qual"][qa] + r["qual"][qb]), 0
                elif m["qual"][qa] >= r["qual"][qb]:
                    m["qual"][qa], r["qual"][qb] = int(0.8 * m["qual"][qa]), 0
                else:
                    m["qual"][qa], r["qual"][qb] = 0, int(0.8 * r["qual"][qb])
        first.setdefault(r["name"], r)
    out = []
    for r in reads:
        loc = r["loc"]
        if loc is None:
            continue
        rev = bool(r["flag"] & 16)
        if loc[0] == "D":
            bq = min(r["qual"][loc[1]], 93) if loc[1] < len(r["qual"]) else 0
            if bq >= min_bq:
                out.append("#" if rev else "*")
            continue
        q, i, last = loc[1], loc[2], loc[3]
        if min(r["qual"][q], 93) < min_bq:
            continue
        b = r["seq"][q].upper()
        tok = b if b in "ACGT" else "N"
        if last:
            j = i + 1
            while j < len(r["ops"]) and r["ops"][j][0] == "P":
                j += 1
            if j < len(r["ops"]) and r["ops"][j][0] == "I":
                tok += "+" + r["seq"][q + 1:q + 1 + r["ops"][j][1]].upper()
            elif j < len(r["ops"]) and r["ops"][j][0] == "D":
                tok += "-" + "N" * r["ops"][j][1]
        out.append(tok)
    return out


def _mpileup(args, source, region_or_none, stdin_text=None, reverse_del=False):
    cmd = [args.samtools, "mpileup", source] + (["--reverse-del"] if reverse_del else []) + \
          ["--min-MQ", str(args.min_mq), "--min-BQ", str(args.min_bq), "--excl-flags", str(EXCL_FLAGS)] + (["-r", region_or_none] if region_or_none else [])
    return subprocess.run(cmd, input=stdin_text, stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True).stdout


def word_match(line, word):
    """`grep -w`: the number as a whole word somewhere in the line"""
    import re
    return re.search(r"(?<![A-Za-z0-9_])%s(?![A-Za-z0-9_])" % re.escape(word), line) is not None


def evaluate_call(args, rec, realign_fn=None):
    """extract_base (realign_variants.py:59-123) for one call -> (ctg, pos, passes, counts)"""
    ctg, pos = (args.ctg_name if args.ctg_name is not None else rec["ctg"]), rec["pos"]
    try:
        qual = float(rec["qual"]) if rec["qual"] is not None else None
    except ValueError:
        qual = None
    if qual is not None and qual >= QUAL_BAR:
        return ctg, pos, True, (-1, -1, -1, -1)
    native = getattr(args, "bam_reader", "samtools") == "native"
    if native:
        raw = bam_column_alleles(args, ctg, pos)
        if not raw:
            return ctg, pos, True, (-1, -1, -1, -1)
    else:
        cols = _mpileup(args, args.bam_fn, "{}:{}-{}".format(ctg, pos, pos)).rstrip().split("\t")
        if len(cols) < 4:
            return ctg, pos, True, (-1, -1, -1, -1)
    # the reference's inner command: realign_reads --pos P | samtools mpileup - --reverse-del ... | grep -w P
    sam = StringIO()
    inner = ArgumentParser()
    ns = inner.parse_args([])
    ns.pos, ns.ctg_name, ns.bam_fn, ns.ref_fn, ns.samtools = pos, ctg, args.bam_fn, args.ref_fn, args.samtools
    ns.min_mq, ns.min_coverage, ns.realign_flanking_window, ns.max_distance = 20, 2.0, 100, 50        # realign_reads' own defaults (:690-711)
    ns.bam_reader = "native" if native else "samtools"
    ns.realign_fn = realign_fn                # None: one cto_realign_reads per window; a WindowBatcher: windows of many calls per launch
    rr.reads_realignment(ns, out=sam)
    if native:
        new = sam_column_alleles(sam.getvalue(), pos, args.min_mq, args.min_bq)
        if not new:
            return ctg, pos, True, (-1, -1, -1, -1)
        ok, counts = decide(raw, new, rec["alt"])
        return ctg, pos, ok, counts
    text = _mpileup(args, "-", None, stdin_text=sam.getvalue(), reverse_del=True)
    hits = [ln for ln in text.split("\n") if word_match(ln, str(pos))]
    new_cols = "\n".join(hits).rstrip().split("\t")
    if len(new_cols) < 4:
        return ctg, pos, True, (-1, -1, -1, -1)
    ok, counts = decide(cols[4], new_cols[4], rec["alt"])
    return ctg, pos, ok, counts


def _evaluate_call_star(job):
    return evaluate_call(*job)


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
    batcher = None
    realigner = getattr(args, "realigner", "host")
    chose_auto = realigner == "auto"
    explicit_process_pool = getattr(args, "pool", "thread") == "process" and getattr(args, "pool_given", False)
    if realigner == "auto" and explicit_process_pool:
        realigner = "host"                        # a --pool process that was asked for is honoured: the batcher needs threads
    if realigner == "auto":                       # the device form where a HIP device is visible to this process
        try:
            import torch
            realigner = "device" if torch.cuda.is_available() else "host"
        except ImportError:
            realigner = "host"
    if realigner == "device" and todo:
        # the calls are worker threads that park their windows at a WindowBatcher (realign_reads.py): when all of them wait, one
        # cto_realign_windows call - k_fast_pass + k_sw on the current HIP device - serves the lot.  More threads than cores on
        # purpose: they sleep while the batch runs, and a batch is only as large as the number of calls in flight.
        batcher = rr.WindowBatcher("device", threads=threads, host_fallback=chose_auto)
        workers = max(threads, min(256, len(todo)))
        pool = ThreadPoolExecutor(max_workers=workers)

        def one(r):
            with batcher.worker():
                return evaluate_call(args, r, realign_fn=batcher)
        results = pool.map(one, todo)
    elif getattr(args, "pool", "thread") == "process" and threads > 1 and len(todo) > 1:
        import multiprocessing as mp
        pool = ProcessPoolExecutor(max_workers=threads, mp_context=mp.get_context("spawn"))
        results = pool.map(_evaluate_call_star, [(args, r) for r in todo], chunksize=max(1, min(64, len(todo) // (4 * threads) or 1)))
    else:
        pool = ThreadPoolExecutor(max_workers=threads)           # samtools children and the C calls run outside the GIL
        results = pool.map(lambda r: evaluate_call(args, r), todo)
    with pool as ex:
        for ctg, pos, ok, _ in results:
            if not ok:
                failed.add((ctg, pos))
            done += 1
            if done % 1000 == 0:
                print("[INFO] Processing in {}, total processed positions: {}".format(ctg, done), flush=True)
    if batcher is not None:
        batcher.close()
        print("[INFO] Realigner on the device: {} windows in {} batches{}".format(
            batcher.windows, batcher.batches, ", {} of them redone by the host form".format(batcher.fell_back) if batcher.fell_back else ""), flush=True)
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


def build_parser():
    p = ArgumentParser(description="Reads realignment workflow for all input variants")
    p.add_argument("--bam_fn", type=str, default=None)
    p.add_argument("--ref_fn", type=str, default="ref.fa")
    p.add_argument("--ctg_name", type=str, default=None)
    p.add_argument("--pileup_vcf_fn", type=str, default=None)
    p.add_argument("--output_dir", type=str, default=None)
    p.add_argument("--output_vcf_fn", type=str, default=None)
    p.add_argument("--samtools", type=str, default="samtools")
    p.add_argument("--bam_reader", type=str, default="samtools", choices=["samtools", "native"],
                   help="samtools: the reference's three samtools commands per call; native: the built-in BAM / FASTA readers and pileup "
                        "(no samtools needed; parity against samtools unpinned)")
    p.add_argument("--threads", type=int, default=1)
    p.add_argument("--pool", type=str, default="thread", choices=["thread", "process"],
                   help="workers are threads (default) or spawned processes (the reference's ProcessPoolExecutor; scales with the cores)")
    p.add_argument("--realigner", type=str, default="auto", choices=["auto", "host", "device"],
                   help="auto (default): device where a HIP device is visible (windows the device form gives back with an error are redone by the host form, same output; an explicit --pool process selects the host form), host otherwise; host: every window through cto_realign_reads on the worker that needs it (SSE2); device: the windows of all calls "
                        "in flight are batched into cto_realign_windows calls on the current HIP device (k-mer fast pass and both striped "
                        "Smith-Waterman passes as kernels; same output)")
    p.add_argument("--python", type=str, default="python3", help="accepted for compatibility: the realignment runs in-process")
    p.add_argument("--show_ref", action="store_true")
    p.add_argument("--min_mq", type=int, default=20)             # shared/param.py:17
    p.add_argument("--min_bq", type=int, default=0)              # shared/param.py:19
    p.add_argument("--enable_realignment", type=str2bool, default=True)
    p.add_argument("--qual", type=float, default=None, help="accepted for compatibility (unused by the reference too)")
    p.add_argument("--pos", type=int, default=None, help=SUPPRESS)
    p.add_argument("--is_indel", action="store_true", help=SUPPRESS)
    return p


def main(argv=None):
    p = build_parser()
    if len(sys.argv[1:] if argv is None else argv) == 0:
        p.print_help()
        sys.exit(1)
    a = p.parse_args(argv)
    a.pool_given = any(t == "--pool" or t.startswith("--pool=") for t in (sys.argv[1:] if argv is None else argv))
    realign_variants(a)


if __name__ == "__main__":
    main()

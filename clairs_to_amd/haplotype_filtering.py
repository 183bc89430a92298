"""Drop-in counterpart of `clairs_to.py haplotype_filtering` (reference: src/haplotype_filtering.py; STEP 4 / 8 of
run_clairs_to for long reads): tags the PASS calls of a pileup VCF with the long-read hard filters
(LowAltBQ, LowAltMQ, ReadStartEnd, VariantCluster, NoAncestry, MultiHap, StrandBias, LowSeqEntropy), the phaseable flag `H`
and the strand-bias p-value `SB`, from the haplotagged tumour BAM and a phased germline VCF.

Same inputs, options and output VCF as the reference.  What differs is how the work is done: the reference starts one
`pypy3` process per call under GNU parallel (or, in its chunk mode, one Python dict-of-dicts pass per <= 200 calls); here the
calls of a contig are cut into mpileup jobs, each job's nine-column `samtools mpileup --output-QNAME --output-extra HP` text is
handed to ONE C call (cto_haplotype_filter, csrc/hapfilter.cpp) that evaluates every read-level rule for every call of the
job, and Python keeps what is exact integer arithmetic in the reference (Fisher's test on the returned 2x2 table) and the VCF
text.  There is no device work in this stage (thousands of calls, not millions of sites).
"""
import bisect
import ctypes as C
import gzip
import os
import shlex
import subprocess
import sys
import tempfile
from argparse import ArgumentParser
from concurrent.futures import ThreadPoolExecutor

import numpy as np

from ._cli import add_ignored, add_unsupported, check_unsupported, str2bool, str_none
from ._lib import lib, check
from .fasta import read_region

LAST_FORMAT_LINE = '##FORMAT=<ID=TU,Number=1,Type=Integer,Description="Count of T in the tumor BAM">'
FLAG_NAMES = ("phaseable", "hetero", "homo", "read_start_end", "bq", "mq", "co_exist", "hetero_both_side", "sequence_entropy")
MAX_SITES_PER_JOB, MAX_SPAN_PER_JOB = 200, 5000000          # haplotype_filtering.py:190-191


# ------------------------------------------------------------------------------------------ Fisher's exact test
def n_choose_k(n, k):
    """exact binomial coefficient (0 when k > n), integers throughout"""
    if k > n:
        return 0
    k = min(k, n - k)
    out = 1
    for i in range(1, k + 1):
        out = out * (n - i + 1) // i
    return out


def fisher_exact_two_sided(a, b, c, d):
    """Two-sided p of the 2x2 table [[a, b], [c, d]] as haplotype_filtering.py:60-97 evaluates it: the probability t of the
    observed table from exact binomials (one correctly rounded integer division), then the tables on either side of it by
    the hypergeometric recurrence in floating point, added when they are no more likely than the observed one.  The order of
    the floating-point operations is the reference's (the value is printed into the VCF with five decimals)."""
    if a == b == c == d:
        return 1.0
    t = n_choose_k(a + b, a) * n_choose_k(c + d, c) / n_choose_k(a + b + c + d, a + c)
    total = t
    for towards_b in (True, False):              # first shrink a (and d), then shrink b (and c)
        w, x, y, z = a, b, c, d
        cur, side = float(t), 0.0
        while (w > 0 and z > 0) if towards_b else (x > 0 and y > 0):
            if towards_b:
                cur *= w * z
                w, x, y, z = w - 1, x + 1, y + 1, z - 1
                cur /= x * y
            else:
                cur *= x * y
                w, x, y, z = w + 1, x - 1, y - 1, z + 1
                cur /= w * z
            if cur <= t:
                side += cur
        total += side
    return total


# ------------------------------------------------------------------------------------------ VCF input
def _open_text(fn):
    with open(fn, "rb") as f:
        gz = f.read(2) == b"\x1f\x8b"
    return gzip.open(fn, "rt") if gz else open(fn)


def read_vcf(fn, ctg_name, show_ref=False, discard_indel=False, filter_tag=None, skip_genotype=False):
    """The subset of shared/vcf.py:VcfReader.read_vcf this stage relies on.  -> (header text, {pos: record}) with record =
    dict(ref, alt (first ALT allele), gt (sorted int pair), filter, af, qual, row)."""
    header, out = "", {}
    if fn is None or not os.path.exists(fn):
        return header, out
    contigs = None if ctg_name is None else set(x.strip() for x in ctg_name.split(",") if x.strip())
    allowed = None if filter_tag is None else filter_tag.split(",")
    with _open_text(fn) as f:
        for row in f:
            c = row.strip().split()
            if not c:
                continue
            if c[0][0] == "#":
                header += row
                continue
            if contigs is not None and c[0] not in contigs:
                continue
            flt = c[6] if len(c) >= 7 else None
            if allowed is not None and flt not in allowed:
                continue
            ref, alt = c[3], c[4]
            if discard_indel and (len(ref) > 1 or len(alt) > 1):
                continue
            qual = c[5] if len(c) > 5 else None
            gt = c[-1].split(":")[0].replace("/", "|").replace(".", "0").split("|")
            try:
                g1, g2 = gt
                if int(g1) > int(g2):
                    g1, g2 = g2, g1
                if "*" in alt:
                    alts = alt.split(",")
                    if int(g1) + int(g2) != 3 or len(alts) != 2:
                        continue
                    alt, g1, g2 = "".join(x for x in alts if x != "*"), "0", "1"
            except ValueError:
                g1 = g2 = -1
            af = None
            tags = c[8].split(":") if len(c) > 9 else []
            if "AF" in tags or "VAF" in tags:
                af = float(c[9].split(":")[tags.index("AF") if "AF" in tags else tags.index("VAF")])
            if g1 == "0" and g2 == "0" and not show_ref and not skip_genotype:
                continue
            key = int(c[1]) if (contigs is not None and len(contigs) == 1 and "," not in ctg_name) else (c[0], int(c[1]))
            out[key] = dict(ctg=c[0], pos=int(c[1]), ref=ref, alt=alt.split(",")[0] if "," in alt else alt, gt=(int(g1), int(g2)),
                            filter=flt, af=af, qual=qual, row=row)
    return header, out


def header_up_to_last_format(header):
    """haplotype_filtering.py:33-43: keep the header up to and including the TU FORMAT line (only the first line when it is
    missing)"""
    lines = header.split("\n")
    idx = 0
    for i, ln in enumerate(lines):
        if LAST_FORMAT_LINE in ln:
            idx = i
            break
    return "\n".join(lines[:idx + 1]) + "\n"


# ------------------------------------------------------------------------------------------ jobs
def partition_jobs(positions, flanking, max_sites=MAX_SITES_PER_JOB, max_span=MAX_SPAN_PER_JOB):
    """sorted call positions -> [(lo, hi, [positions])]: at most max_sites calls and max_span bp of region per mpileup job
    (what :196-221 bounds; the decisions do not depend on the cut, every call only sees its own +-flanking window)"""
    jobs, cur = [], []
    for p in sorted(positions):
        if cur and (len(cur) >= max_sites or (max_span > 0 and (p + flanking + 1) - max(cur[0] - flanking, 1) > max_span)):
            jobs.append(cur)
            cur = []
        cur.append(p)
    if cur:
        jobs.append(cur)
    return [(max(j[0] - flanking, 1), j[-1] + flanking + 1, j) for j in jobs]


def flank_bed(positions, flanking):
    """merged 0-based [start, end) intervals covering pos +- flanking of every call (:290-308)"""
    out = []
    for p in sorted(positions):
        s, e = max(p - flanking, 1) - 1, p + flanking
        if out and s <= out[-1][1]:
            out[-1][1] = max(out[-1][1], e)
        else:
            out.append([s, e])
    return out


def mpileup_text(args, contig, lo, hi, positions, flanking):
    """the nine-column text of one job: `--mpileup_fn` (prepared text, rows outside the job are ignored by the C side) or the
    reference's own samtools command (:322-345)"""
    if getattr(args, "mpileup_fn", None):
        with open(args.mpileup_fn, "rb") as f:
            return f.read()
    bam = args.tumor_bam_fn
    if not os.path.isfile(bam) and not bam.endswith(".bam"):
        bam = bam + contig + ".bam"                      # phased-prefix mode of run_clairs_to --phase_tumor (:275-280)
    fd, bed = tempfile.mkstemp(suffix=".bed", prefix="hf_mpileup_")
    try:
        with os.fdopen(fd, "w") as f:
            f.write("".join("%s\t%d\t%d\n" % (contig, s, e) for s, e in flank_bed(positions, flanking)))
        # --haplotype_chunk_mpileup_bed False: the job's whole span without -l (:1085-1093) - more rows, the same evidence
        bed_opt = "-l {} ".format(shlex.quote(bed)) if getattr(args, "haplotype_chunk_mpileup_bed", True) else ""
        cmd = "{} mpileup --min-MQ {} --min-BQ {} --excl-flags 2316 {}-r {} --output-MQ --output-QNAME --output-extra HP {}".format(
            shlex.quote(args.samtools), args.min_mq, args.min_bq, bed_opt, shlex.quote("%s:%d-%d" % (contig, lo, hi)), shlex.quote(bam))
        res = subprocess.run(shlex.split(cmd), stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        if res.returncode != 0:
            print("[ERROR] samtools mpileup failed (exit {}). Command (trunc): {} stderr: {}".format(
                res.returncode, cmd[:400], res.stderr.decode(errors="replace").strip().replace("\n", " ")[:800]), flush=True)
        return res.stdout
    finally:
        try:
            os.unlink(bed)
        except OSError:
            pass


def evaluate_job(text, ref_seq, region_lo, calls, flanking, max_co_exist_read_num, disable_rse):
    """calls: [(pos, ref, alt, af, hetero_info, homo_info)] -> [dict(flags..., strand table)] through cto_haplotype_filter"""
    n = len(calls)
    pos = np.array([c[0] for c in calls], dtype=np.int32)
    af = np.array([1.0 if c[3] is None else float(c[3]) for c in calls], dtype=np.float64)
    fields, off = [], np.zeros(n + 1, dtype=np.int64)
    for i, c in enumerate(calls):
        b = ("%s\t%s\t%s\t%s" % (c[1], c[2], c[4], c[5])).encode()
        fields.append(b)
        off[i + 1] = off[i] + len(b)
    blob = b"".join(fields)
    flags = np.zeros((n, len(FLAG_NAMES)), dtype=np.uint8)
    strand = np.zeros((n, 4), dtype=np.int64)
    tb = text if isinstance(text, (bytes, bytearray)) else text.encode()
    rb = ref_seq.encode() if isinstance(ref_seq, str) else ref_seq
    fb = np.frombuffer(blob, dtype=np.uint8) if blob else np.zeros(1, dtype=np.uint8)
    tarr = np.frombuffer(tb, dtype=np.uint8) if len(tb) else np.zeros(1, dtype=np.uint8)
    check(lib.cto_haplotype_filter(tarr.ctypes.data, len(tb), rb, int(region_lo), len(rb), n, pos.ctypes.data, fb.ctypes.data,
                                   off.ctypes.data, af.ctypes.data, int(flanking), int(max_co_exist_read_num), int(bool(disable_rse)),
                                   flags.ctypes.data, strand.ctypes.data))
    out = []
    for i, c in enumerate(calls):
        a0, r0, a1, r1 = (int(v) for v in strand[i])
        p_value = fisher_exact_two_sided(a0, r0, a1, r1)
        is_snp = len(c[1]) == 1 and len(c[2]) == 1
        # `is_snp and p < 0.001 or (a0 == 0 or a1 == 0)` / `not is_snp and p < 0.01 or (...)` (:584-587)
        sb_fail = (p_value < (0.001 if is_snp else 0.01)) or a0 == 0 or a1 == 0
        d = {k: bool(flags[i, j]) for j, k in enumerate(FLAG_NAMES)}
        d.update(pos=c[0], strand_bias=not sb_fail, p_value=str(round(p_value, 5)))
        d["pass_hap"] = all(d[k] for k in FLAG_NAMES[1:]) and d["strand_bias"]
        out.append(d)
    return out


# ------------------------------------------------------------------------------------------ the stage
def tag_row(row, res):
    """update_filter_info (:765-823) for one evaluated call"""
    c = row.rstrip().split("\t")
    if res["phaseable"]:
        c[7] = "H;" + c[7]
    if not res["pass_hap"]:
        c[5], c[6] = "0.0000", "LowQual"
    for ok, tag in ((res["bq"], "LowAltBQ"), (res["mq"], "LowAltMQ"), (res["read_start_end"], "ReadStartEnd"),
                    (res["co_exist"], "VariantCluster"), (res["hetero"] and res["homo"], "NoAncestry"),
                    (res["hetero_both_side"], "MultiHap"), (res["strand_bias"], "StrandBias"), (res["sequence_entropy"], "LowSeqEntropy")):
        if not ok:
            c[6] += ";" + tag
    c[7] += ";SB={}".format(res["p_value"])
    return "\t".join(c)


def haplotype_filter(args):
    ctg_name = args.ctg_name
    if ctg_name is None or "," in ctg_name:
        sys.exit("[ERROR] clairs_to_amd haplotype_filtering handles one contig per invocation (--ctg_name)")
    flanking, max_co = args.flanking, args.min_alt_coverage
    os.makedirs(args.output_dir, exist_ok=True)
    if not args.apply_haplotype_filtering:
        if os.path.lexists(args.output_vcf_fn):
            os.remove(args.output_vcf_fn)
        os.symlink(args.pileup_vcf_fn, args.output_vcf_fn)
        return
    _, germ = read_vcf(args.germline_vcf_fn, ctg_name, show_ref=False, filter_tag="PASS")
    germ_list = sorted((p, r["alt"], sum(r["gt"])) for p, r in germ.items() if sum(r["gt"]) in (1, 2))
    germ_pos = [g[0] for g in germ_list]
    header, pileup = read_vcf(args.pileup_vcf_fn, ctg_name, show_ref=args.show_ref, discard_indel=not args.is_indel,
                              filter_tag=args.input_filter_tag)
    allowed = None if args.input_filter_tag is None else frozenset(s.strip() for s in args.input_filter_tag.split(",") if s.strip())
    calls = {}
    hap_info_fn = os.path.join(args.output_dir, "HAP_INFO_INDEL" if args.is_indel else "HAP_INFO_SNV")
    with open(hap_info_fn, "w") as f:
        for p, r in pileup.items():
            if (r["filter"] not in allowed) if allowed is not None else (r["filter"] != "PASS"):
                continue
            if args.test_pos and p != args.test_pos:
                continue
            lo, hi = bisect.bisect_right(germ_pos, p - flanking), bisect.bisect_right(germ_pos, p + flanking)
            het = ",".join("%d-%s" % (g[0], g[1]) for g in germ_list[lo:hi] if g[0] != p and g[2] == 1)
            hom = ",".join("%d-%s" % (g[0], g[1]) for g in germ_list[lo:hi] if g[0] != p and g[2] == 2)
            calls[p] = (p, r["ref"], r["alt"], r["af"], het, hom)
            f.write(" ".join([ctg_name, str(p), r["ref"], r["alt"], str(r["af"]), str(r["qual"]), het, hom]) + "\n")
    results = {}
    jobs = partition_jobs(calls.keys(), flanking, args.haplotype_chunk_max_sites, args.haplotype_chunk_max_span)

    def run_job(job):
        lo, hi, ps = job
        text = mpileup_text(args, ctg_name, lo, hi, ps, flanking)
        ref = read_region(args.ref_fn, ctg_name, lo, hi)
        return evaluate_job(text, ref, lo, [calls[p] for p in ps], flanking, max_co, args.disable_read_start_end_filtering)
    threads = max(1, int(args.threads * 4 / 5))
    if len(jobs) <= 1 or threads <= 1:
        done = [run_job(j) for j in jobs]
    else:
        with ThreadPoolExecutor(max_workers=min(threads, len(jobs))) as ex:      # samtools + the C call run outside the GIL
            done = list(ex.map(run_job, jobs))
    n_done = 0
    for part in done:
        for r in part:
            results[r["pos"]] = r
            n_done += 1
            if n_done % 1000 == 0:
                print("[INFO] Haplotype filtering: {} candidates processed".format(n_done), flush=True)
    out_header = header_up_to_last_format(header)
    fai = args.ref_fn + ".fai" if os.path.exists(args.ref_fn + ".fai") else ".".join(args.ref_fn.split(".")[:-1]) + ".fai"
    for row in open(fai):
        c = row.strip().split("\t")
        if c[0] == ctg_name:
            out_header += "##contig=<ID=%s,length=%s>\n" % (c[0], c[1])
    out_header += "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tSAMPLE\n"
    os.makedirs(os.path.dirname(os.path.abspath(args.output_vcf_fn)), exist_ok=True)
    with open(args.output_vcf_fn, "w") as out:
        out.write(out_header)
        for p in sorted(pileup):
            row = pileup[p]["row"].rstrip()
            out.write((tag_row(row, results[p]) if p in results else row) + "\n")
    n_in = len(calls)
    fails = lambda *keys: sum(1 for r in results.values() if not all(r[k] for k in keys))
    for label, keys in (("all hard filters", ("pass_hap",)), ("low alt bq", ("bq",)), ("low alt mq", ("mq",)),
                        ("read start and end", ("read_start_end",)), ("variant cluster", ("co_exist",)), ("no ancestry", ("hetero", "homo")),
                        ("multi haplotypes", ("hetero_both_side",)), ("strand bias", ("strand_bias",)),
                        ("low sequence entropy", ("sequence_entropy",))):
        print("[INFO] Total input calls: {}, filtered by {}: {}".format(n_in, label, fails(*keys)), flush=True)
    return results


def build_parser():
    p = ArgumentParser(description="Haplotype filtering for long-read data (C evaluation of the read-level rules)")
    p.add_argument("--tumor_bam_fn", type=str, default=None)
    p.add_argument("--ref_fn", type=str, default=None)
    p.add_argument("--ctg_name", type=str, default=None)
    p.add_argument("--pileup_vcf_fn", type=str, default=None)
    p.add_argument("--output_vcf_fn", type=str, default=None)
    p.add_argument("--germline_vcf_fn", type=str, default=None)
    p.add_argument("--output_dir", type=str, default=None)
    p.add_argument("--threads", type=int, default=4)
    p.add_argument("--input_filter_tag", type=str_none, default=None)
    p.add_argument("--show_ref", action="store_true")
    p.add_argument("--samtools", type=str, default="samtools")
    p.add_argument("--mpileup_fn", type=str, default=None, help="prepared nine-column mpileup text instead of running samtools")
    p.add_argument("--apply_haplotype_filtering", type=str2bool, default=True)
    p.add_argument("--min_mq", type=int, default=20)             # shared/param.py:17
    p.add_argument("--min_bq", type=int, default=0)              # shared/param.py:19
    p.add_argument("--min_alt_coverage", type=int, default=2)
    p.add_argument("--is_indel", action="store_true")
    p.add_argument("--test_pos", type=int, default=None)
    p.add_argument("--flanking", type=int, default=100)
    p.add_argument("--haplotype_filtering_chunk_mode", type=str2bool, default=True,
                   help="the reference's default is False (one pypy3 process per call under GNU parallel, :800-880); both of its modes evaluate the "
                        "same rules on the same rows, and here every call is evaluated in mpileup jobs either way")
    p.add_argument("--haplotype_chunk_max_sites", type=int, default=MAX_SITES_PER_JOB)
    p.add_argument("--haplotype_chunk_max_span", type=int, default=MAX_SPAN_PER_JOB)
    p.add_argument("--haplotype_chunk_mpileup_bed", type=str2bool, default=True)
    p.add_argument("--disable_read_start_end_filtering", type=str2bool, default=False)
    # src/haplotype_filtering.py:1213-1312.  --python / --pypy3 / --parallel name the interpreters of its per-call worker processes;
    # --debug, --add_phasing_info, --is_happy_format, --max_overlap_distance, --hap_info_fn are declared there and read nowhere.
    add_ignored(p, python="str", pypy3="str", parallel="str", hap_info_fn="str", debug="flag", add_phasing_info="bool", is_happy_format="bool",
                max_overlap_distance="int")
    # its worker mode: ONE call per process, described on the command line (:708-740) - only the reference's own driver starts those
    add_unsupported(p, pos=("int", None), ref_base=("str", None), alt_base=("str", None), af=("float", None), qual=("float", None),
                    hetero_info=("str", None), homo_info=("str", None))
    return p


def main(argv=None):
    p = build_parser()
    a = p.parse_args(argv)
    check_unsupported(p, a)
    haplotype_filter(a)


if __name__ == "__main__":
    main()

"""Candidate extraction on the GPU: counterpart of `clairs_to.py extract_candidates_calling`
(reference: src/extract_candidates_calling.py, STEP 1 of run_clairs_to:1194-1226).

`extract_candidates()` runs the gates on a column pack that is already in HBM - the same pack tensor creation uses, so
candidate sites become an internal product of the engine (one pileup instead of three).  `main()` mirrors the CLI at
its file seam and takes the argv run_clairs_to builds for it: the region of `--chunk_id / --chunk_num` (from the .fai or from the
span of the confident BED, :240-270), the confident BED's restriction (`-l`, :302), `--call_indels_only_in_these_regions` (:437-446),
the positions of `--hybrid_mode_vcf_fn / --genotyping_mode_vcf_fn` (:225-238, 347-383) - each an interval test or a marker on the
flags in HBM - and it writes what the reference writes: the `<ctg>.<chunk>_<i>_<n>_snv` / `_indel` BED chunk files (<= 10 000 windows
`x-17 .. x+17` each), the `SNV_CANDIDATES_FILE_*` / `INDEL_CANDIDATES_FILE_*` lists the next step reads, `bed/<ctg>_<chunk>.bed` and
`<ctg>.<chunk>_hybrid_info`."""
import ctypes as C
import gzip
import os
from argparse import ArgumentParser

import numpy as np
import torch

from ._cli import add_ignored, add_unsupported, check_unsupported, str2bool, str_none
from ._lib import lib, check, current_stream_ptr
from .fasta import read_region
from .pack import ColumnPack

SPLIT_BED_SIZE, FLANK, EXPAND_REF = 10000, 16, 1000       # shared/param.py:21, 60, 101


def extract_candidates(dev_pack, min_bq, min_mq=20, snv_min_af=0.05, indel_min_af=0.05, min_coverage=4, alt_base_num=3,
                       select_indel=True):
    """-> (flags uint8 [n_cols]: bit0 SNV candidate, bit1 indel candidate, bit2 pass_af; depth int32 [n_cols])."""
    n = max(dev_pack.n_cols, 1)
    flags = torch.zeros((n,), dtype=torch.uint8, device=dev_pack.device)
    depth = torch.zeros((n,), dtype=torch.int32, device=dev_pack.device)
    with torch.cuda.device(dev_pack.device):
        check(lib.cto_extract_candidates(C.byref(dev_pack.view), int(min_mq), int(min_bq), float(snv_min_af),
                                         float(indel_min_af), float(min_coverage), int(alt_base_num), int(bool(select_indel)),
                                         flags.data_ptr(), depth.data_ptr(), current_stream_ptr()))
    return flags[:dev_pack.n_cols], depth[:dev_pack.n_cols]


def candidate_positions(dev_pack, flags, bit=1, lo=1, hi=2 ** 31 - 1):
    """Sorted 1-based positions of the columns whose flag has `bit` set (1 = SNV list, 2 = indel list) and whose position lies in
    [lo, hi]: device tensor, int32, compacted on the device in position order (cto_candidate_positions)."""
    dev = dev_pack.device
    nc = dev_pack.n_cols
    if nc == 0:
        return torch.empty((0,), dtype=torch.int32, device=dev)
    flags = flags.contiguous()
    out = torch.empty((nc,), dtype=torch.int32, device=dev)
    scratch = torch.empty(((nc + 255) // 256 + 2,), dtype=torch.int32, device=dev)
    n_out = torch.empty((1,), dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        check(lib.cto_candidate_positions(C.byref(dev_pack.view), flags.data_ptr(), int(bit), int(lo), int(hi), out.data_ptr(), nc,
                                          scratch.data_ptr(), n_out.data_ptr(), current_stream_ptr()))
    return out[:int(n_out.item())]


def restrict_to_intervals(dev_pack, flags, depth, intervals, clear):
    """flags of the columns outside `intervals` (sorted, merged, 0-based half-open [begin, end) pairs) lose the bits of `clear`
    (cto_extract_restrict): 0xff = the confident BED of `samtools mpileup -l`, 2 | 16 = --call_indels_only_in_these_regions."""
    if dev_pack.n_cols == 0:
        return
    iv = torch.tensor(np.asarray(intervals, dtype=np.int32).reshape(-1), dtype=torch.int32, device=dev_pack.device)
    with torch.cuda.device(dev_pack.device):
        check(lib.cto_extract_restrict(C.byref(dev_pack.view), flags.data_ptr(), depth.data_ptr() if depth is not None else None,
                                       iv.data_ptr(), int(iv.numel() // 2), int(clear), current_stream_ptr()))


def mark_positions(dev_pack, flags, positions, bit=64):
    """bit 6 on the columns of `positions` (sorted int32): the hybrid / genotyping list (cto_extract_mark)"""
    if dev_pack.n_cols == 0 or len(positions) == 0:
        return None
    d_pos = torch.tensor(np.asarray(positions, dtype=np.int32), dtype=torch.int32, device=dev_pack.device)
    with torch.cuda.device(dev_pack.device):
        check(lib.cto_extract_mark(C.byref(dev_pack.view), flags.data_ptr(), d_pos.data_ptr(), int(d_pos.numel()), int(bit), current_stream_ptr()))
    return d_pos


def hybrid_info_rows(pack, dev_pack, flags, positions, ctg, min_mq, min_bq, select_indel):
    """The text of `<ctg>.<chunk>_hybrid_info` for `positions` (sorted): counts on the device (cto_hybrid_info), strings on the host."""
    n = len(positions)
    if n == 0 or dev_pack.n_cols == 0:
        return ""
    dev = dev_pack.device
    d_pos = torch.tensor(np.asarray(positions, dtype=np.int32), dtype=torch.int32, device=dev)
    rec = torch.empty((n, 16), dtype=torch.int32, device=dev)
    nk = max(dev_pack.n_keys, 1)
    gcnt = torch.zeros((nk,), dtype=torch.int32, device=dev)
    gfirst = torch.zeros((nk,), dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        check(lib.cto_hybrid_info(C.byref(dev_pack.view), flags.data_ptr(), d_pos.data_ptr(), n, int(min_mq), int(min_bq), int(bool(select_indel)),
                                  rec.data_ptr(), gcnt.data_ptr(), gfirst.data_ptr(), current_stream_ptr()))
    h_rec, h_cnt, h_first = rec.cpu().numpy(), gcnt.cpu().numpy(), gfirst.cpu().numpy()
    h_pos = np.ascontiguousarray(positions, dtype=np.int32)
    need = int(check(lib.cto_hybrid_info_rows(pack._h, ctg.encode(), n, h_pos.ctypes.data, h_rec.ctypes.data, int(bool(select_indel)),
                                              h_cnt.ctypes.data, h_first.ctypes.data, None, 0)))
    if need == 0:
        return ""
    buf = C.create_string_buffer(need)
    check(lib.cto_hybrid_info_rows(pack._h, ctg.encode(), n, h_pos.ctypes.data, h_rec.ctypes.data, int(bool(select_indel)),
                                   h_cnt.ctypes.data, h_first.ctypes.data, C.addressof(buf), need))
    return buf.raw[:need].decode()


def _open_text(fn):
    with open(fn, "rb") as f:
        gz = f.read(2) == b"\x1f\x8b"
    return gzip.open(fn, "rt") if gz else open(fn)


def read_bed_rows(bed_fn, ctg_name):
    """rows of `ctg_name` as (begin, end), fields split on any white space (bed_tree_from, shared/interval_tree.py:42-71; run_clairs_to's
    split BEDs are space-separated); a row with end < begin or a negative bound ends the run as it ends the reference's"""
    rows = []
    with _open_text(bed_fn) as f:
        for row_id, row in enumerate(f):
            if row[:1] == "#":
                continue
            c = row.strip().split()
            if not c or c[0] != ctg_name:
                continue
            b, e = int(c[1]), int(c[2])
            if e < b or b < 0 or e < 0:
                raise SystemExit("[ERROR] Invalid bed input in {}-th row {} {} {}".format(row_id + 1, c[0], b, e))
            rows.append((b, e))
    return rows


def merged_intervals(rows, widen_empty):
    """sorted, merged [begin, end) pairs.  widen_empty: a row with begin == end covers one base (the interval tree of the reference,
    interval_tree.py:68-69); without it such a row covers nothing (`samtools mpileup -l`)."""
    iv = sorted((b, e + 1 if (widen_empty and b == e) else e) for b, e in rows)
    out = []
    for b, e in iv:
        if e <= b:
            continue
        if out and b <= out[-1][1]:
            out[-1][1] = max(out[-1][1], e)
        else:
            out.append([b, e])
    return out


def read_known_vcf(vcf_fn, ctg_name, select_indel):
    """VcfReader(vcf_fn, ctg_name).read_vcf() as extract_candidates_calling uses it (:225-238; shared/vcf.py:239-352): the positions of the
    contig's records (the last record of a position decides) and which of them are indel records - REF or the first ALT longer than one
    base.  A `*` allele is dropped from ALT when the genotype parses as 1/2 with two alleles (the record is skipped when it parses as
    anything else); a genotype that does not parse leaves ALT as it is."""
    recs = {}
    tumor_last = False
    with _open_text(vcf_fn) as f:
        for row in f:
            c = row.strip().split()
            if not c:
                continue
            if c[0][0] == "#":
                tumor_last = c[-1].rstrip().lower() == "tumor"
                continue
            if c[0] != ctg_name:
                continue
            ref, alt = c[3], c[4]
            last = c[-2] if tumor_last else c[-1]
            gt = last.split(":")[0].replace("/", "|").replace(".", "0").split("|")
            try:
                g1, g2 = gt
                if int(g1) > int(g2):
                    g1, g2 = g2, g1
                if "*" in alt:
                    alts = alt.split(",")
                    if int(g1) + int(g2) != 3 or len(alts) != 2:
                        continue
                    alt = "".join(x for x in alts if x != "*")
            except ValueError:
                pass
            recs[int(c[1])] = len(ref) > 1 or len(alt.split(",")[0]) > 1
    positions = sorted(recs)
    return positions, [p for p in positions if recs[p]] if select_indel else []


def contig_length(ref_fn, ctg_name):
    fai = ref_fn + ".fai" if os.path.exists(ref_fn + ".fai") else ".".join(ref_fn.split(".")[:-1]) + ".fai"
    if not os.path.exists(fai):
        raise SystemExit("[ERROR] file %s not found" % (ref_fn + ".fai"))
    n = 0
    for row in open(fai):
        c = row.strip().split("\t")
        if c[0] == ctg_name:
            n = int(c[1])
    return n


def chunk_region(a, confident_rows):
    """(ctg_start, ctg_end, chunk_id as the file names carry it): --chunk_id / --chunk_num cut the contig's length from the .fai - or, with
    a confident BED, the span of its rows - into chunk_num equal parts (:240-270); otherwise --ctg_start / --ctg_end as given."""
    chunk_id = a.chunk_id - 1 if a.chunk_id else None               # 1-based on the command line, 0-based from here on (:182)
    if chunk_id is None:
        return a.ctg_start, a.ctg_end, chunk_id
    if not a.chunk_num:
        raise SystemExit("[ERROR] --chunk_id needs --chunk_num")
    if confident_rows is not None:
        if not confident_rows:
            raise SystemExit("[ERROR] the BED {} has no row of {}".format(a.bed_fn, a.ctg_name))
        b0, b1 = min(b for b, _ in confident_rows), max(e for _, e in confident_rows)
        size = (b1 - b0) // a.chunk_num + 1 if (b1 - b0) % a.chunk_num else (b1 - b0) // a.chunk_num
        start = b0 + 1 + size * chunk_id
    else:
        n = contig_length(a.ref_fn, a.ctg_name)
        size = n // a.chunk_num + 1 if n % a.chunk_num else n // a.chunk_num
        start = size * chunk_id
    return start, start + size, chunk_id


def write_bed_chunks(folder, ctg, chunk_id, positions, suffix, list_prefix, flank=None):
    """The chunk files of extract_candidates_calling.py:450-488.  chunk_id as the reference prints it: 0-based, or None without --chunk_id."""
    flank = FLANK if flank is None else flank
    if not len(positions):
        return []
    os.makedirs(folder, exist_ok=True)
    n_regions = -(-len(positions) // SPLIT_BED_SIZE)
    paths = []
    for i in range(n_regions):
        part = positions[i * SPLIT_BED_SIZE:(i + 1) * SPLIT_BED_SIZE]
        path = os.path.join(folder, "{}.{}_{}_{}_{}".format(ctg, chunk_id, i, n_regions, suffix))
        with open(path, "w") as f:
            f.write("\n".join("\t".join([ctg, str(max(x - flank - 1, 1)), str(x + flank + 1)]) for x in part) + "\n")
        paths.append(path)
    with open(os.path.join(folder, "{}_{}_{}".format(list_prefix, ctg, chunk_id)), "w") as f:
        f.write("\n".join(paths) + "\n")
    return paths


def pack_of_region(a, lo, hi, confident_bed=None):
    """The MQ- and BQ-unfiltered pack of the rows lo .. hi (the gates run on the device): pre-made text, the native BAM reader, or
    `samtools mpileup` - the reference's own command (extract_candidates_calling.py:298-309) but with `--min-MQ 0 --min-BQ 0
    --output-MQ` so that the same pack also serves tensor creation."""
    import shlex
    import subprocess
    if a.mpileup_fn:
        opener = gzip.open if a.mpileup_fn.endswith(".gz") else open
        with opener(a.mpileup_fn, "rb") as f:
            text = f.read()
        if not text.strip():
            return ColumnPack.from_mpileup(b"", "A", 1)
        first = int(text.split(b"\t", 2)[1])
        last = int(text.rstrip(b"\n").rsplit(b"\n", 1)[-1].split(b"\t", 2)[1])
        ref_start = max(1, first - EXPAND_REF)
        ref = read_region(a.ref_fn, a.ctg_name, ref_start, last + EXPAND_REF)
        return ColumnPack.from_mpileup(text, ref, ref_start)
    if lo is None or hi is None:
        raise SystemExit("[ERROR] a region (--chunk_id / --chunk_num, or --ctg_start / --ctg_end) is required with --tumor_bam_fn")
    ref_start = max(1, lo - EXPAND_REF)
    ref = read_region(a.ref_fn, a.ctg_name, ref_start, hi + EXPAND_REF)
    if not ref:
        raise SystemExit("[ERROR] Failed to load reference sequence from file ({}).".format(a.ref_fn))
    if a.bam_reader == "native":
        return ColumnPack.from_bam(a.tumor_bam_fn, a.ctg_name, lo, hi, ref, ref_start,
                                   max_depth=a.max_depth if a.max_depth is not None else 8000)
    cmd = "{} mpileup --reverse-del --output-MQ -r {}:{}-{} --min-MQ 0 --min-BQ 0{} --excl-flags 2316".format(
        a.samtools, a.ctg_name, lo, hi, " -l {}".format(confident_bed) if confident_bed else "")
    if a.max_depth is not None:
        cmd += " --max-depth {}".format(a.max_depth)
    text = subprocess.run(shlex.split(cmd) + [a.tumor_bam_fn], stdout=subprocess.PIPE, check=True).stdout
    return ColumnPack.from_mpileup(text, ref, ref_start)


def extract_to_files(a, device="cuda"):
    """extract_pair_candidates (:172-503) with every gate on the device; returns (SNV positions, indel positions)."""
    g = lambda k, d=None: getattr(a, k, d)
    flank = FLANK if g("flanking") is None else g("flanking")
    n_pos = 2 * flank + 1
    select_indel = bool(a.select_indel_candidates)
    # the optional inputs; a path that does not exist is no input (file_path_from(..., exit_on_not_found=False), :205-210) - run_clairs_to
    # always passes `--bed_fn <work>/split_beds/<ctg>` and `--call_indels_only_in_these_regions <work>/split_indel_beds/<ctg>`
    conf_fn = g("bed_fn") if g("bed_fn") and os.path.exists(g("bed_fn")) else None
    indel_fn = g("call_indels_only_in_these_regions")
    indel_fn = indel_fn if indel_fn and os.path.exists(indel_fn) else None
    conf_rows = read_bed_rows(conf_fn, a.ctg_name) if conf_fn else None
    known_fn = g("hybrid_mode_vcf_fn") or g("genotyping_mode_vcf_fn")
    known, known_indel = read_known_vcf(known_fn, a.ctg_name, select_indel) if known_fn else ([], [])
    ctg_start, ctg_end, chunk_id = chunk_region(a, conf_rows)
    if ctg_start is not None and ctg_end is not None:
        lo, hi = max(ctg_start - n_pos, 1), ctg_end + n_pos                     # the rows the reference asks samtools for (:289-292)
    elif a.mpileup_fn:
        lo, hi = 1, 2 ** 31 - 1
    else:
        lo, hi = 1, contig_length(a.ref_fn, a.ctg_name)                         # --ctg_name alone: the whole contig (:296-299)
    pack = pack_of_region(a, None if a.mpileup_fn else lo, hi, conf_fn)
    dp = pack.to_device(device)
    flags, depth = extract_candidates(dp, a.min_bq, a.min_mq, a.snv_min_af, a.indel_min_af, a.min_coverage,
                                      a.alternative_base_num, select_indel)
    if conf_rows is not None:
        restrict_to_intervals(dp, flags, depth, merged_intervals(conf_rows, widen_empty=False), 0xff)
    if select_indel and indel_fn and not g("bed_fn_source"):                    # superseded by --bed_fn (:438)
        rows = read_bed_rows(indel_fn, a.ctg_name)
        if rows:                                                                # a BED without rows of the contig filters nothing (:440)
            restrict_to_intervals(dp, flags, None, merged_intervals(rows, widen_empty=True), 2 | 16)
    if known:
        mark_positions(dp, flags, known)
    snv = candidate_positions(dp, flags, 1, lo, hi).cpu().tolist()
    indel = candidate_positions(dp, flags, 2, lo, hi).cpu().tolist() if select_indel else []
    folder = a.candidates_folder
    os.makedirs(os.path.join(folder, "bed"), exist_ok=True)
    passed = candidate_positions(dp, flags, 4, lo, hi).cpu().tolist()
    with open(os.path.join(folder, "bed", "{}_{}.bed".format(a.ctg_name, chunk_id)), "w") as f:          # candidates_set (:385-392)
        f.write("".join("%s\t%d\t%d\n" % (a.ctg_name, x - 1, x) for x in sorted(set(passed) | set(known))))
    if select_indel:
        print("[INFO] {} chunk {}/{}: Total SNV candidates found: {}, total Indel candidates found: {}".format(
            a.ctg_name, chunk_id, g("chunk_num"), len(snv), len(indel)))
    else:
        print("[INFO] {} chunk {}/{}: Total SNV candidates found: {}".format(a.ctg_name, chunk_id, g("chunk_num"), len(snv)))
    write_bed_chunks(folder, a.ctg_name, chunk_id, snv, "snv", "SNV_CANDIDATES_FILE", flank)
    write_bed_chunks(folder, a.ctg_name, chunk_id, indel, "indel", "INDEL_CANDIDATES_FILE", flank)
    if known_fn:
        in_rows = [x for x in known if lo <= x <= hi]
        with open(os.path.join(folder, "{}.{}_hybrid_info".format(a.ctg_name, chunk_id)), "w") as f:
            f.write(hybrid_info_rows(pack, dp, flags, in_rows, a.ctg_name, a.min_mq, a.min_bq, select_indel))
    return snv, indel


def build_parser():
    p = ArgumentParser(description="Extract candidate sites from a pileup (GPU gates)")
    p.add_argument("--platform", type=str, default="ont")
    p.add_argument("--candidates_folder", type=str, required=True)
    p.add_argument("--mpileup_fn", type=str, default=None, help="samtools mpileup --reverse-del --output-MQ --min-MQ 0 --min-BQ 0 text")
    p.add_argument("--tumor_bam_fn", type=str, default=None)
    p.add_argument("--samtools", type=str, default="samtools")
    p.add_argument("--bam_reader", type=str, default="samtools", choices=["samtools", "native"])
    p.add_argument("--max_depth", type=int, default=None,
                   help="EXPERIMENTAL: maximum tumor depth handed to samtools mpileup / the built-in reader (the reference's default: 8000). Engine limits, reported as an error of the chunk (CTO_EUNSUPPORTED), not silently: a pileup column may hold at most 32767 read-bases and 2048 distinct indel alleles")
    p.add_argument("--ref_fn", type=str, required=True)
    p.add_argument("--ctg_name", type=str, required=True)
    p.add_argument("--ctg_start", type=int, default=None)
    p.add_argument("--ctg_end", type=int, default=None)
    p.add_argument("--chunk_id", type=int, default=None, help="1-based; the file names carry it 0-based, as the reference's do")
    p.add_argument("--chunk_num", type=int, default=None)
    p.add_argument("--bed_fn", type=str, default=None, help="confident regions of this contig (run_clairs_to: <work>/split_beds/<ctg>); a missing file is no BED")
    p.add_argument("--bed_fn_source", type=str_none, default=None, help="the user's --bed_fn, if any: it supersedes --call_indels_only_in_these_regions")
    p.add_argument("--call_indels_only_in_these_regions", type=str, default=None)
    p.add_argument("--hybrid_mode_vcf_fn", type=str_none, default=None)
    p.add_argument("--genotyping_mode_vcf_fn", type=str_none, default=None)
    p.add_argument("--snv_min_af", type=float, default=0.05)
    p.add_argument("--indel_min_af", type=float, default=1.0)
    p.add_argument("--min_coverage", type=float, default=4)
    p.add_argument("--min_mq", type=int, default=20)
    p.add_argument("--min_bq", type=int, default=None, help="default: the platform's (shared/param.py min_bq_dict)")
    p.add_argument("--alternative_base_num", type=int, default=3)
    p.add_argument("--select_indel_candidates", type=str2bool, default=False)
    p.add_argument("--flanking", type=int, default=None)
    # src/extract_candidates_calling.py:506-610.  --output_depth / --output_alt_info only shape the rows of --alt_fn, which run_clairs_to
    # never asks for; the others below it belong to training-set preparation.
    add_ignored(p, output_depth="bool", output_alt_info="bool")
    add_unsupported(p, alt_fn=("str", None), store_tumor_infos=("bool", False), truth_vcf_fn=("str", None), min_truth_snv_af=("float", None),
                    min_truth_indel_af=("float", None))
    return p


def main(argv=None):
    p = build_parser()
    a = p.parse_args(argv)
    check_unsupported(p, a)
    if not a.mpileup_fn and not a.tumor_bam_fn:
        raise SystemExit("[ERROR] one of --mpileup_fn / --tumor_bam_fn is required")
    if a.min_bq is None:
        from .platforms import MIN_BQ
        a.min_bq = MIN_BQ.get(a.platform, 0)
    extract_to_files(a)


if __name__ == "__main__":
    main()

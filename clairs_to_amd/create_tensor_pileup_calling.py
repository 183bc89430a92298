"""Drop-in counterpart of `clairs_to.py create_tensor_pileup_calling` (reference: src/create_tensor_pileup_calling.py,
invoked twice per chunk at run_clairs_to:1228-1271) with the featurisation on the GPU.

Same tensor text out (7 tab-separated fields per site, create_tensor_pileup_calling.py:561-569).  The pileup comes
either from `--mpileup_fn` (text made by `samtools mpileup --reverse-del --output-MQ --min-MQ 0 --min-BQ 0 ...`) or,
when samtools exists, from the BAM with the reference's own command line except `--min-BQ 0`: the per-base BQ gate
of the AFF pass is applied inside the kernel, so ONE pileup serves both passes (`--tensor_can_fn_neg`)."""
import gzip
import shlex
import subprocess
import sys
from argparse import ArgumentParser

import numpy as np
import torch

from ._cli import add_ignored, add_unsupported, check_unsupported
from .fasta import read_region
from .featurize import featurize, alt_infos
from .pack import ColumnPack

FLANK, NPOS, MAX_INDEL, EXPAND_REF = 16, 33, 60, 1000        # shared/param.py:60,61,114,101


def read_candidates(bed_fn, ctg_name):
    """BED rows `ctg x-17 x+17 [type]` -> sorted centres, types, (ctg_start, ctg_end) as at :347-370."""
    opener = gzip.open if bed_fn.endswith(".gz") else open
    centres, ctg_start, ctg_end = {}, float("inf"), 0
    with opener(bed_fn, "rt") as f:
        for row in f:
            c = row.rstrip().split("\t")
            if len(c) < 3 or c[0] != ctg_name:
                continue
            position, end = int(c[1]) + 1, int(c[2]) + 1
            ctg_start, ctg_end = min(position, ctg_start), max(end, ctg_end)
            centre = end - FLANK - 2 if position < 1 else position + (end - position) // 2 - 1
            centres[centre] = c[3] if len(c) == 4 else "unknown"
    return centres, ctg_start, ctg_end


def read_candidate_positions(bed_fn, ctg_name):
    """read_candidates() for callers that only need the positions: sorted unique centres (int32 array), ctg_start, ctg_end.
    One C call on the file's bytes (cto_bed_centres) - the per-row Python loop above costs ~4 ms per 4 096-site chunk and,
    holding the interpreter lock, used to be what bounded a multi-threaded producer pool."""
    import ctypes as C
    from ._lib import lib, check
    opener = gzip.open if bed_fn.endswith(".gz") else open
    with opener(bed_fn, "rb") as f:
        raw = f.read()
    cap = raw.count(b"\n") + 1
    out = np.empty(max(cap, 1), dtype=np.int32)
    span = np.zeros(2, dtype=np.int64)
    has_types = C.c_int(0)
    buf = np.frombuffer(raw, dtype=np.uint8) if raw else np.zeros(1, dtype=np.uint8)
    n = int(lib.cto_bed_centres(buf.ctypes.data, len(raw), ctg_name.encode(), out.ctypes.data, cap, span.ctypes.data, C.byref(has_types)))
    check(n)
    if n == 0:
        return np.zeros(0, dtype=np.int32), float("inf"), 0
    return np.unique(out[:n]), int(span[0]), int(span[1])


def read_bed_intervals(bed_fn, ctg_name):
    """The BED rows of `ctg_name` as 0-based [begin, end) intervals (what `samtools mpileup -l` restricts positions to)."""
    opener = gzip.open if bed_fn.endswith(".gz") else open
    out = []
    with opener(bed_fn, "rt") as f:
        for row in f:
            c = row.rstrip().split("\t")
            if len(c) >= 3 and c[0] == ctg_name:
                out.append((max(0, int(c[1])), int(c[2])))
    return out


def load_pack(args, ref, ref_start, ctg_start, ctg_end, max_indel, device=None, stream=None):
    """The chunk's column pack from (in this order) `--mpileup_fn` text, the native BAM reader (`--bam_reader native`,
    csrc/bam.cpp, parity unpinned; `--bam_reader gpu`: the same with the BGZF blocks inflated on `device`, csrc/inflate.hip) or
    `samtools mpileup` run exactly as the reference runs it except for `--min-BQ 0` (create_tensor_pileup_calling.py:426-446):
    one pileup serves the AFF and the NEG pass."""
    if getattr(args, "mpileup_fn", None):
        if args.mpileup_fn.endswith(".gz"):
            with gzip.open(args.mpileup_fn, "rb") as f:
                return ColumnPack.from_mpileup(f.read(), ref, ref_start, max_indel)
        import os
        if os.path.getsize(args.mpileup_fn) == 0:
            return ColumnPack.from_mpileup(b"", ref, ref_start, max_indel)
        # plain text: mapped, not read - the tokeniser's threads pull the pages straight from the page cache
        return ColumnPack.from_mpileup(np.memmap(args.mpileup_fn, dtype=np.uint8, mode="r"), ref, ref_start, max_indel)
    ext_s, ext_e = max(1, ctg_start - NPOS), ctg_end + NPOS
    reader = getattr(args, "bam_reader", "samtools")
    if reader in ("native", "gpu"):
        from .platforms import warn_unpinned_bam_reader
        warn_unpinned_bam_reader(getattr(args, "platform", "ont"), reader)
        inflated = None
        if reader == "gpu":
            from .bgzf import inflate_span
            inflated = inflate_span(args.tumor_bam_fn, None, args.ctg_name, ext_s, ext_e,
                                    torch.device(device if device is not None else "cuda"), stream)
        return ColumnPack.from_bam(args.tumor_bam_fn, args.ctg_name, ext_s, ext_e, ref, ref_start,
                                   bed=read_bed_intervals(args.candidates_bed_regions, args.ctg_name),
                                   max_depth=args.max_depth if args.max_depth is not None else 8000, max_indel_length=max_indel,
                                   inflated=inflated)
    cmd = "{} mpileup --reverse-del --output-MQ -r {}:{}-{} --min-MQ 0 --min-BQ 0 -l {} --excl-flags 2316".format(
        args.samtools, args.ctg_name, ext_s, ext_e, args.candidates_bed_regions)
    if args.max_depth is not None:
        cmd += " --max-depth {}".format(args.max_depth)
    text = subprocess.run(shlex.split(cmd) + [args.tumor_bam_fn], stdout=subprocess.PIPE, check=True).stdout
    return ColumnPack.from_mpileup(text, ref, ref_start, max_indel)


def create_tensor(args, device="cuda"):
    centres, ctg_start, ctg_end = read_candidates(args.candidates_bed_regions, args.ctg_name)
    if not centres:
        print("[INFO] {} Tensors generated: 0".format(args.ctg_name))
        return 0
    sites = sorted(centres)
    ref_start = max(1, ctg_start - EXPAND_REF)
    ref = read_region(args.ref_fn, args.ctg_name, ref_start, ctg_end + EXPAND_REF)
    if not ref:
        sys.exit("[ERROR] Failed to load reference sequence from file ({}).".format(args.ref_fn))
    max_indel = MAX_INDEL if args.max_indel_length is None else args.max_indel_length
    pack = load_pack(args, ref, ref_start, ctg_start, ctg_end, max_indel, device=device)
    dp = pack.to_device(device)
    feat = featurize(dp, torch.tensor(sites, dtype=torch.int32, device=device), args.min_bq, 0, want_raw=True, want_x=False)
    torch.cuda.synchronize()
    info = feat.site_info.cpu().numpy()
    outs = [(args.tensor_can_fn, feat.raw_aff, 1)]
    if args.tensor_can_fn_neg:
        outs.append((args.tensor_can_fn_neg, feat.raw_neg, 2))
    alts_aff = alt_infos(feat, pack, info)
    n_written = 0
    for fn, raw, dcol in outs:
        raw = raw.cpu().numpy()
        alts = alts_aff if dcol == 1 else alt_infos(feat, pack, info, pass_idx=1)
        n_written = 0
        with (gzip.open(fn, "wt") if fn != "PIPE" else sys.stdout) as out:
            for i, pos in enumerate(sites):
                if info[i, 3] & 1:
                    continue
                o = pos - ref_start
                ref_seq = ref[max(0, o - FLANK): o + FLANK + 1]      # rows with pos - 16 < 1 never get here (flag bit 0)
                out.write("%s\t%d\t%s\t%s\t%s\t%s\t%s\n" % (
                    args.ctg_name, pos, ref_seq, " ".join("%d" % v for v in raw[i].ravel()), alts[i], centres[pos],
                    ref_seq[FLANK]))
                n_written += 1
    print("[INFO] {} Tensors generated: {}".format(args.ctg_name, n_written))
    return n_written


def build_parser():
    p = ArgumentParser(description="Generate tumor pileup tensors for calling (GPU featurisation)")
    p.add_argument("--platform", type=str, default="ont")
    p.add_argument("--tumor_bam_fn", type=str, default=None)
    p.add_argument("--mpileup_fn", type=str, default=None, help="samtools mpileup text (--min-BQ 0) instead of a BAM")
    p.add_argument("--ref_fn", type=str, required=True)
    p.add_argument("--tensor_can_fn", type=str, default="PIPE")
    p.add_argument("--tensor_can_fn_neg", type=str, default=None, help="also write the --min_bq 0 (NEG) tensor")
    p.add_argument("--ctg_name", type=str, required=True)
    p.add_argument("--samtools", type=str, default="samtools")
    p.add_argument("--bam_reader", type=str, default="samtools", choices=["samtools", "native", "gpu"],
                   help="'native': built-in BAM + BAI reader instead of a samtools subprocess (parity unpinned, see csrc/bam.cpp)")
    p.add_argument("--min_bq", type=int, default=0)
    p.add_argument("--max_depth", type=int, default=None,
                   help="EXPERIMENTAL: maximum tumor depth handed to samtools mpileup / the built-in reader (the reference's default: 8000). Engine limits, reported as an error of the chunk (CTO_EUNSUPPORTED), not silently: a pileup column may hold at most 32767 read-bases and 2048 distinct indel alleles")
    p.add_argument("--max_indel_length", type=int, default=None)
    p.add_argument("--candidates_bed_regions", type=str, required=True)
    # the rest of the reference's parser (src/create_tensor_pileup_calling.py:582-661).  With --candidates_bed_regions - the only form
    # run_clairs_to uses (:1228-1271) - the reference itself reads none of the `ignored` ones: the gates below are decode_pileup_bases
    # arguments that calling never consults (has_pileup_candidates, :238-242), --min_mq is overridden by a literal 0 (:421), --zstd
    # names the compressor of the tensor text (gzip here), --bed_fn is read into a variable nothing uses (:322).
    add_ignored(p, snv_min_af="float", indel_min_af="float", min_coverage="float", min_mq="int", zstd="str", bed_fn="str", ctg_start="int",
                ctg_end="int")
    add_unsupported(p, vcf_fn=("str", None), extend_bed=("str?", None), alt_fn=("str", None), truth_vcf_fn=("str", None), chunk_num=("int", None),
                    chunk_id=("int", None), phase_tumor=("bool", False), flanking=("int", None))
    return p


def main(argv=None):
    p = build_parser()
    a = p.parse_args(argv)
    check_unsupported(p, a)
    create_tensor(a)


if __name__ == "__main__":
    main()

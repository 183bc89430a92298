"""Candidate extraction on the GPU: counterpart of `clairs_to.py extract_candidates_calling`
(reference: src/extract_candidates_calling.py, STEP 1 of run_clairs_to:1194-1226).

`extract_candidates()` runs the gates on a column pack that is already in HBM - the same pack tensor creation uses, so
candidate sites become an internal product of the engine (one pileup instead of three).  `main()` mirrors the CLI at
its file seam: it writes the `<ctg>.<chunk>_<i>_<n>_snv` / `_indel` BED chunk files (<= 10 000 windows
`x-17 .. x+17` each) and the `SNV_CANDIDATES_FILE_*` list the next step reads."""
import ctypes as C
import gzip
import os
from argparse import ArgumentParser

import numpy as np
import torch

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


def write_bed_chunks(folder, ctg, chunk_id, positions, suffix, list_prefix):
    """The chunk files of extract_candidates_calling.py:450-488."""
    if not len(positions):
        return []
    os.makedirs(folder, exist_ok=True)
    n_regions = -(-len(positions) // SPLIT_BED_SIZE)
    paths = []
    for i in range(n_regions):
        part = positions[i * SPLIT_BED_SIZE:(i + 1) * SPLIT_BED_SIZE]
        path = os.path.join(folder, "{}.{}_{}_{}_{}".format(ctg, chunk_id, i, n_regions, suffix))
        with open(path, "w") as f:
            f.write("\n".join("\t".join([ctg, str(max(x - FLANK - 1, 1)), str(x + FLANK + 1)]) for x in part) + "\n")
        paths.append(path)
    with open(os.path.join(folder, "{}_{}_{}".format(list_prefix, ctg, chunk_id)), "w") as f:
        f.write("\n".join(paths) + "\n")
    return paths


def pack_of_region(a):
    """The MQ- and BQ-unfiltered pack of the chunk (the gates run on the device): pre-made text, the native BAM reader, or
    `samtools mpileup` - the reference's own command (extract_candidates_calling.py:298-309) but with `--min-MQ 0 --min-BQ 0
    --output-MQ` so that the same pack also serves tensor creation."""
    import shlex
    import subprocess
    if a.mpileup_fn:
        opener = gzip.open if a.mpileup_fn.endswith(".gz") else open
        with opener(a.mpileup_fn, "rb") as f:
            text = f.read()
        first = int(text.split(b"\t", 2)[1])
        last = int(text.rstrip(b"\n").rsplit(b"\n", 1)[-1].split(b"\t", 2)[1])
        ref_start = max(1, first - EXPAND_REF)
        ref = read_region(a.ref_fn, a.ctg_name, ref_start, last + EXPAND_REF)
        return ColumnPack.from_mpileup(text, ref, ref_start)
    if a.ctg_start is None or a.ctg_end is None:
        raise SystemExit("[ERROR] --ctg_start / --ctg_end are required with --tumor_bam_fn")
    ref_start = max(1, a.ctg_start - EXPAND_REF)
    ref = read_region(a.ref_fn, a.ctg_name, ref_start, a.ctg_end + EXPAND_REF)
    if a.bam_reader == "native":
        return ColumnPack.from_bam(a.tumor_bam_fn, a.ctg_name, a.ctg_start, a.ctg_end, ref, ref_start,
                                   max_depth=a.max_depth if a.max_depth is not None else 8000)
    cmd = "{} mpileup --reverse-del --output-MQ -r {}:{}-{} --min-MQ 0 --min-BQ 0 --excl-flags 2316".format(
        a.samtools, a.ctg_name, a.ctg_start, a.ctg_end)
    if a.max_depth is not None:
        cmd += " --max-depth {}".format(a.max_depth)
    text = subprocess.run(shlex.split(cmd) + [a.tumor_bam_fn], stdout=subprocess.PIPE, check=True).stdout
    return ColumnPack.from_mpileup(text, ref, ref_start)


def extract_to_files(a, device="cuda"):
    pack = pack_of_region(a)
    dp = pack.to_device(device)
    flags, _ = extract_candidates(dp, a.min_bq, a.min_mq, a.snv_min_af, a.indel_min_af, a.min_coverage,
                                  a.alternative_base_num, a.select_indel_candidates)
    snv = candidate_positions(dp, flags, 1).cpu().tolist()
    indel = candidate_positions(dp, flags, 2).cpu().tolist() if a.select_indel_candidates else []
    write_bed_chunks(a.candidates_folder, a.ctg_name, a.chunk_id, snv, "snv", "SNV_CANDIDATES_FILE")
    write_bed_chunks(a.candidates_folder, a.ctg_name, a.chunk_id, indel, "indel", "INDEL_CANDIDATES_FILE")
    return snv, indel


def main():
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
    p.add_argument("--chunk_id", type=int, default=None)
    p.add_argument("--snv_min_af", type=float, default=0.05)
    p.add_argument("--indel_min_af", type=float, default=1.0)
    p.add_argument("--min_coverage", type=float, default=4)
    p.add_argument("--min_mq", type=int, default=20)
    p.add_argument("--min_bq", type=int, default=0)
    p.add_argument("--alternative_base_num", type=int, default=3)
    p.add_argument("--select_indel_candidates", type=lambda v: str(v).lower() in ("yes", "true", "t", "y", "1"), default=False)
    a = p.parse_args()
    if not a.mpileup_fn and not a.tumor_bam_fn:
        raise SystemExit("[ERROR] one of --mpileup_fn / --tumor_bam_fn is required")
    extract_to_files(a)


if __name__ == "__main__":
    main()

"""All candidate chunks of a run on N GPUs: the multi-GPU form of STEP 2 of the reference orchestrator.

The reference hands its <= 10 000-site chunk files (`<ctg>.<i>_<n>_snv`, written by extract_candidates_calling.py:450-488 and
listed in CANDIDATES_FILES) to GNU parallel, four commands per chunk (run_clairs_to:1228-1308).  Here one process per GPU
(`python -m torch.distributed.run --nproc-per-node N -m clairs_to_amd call_chunks ...`, or a single process) takes a
contiguous share of the chunk list (dist.shard_range: neighbouring chunks stay on one GPU), keeps one Engine alive - the
checkpoints are read once - and writes `p_<chunk>.vcf` per chunk exactly as `pileup_call` does; there is no data-path
collective (sites are independent), only a barrier before rank 0 merges the chunk VCFs (`sort_vcf`) and optionally applies
`postprocess_vcf`.
"""
import os
import sys
from argparse import ArgumentParser, Namespace

import torch

from .dist import shard_range
from .pileup_call import add_common_arguments, make_engine, pileup_call
from .postprocess_vcf import postprocess_vcf, sort_vcf


def chunk_contig(bed_fn):
    """contig of a chunk file = first column of its first row (chunk files hold one contig, extract_candidates_calling.py:455)."""
    import gzip
    opener = gzip.open if bed_fn.endswith(".gz") else open
    with opener(bed_fn, "rt") as f:
        for row in f:
            if row.strip():
                return row.split("\t")[0]
    return None


def call_chunks(args):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # the control plane only needs a barrier: gloo keeps the GPUs' streams out of it
        dist.init_process_group("gloo")
    if not torch.cuda.is_available():
        sys.exit("[ERROR] clairs_to_amd call_chunks needs a HIP device; there is no CPU fallback")
    torch.cuda.set_device(local_rank % torch.cuda.device_count())
    device = torch.device("cuda", local_rank % torch.cuda.device_count())
    chunks = [r.strip() for r in open(args.chunk_list) if r.strip()]
    lo, hi = shard_range(len(chunks), world, rank)
    os.makedirs(args.output_dir, exist_ok=True)
    eng = make_engine(args, device)
    n_rows = 0

    def chunk_args(bed):
        ctg = chunk_contig(bed)
        if ctg is None:
            return None
        a = Namespace(**vars(args))
        a.candidates_bed_regions, a.ctg_name, a.mpileup_fn, a.predict_fn = bed, ctg, None, None
        a.call_fn = os.path.join(args.output_dir, "p_%s.vcf" % os.path.basename(bed))
        return a
    # two-stage pipeline: the pack of the next chunk is produced on a host thread (BAM decoding / samtools + tokenising run
    # outside the GIL) while the current chunk is on the GPU and its VCF rows are written
    from concurrent.futures import ThreadPoolExecutor
    from .pileup_call import prepare_chunk
    mine = [a for a in (chunk_args(b) for b in chunks[lo:hi]) if a is not None]
    with ThreadPoolExecutor(max_workers=1) as pool:
        nxt = pool.submit(prepare_chunk, mine[0]) if mine else None
        for i, a in enumerate(mine):
            prep = nxt.result()
            nxt = pool.submit(prepare_chunk, mine[i + 1]) if i + 1 < len(mine) else None
            n_rows += pileup_call(a, engine=eng, prepared=prep) if prep is not None else 0
    print("[INFO] rank %d/%d: chunks %d..%d, %d VCF records" % (rank, world, lo, hi, n_rows), file=sys.stderr)
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
    if rank == 0 and args.merged_vcf_fn:
        contigs = []
        for bed in chunks:
            c = chunk_contig(bed)
            if c is not None and c not in contigs:
                contigs.append(c)
        n = sort_vcf(args.output_dir, args.merged_vcf_fn, contigs, vcf_fn_prefix="p_", ref_fn=args.ref_fn, sample_name=args.sample_name)
        print("[INFO] merged %d records into %s" % (n, args.merged_vcf_fn), file=sys.stderr)
        if args.final_vcf_fn:
            from .platforms import resolve_platform
            postprocess_vcf(args.merged_vcf_fn, args.final_vcf_fn, platform=resolve_platform(args.platform)[1],
                            ref_fn=args.ref_fn, sample_name=args.sample_name)
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()
    return n_rows


def main():
    p = ArgumentParser(description="Pileup calling of all candidate chunks of a run, one process per GPU")
    add_common_arguments(p)
    p.add_argument("--chunk_list", type=str, required=True, help="file with one candidate BED chunk path per line (CANDIDATES_FILES)")
    p.add_argument("--output_dir", type=str, required=True, help="directory for the p_<chunk>.vcf files")
    p.add_argument("--merged_vcf_fn", type=str, default=None, help="rank 0: sort_vcf of all chunk VCFs")
    p.add_argument("--final_vcf_fn", type=str, default=None, help="rank 0: postprocess_vcf of the merged VCF")
    call_chunks(p.parse_args())


if __name__ == "__main__":
    main()

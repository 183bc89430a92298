"""All candidate chunks of a run on N GPUs: the multi-GPU form of STEP 2 of the reference orchestrator.

The reference hands its <= 10 000-site chunk files (`<ctg>.<i>_<n>_snv`, written by extract_candidates_calling.py:450-488 and
listed in CANDIDATES_FILES) to GNU parallel, four commands per chunk (run_clairs_to:1228-1308).  Here one process per GPU
(`python -m torch.distributed.run --nproc-per-node N -m clairs_to_amd call_chunks ...`, or a single process) takes a
contiguous share of the chunk list (dist.shard_range: neighbouring chunks stay on one GPU), keeps one Engine alive - the
checkpoints are read once - and writes `p_<chunk>.vcf` per chunk exactly as `pileup_call` does; there is no data-path
collective (sites are independent), only a status exchange before rank 0 merges the chunk VCFs (`sort_vcf`) and optionally
applies `postprocess_vcf`.

Inside a rank the chunks flow through a three-stage pipeline, several chunks in flight:
    producers (thread pool)   BED + reference slice + column pack (BAM decoding / samtools + tokenising; C code, GIL released)
                              and the pack's upload on a copy stream                                   pileup_call.prepare_chunk
    launcher (this thread)    wait for the upload event, 12 kernel launches, asynchronous copies of the per-site outputs into
                              re-used page-locked buffers                                               pileup_call.launch_chunk
    writers (thread pool)     alt_info strings + every VCF record in two C calls, file write            pileup_call.finish_chunk
The GPU needs ~2 ms per 4 096-site chunk; one producer delivers a chunk in 5 ms (mpileup text, 8 tokeniser threads) to 100 ms
(BAM, 2 decoding threads), so the rate is set by how many producers the host can run - `--producers` (default_producers()).

The same three stages exist in C (`cto_run_chunks`, csrc/pipeline.hip: native threads, one page-locked staging copy per chunk,
buffers kept from chunk to chunk) and are what `--pipeline auto` runs whenever the inputs are plain files - mpileup text
(`--mpileup_dir`, plain or .gz), a BAM through the built-in reader (`--bam_reader native`) or through a `samtools mpileup` child per chunk;
the Python-side device inflate (`--bam_reader gpu`) and the `--predict_fn` tap stay on the thread pools of this module.  Both write the same files, byte for byte.
"""
import os
import sys
import threading
from argparse import ArgumentParser, Namespace
from collections import deque
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

from .dist import shard_range
from .pileup_call import add_common_arguments, finish_chunk, launch_chunk, make_engine, prepare_chunk
from .platforms import resolve_platform
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


def usable_cores():
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = max(1, min(n, int(float(quota) / float(period) + 0.5)))
    except Exception:
        pass
    try:                                     # one process per GPU on the node: each rank plans with its share of the cores
        n = max(1, n // max(1, int(os.environ.get("LOCAL_WORLD_SIZE", "1"))))
    except ValueError:
        pass
    return n


def default_producers(native_bam, pipeline="python"):
    """pack-producer threads per rank: a quarter of the usable cores for mpileup text (the tokeniser saturates memory bandwidth early),
    half of them for the native BAM reader on the Python pipeline (inflate-bound, and its per-call serial parts - index, header,
    merge - want more calls in flight: 8 producers x 8 threads gave 280 k sites/s on 16 cores where 4 x 8 gave 218 k).  The C
    pipeline runs one BAM producer per usable core with two decoding threads each (pack_threads()): a call split over 8 threads
    decodes the long reads that straddle its 7 inner boundaries twice - 320 ms of CPU per 1 Mb x 50x chunk against 243 ms
    unsplit - and 16 x 2 measured 264 k sites/s where 8 x 8 gave 205 k; with DEVICE_INFLATE chunks in flight through the device
    inflate (their producers asleep meanwhile) 16 x 2 measured 390-490 k.  (20-24 producers gave 190-350 k: with more runnable
    threads than cores the chunks that wait for the device come back to a busy host.)"""
    if native_bam and pipeline == "native":
        # round 4: the device chunks' producers sleep while the device works (no spinning waits left in the pile-up driver: 55 ms of CPU
        # per chunk instead of 74), so a quarter more producers than cores keeps the cores busy: 16 / 20 / 24 producers on 16 cores
        # 664-708 / 736 / 576 k sites/s
        # round 6: never fewer than 8 - a device chunk costs its producer 14 ms of CPU and 25 ms of waiting (file read, inflate, pile-up), so a
        # rank with two cores to itself (an 8-GPU node's share) keeps the device busy with eight chunks in flight, not with two; eight stay
        # below the inflate contexts (DEVICE_INFLATE[1]), so none of them falls back to decoding on the host
        return max(1, min(40, max(usable_cores() + usable_cores() // 4, 8)))
    if pipeline == "native" and not native_bam:
        # mpileup text on the C pipeline: round 6's sweep on 16 cores (tools/experiments/writers_sweep.py, 4096-site chunks, M sites/s;
        # producers x writers): 4 x 2 1.91, 4 x 3 1.94, 4 x 6 1.97, 6 x 2 1.95, 6 x 3 1.99, 6 x 6 2.01 - of the networks' 2.16.  A chunk holds its
        # slot from tokenising to the written VCF, and with two writers the writers' 4.2 ms per chunk was the period
        return max(1, min(16, usable_cores() * 3 // 8))
    return max(1, min(16, usable_cores() // (2 if native_bam else 4)))


def default_writers():
    """VCF-writer threads per rank: a quarter of the usable cores, at least two (see default_producers)"""
    return max(2, min(8, usable_cores() // 4))


# BAM chunks on the C pipeline: up to DEVICE_INFLATE[1] chunks at a time have their BGZF blocks inflated on the GPU, on streams confined to
# DEVICE_INFLATE[0] of its 256 compute units (the networks keep the rest), the others on the host cores.  Measured on 64 chunk files, 16 usable
# cores, 16 producers (sites/s; profiles/round2_d_bam_hybrid.txt, two passes per setting on one box): host only 246-250 k; 128 CUs x 8 / 10
# chunks 347-402 / 408-429 k; 144 CUs 432-488 / 386-391 k; 160 CUs 368-396 / 323-328 k (507 k once on another box); all 64 chunks through the
# device (16 in flight) 191-385 k; 20+ producers 263-349 k (more runnable threads than cores).
# Round 4, with the 2.4x faster inflate kernel (32 chunk files, one process per setting): 112 / 128 / 144 / 160 CUs x 8 chunks 584 / 696 / 696 / 708 k;
# 144 CUs x 10 / 12 chunks 743 / 629 k; 160 x 10 739 k; 18 producers 714 k.
DEVICE_INFLATE = (144, 10)


def pack_threads(native_bam, pipeline="python"):
    """CTO_PACK_THREADS for the C producers (they split one chunk over up to 32 threads of their own): with several producers running
    side by side that oversubscribes a small host (16 usable cores on the bench box: 4 producers x 8 threads measured 870-960 k
    sites/s from text, 4 x 32 threads 660-715 k), so each call gets half of the usable cores - two for BAM chunks on the C pipeline"""
    return 2 if (native_bam and pipeline == "native") else max(2, usable_cores() // 2)


def run_pipeline(eng, chunk_args, producers=4, writers=2, depth=None, stats=None):
    """chunk_args: Namespaces as pileup_call takes them, in order.  Returns the number of VCF records written.
    `depth` bounds the chunks prepared ahead of the launcher (packs live in host + device memory while they wait)."""
    device = eng.device
    depth = depth if depth is not None else producers + 2
    n_rows = 0
    local = threading.local()
    from ._lib import lib
    per_call = 0 if "CTO_PACK_THREADS" in os.environ else pack_threads(False)      # the user's setting wins

    import time

    def clock(key, t0):
        if stats is not None:
            stats[key] = stats.get(key, 0.0) + time.perf_counter() - t0      # summed over threads: thread-seconds per stage

    def produce(a):
        t0 = time.perf_counter()
        if getattr(local, "stream", None) is None:
            local.stream = torch.cuda.Stream(device)            # one copy stream per producer thread
            lib.cto_set_pack_threads(per_call)                  # ... and its share of the cores for the C producers' own threads
        try:
            return prepare_chunk(a, device=device, copy_stream=local.stream)
        finally:
            clock("produce_s", t0)

    free_pinned = deque()                                        # page-locked buffer sets, re-used chunk after chunk

    def finish(a, prep, launched):
        t0 = time.perf_counter()
        try:
            return finish_chunk(a, eng.K, prep, launched)
        finally:
            free_pinned.append(launched["pinned"])
            clock("finish_s", t0)

    with ThreadPoolExecutor(max_workers=producers) as prod, ThreadPoolExecutor(max_workers=writers) as wr:
        pending, writing = deque(), deque()
        it = iter(chunk_args)

        def feed():
            while len(pending) < depth:
                a = next(it, None)
                if a is None:
                    return
                pending.append((a, prod.submit(produce, a)))
        feed()
        while pending:
            a, fut = pending.popleft()
            t0 = time.perf_counter()
            prep = fut.result()
            clock("launcher_waits_for_producer_s", t0)
            feed()
            if prep is None:
                print("[INFO] {} total processed positions: 0".format(a.ctg_name), file=sys.stderr)
                continue
            t0 = time.perf_counter()
            launched = launch_chunk(eng, prep, want_probs=bool(getattr(a, "predict_fn", None)) or getattr(a, "site_sink", None) is not None,
                                    pinned=free_pinned.popleft() if free_pinned else None)
            clock("launch_s", t0)
            writing.append(wr.submit(finish, a, prep, launched))
            if stats is not None:
                stats["sites"] = stats.get("sites", 0) + len(prep["sites"])
            while len(writing) > writers + 1:                    # bound the results waiting for a writer
                n_rows += writing.popleft().result()
        while writing:
            n_rows += writing.popleft().result()
    return n_rows


def native_eligible(chunk_args):
    """cto_run_chunks reads the files itself (BED and pileup text plain or .gz, BAM through the built-in reader or a `samtools
    mpileup` child process per chunk): not the Python-side device inflate (`--bam_reader gpu`), not the --predict_fn tap"""
    for a in chunk_args:
        if getattr(a, "predict_fn", None):
            return False
        if not getattr(a, "mpileup_fn", None) and getattr(a, "bam_reader", "samtools") not in ("native", "samtools"):
            return False
    return True


class RegionModes(object):
    """What REGION jobs take from extract_candidates_calling's optional inputs, per contig and read once: the confident BED's rows
    (--bed_fn; a path that does not exist is no BED, as for the reference, :205-210) as sorted merged int32 pairs, and the positions of
    the --hybrid_mode_vcf_fn / --genotyping_mode_vcf_fn records (read_known_vcf = VcfReader as :225-238 uses it)."""

    def __init__(self, args):
        from .extract_candidates_calling import merged_intervals, read_bed_rows, read_known_vcf
        self._rows, self._merged, self._known_of = read_bed_rows, merged_intervals, read_known_vcf
        bed = getattr(args, "bed_fn", None)
        self.bed_fn = bed if bed and os.path.exists(bed) else None
        self.known_fn = getattr(args, "hybrid_mode_vcf_fn", None) or getattr(args, "genotyping_mode_vcf_fn", None)
        self.select_indel = not args.disable_indel_calling
        self._conf, self._known, self.rows = {}, {}, {}

    def confident_rows(self, ctg):
        if self.bed_fn is None:
            return None
        if ctg not in self.rows:
            self.rows[ctg] = self._rows(self.bed_fn, ctg)
        return self.rows[ctg]

    def confident(self, ctg):
        rows = self.confident_rows(ctg)
        if rows is None:
            return None
        if ctg not in self._conf:
            self._conf[ctg] = np.ascontiguousarray(np.asarray(self._merged(rows, widen_empty=False), dtype=np.int32).reshape(-1))
        return self._conf[ctg]

    def known(self, ctg):
        if not self.known_fn:
            return None
        if ctg not in self._known:
            self._known[ctg] = np.ascontiguousarray(np.asarray(self._known_of(self.known_fn, ctg, self.select_indel)[0], dtype=np.int32))
        return self._known[ctg]


def region_rows(args):
    """--region_list rows -> (ctg, start, end) strings.  `ctg start end` as given; `ctg i/n` = --chunk_id i --chunk_num n of
    extract_candidates_calling (chunk_region: the .fai's length or, with --bed_fn, the span of the confident rows)."""
    from .extract_candidates_calling import chunk_region
    modes = RegionModes(args)
    out = []
    for r in open(args.region_list):
        c = r.split()
        if not c or r.startswith("#"):
            continue
        if len(c) >= 3:
            out.append((c[0], c[1], c[2]))
            continue
        if len(c) != 2 or "/" not in c[1]:
            sys.exit("[ERROR] --region_list: a row is `ctg start end` or `ctg i/n`: " + r.strip())
        i, n = (int(x) for x in c[1].split("/"))
        a = Namespace(chunk_id=i, chunk_num=n, ctg_name=c[0], ref_fn=args.ref_fn, bed_fn=modes.bed_fn, ctg_start=None, ctg_end=None)
        s, e, _ = chunk_region(a, modes.confident_rows(c[0]))
        out.append((c[0], str(s), str(e)))
    return out


def run_pipeline_native(eng, chunk_args, producers=4, writers=2, depth=None, stats=None, verbose=True, inflate_cus=None, inflate_jobs=None,
                        two_streams=False, device_tokenise=None):
    """run_pipeline() as ONE C call (cto_run_chunks, csrc/pipeline.hip): the same stages on native threads, with page-locked staging,
    buffers kept from chunk to chunk and - for BAM input - some chunks' BGZF blocks inflated on the device.  Same files, byte for byte.
    The kernels run on a stream of their own: the legacy default stream would synchronise with the CU-masked inflate streams."""
    import ctypes as C
    from ._lib import ChunkJob, RunCfg, RunStats, check, lib
    from .call_variants import chunk_vcf_header
    from .create_tensor_pileup_calling import MAX_INDEL
    chunk_args = list(chunk_args)
    if not chunk_args:
        return 0
    a0 = chunk_args[0]
    jobs = (ChunkJob * len(chunk_args))()
    keep = []                                  # the arrays the jobs point into, alive until the call returns
    for j, a in zip(jobs, chunk_args):
        os.makedirs(os.path.dirname(os.path.abspath(a.call_fn)), exist_ok=True)
        mp = getattr(a, "mpileup_fn", None)
        j.ctg_name, j.vcf_path = a.ctg_name.encode(), a.call_fn.encode()
        region = getattr(a, "region", None)
        if region is not None:                 # REGION job: no candidate BED - the candidates are extracted from the same pile-up
            j.bed_path, j.region_start, j.region_end = None, int(region[0]), int(region[1])
            j.candidates_path = a.candidates_out_fn.encode() if getattr(a, "candidates_out_fn", None) else None
            # the other modes of extract_candidates_calling: the job carries its contig's confident rows and hybrid / genotyping positions
            conf, known = getattr(a, "confident_intervals", None), getattr(a, "known_pos", None)
            if conf is not None:
                keep.append(conf)
                j.confident_intervals, j.n_confident_intervals, j.restrict_to_confident = conf.ctypes.data, len(conf) // 2, 1
            if known is not None and len(known):
                keep.append(known)
                j.known_pos, j.n_known_pos = known.ctypes.data, len(known)
            j.hybrid_info_path = a.hybrid_info_fn.encode() if getattr(a, "hybrid_info_fn", None) else None
        else:
            j.bed_path = a.candidates_bed_regions.encode()
        j.mpileup_path = mp.encode() if mp else None
        j.bam_path = None if mp else str(a.tumor_bam_fn).encode()
    cfg = RunCfg()
    cfg.aff, cfg.neg = eng.h_aff, eng.h_neg
    cfg.d_lik, cfg.d_edges = eng.posterior.lik.data_ptr(), eng.posterior.edges.data_ptr()
    cfg.K, cfg.min_bq, cfg.min_rescale_cov = eng.K, eng.min_bq, eng.min_rescale_cov or 0
    cfg.max_indel_length = MAX_INDEL if getattr(a0, "max_indel_length", None) is None else a0.max_indel_length
    cfg.max_depth = 8000 if getattr(a0, "max_depth", None) is None else a0.max_depth
    cfg.neg_reads_aff, cfg.show_ref, cfg.verbose = int(eng.neg_reads_aff), int(bool(a0.show_ref)), int(bool(verbose))
    cfg.qual_pass = -1.0 if a0.qual is None else float(a0.qual)
    cfg.ref_fa = str(a0.ref_fn).encode()
    cfg.vcf_header = chunk_vcf_header(str(a0.ref_fn), eng.K, a0.sample_name).encode()
    cfg.producers, cfg.writers, cfg.depth = int(producers), int(writers), int(depth or os.environ.get("CTO_PIPELINE_DEPTH", 0))
    if not getattr(a0, "mpileup_fn", None) and getattr(a0, "bam_reader", "samtools") == "samtools":
        cfg.samtools = str(a0.samtools).encode()                  # the reference's producer, one child process per chunk
        cfg.samtools_max_depth = int(a0.max_depth or 0)
    # threads per producer call: the environment's CTO_PACK_THREADS if the user set one, else this module's plan for the host
    cfg.pack_threads = 0 if "CTO_PACK_THREADS" in os.environ else pack_threads(not getattr(a0, "mpileup_fn", None), "native")
    cfg.inflate_cus = DEVICE_INFLATE[0] if inflate_cus is None else int(inflate_cus)       # only BAM jobs use it
    cfg.inflate_jobs = DEVICE_INFLATE[1] if inflate_jobs is None else int(inflate_jobs)
    cfg.device_pileup = int(os.environ.get("CTO_DEVICE_PILEUP", "1") != "0")     # the device-inflated chunks are piled up on the device too
    # mpileup text (files or the samtools child's output) can go up as it is and be tokenised on the device (csrc/tokenise.hip; texts its
    # single pass declines are tokenised on the host; same packs).  It trades GPU time for host cores - ~0.27 ms of kernels per 22 MB chunk
    # beside the networks' 1.9 ms, against 28 ms of host CPU - so the default follows the cores this rank has: with 16 cores to one GPU the
    # host tokeniser keeps the GPU for the networks (measured 1.9 M sites/s against 1.6-1.74 M); the host form is bound by cores / 28 ms
    # per chunk (12 cores: 1.7 M sites/s, 4 cores: 0.58 M), so up to 12 cores per rank - an 8-GPU node's share - the device form is used.
    # CTO_DEVICE_TOKENISE=0|1 / device_tokenise=False|True decide by hand.
    if device_tokenise is None:
        env = os.environ.get("CTO_DEVICE_TOKENISE")
        device_tokenise = (env != "0") if env is not None else usable_cores() <= 12
    cfg.device_tokenise = int(bool(device_tokenise))
    # gates of REGION jobs = extract_candidates_calling's options as run_clairs_to:1196-1220 passes them
    from .synth import PLATFORMS as _PF
    fam = resolve_platform(getattr(a0, "platform", "ont"))[1]
    cfg.extract_min_mq = int(getattr(a0, "extract_min_mq", None) or 20)
    cfg.extract_min_bq = int(eng.min_bq if getattr(a0, "extract_min_bq", None) is None else a0.extract_min_bq)
    cfg.alt_base_num = int(getattr(a0, "alternative_base_num", None) or 3)
    cfg.snv_min_af = float(getattr(a0, "snv_min_af", None) or 0.05)
    cfg.indel_min_af = float(getattr(a0, "indel_min_af", None) or _PF.get(fam, _PF["ont"])["indel_min_af"])
    cfg.min_coverage = float(4 if getattr(a0, "min_coverage", None) is None else a0.min_coverage)
    ib = getattr(a0, "call_indels_only_in_these_regions", None)
    cfg.indel_regions_bed = str(ib).encode() if ib and os.path.exists(str(ib)) else None       # a path that does not exist is no BED (:205-210)
    cfg.indel_bed_superseded = int(bool(getattr(a0, "bed_fn_source", None)))
    if two_streams:                               # consecutive chunks on two compute streams (a second pair of handles of the same weights)
        with torch.cuda.device(eng.device):
            cfg.aff2, cfg.neg2 = eng.aff._handle2(), eng.neg._handle2()
    st = RunStats()
    with torch.cuda.device(eng.device):
        # CTO_MAIN_PRIORITY=1: the compute stream at high priority.  Measured at the end of round 6: with BAM input the networks' launches then
        # go ahead of the inflate / pile-up kernels they share the chip with (BAM -> VCF 0.72-0.73 -> 0.74-0.78 M sites/s, REGION jobs 0.57 ->
        # 0.61 M) - but torch creates its whole pool of high-priority streams with the first one, their hardware queues stay for the life of
        # the process, and beside the CU-masked inflate streams' queues that is more queues than the chip keeps resident: runs that follow in
        # the same process lose (10 000-site text chunks 1.95 -> 1.31 M, all-device BAM on two cores 0.55 -> 0.48 M).  Not the default.
        main = torch.cuda.Stream(eng.device, priority=-1 if os.environ.get("CTO_MAIN_PRIORITY") == "1" else 0)
        main.wait_stream(torch.cuda.current_stream())
        rc = lib.cto_run_chunks(C.byref(cfg), jobs, len(chunk_args), C.c_void_p(main.cuda_stream), C.byref(st))
        main.synchronize()
    sys.stdout.flush()
    check(rc)
    if stats is not None:
        for k, v in (("sites", st.candidates), ("produce_s", st.produce_s), ("finish_s", st.finish_s), ("launch_s", st.launch_s),
                     ("launcher_waits_for_producer_s", st.launcher_wait_s), ("pack_s", st.pack_s), ("upload_s", st.upload_s), ("device_s", st.device_s), ("device_inflated", st.device_inflated), ("device_piled", st.device_piled), ("device_tokenised", st.device_tokenised), ("low_coverage", st.low_coverage), ("clamped", st.clamped)):
            stats[k] = stats.get(k, 0) + v
    return int(st.rows)


def call_chunks(args):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import datetime
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # the control plane only exchanges a status word: gloo keeps the GPUs' streams out of it; the timeout covers shards that
        # are legitimately hours out of balance (a failed rank reports through the status exchange below, not by timing out)
        # --gather_outputs adds the data-path exchange of the per-site outputs: device tensors over nccl (= RCCL over xGMI), the status
        # objects stay on gloo.  CTO_GATHER_BACKEND=gloo (tests on a one-GPU box) sends the rows through host memory instead.
        gather_backend = os.environ.get("CTO_GATHER_BACKEND", "nccl") if getattr(args, "gather_outputs", False) else None
        dist.init_process_group("cpu:gloo,cuda:nccl" if gather_backend == "nccl" else "gloo", timeout=datetime.timedelta(hours=24))
    if not torch.cuda.is_available():
        sys.exit("[ERROR] clairs_to_amd call_chunks needs a HIP device; there is no CPU fallback")
    torch.cuda.set_device(local_rank % torch.cuda.device_count())
    device = torch.device("cuda", local_rank % torch.cuda.device_count())
    region_mode = bool(getattr(args, "region_list", None))
    if region_mode:
        # REGION jobs: rows `ctg start end` (1-based, inclusive: the --ctg_start / --ctg_end of extract_candidates_calling); no BED
        # or rows `ctg i/n`: part i (1-based) of n equal parts of the contig's length from the .fai - with --bed_fn, of the span of its rows
        # (--chunk_id / --chunk_num, extract_candidates_calling.py:240-270)
        chunks = region_rows(args)
    else:
        chunks = [r.strip() for r in open(args.chunk_list) if r.strip()]
    lo, hi = shard_range(len(chunks), world, rank)
    os.makedirs(args.output_dir, exist_ok=True)
    n_rows, failure = 0, None
    gathering = bool(getattr(args, "gather_outputs", False))
    collected = {}                       # chunk index -> the chunk's per-site outputs (finish_chunk's site_sink)

    def region_name(r):
        return "%s_%s_%s" % (r[0], r[1], r[2])
    modes = RegionModes(args) if region_mode else None

    def chunk_args(bed):
        if region_mode:
            a = Namespace(**vars(args))
            a.candidates_bed_regions, a.ctg_name, a.predict_fn, a.region = None, bed[0], None, (int(bed[1]), int(bed[2]))
            mp_dir = getattr(args, "mpileup_dir", None)
            a.mpileup_fn = os.path.join(mp_dir, region_name(bed) + ".mpileup") if mp_dir else None
            a.call_fn = os.path.join(args.output_dir, "p_%s.vcf" % region_name(bed))
            cd = getattr(args, "candidates_dir", None)
            a.candidates_out_fn = os.path.join(cd, region_name(bed) + (".snv" if args.disable_indel_calling else ".indel")) if cd else None
            a.confident_intervals, a.known_pos = modes.confident(bed[0]), modes.known(bed[0])
            a.hybrid_info_fn = os.path.join(cd, region_name(bed) + "_hybrid_info") if cd and modes.known_fn else None
            return a
        ctg = chunk_contig(bed)
        if ctg is None:
            return None
        a = Namespace(**vars(args))
        a.candidates_bed_regions, a.ctg_name, a.predict_fn = bed, ctg, None
        mp_dir = getattr(args, "mpileup_dir", None)
        a.mpileup_fn = os.path.join(mp_dir, os.path.basename(bed) + ".mpileup") if mp_dir else None
        a.call_fn = os.path.join(args.output_dir, "p_%s.vcf" % os.path.basename(bed))
        return a
    try:
        eng = make_engine(args, device)
        mine = []
        for ci, b in enumerate(chunks[lo:hi]):
            a = chunk_args(b)
            if a is None:
                continue
            if gathering:
                a.site_sink = (lambda rec, ci=lo + ci, ctg=a.ctg_name: collected.__setitem__(ci, dict(rec, ctg=ctg)))
            mine.append(a)
        for a in mine:                       # a chunk VCF left by an earlier run must not survive into this run's merge
            if os.path.exists(a.call_fn):
                os.remove(a.call_fn)
        how = getattr(args, "pipeline", None) or "auto"
        if gathering:
            if region_mode or how == "native":
                sys.exit("[ERROR] --gather_outputs runs on the thread-pool pipeline of this module (--pipeline python, --chunk_list): "
                         "cto_run_chunks keeps the per-site outputs to itself")
            how = "python"
        if region_mode:
            if how == "python" or not native_eligible(mine):
                sys.exit("[ERROR] --region_list runs in the C pipeline only (cto_run_chunks): --mpileup_dir text, --bam_reader native or samtools")
            how = "native"
            if getattr(args, "candidates_dir", None):
                os.makedirs(args.candidates_dir, exist_ok=True)
        if how == "native" and not native_eligible(mine):
            sys.exit("[ERROR] --pipeline native does not do --bam_reader gpu or --predict_fn")
        native = how == "native" or (how == "auto" and native_eligible(mine))
        run = run_pipeline_native if native else run_pipeline
        reader = None if getattr(args, "mpileup_dir", None) else getattr(args, "bam_reader", "samtools")
        from .platforms import warn_unpinned_bam_reader
        warn_unpinned_bam_reader(getattr(args, "platform", "ont"), reader)
        if getattr(args, "producers", None):
            producers = args.producers
        elif reader == "samtools":        # one `samtools mpileup` child per producer (a core each; the producer sleeps on its pipe)
            producers = max(1, min(64, usable_cores()))
        else:
            producers = default_producers(reader in ("native", "gpu"), "native" if native else "python")
        kw = dict(inflate_cus=getattr(args, "device_inflate_cus", None)) if native else {}
        n_rows = run(eng, mine, producers=producers, writers=getattr(args, "writers", None) or default_writers(), **kw)
    except (Exception, SystemExit) as e:     # a bad reference, a corrupt BAM, CTO_EUNSUPPORTED ...: report, do not leave the others waiting
        failure = "%s: %s" % (type(e).__name__, e)
        print("[ERROR] rank %d/%d failed: %s" % (rank, world, failure), file=sys.stderr)
    print("[INFO] rank %d/%d: chunks %d..%d, %d VCF records" % (rank, world, lo, hi, n_rows), file=sys.stderr)
    if world > 1:
        import torch.distributed as dist
        status = [None] * world
        dist.all_gather_object(status, failure)       # doubles as the barrier before the merge
        failed = [(r, s) for r, s in enumerate(status) if s]
    else:
        failed = [(0, failure)] if failure else []
    if failed:
        if world > 1:
            import torch.distributed as dist
            dist.destroy_process_group()
        sys.exit("[ERROR] call_chunks: %s" % "; ".join("rank %d: %s" % f for f in failed))
    if gathering:
        n_g = gather_and_write(args, collected, chunks, world, rank, device)
        if rank == 0:
            print("[INFO] gathered the outputs of %d sites from %d rank(s); %d records in %s" % (n_g[0], world, n_g[1], args.merged_vcf_fn), file=sys.stderr)
        if rank == 0 and args.final_vcf_fn:
            final_vcf(args)
    elif rank == 0 and args.merged_vcf_fn:
        contigs = []
        for bed in chunks:
            c = bed[0] if region_mode else chunk_contig(bed)
            if c is not None and c not in contigs:
                contigs.append(c)
        # merge exactly the chunk VCFs of THIS chunk list (a stale p_*.vcf of another run in the same directory stays out)
        names = ["p_%s.vcf" % (region_name(b) if region_mode else os.path.basename(b)) for b in chunks]
        n = sort_vcf(args.output_dir, args.merged_vcf_fn, contigs, vcf_fn_prefix="p_", ref_fn=args.ref_fn, sample_name=args.sample_name,
                     only_files=names)
        print("[INFO] merged %d records into %s" % (n, args.merged_vcf_fn), file=sys.stderr)
        if args.final_vcf_fn:
            final_vcf(args)
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()
    return n_rows


def final_vcf(args):
    """--final_vcf_fn: postprocess_vcf of the merged VCF with run_clairs_to's gates (:1519-1529: --qual, the two region cut-offs, --af,
    --cmdline = the file holding the command line for the header); a gate left out is the platform's default, as in the reference."""
    cmd = None
    if args.cmdline is not None and os.path.exists(args.cmdline):
        cmd = open(args.cmdline).read().rstrip()
    return postprocess_vcf(args.merged_vcf_fn, args.final_vcf_fn, platform=resolve_platform(args.platform)[1], qual=args.postprocess_qual,
                           qual_cutoff_phaseable_region=args.postprocess_qual_cutoff_phaseable_region,
                           qual_cutoff_unphaseable_region=args.postprocess_qual_cutoff_unphaseable_region, af=args.postprocess_af, ref_fn=args.ref_fn,
                           sample_name=args.sample_name, cmdline=cmd)


def gather_and_write(args, collected, chunks, world, rank, device):
    """--gather_outputs: the exchange step north_star names, in a real run.  Every rank's per-site outputs - probabilities [2K][2],
    decision, QUAL, strand / depth words, the candidate's position and reference base, its alt_info string - are all_gathered in
    rank-major order (= the chunk list's order: ranks hold contiguous runs of it) and rank 0 formats the merged VCF from the
    gathered buffer with the same C call that formats a chunk's (cto_vcf_rows_batch), in sort_vcf's order.  The reference's
    equivalent is files: p_<chunk>.vcf per GNU-parallel job, then sort_vcf over the directory (run_clairs_to:1293-1317).
    Returns (sites gathered, records written)."""
    import numpy as np
    import torch.distributed as dist
    from .call_variants import vcf_rows_batch
    from .dist import gather_site_rows
    from .call_variants import chunk_vcf_header
    from .postprocess_vcf import contig_order, _header_from_fai
    K = 4 if args.disable_indel_calling else 6
    recs = [collected[i] for i in sorted(collected)]
    names = sorted({c for c in (chunk_contig(b) for b in chunks) if c is not None})        # the same on every rank
    on_gpu = world > 1 and os.environ.get("CTO_GATHER_BACKEND", "nccl") == "nccl"
    dev = device if on_gpu else torch.device("cpu")

    def cat(key, dtype, tail):
        parts = [np.asarray(r[key]).reshape((-1,) + tail) for r in recs]
        a = np.concatenate(parts) if parts else np.zeros((0,) + tail, dtype=dtype)
        return torch.from_numpy(np.ascontiguousarray(a.astype(dtype, copy=False))).to(dev)
    n_local = sum(len(r["pos"]) for r in recs)
    ctg_idx = np.concatenate([np.full(len(r["pos"]), names.index(r["ctg"]), dtype=np.int64) for r in recs]) if recs else np.zeros(0, dtype=np.int64)
    alt_max = max([int(r["alt_len"].max()) if len(r["alt_len"]) else 0 for r in recs] + [1])
    if world > 1:
        m = torch.tensor([alt_max], dtype=torch.int64, device=dev)
        dist.all_reduce(m, op=dist.ReduceOp.MAX)
        alt_max = int(m.item())
    alt = np.zeros((n_local, alt_max), dtype=np.uint8)
    row = 0
    for r in recs:
        off = np.concatenate([[0], np.cumsum(r["alt_len"])])
        buf = np.frombuffer(r["alt_buf"], dtype=np.uint8)
        for i in range(len(r["pos"])):
            alt[row, : r["alt_len"][i]] = buf[off[i]:off[i + 1]]
            row += 1
    local = {"key": torch.from_numpy(np.stack([ctg_idx, np.concatenate([r["pos"] for r in recs]) if recs else np.zeros(0, dtype=np.int64)], axis=1)).to(dev),
             "centre": cat("centre", np.uint8, ()), "info": cat("info", np.int32, (12,)), "decision": cat("decision", np.int32, (4,)),
             "qual": cat("qual", np.float64, ()), "probs": cat("probs", np.float32, (2 * K, 2)), "alt_len": cat("alt_len", np.int32, ()),
             "alt": torch.from_numpy(alt).to(dev)}
    if world > 1:
        got = {k: gather_site_rows(v)[0] for k, v in local.items()}
        # the exchange is checked where it is cheap: this rank's block sits bit for bit at its offset of the gathered probabilities
        counts = gather_site_rows(local["qual"])[1]
        base = sum(counts[:rank])
        ok = torch.tensor([int(torch.equal(got["probs"][base:base + n_local], local["probs"]))], dtype=torch.int64, device=dev)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) != 1:
            sys.exit("[ERROR] call_chunks --gather_outputs: a rank's block is not where the gathered buffer should hold it")
    else:
        got = local
    n_sites, n_records = int(got["qual"].shape[0]), 0
    if rank == 0:
        g = {k: v.cpu().numpy() for k, v in got.items()}
        if getattr(args, "gathered_probs_fn", None):
            np.save(args.gathered_probs_fn, g["probs"])
        body = []
        for ctg in contig_order(list(names)):
            sel = np.nonzero(g["key"][:, 0] == names.index(ctg))[0]
            if not len(sel):
                continue
            al = g["alt_len"][sel].astype(np.int64)
            off = np.concatenate([[0], np.cumsum(al)]).astype(np.int64)
            buf = b"".join(g["alt"][i, : g["alt_len"][i]].tobytes() for i in sel)
            text, cnt = vcf_rows_batch(ctg, np.ascontiguousarray(g["key"][sel, 1]), np.ascontiguousarray(g["centre"][sel]), buf, off,
                                       np.ascontiguousarray(g["info"][sel]), np.ascontiguousarray(g["decision"][sel]),
                                       np.ascontiguousarray(g["qual"][sel]), K, show_ref=args.show_ref, qual_pass=args.qual)
            at = {}
            for r in text.split("\n"):
                if r:
                    at[int(r.split("\t", 2)[1])] = r + "\n"
            body.extend(at[p] for p in sorted(at))
            n_records += len(at)
        os.makedirs(os.path.dirname(os.path.abspath(args.merged_vcf_fn)), exist_ok=True)
        with open(args.merged_vcf_fn, "w") as out:
            if n_records == 0:
                out.write(_header_from_fai(args.ref_fn, args.sample_name))
            else:
                out.write(chunk_vcf_header(args.ref_fn, K, args.sample_name))
                out.write("".join(body))
    return n_sites, n_records


def main(argv=None):
    p = ArgumentParser(description="Pileup calling of all candidate chunks of a run, one process per GPU")
    add_common_arguments(p)
    p.add_argument("--chunk_list", type=str, default=None, help="file with one candidate BED chunk path per line (CANDIDATES_FILES)")
    p.add_argument("--region_list", type=str, default=None,
                   help="instead of --chunk_list: rows `ctg start end` (1-based, inclusive) or `ctg i/n` (part i of n of the contig: extract_candidates_calling's "
                        "--chunk_id i --chunk_num n, from the .fai or - with --bed_fn - from the span of its rows). No candidate BEDs: every region is piled up once "
                        "and candidate extraction (extract_candidates_calling's gates) runs on that pile-up in HBM, in front of tensor creation")
    p.add_argument("--candidates_dir", type=str, default=None, help="--region_list: also write each region's candidates as BED window rows here")
    p.add_argument("--snv_min_af", type=float, default=0.05, help="--region_list: extract_candidates_calling --snv_min_af")
    p.add_argument("--indel_min_af", type=float, default=None, help="--region_list: --indel_min_af (default: the platform's)")
    p.add_argument("--min_coverage", type=float, default=4, help="--region_list: --min_coverage")
    p.add_argument("--alternative_base_num", type=int, default=3, help="--region_list: --alternative_base_num")
    p.add_argument("--extract_min_mq", type=int, default=20, help="--region_list: --min_mq of extract_candidates_calling")
    p.add_argument("--extract_min_bq", type=int, default=None, help="--region_list: --min_bq of extract_candidates_calling (default: --min_bq, as run_clairs_to passes the same value to both)")
    p.add_argument("--bed_fn", type=str, default=None,
                   help="--region_list: confident regions (extract_candidates_calling --bed_fn, :249-260, 302): positions outside its rows have no pileup row - "
                        "no candidate there, and `ctg i/n` rows of --region_list cut the span of its rows; a missing file is no BED")
    p.add_argument("--bed_fn_source", type=str, default=None, help="--region_list: the user's own --bed_fn, if any: it supersedes --call_indels_only_in_these_regions (:438)")
    p.add_argument("--hybrid_mode_vcf_fn", type=str, default=None,
                   help="--region_list: positions of this VCF's records are candidates whenever they show an alternative base / an indel, AF gates or not (:347-349, 370-383)")
    p.add_argument("--genotyping_mode_vcf_fn", type=str, default=None, help="--region_list: the same list under the reference's other name (:225-238)")
    p.add_argument("--call_indels_only_in_these_regions", type=str, default=None,
                   help="--region_list, indel mode: keep an indel candidate only inside the rows of this BED (extract_candidates_calling.py:437-446)")
    p.add_argument("--output_dir", type=str, required=True, help="directory for the p_<chunk>.vcf files")
    p.add_argument("--merged_vcf_fn", type=str, default=None, help="rank 0: sort_vcf of all chunk VCFs")
    p.add_argument("--final_vcf_fn", type=str, default=None, help="rank 0: postprocess_vcf of the merged VCF")
    # (--qual is call_variants' option here, as in pileup_call: the gates of the final step carry postprocess_vcf's names behind a prefix)
    p.add_argument("--postprocess_qual", type=float, default=None, help="--final_vcf_fn: postprocess_vcf --qual (default: the platform's)")
    p.add_argument("--postprocess_qual_cutoff_phaseable_region", type=float, default=None, help="--final_vcf_fn: postprocess_vcf --qual_cutoff_phaseable_region")
    p.add_argument("--postprocess_qual_cutoff_unphaseable_region", type=float, default=None, help="--final_vcf_fn: postprocess_vcf --qual_cutoff_unphaseable_region")
    p.add_argument("--postprocess_af", type=float, default=None, help="--final_vcf_fn: postprocess_vcf --af (default: the platform's)")
    p.add_argument("--cmdline", type=str, default=None, help="--final_vcf_fn: file holding the command line for the ##cmdline header row (tmp/CMD)")
    p.add_argument("--gather_outputs", action="store_true",
                   help="exchange step in the data path: every rank's per-site outputs (probabilities, decision, QUAL, counts, alt_info) are "
                        "all_gathered in rank-major = chunk-list order (RCCL over xGMI) and rank 0 writes --merged_vcf_fn from the gathered "
                        "buffer instead of merging the p_<chunk>.vcf files (which are still written); same file either way")
    p.add_argument("--gathered_probs_fn", type=str, default=None, help="--gather_outputs: rank 0 also saves the gathered probabilities [sites][2K][2] (.npy)")
    p.add_argument("--mpileup_dir", type=str, default=None,
                   help="read <dir>/<chunk file name>.mpileup (samtools mpileup --min-BQ 0 text of the chunk) instead of the BAM")
    p.add_argument("--producers", type=int, default=None, help="pack-producer threads per rank (default: usable cores / 4, / 2 with --bam_reader native; <= 16)")
    p.add_argument("--writers", type=int, default=None, help="VCF-writer threads per rank (default: a quarter of the usable cores, at least 2)")
    p.add_argument("--device_inflate_cus", type=int, default=None,
                   help="C pipeline, BAM input: compute units the device BGZF inflate is confined to (default %d; 0: inflate on the host only)" % DEVICE_INFLATE[0])
    p.add_argument("--pipeline", type=str, default="auto", choices=["auto", "native", "python"],
                   help="'native': the chunk loop as one C call (cto_run_chunks; plain-text inputs, --mpileup_dir or --bam_reader native); "
                        "'python': the thread-pool pipeline of this module; 'auto': native when the inputs allow it")
    args = p.parse_args(argv)
    if bool(args.chunk_list) == bool(args.region_list):
        p.error("exactly one of --chunk_list / --region_list is required")
    if args.gather_outputs and not args.merged_vcf_fn:
        p.error("--gather_outputs writes --merged_vcf_fn from the gathered outputs: name it")
    call_chunks(args)


if __name__ == "__main__":
    main()

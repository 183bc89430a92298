"""File-to-file throughput of the hot path: candidate chunk files + pileup source (mpileup text or BAM) -> p_<chunk>.vcf,
through the same pipeline `call_chunks` runs (producers -> launcher -> writers: cto_run_chunks in C, or call_chunks.run_pipeline
on Python thread pools for the inputs the C loop does not read itself), with a resident Engine.  Everything a real run pays is inside the timed region: reading the BED / FASTA / text or BAM from disk, tokenising or
BAM decoding, the pack upload over PCIe, the 12 kernels, the device-to-host copies, the alt_info strings, the VCF rows and the
file writes.  Used by bench.py (`e2e` object of its JSON line; never its `value`) and tools/e2e_bench.py."""
import os
import resource
import shutil
import tempfile
import time
from argparse import Namespace


def chunk_namespaces(run, out_dir, bam=False, bam_reader="native"):
    """one pileup_call Namespace per chunk file of a synthetic run directory (synth_run.make_text_run / make_bam_run)"""
    from .call_chunks import chunk_contig
    out = []
    for bed in run["chunks"]:
        a = Namespace(platform="ont", ref_fn=run["ref_fn"], samtools="samtools", bam_reader=bam_reader if bam else "samtools",
                      tumor_bam_fn=run.get("bam_fn"), min_bq=None, max_depth=None, max_indel_length=None, min_rescale_cov=50,
                      disable_indel_calling=True, sample_name="SAMPLE", show_ref=False, qual=0, pileup=True, predict_fn=None,
                      output_dir=out_dir)
        a.candidates_bed_regions, a.ctg_name = bed, chunk_contig(bed)
        a.mpileup_fn = None if bam else os.path.join(run["mpileup_dir"], os.path.basename(bed) + ".mpileup")
        a.call_fn = os.path.join(out_dir, "p_%s.vcf" % os.path.basename(bed))
        out.append(a)
    return out


def region_namespaces(run, out_dir, n_regions, bam_reader="native"):
    """REGION jobs over the BAM of a synthetic run: the contig cut into n_regions ranges, no candidate BEDs - candidate extraction is
    an internal product of the run (cto_run_chunks REGION jobs; call_chunks --region_list)"""
    L = run["contig_len"]
    per = (L + n_regions - 1) // n_regions
    out = []
    for i in range(n_regions):
        a = Namespace(platform="ont", ref_fn=run["ref_fn"], samtools="samtools", bam_reader=bam_reader, tumor_bam_fn=run["bam_fn"], min_bq=None,
                      max_depth=None, max_indel_length=None, min_rescale_cov=50, disable_indel_calling=True, sample_name="SAMPLE", show_ref=False,
                      qual=0, pileup=True, predict_fn=None, output_dir=out_dir, candidates_bed_regions=None, mpileup_fn=None, ctg_name=run["ctg"],
                      region=(max(1, i * per + 1), min(L, (i + 1) * per)))
        a.call_fn = os.path.join(out_dir, "p_%s_%d_%d.vcf" % (a.ctg_name, a.region[0], a.region[1]))
        out.append(a)
    return out


def build_run(d, kind="text", n_chunks=16, sites_per_chunk=4096, distinct=3, region_kb=None):
    """synthetic run directory under `d` (input synthesis, untimed) -> (run dict, description)"""
    from .synth_run import make_bam_run, make_text_run
    if kind == "text":
        run = make_text_run(os.path.join(d, "run_text"), n_chunks=n_chunks, sites_per_chunk=sites_per_chunk, distinct=distinct)
        mp = os.path.join(run["mpileup_dir"], os.path.basename(run["chunks"][0]) + ".mpileup")
        return run, "samtools-mpileup text, %d chunk files x %d candidates (%d distinct pileups), %.1f MB of text per chunk" % (
            n_chunks, sites_per_chunk, distinct, os.path.getsize(mp) / 1e6)
    region_kb = region_kb if region_kb is not None else max(200, n_chunks * sites_per_chunk // 4)
    run = make_bam_run(os.path.join(d, "run_bam"), region_kb=region_kb, n_chunks=n_chunks)
    return run, "synthetic 50x long-read BAM + BAI over %d kb (%.0f MB), %d chunk files, candidates every 250 bp, native BAM reader" % (
        region_kb, os.path.getsize(run["bam_fn"]) / 1e6, len(run["chunks"]))


def time_run(eng, run, kind, out_dir, producers=None, writers=2, repeats=4, bam_reader="native", pipeline="python", inflate_cus=None, inflate_jobs=None, two_streams=False,
             regions=0, times=1, device_tokenise=None):
    """the pipeline over a prepared run directory -> dict(sites_per_s, ...); best of `repeats` passes (the first one warms the page
    cache, the pinned buffers and the model workspaces)"""
    from .call_chunks import default_producers, default_writers, run_pipeline, run_pipeline_native
    producers = producers if producers else default_producers(kind == "bam", pipeline)
    writers = writers if writers else default_writers()
    os.makedirs(out_dir, exist_ok=True)
    chunk_args = region_namespaces(run, out_dir, regions, bam_reader) if regions else chunk_namespaces(run, out_dir, bam=(kind == "bam"), bam_reader=bam_reader)
    if times > 1:
        # the same chunks `times` times over (each copy decoded from the file again, its own output file): a run of a few dozen chunks is
        # mostly its own start-up - the first wave of host-decoded chunks takes as long as the whole run - and a genome is thousands
        more = []
        for t in range(1, times):
            for a in chunk_args:
                b = Namespace(**vars(a))
                b.call_fn = a.call_fn[:-4] + ".%d.vcf" % t
                more.append(b)
        chunk_args = chunk_args + more
    best, rows, best_stats = None, 0, {}
    for _ in range(max(1, repeats)):
        stats = {}
        ru0 = resource.getrusage(resource.RUSAGE_SELF)
        t0 = time.perf_counter()
        if pipeline == "native":
            rows = run_pipeline_native(eng, chunk_args, producers=producers, writers=writers, stats=stats, verbose=False,
                                       inflate_cus=inflate_cus, inflate_jobs=inflate_jobs, two_streams=two_streams,
                                       device_tokenise=bool(device_tokenise))
        else:
            rows = run_pipeline(eng, chunk_args, producers=producers, writers=writers, stats=stats)
        dt = time.perf_counter() - t0
        ru1 = resource.getrusage(resource.RUSAGE_SELF)
        if best is None or dt < best:
            best, best_stats = dt, stats
            host = dict(user_cpu_ms_per_chunk=round((ru1.ru_utime - ru0.ru_utime) * 1e3 / max(1, len(chunk_args)), 2),
                        sys_cpu_ms_per_chunk=round((ru1.ru_stime - ru0.ru_stime) * 1e3 / max(1, len(chunk_args)), 2),
                        minor_faults_per_chunk=int((ru1.ru_minflt - ru0.ru_minflt) / max(1, len(chunk_args))))
    extra = {k: int(best_stats[k]) for k in ("device_inflated", "device_piled", "device_tokenised") if k in best_stats}
    n_sites = run["n_sites"] * times
    if regions:                              # the candidates are the run's own product; every position of the contig was scanned for them
        n_sites = int(best_stats.get("sites", 0))
        extra["positions_scanned"] = int(run["contig_len"]) * times
        extra["positions_per_s"] = round(run["contig_len"] * times / best, 1)
        extra["candidates_extracted"] = n_sites
    per_chunk = {k[:-2] + "_ms_per_chunk": round(v * 1e3 / max(1, len(chunk_args)), 3) for k, v in best_stats.items() if k.endswith("_s")}
    return dict(sites_per_s=round(n_sites / best, 1), sites=int(n_sites), chunks=len(chunk_args), seconds=round(best, 4),
                producers=producers, writers=writers, pipeline=pipeline, vcf_records=int(rows), stage_thread_time=per_chunk, host_process=host, **extra,
                includes="disk reads, tokenise / BAM decode, PCIe both ways, kernels, alt_info + VCF rows (C), file writes")


class confined(object):
    """the calling thread - and every thread it starts from here on: the producers and writers of cto_run_chunks - on the first `n` of the
    cores it may use now (sched_setaffinity); usable_cores() then plans with n, as a rank of a node with that share of the cores would"""

    def __init__(self, n):
        self.n = int(n)

    def __enter__(self):
        self.old = os.sched_getaffinity(0)
        os.sched_setaffinity(0, set(sorted(self.old)[:max(1, self.n)]))
        return self

    def __exit__(self, *exc):
        os.sched_setaffinity(0, self.old)


def rank_share_legs(eng, run, kind, d, writers, pipeline, times, share):
    """What ONE rank of a node with `share` GPUs gets from this host: the same files with the process confined to usable_cores / share
    cores (an 8-GPU node leaves a rank an eighth of the cores; bench.py has one GPU, so this is the per-rank budget of the file -> VCF path,
    not a scaling curve).  Text: the host tokeniser and the device tokeniser (call_chunks' choice at <= 12 cores); BAM: the device inflate +
    pile-up path with that many cores behind it."""
    from .call_chunks import default_producers, usable_cores
    n = max(1, usable_cores() // max(1, share))
    out = {"cores": n, "of_usable": usable_cores(), "share": "1/%d" % share}
    pick = lambda h: {k: h[k] for k in ("sites_per_s", "seconds", "producers", "stage_thread_time", "host_process") if k in h} | \
        {k: h[k] for k in ("device_inflated", "device_piled", "device_tokenised") if k in h}
    with confined(n):
        if kind == "text":
            out["host_tokeniser"] = pick(time_run(eng, run, kind, os.path.join(d, "vcf_rank_host"), max(1, n), writers, 2, pipeline=pipeline, times=times,
                                                  device_tokenise=False))
            out["device_tokeniser"] = pick(time_run(eng, run, kind, os.path.join(d, "vcf_rank_dev"), max(2, n), writers, 2, pipeline=pipeline, times=times,
                                                    device_tokenise=True))
        else:
            out["device_inflate_pileup"] = pick(time_run(eng, run, kind, os.path.join(d, "vcf_rank_bam"), default_producers(True, pipeline), writers, 2,
                                                         pipeline=pipeline, times=times))
    return out


def measure(eng, kind="text", n_chunks=16, sites_per_chunk=4096, distinct=3, region_kb=None, producers=None, writers=None, workdir=None,
            repeats=4, pipeline="python", with_extraction=False, host_tokeniser_too=False, rank_share=0):
    """build_run + time_run in a temporary directory.  with_extraction (kind "bam"): -> (BED-driven leg, REGION-job leg on the same
    BAM: no candidate BEDs, the candidates are extracted from the pile-up inside the run)"""
    d = tempfile.mkdtemp(prefix="cto_e2e_", dir=workdir)
    try:
        t0 = time.perf_counter()
        run, source = build_run(d, kind, n_chunks, sites_per_chunk, distinct, region_kb)
        prep_s = time.perf_counter() - t0
        # a pass over the prepared files runs them several times over: its fill and drain (~12 ms: the first chunk's tokenising, the last one's
        # records) are a fixed cost that a run of 96 chunks (0.2 s) would carry as 6 % of its rate, a genome's run not at all
        times = 3 if kind == "bam" else 2
        r = time_run(eng, run, kind, os.path.join(d, "vcf_output"), producers, writers, repeats, pipeline=pipeline, times=times)
        if times > 1:
            source += "; every chunk %d times per pass (%d jobs)" % (times, times * len(run["chunks"]))
        r.update(source=source, input_synthesis_s=round(prep_s, 1))
        if host_tokeniser_too and kind == "text" and pipeline == "native":
            # the same files with the text tokenised on the device (cto_tokenise_device) instead of on the producer threads: what a rank with
            # a few cores to itself runs (call_chunks' default there); four producer threads (they mostly wait for the device: ~5 ms of CPU
            # per chunk all threads together; measured 1.42 / 1.59 / 1.66 M sites/s with 2 / 3 / 4)
            h = time_run(eng, run, kind, os.path.join(d, "vcf_output_dev_tok"), 4, writers, repeats, pipeline=pipeline, times=times,
                         device_tokenise=True)
            r["device_tokeniser"] = {k: h[k] for k in ("sites_per_s", "seconds", "producers", "stage_thread_time", "host_process", "device_tokenised")}
        if rank_share and pipeline == "native" and hasattr(os, "sched_setaffinity"):
            r["rank_of_%d" % rank_share] = rank_share_legs(eng, run, kind, d, writers, pipeline, times, rank_share)
        if with_extraction and kind == "bam":
            r2 = time_run(eng, run, kind, os.path.join(d, "vcf_output_regions"), producers, writers, repeats, pipeline="native", regions=len(run["chunks"]),
                          times=times)
            r2.update(source=source.replace("chunk files, candidates every 250 bp", "REGION jobs (no candidate BEDs: extract_candidates_calling's gates "
                                                                                    "run on the pile-up of every position, in HBM)"),
                      includes=r2["includes"] + ", candidate extraction")
            return r, r2
        return r
    finally:
        shutil.rmtree(d, ignore_errors=True)


def main():
    """`python -m clairs_to_amd.e2e [--chunks N] [--batch B] [--kinds text,bam]`: both file-to-file legs with a fresh Engine (seeded
    synthetic SNV models), one JSON object on stdout.  bench.py runs this as a child process so that the legs' kernel launches
    (pipelined, overlapping PCIe copies) stay out of the parent's kernel statistics when a profiler is attached to it."""
    import argparse
    import json
    import torch
    from .call_chunks import usable_cores
    from .engine import Engine, synthetic_models
    from .synth import likelihood_table, lik_and_edges
    ap = argparse.ArgumentParser()
    ap.add_argument("--chunks", type=int, default=12)
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--bam-chunks", type=int, default=None, help="chunk files of the BAM leg (default: a third of --chunks)")
    ap.add_argument("--kinds", default="text,bam")
    ap.add_argument("--reference-chunk-sites", type=int, default=0,
                    help="also run the text leg on chunk files of this many candidates (the reference cuts 10 000: shared/param.py:21)")
    ap.add_argument("--producers", type=int, default=None)
    ap.add_argument("--writers", type=int, default=None, help="default: call_chunks.default_writers()")
    ap.add_argument("--rank-share", type=int, default=8, help="also run each kind confined to usable_cores / N cores: one rank's share of an N-GPU node (0: skip)")
    ap.add_argument("--pipeline", default="native", choices=["native", "python"],
                    help="cto_run_chunks (csrc/pipeline.hip) or call_chunks.run_pipeline; same files either way")
    a = ap.parse_args()
    dev = torch.device("cuda", torch.cuda.current_device())
    models = synthetic_models(4, seed=0)
    lik, edges = lik_and_edges(likelihood_table(4), 4)
    eng = Engine(models["aff"], models["neg"], lik, edges, min_bq=20, device=dev)
    out = {"host_cores_usable": usable_cores(), "host_cores_visible": os.cpu_count()}
    for kind in a.kinds.split(","):
        n = a.chunks if kind == "text" else (a.bam_chunks or max(2, a.chunks // 3))
        r = measure(eng, kind=kind, n_chunks=n, sites_per_chunk=a.batch, producers=a.producers, writers=a.writers, pipeline=a.pipeline,
                    with_extraction=(kind == "bam" and a.pipeline == "native"), host_tokeniser_too=True, rank_share=a.rank_share)
        if isinstance(r, tuple):
            out["bam_to_vcf"], out["bam_to_vcf_with_extraction"] = r
        else:
            out["mpileup_text_to_vcf" if kind == "text" else "bam_to_vcf"] = r
    if a.reference_chunk_sites > 0 and "text" in a.kinds.split(","):
        n = max(4, a.chunks * a.batch // a.reference_chunk_sites)
        out["mpileup_text_to_vcf_reference_chunks"] = measure(eng, kind="text", n_chunks=n, sites_per_chunk=a.reference_chunk_sites,
                                                              producers=a.producers, writers=a.writers, pipeline=a.pipeline, host_tokeniser_too=True)
    print(json.dumps(out))


if __name__ == "__main__":
    main()

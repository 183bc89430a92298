"""File-to-file throughput (chunk files + mpileup text or BAM -> p_<chunk>.vcf) through the call_chunks pipeline, swept over
the number of producer threads; one JSON line per measurement (the logs kept under profiles/ come from this).
python tools/e2e_bench.py [--kind text,bam] [--chunks N] [--sites N] [--producers 2,4,8,16] [--writers 2] [--pack-threads T] [--pipeline python,native]"""
import argparse
import json
import os
import shutil
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kind", default="text,bam")
    ap.add_argument("--chunks", type=int, default=16)
    ap.add_argument("--sites", type=int, default=4096)
    ap.add_argument("--producers", default="2,4,8,16")
    ap.add_argument("--writers", default="2")
    ap.add_argument("--bam-reader", default="native", help="native (host inflate) or gpu (device inflate), comma separated")
    ap.add_argument("--two-streams", default="0", help="native pipeline: 1 = consecutive chunks alternate between two compute streams; comma separated")
    ap.add_argument("--repeats", type=int, default=4, help="passes over the chunk list per measurement (best one is reported; large values = soak test)")
    ap.add_argument("--pipeline", default="python", help="python (call_chunks.run_pipeline) and / or native (cto_run_chunks), comma separated")
    ap.add_argument("--inflate-cus", default="144", help="BAM + native pipeline: compute units the device inflate is confined to (0 = host inflate only), comma separated")
    ap.add_argument("--inflate-jobs", default="8", help="chunks in flight through the device inflate, comma separated")
    ap.add_argument("--pack-threads", default=None, help="CTO_PACK_THREADS values for the C producers, comma separated (default: their own, <= 32)")
    a = ap.parse_args()
    import torch
    from clairs_to_amd.e2e import build_run, time_run
    from clairs_to_amd.engine import Engine, synthetic_models
    from clairs_to_amd.synth import likelihood_table, lik_and_edges
    dev = torch.device("cuda:0")
    models = synthetic_models(4)
    lik, edges = lik_and_edges(likelihood_table(4), 4)
    eng = Engine(models["aff"], models["neg"], lik, edges, min_bq=20, device=dev)
    d = tempfile.mkdtemp(prefix="cto_e2e_")
    try:
        for kind in a.kind.split(","):
            t0 = time.perf_counter()
            run, source = build_run(d, kind, a.chunks, a.sites)
            print(json.dumps(dict(kind=kind, source=source, input_synthesis_s=round(time.perf_counter() - t0, 1), host_cores=os.cpu_count())), flush=True)
            for pt in (a.pack_threads.split(",") if a.pack_threads else [None]):
                if pt:
                    os.environ["CTO_PACK_THREADS"] = pt
                for w in [int(x) for x in a.writers.split(",")]:
                  for br in (a.bam_reader.split(",") if kind == "bam" else ["-"]):
                   for pl in a.pipeline.split(","):
                    if pl == "native" and br == "gpu":
                        continue
                    for p in [int(x) for x in a.producers.split(",")]:
                      dev_inflate = kind == "bam" and pl == "native"
                      for cus in ([int(x) for x in a.inflate_cus.split(",")] if dev_inflate else [0]):
                       for ij in ([int(x) for x in a.inflate_jobs.split(",")] if (dev_inflate and cus) else [0]):
                        for ts in ([int(x) for x in a.two_streams.split(",")] if pl == "native" else [0]):
                         r = time_run(eng, run, kind, os.path.join(d, "vcf_%s" % kind), producers=p, writers=w, bam_reader=br, pipeline=pl,
                                     inflate_cus=cus, inflate_jobs=ij, repeats=a.repeats, two_streams=bool(ts))
                         import resource
                         r.update(kind=kind, pack_threads=pt or "auto", bam_reader=br, inflate_cus=cus, inflate_jobs=ij, repeats=a.repeats,
                                  two_streams=ts, max_rss_mb=resource.getrusage(resource.RUSAGE_SELF).ru_maxrss // 1024)
                         r.pop("includes")
                         print(json.dumps(r), flush=True)
    finally:
        shutil.rmtree(d, ignore_errors=True)


if __name__ == "__main__":
    main()

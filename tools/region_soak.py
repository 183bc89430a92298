"""Soak of the REGION-job path: many passes over the same BAM + regions in one process (no hang, no failure, flat memory, stable rate).
python tools/region_soak.py [regions] [passes]"""
import os
import resource
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from clairs_to_amd.e2e import build_run, time_run
    from clairs_to_amd.engine import Engine, synthetic_models
    from clairs_to_amd.synth import likelihood_table, lik_and_edges
    n_reg = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    passes = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    dev = torch.device("cuda:0")
    models = synthetic_models(4)
    lik, edges = lik_and_edges(likelihood_table(4), 4)
    eng = Engine(models["aff"], models["neg"], lik, edges, min_bq=20, device=dev)
    d = tempfile.mkdtemp(prefix="cto_soak_")
    run, source = build_run(d, "bam", n_reg, 4096)
    print(source, flush=True)
    rates = []
    for block in range(passes // 5):
        t0 = time.perf_counter()
        r = time_run(eng, run, "bam", os.path.join(d, "vcf_regions"), repeats=5, pipeline="native", regions=n_reg)
        rss = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e6
        free, total = torch.cuda.mem_get_info(dev)
        rates.append(r["sites_per_s"])
        print("passes %3d-%3d: best %.0f candidate sites/s (%.1f M positions/s), %d records, 5 passes in %.1f s, max RSS %.2f GB, device memory in use %.1f GB"
              % (block * 5 + 1, block * 5 + 5, r["sites_per_s"], r["positions_per_s"] / 1e6, r["vcf_records"], time.perf_counter() - t0, rss,
                 (total - free) / 1e9), flush=True)
    print("best %.0f, worst block %.0f" % (max(rates), min(rates)))


if __name__ == "__main__":
    main()

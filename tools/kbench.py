#!/usr/bin/env python3
"""Stage-level timing on one GPU (not the judged bench): featurize / AFF / NEG alone and together."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from clairs_to_amd._lib import lib, check  # noqa: E402
from clairs_to_amd.engine import Engine, synthetic_models  # noqa: E402
from clairs_to_amd.featurize import featurize  # noqa: E402
from clairs_to_amd.synth import SynthChunk, PLATFORMS, likelihood_table, lik_and_edges  # noqa: E402


def timeit(fn, reps):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--n-out", type=int, default=4)
    ap.add_argument("--platform", default="ont", choices=["ont", "ilmn", "hifi"], help="generator preset (SURVEY 8d)")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    K = a.n_out
    models = synthetic_models(K)
    lik, edges = lik_and_edges(likelihood_table(K), K)
    eng = Engine(models["aff"], models["neg"], lik, edges, min_bq=PLATFORMS[a.platform]["min_bq"], device=dev)
    ch = SynthChunk.for_platform(a.platform, a.batch, seed=1)
    dp = eng.upload(ch.arrays())
    sp = torch.from_numpy(ch.site_pos).to(dev)
    feat = featurize(dp, sp, 20, 50)
    B = a.batch
    la = torch.empty((K, B, 2), device=dev)
    ln = torch.empty((K, B, 2), device=dev)
    s = int(torch.cuda.current_stream().cuda_stream)
    res = {}
    res["featurize_ms"] = timeit(lambda: featurize(dp, sp, 20, 50), a.reps)
    res["aff_ms"] = timeit(lambda: check(lib.cto_model_forward(eng.h_aff, feat.x_aff.data_ptr(), B, la.data_ptr(), s)), a.reps)
    res["neg_ms"] = timeit(lambda: check(lib.cto_model_forward(eng.h_neg, feat.x_neg.data_ptr(), B, ln.data_ptr(), s)), a.reps)
    from clairs_to_amd.extract_candidates_calling import extract_candidates
    res["extract_ms"] = timeit(lambda: extract_candidates(dp, 20), a.reps)
    res["post_ms"] = timeit(lambda: eng.posterior(la, ln), a.reps)
    res["all_ms"] = timeit(lambda: eng.run_device(dp, sp), a.reps)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.reps):
        eng.run_device(dp, sp)
    res["all_host_issue_ms"] = (time.perf_counter() - t0) / a.reps * 1e3
    torch.cuda.synchronize()
    eng1 = Engine(models["aff"], models["neg"], lik, edges, min_bq=PLATFORMS[a.platform]["min_bq"], device=dev, two_streams=False)
    res["all_1stream_ms"] = timeit(lambda: eng1.run_device(dp, sp), a.reps)
    gf_aff = 2e-9 * lib.cto_model_macs_per_site(eng.h_aff) * B
    gf_neg = 2e-9 * lib.cto_model_macs_per_site(eng.h_neg) * B
    res["aff_tflops"] = gf_aff / res["aff_ms"]
    res["neg_tflops"] = gf_neg / res["neg_ms"]
    res["all_tflops"] = (gf_aff + gf_neg) / res["all_ms"]
    res["sites_per_s"] = B / res["all_ms"] * 1e3
    # PCIe-inclusive: host packs in, host results out (Engine.run_stream), pageable vs pinned host buffers
    from clairs_to_amd.pack import pin_arrays
    n_chunks = max(8, a.reps)
    for tag, arrs in (("pageable", ch.arrays()), ("pinned", pin_arrays(ch.arrays()))):
        for _ in eng.run_stream(((arrs, ch.site_pos) for _ in range(3))):
            pass
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = sum(1 for _ in eng.run_stream(((arrs, ch.site_pos) for _ in range(n_chunks))))
        dt = time.perf_counter() - t0
        res["pcie_%s_sites_per_s" % tag] = n * B / dt
    print({k: round(v, 4) for k, v in res.items()})


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""What the live HIP-event bracket of bench.py's timed region costs: 20-step regions with cto_model_profile on and off, alternating."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import ctypes as C
    import torch
    from clairs_to_amd._lib import lib, check
    from clairs_to_amd.engine import Engine, synthetic_models
    from clairs_to_amd.synth import SynthChunk, likelihood_table, lik_and_edges
    dev = torch.device("cuda:0")
    K, B, pool = 4, 4096, 16
    models = synthetic_models(K)
    lik, edges = lik_and_edges(likelihood_table(K), K)
    eng = Engine(models["aff"], models["neg"], lik, edges, min_bq=20, device=dev)
    chunks = [SynthChunk(B, seed=s) for s in range(pool)]
    packs = [eng.upload(c.arrays()) for c in chunks]
    sites = [torch.from_numpy(c.site_pos).to(dev) for c in chunks]
    eng.run_device(packs[0], sites[0])
    torch.cuda.synchronize()
    pre = sys.argv[1] if len(sys.argv) > 1 else "none"
    if pre == "touch":                     # tensor creation once per resident pack
        from clairs_to_amd.featurize import featurize
        for i in range(pool):
            featurize(packs[i], sites[i], 20, 50)
    elif pre == "long":
        for i in range(60):
            eng.run_device(packs[i % pool], sites[i % pool])
    elif pre == "sleep":
        torch.cuda.synchronize()
        time.sleep(2.0)
    torch.cuda.synchronize()
    res = {"pre": pre, "on": [], "off": []}
    for rep in range(6):
        for mode in ("on", "off"):
            for i in range(5):
                eng.run_device(packs[i % pool], sites[i % pool])
            torch.cuda.synchronize()
            check(lib.cto_model_profile(eng.h_neg, 1 if mode == "on" else 0))
            t0 = time.perf_counter()
            for i in range(20):
                eng.run_device(packs[i % pool], sites[i % pool])
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            check(lib.cto_model_profile(eng.h_neg, 0))
            ms, macs = C.c_double(0.0), C.c_int64(0)
            lib.cto_model_profile_read(eng.h_neg, C.byref(ms), C.byref(macs))
            lib.cto_model_profile_read_stage(eng.h_neg, 1, C.byref(ms), C.byref(macs))
            res[mode].append(round(dt / 20 * 1e3, 4))
    print(json.dumps(res))


if __name__ == "__main__":
    main()

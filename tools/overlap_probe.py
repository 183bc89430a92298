#!/usr/bin/env python3
"""Does tensor creation of chunk i+1 hide under the networks of chunk i?  Serial loop against a two-stream loop (HIP events)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from clairs_to_amd._lib import lib, check
    from clairs_to_amd.engine import Engine, synthetic_models
    from clairs_to_amd.featurize import featurize
    from clairs_to_amd.synth import SynthChunk, likelihood_table, lik_and_edges
    dev = torch.device("cuda:0")
    K, B, pool, steps = 4, 4096, 8, 40
    models = synthetic_models(K)
    lik, edges = lik_and_edges(likelihood_table(K), K)
    eng = Engine(models["aff"], models["neg"], lik, edges, min_bq=20, device=dev)
    chunks = [SynthChunk(B, seed=s) for s in range(pool)]
    packs = [eng.upload(c.arrays()) for c in chunks]
    sites = [torch.from_numpy(c.site_pos).to(dev) for c in chunks]
    la = torch.empty((K, B, 2), device=dev)
    ln = torch.empty((K, B, 2), device=dev)
    main_s = torch.cuda.current_stream()
    side = torch.cuda.Stream(dev)

    def nets(feat):
        s = int(main_s.cuda_stream)
        check(lib.cto_model_forward(eng.h_neg, feat.x_neg.data_ptr(), B, ln.data_ptr(), s))
        check(lib.cto_model_forward(eng.h_aff, feat.x_aff.data_ptr(), B, la.data_ptr(), s))
        return eng.posterior(la, ln)

    def serial(n):
        for i in range(n):
            nets(featurize(packs[i % pool], sites[i % pool], 20, 50))

    def overlapped(n):
        feat = featurize(packs[0], sites[0], 20, 50)
        for i in range(n):
            nxt = None
            if i + 1 < n:
                side.wait_stream(main_s) if i == 0 else None
                with torch.cuda.stream(side):
                    nxt = featurize(packs[(i + 1) % pool], sites[(i + 1) % pool], 20, 50)
                    done = torch.cuda.Event()
                    done.record(side)
            nets(feat)
            if nxt is not None:
                main_s.wait_event(done)
                feat = nxt

    res = {}
    for name, fn in (("serial_ms_per_step", serial), ("overlapped_ms_per_step", overlapped), ("serial_again", serial)):
        fn(5)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn(steps)
        e1.record()
        torch.cuda.synchronize()
        res[name] = round(e0.elapsed_time(e1) / steps, 4)
    print(json.dumps(res))


if __name__ == "__main__":
    main()

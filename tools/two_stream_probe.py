#!/usr/bin/env python3
"""AFF (CvT) and NEG (BiGRU) on two streams against one, per step (HIP events), with the layer-1 tile height from CTO_GRU_L1_MS."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from clairs_to_amd.engine import Engine, synthetic_models
    from clairs_to_amd.synth import SynthChunk, likelihood_table, lik_and_edges
    dev = torch.device("cuda:0")
    K, B, pool, steps = 4, 4096, 8, 40
    models = synthetic_models(K)
    lik, edges = lik_and_edges(likelihood_table(K), K)
    chunks = [SynthChunk(B, seed=s) for s in range(pool)]
    res = {"l1_ms_env": os.environ.get("CTO_GRU_L1_MS", "2")}
    for two in (False, True, False, True):
        eng = Engine(models["aff"], models["neg"], lik, edges, min_bq=20, device=dev, two_streams=two)
        packs = [eng.upload(c.arrays()) for c in chunks]
        sites = [torch.from_numpy(c.site_pos).to(dev) for c in chunks]
        for i in range(6):
            eng.run_device(packs[i % pool], sites[i % pool])
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(steps):
            out = eng.run_device(packs[i % pool], sites[i % pool])
        e1.record()
        torch.cuda.synchronize()
        res.setdefault("two_streams" if two else "one_stream", []).append(round(e0.elapsed_time(e1) / steps, 4))
        res["checksum_%d" % int(two)] = float(out["probs"].double().sum().item())
    print(json.dumps(res))


if __name__ == "__main__":
    main()

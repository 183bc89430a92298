#!/usr/bin/env python3
"""Tensor creation, one kernel against the two-stage path, on the bench's 4096-site chunk (HIP events, mean of --reps launches):
python tools/feat_ab.py [--reps 50] [--batch 4096] [--spacing 40]"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--reps", type=int, default=50)
    p.add_argument("--batch", type=int, default=4096)
    p.add_argument("--spacing", type=int, default=None)
    a = p.parse_args()
    import torch
    from clairs_to_amd.pack import DevicePack
    from clairs_to_amd.featurize import featurize
    from clairs_to_amd.synth import SynthChunk
    dev = torch.device("cuda:0")
    kw = {} if a.spacing is None else {"spacing": a.spacing}
    ch = SynthChunk(a.batch, seed=1, **kw)
    dp = DevicePack(ch.arrays(), dev)
    sp = torch.from_numpy(ch.site_pos).to(dev)
    res = {"n_cols": int(dp.n_cols), "n_entries": int(ch.arrays()["entries"].size), "n_sites": a.batch}
    for name, fused, want_x in (("two_stage_ms", False, True), ("one_kernel_ms", True, True), ("one_kernel_no_tensor_stores_ms", True, False)):
        for _ in range(5):
            featurize(dp, sp, 20, 50, fused=fused, want_x=want_x)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.reps):
            featurize(dp, sp, 20, 50, fused=fused, want_x=want_x)
        e1.record()
        torch.cuda.synchronize()
        res[name] = round(e0.elapsed_time(e1) / a.reps, 4)
    print(json.dumps(res))


if __name__ == "__main__":
    main()

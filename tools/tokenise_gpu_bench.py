#!/usr/bin/env python3
"""cto_tokenise_device alone on one 4096-site chunk's mpileup text (22 MB), for rocprofv3 / timing:
python tools/tokenise_gpu_bench.py [--sites 4096] [--reps 5]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--sites", type=int, default=4096)
    p.add_argument("--reps", type=int, default=5)
    a = p.parse_args()
    import torch
    import oracle
    from clairs_to_amd.pack import ColumnPack, DeviceTokeniser
    from clairs_to_amd.synth import SynthChunk
    torch.cuda.set_device(0)
    ch = SynthChunk(a.sites, seed=3)
    text = oracle.synth_mpileup_text(ch, 0)
    ref, lo = ch.ref_window()
    tok = DeviceTokeniser()
    for i in range(a.reps):
        t = time.perf_counter()
        r = tok(text, ref, lo)
        torch.cuda.synchronize()
        print("device: %.2f ms for %.1f MB (%s)" % ((time.perf_counter() - t) * 1e3, len(text) / 1e6, "ok" if r else "fallback"))
    t = time.perf_counter()
    ColumnPack.from_mpileup(text, ref, lo)
    print("host:   %.2f ms" % ((time.perf_counter() - t) * 1e3))


if __name__ == "__main__":
    main()

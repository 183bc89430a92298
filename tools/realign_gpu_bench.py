#!/usr/bin/env python3
"""The device form of the realigner alone (cto_realign_windows, where = device) on the bench leg's windows, for rocprofv3:
cd /tmp && rocprofv3 --kernel-trace --stats -d out -- python /root/repo/tools/realign_gpu_bench.py [--windows 1500] [--reps 3]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--windows", type=int, default=1500)
    p.add_argument("--reps", type=int, default=3)
    p.add_argument("--threads", type=int, default=16)
    a = p.parse_args()
    import numpy as np
    import torch
    from clairs_to_amd.synth_realign import gen_window
    from clairs_to_amd.realign_reads import realign_windows
    torch.cuda.set_device(0)
    rng = np.random.default_rng(20260930)
    ws = [gen_window(rng) for _ in range(a.windows)]
    args = [(w["seqs"], w["positions"], w["cigars"], w["reference"], w["haplotypes"], w["ref_start"], w["ref_prefix"], w["ref_suffix"]) for w in ws]
    for _ in range(a.reps):
        st = {}
        t = time.perf_counter()
        realign_windows(args, where="device", threads=a.threads, stats=st)
        st["wall_s"] = round(time.perf_counter() - t, 4)
        print(json.dumps(st))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Reads per second of the Illumina realigner on synthetic windows (tests/realignutil.gen_window): cto_realign_reads, one thread and
--threads, beside the reference's own `realign_reads` when oracle/_ref/librealigner_ref.so is built.  CPU only.
python tools/realign_bench.py [--windows 400] [--threads 4]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--windows", type=int, default=400)
    p.add_argument("--reads", type=int, default=None, help="reads per window (default: the generator's mix)")
    p.add_argument("--threads", type=int, default=4)
    a = p.parse_args()
    import numpy as np
    import realignutil as ru
    from clairs_to_amd._lib import lib, check
    rng = np.random.default_rng(5)
    wins = [ru.gen_window(rng, n_reads=a.reads) for _ in range(a.windows)]
    n = sum(len(w["seqs"]) for w in wins)
    res = {"windows": a.windows, "reads": n}

    def rate(fn):
        fn(wins[0])
        t = time.time()
        out = [fn(w) for w in wins]
        return round(n / (time.time() - t)), out
    check(lib.cto_set_realign_threads(1))
    res["cto_realign_reads_1_thread_reads_per_s"], got = rate(ru.amd_realign)
    check(lib.cto_set_realign_threads(a.threads))
    res["cto_realign_reads_%d_threads_reads_per_s" % a.threads], got_t = rate(ru.amd_realign)
    check(lib.cto_set_realign_threads(1))
    res["threads_equal"] = got == got_t
    if ru.ref_lib() is not None:
        devnull = os.open(os.devnull, os.O_WRONLY)          # the reference prints a notice per short read
        saved = os.dup(2)
        os.dup2(devnull, 2)
        try:
            res["reference_realign_reads_per_s"], want = rate(ru.ref_realign)
        finally:
            os.dup2(saved, 2)
        res["equal_to_reference"] = got == want
    print(json.dumps(res))


if __name__ == "__main__":
    main()

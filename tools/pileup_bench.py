"""Time of the device pileup (csrc/pileup.hip) on one 1 Mb x 50x chunk of a synthetic long-read BAM, candidates every 250 bp
(the BED of their +-16 windows), against the host reader on the same chunk.  Under rocprofv3 --kernel-trace --stats this gives the
per-kernel split.   python tools/pileup_bench.py [region_kb] [reps]"""
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from clairs_to_amd._lib import lib
    from clairs_to_amd.bgzf import DevicePileup
    from clairs_to_amd.fasta import read_region
    from clairs_to_amd.pack import ColumnPack
    from clairs_to_amd.synth_run import make_bam_run
    dev = torch.device("cuda:0")
    d = tempfile.mkdtemp(prefix="cto_pile_")
    kb = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    run = make_bam_run(os.path.join(d, "run"), region_kb=kb, n_chunks=2)
    bam = run["bam_fn"]
    lo, hi = 500001, 1500000
    sites = np.arange(lo + 100, hi - 100, 250)
    bed = [(int(x) - 17, int(x) + 16) for x in sites]
    every = len(sys.argv) > 3 and sys.argv[3] == "all"
    if every:
        bed = None
    ref = read_region(run["ref_fn"], "chr1", max(1, lo - 2000), hi + 2000)
    ref_start = max(1, lo - 2000)
    dp = DevicePileup()
    for rep in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        pv, lite, fb = dp.pileup(bam, None, "chr1", lo, hi, ref, ref_start, dev, bed=bed)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        print("device: %.1f ms (read + H2D + inflate + pile-up), %d columns, %d entries, %d keys, fallback %d" % (
            (t1 - t0) * 1e3, pv.n_cols, pv.n_entries, pv.n_keys, fb), flush=True)
        if every and not fb:
            import ctypes as C
            from clairs_to_amd._lib import check, current_stream_ptr
            nc = int(pv.n_cols)
            flags = torch.zeros((nc,), dtype=torch.uint8, device=dev)
            depth = torch.zeros((nc,), dtype=torch.int32, device=dev)
            out = torch.empty((nc,), dtype=torch.int32, device=dev)
            scr = torch.empty(((nc + 255) // 256 + 2,), dtype=torch.int32, device=dev)
            n_out = torch.empty((1,), dtype=torch.int32, device=dev)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            check(lib.cto_extract_candidates(C.byref(pv), 20, 20, 0.05, 1.0, 4.0, 3, 0, flags.data_ptr(), depth.data_ptr(), current_stream_ptr()))
            check(lib.cto_candidate_positions(C.byref(pv), flags.data_ptr(), 1, lo, hi, out.data_ptr(), nc, scr.data_ptr(), n_out.data_ptr(),
                                              current_stream_ptr()))
            e1.record()
            torch.cuda.synchronize()
            print("   gates + compaction: %.3f ms, %d SNV candidates (pack %.1f MB + 5 B / column)" % (
                e0.elapsed_time(e1), int(n_out.item()), (pv.n_entries * 4 + nc * 21) / 1e6), flush=True)
        lib.cto_pack_free(lite)
    if every:
        return
    t0 = time.perf_counter()
    p = ColumnPack.from_bam(bam, "chr1", lo, hi, ref, ref_start, bed=bed)
    print("host reader: %.1f ms, %d columns, %d entries" % ((time.perf_counter() - t0) * 1e3, p.n_cols, p.n_entries))


if __name__ == "__main__":
    main()

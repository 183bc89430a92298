"""Where the time of bgzf.inflate_span goes: file read, block scan, H2D, inflate kernel, D2H (one 1 Mb x 50x chunk of a synthetic BAM)."""
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import ctypes as C
    import torch
    from clairs_to_amd._lib import check, lib
    from clairs_to_amd import bgzf
    from clairs_to_amd.synth_run import make_bam_run
    dev = torch.device("cuda:0")
    d = tempfile.mkdtemp(prefix="cto_inf_")
    kb = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    run = make_bam_run(os.path.join(d, "run"), region_kb=kb, n_chunks=2)
    bam = run["bam_fn"]
    lo, hi = 500001, 1500000
    fb, fe = C.c_int64(0), C.c_int64(0)
    check(lib.cto_bam_chunk_span(bam.encode(), None, b"chr1", lo, hi, C.byref(fb), C.byref(fe)))
    n = fe.value - fb.value
    for rep in range(3):
        t0 = time.perf_counter()
        h_in = bgzf._pinned("h_in", n + bgzf.BGZF_PAD)
        view = h_in.numpy()
        with open(bam, "rb", buffering=0) as f:
            f.seek(fb.value)
            got = f.readinto(memoryview(view)[:n])
        t1 = time.perf_counter()
        blocks, out_bytes = bgzf.scan(view, n, fb.value)
        t2 = time.perf_counter()
        d_in = h_in[:n + bgzf.BGZF_PAD].to(dev, non_blocking=True)
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        d_out, d_status = bgzf.inflate_device(d_in, blocks, out_bytes, dev)
        e1.record()
        torch.cuda.synchronize()
        t4 = time.perf_counter()
        h_out = bgzf._pinned("h_out", out_bytes)
        h_out[:out_bytes].copy_(d_out, non_blocking=True)
        torch.cuda.synchronize()
        t5 = time.perf_counter()
        st = d_status.cpu().numpy()
        print("bytes %.1f MB -> %.1f MB in %d blocks (got %d): read %.1f ms, scan %.1f, H2D %.1f, inflate %.1f (kernel %.2f ms), D2H %.1f; bad blocks %d" % (
            n / 1e6, out_bytes / 1e6, len(blocks), got, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3, e0.elapsed_time(e1),
            (t5 - t4) * 1e3, int((st != 0).sum())), flush=True)
    t0 = time.perf_counter()
    r = bgzf.inflate_span(bam, None, "chr1", lo, hi, dev)
    print("inflate_span: %.1f ms" % ((time.perf_counter() - t0) * 1e3))


if __name__ == "__main__" and not (len(sys.argv) > 1 and sys.argv[1] == "concurrency"):
    main()


def concurrency():
    """K chunks' inflate kernels in flight on K streams; then the whole producer (inflate_span + pack) from T threads"""
    import ctypes as C
    import threading
    import torch
    from clairs_to_amd import bgzf
    from clairs_to_amd._lib import check, lib
    from clairs_to_amd.fasta import read_region
    from clairs_to_amd.pack import ColumnPack
    from clairs_to_amd.synth_run import make_bam_run
    dev = torch.device("cuda:0")
    d = tempfile.mkdtemp(prefix="cto_inf_")
    run = make_bam_run(os.path.join(d, "run"), region_kb=2000, n_chunks=2)
    bam = run["bam_fn"]
    lo, hi = 500001, 1500000
    fb, fe = C.c_int64(0), C.c_int64(0)
    check(lib.cto_bam_chunk_span(bam.encode(), None, b"chr1", lo, hi, C.byref(fb), C.byref(fe)))
    n = fe.value - fb.value
    host = np.zeros(n + bgzf.BGZF_PAD, dtype=np.uint8)
    with open(bam, "rb") as f:
        f.seek(fb.value)
        host[:n] = np.frombuffer(f.read(n), dtype=np.uint8)
    blocks, out_bytes = bgzf.scan(host, n, fb.value)
    d_in = torch.from_numpy(host).to(dev)
    for K in (1, 2, 4, 8):
        streams = [torch.cuda.Stream(dev) for _ in range(K)]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        outs = [bgzf.inflate_device(d_in, blocks, out_bytes, dev, s) for s in streams]
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print("%d inflate kernels in flight: %.1f ms = %.1f ms per chunk" % (K, dt * 1e3, dt * 1e3 / K), flush=True)
    ref = read_region(run["ref_fn"], "chr1", 1, 2000000, as_bytes=True)
    bed = [(p - 17, p + 16) for p in range(lo + 500, hi - 500, 250)]
    for T in (1, 4, 8):
        def work():
            s = torch.cuda.Stream(dev)
            for _ in range(3):
                t0 = time.perf_counter()
                inf = bgzf.inflate_span(bam, None, "chr1", lo, hi, dev, s)
                t1 = time.perf_counter()
                ColumnPack.from_bam(bam, "chr1", lo, hi, ref, 1, bed=bed, inflated=inf)
                t2 = time.perf_counter()
            res.append(((t1 - t0) * 1e3, (t2 - t1) * 1e3))
        res = []
        th = [threading.Thread(target=work) for _ in range(T)]
        t0 = time.perf_counter()
        for x in th:
            x.start()
        for x in th:
            x.join()
        dt = time.perf_counter() - t0
        print("%d producer threads x 3 chunks: %.1f ms per chunk overall; last chunk of each: inflate_span %s ms, pack %s ms" % (
            T, dt * 1e3 / (3 * T), ["%.0f" % r[0] for r in res], ["%.0f" % r[1] for r in res]), flush=True)
    t0 = time.perf_counter()
    ColumnPack.from_bam(bam, "chr1", lo, hi, ref, 1, bed=bed)
    print("host-inflate producer, same chunk: %.1f ms" % ((time.perf_counter() - t0) * 1e3))


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "concurrency":
    concurrency()

"""A/B of the BGZF inflate kernel: K launches of one 1 Mb x 50x chunk's blocks in flight, for the product library and for a variant
library given in argv (CTO_LIB_PATH of a child process each).   python tools/inflate_ab.py [variant.so ...]
Build the round-3 kernel as a variant:  bash tools/inflate_ab.sh"""
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(bam):
    import ctypes as C
    import torch
    from clairs_to_amd import bgzf
    from clairs_to_amd._lib import check, lib
    dev = torch.device("cuda:0")
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
    ref = None
    for K in [int(x) for x in os.environ.get("CTO_AB_K", "1,1,4,8,16").split(",")]:
        streams = [torch.cuda.Stream(dev) for _ in range(K)]
        best = 1e9
        for rep in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            outs = [bgzf.inflate_device(d_in, blocks, out_bytes, dev, s) for s in streams]
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        bad = int(sum(int((o[1] != 0).sum()) for o in outs))
        digest = int(outs[0][0].to(torch.int64).sum().item())
        print("  %2d in flight: %7.2f ms = %6.2f ms per chunk  (%.1f MB -> %.1f MB, %d blocks, bad %d, sum %d)" % (
            K, best * 1e3, best * 1e3 / K, n / 1e6, out_bytes / 1e6, len(blocks), bad, digest), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--child":
        child(sys.argv[2])
    elif len(sys.argv) > 2 and sys.argv[1] == "--make-bam":
        from clairs_to_amd.synth_run import make_bam_run
        print(make_bam_run(sys.argv[2], region_kb=2000, n_chunks=2)["bam_fn"])
    else:
        from clairs_to_amd.synth_run import make_bam_run
        d = tempfile.mkdtemp(prefix="cto_infab_")
        run = make_bam_run(os.path.join(d, "run"), region_kb=2000, n_chunks=2)
        for lib_path in [None] + sys.argv[1:]:
            env = dict(os.environ)
            if lib_path:
                env["CTO_LIB_PATH"] = os.path.abspath(lib_path)
            print("library: %s" % (lib_path or "product (clairs_to_amd/libclairsto_amd.so)"), flush=True)
            subprocess.run([sys.executable, os.path.abspath(__file__), "--child", run["bam_fn"]], env=env, check=False)

import os, sys, tempfile, time, ctypes as C
import numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
import torch
from clairs_to_amd import bgzf
from clairs_to_amd._lib import check, lib
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
ref = None
for K in (1, 2, 4, 8, 16, 24):
    streams = [torch.cuda.Stream(dev) for _ in range(K)]
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        outs = [bgzf.inflate_device(d_in, blocks, out_bytes, dev, s) for s in streams]
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    bad = int((outs[0][1] != 0).sum().item())
    print("%s: %d blocks, %.1f -> %.1f MB; %2d launches in flight: %.1f ms = %.2f ms per chunk; bad blocks %d" % (
        "LANES" if os.environ.get("CTO_INFLATE_LANES") else "waves", len(blocks), n / 1e6, out_bytes / 1e6, K, dt * 1e3, dt * 1e3 / K, bad), flush=True)
    if K == 1:
        import hashlib
        print("   sha of output", hashlib.sha256(outs[0][0].cpu().numpy().tobytes()).hexdigest()[:16])

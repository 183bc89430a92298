"""Throughput of the native BAM -> pack producer on a synthetic long-read BAM, for 1 thread and for the default thread count
(CTO_PACK_THREADS overrides).  python tools/bam_bench.py [region_kb] [coverage]"""
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    from bamutil import write_bam
    from clairs_to_amd.pack import ColumnPack
    kb = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    cov = float(sys.argv[2]) if len(sys.argv) > 2 else 50.0
    L = kb * 1000
    rng = np.random.default_rng(1)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    ref = acgt[rng.integers(0, 4, size=L)]
    reads, bases, i = [], 0, 0
    while bases < cov * L:
        n = int(np.clip(rng.lognormal(9.0, 0.5), 1000, 30000))
        pos = int(rng.integers(0, max(1, L - n)))
        n = min(n, L - pos)
        seg = ref[pos:pos + n].copy()
        mm = rng.random(n) < 0.01
        seg[mm] = acgt[rng.integers(0, 4, size=int(mm.sum()))]
        # one insertion / deletion every ~150 bases
        cigar, seq, rp = [], [], 0
        cuts = np.sort(rng.choice(np.arange(10, max(11, n - 10)), size=max(1, n // 150), replace=False)) if n > 40 else []
        for c in cuts:
            if c - rp <= 0:
                continue
            cigar.append(("M", int(c - rp)))
            seq.append(seg[rp:c])
            rp = int(c)
            if rng.random() < 0.4:
                k = int(rng.integers(1, 4))
                cigar.append(("I", k))
                seq.append(acgt[rng.integers(0, 4, size=k)])
            else:
                k = int(min(rng.integers(1, 4), n - rp - 1))
                if k > 0:
                    cigar.append(("D", k))
                    rp += k
        if n - rp > 0:
            cigar.append(("M", int(n - rp)))
            seq.append(seg[rp:n])
        s = np.concatenate(seq)
        q = np.clip(np.rint(rng.normal(28, 8, size=s.size)), 1, 50).astype(np.uint8)
        reads.append(dict(name="r%d" % i, flag=16 * int(rng.random() < 0.5), ref=0, pos=pos, mapq=60, cigar=cigar,
                          seq=s.tobytes().decode(), qual=q.tolist()))
        bases += n
        i += 1
    reads.sort(key=lambda r: r["pos"])
    d = tempfile.mkdtemp()
    bam = os.path.join(d, "b.bam")
    t0 = time.perf_counter()
    write_bam(bam, [("chr1", L)], reads, block_payload=65000)
    print("synthetic BAM: %d reads, %.1f Mbases, %.1f MB on disk (written in %.1f s)" % (len(reads), bases / 1e6, os.path.getsize(bam) / 1e6,
                                                                                       time.perf_counter() - t0))
    sites = list(range(1000, L - 1000, 250))
    bed = [(x - 17, x + 17) for x in sites]
    refs = ref.tobytes().decode()
    for threads in ("1", None):
        if threads:
            os.environ["CTO_PACK_THREADS"] = threads
        else:
            os.environ.pop("CTO_PACK_THREADS", None)
        for tag, b in (("BED windows of %d candidates" % len(sites), bed), ("every position", None)):
            best, pack = 1e9, None
            for _ in range(3):
                t0 = time.perf_counter()
                pack = ColumnPack.from_bam(bam, "chr1", 1, L, refs, 1, bed=b)
                best = min(best, time.perf_counter() - t0)
            print("threads %-4s %s: %.3f s  -> %d columns, %.2f M read-bases; %.0f candidate sites/s, %.1f MB/s of BAM" % (
                threads or "auto", tag, best, pack.n_cols, pack.n_entries / 1e6, len(sites) / best, os.path.getsize(bam) / best / 1e6))


if __name__ == "__main__":
    main()

"""Host tokeniser throughput (mpileup text -> column pack), thread scaling.  python tools/tokenise_bench.py [n_sites]"""
import ctypes as C
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    n_sites = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    if len(sys.argv) > 2:      # child: one measurement with the thread count of the environment
        import oracle
        from clairs_to_amd._lib import lib, check, c_vp
        from clairs_to_amd.synth import SynthChunk
        ch = SynthChunk(n_sites, seed=1)
        text = oracle.synth_mpileup_text(ch, 0)
        text = text if isinstance(text, bytes) else text.encode()
        ref, lo = ch.ref_window()
        rb = ref.encode()
        best = 1e9
        for _ in range(5):
            out = c_vp()
            t0 = time.perf_counter()
            check(lib.cto_pack_from_mpileup(text, len(text), rb, lo, len(rb), 60, C.byref(out)))
            best = min(best, time.perf_counter() - t0)
            lib.cto_pack_free(out)
        print("threads %s: best of 5 %.1f ms  %.0f MB/s  %.0f sites/s  (%.1f MB text)" % (
            os.environ.get("CTO_PACK_THREADS", "auto"), best * 1e3, len(text) / best / 1e6, n_sites / best, len(text) / 1e6))
        return
    print("host cpus:", os.cpu_count())
    for nt in ("1", "2", "4", "8", "16", "32"):
        subprocess.run([sys.executable, __file__, str(n_sites), "child"], env=dict(os.environ, CTO_PACK_THREADS=nt))


if __name__ == "__main__":
    main()

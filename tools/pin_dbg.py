"""The comparison half of tools/pin_dbg.sh: two shared objects with the reference's get_consensus ABI (src/realign/debruijn_graph.cpp:387-428,
called as src/realign_reads.py:519-539 calls it), the same windows into both, every consensus list compared.
    python tools/pin_dbg.py <reference .so> <product .so> [extra windows]
Exit code 0 = equal everywhere, 1 = a difference."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)


class Dbg(C.Structure):
    _fields_ = [("consensus_size", C.c_int), ("consensus", C.c_char_p * 200)]            # src/realign_reads.py:80-83


def load(path):
    lib = C.CDLL(os.path.abspath(path))
    lib.get_consensus.restype = C.POINTER(Dbg)
    lib.get_consensus.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_int]
    lib.free_memory.argtypes = [C.POINTER(Dbg), C.c_int]

    def consensus(ref, reads, lowbq):
        bq = ",".join(" ".join(str(x) for x in row) for row in lowbq)
        q = lib.get_consensus(ref.encode(), ",".join(reads).encode(), bq.encode(), len(reads))
        n = q.contents.consensus_size
        out = [c.decode() for c in q.contents.consensus[:min(n, 200)]]
        lib.free_memory(q, n)
        return out
    return consensus


def main():
    from consensus_enum import consensus_windows
    ref_fn, got_fn = sys.argv[1], sys.argv[2]
    extra = int(sys.argv[3]) if len(sys.argv) > 3 else 2000
    want, got = load(ref_fn), load(got_fn)
    n = diff = order_only = multi = 0
    for seed, count in ((2026, 250), (77, extra)):
        for ref, reads, lowbq in consensus_windows(seed, count):
            a, b = want(ref, reads, lowbq), got(ref, reads, lowbq)
            n += 1
            multi += len(a) > 1
            if a != b:
                if sorted(a) == sorted(b):
                    order_only += 1
                diff += 1
                if diff <= 5:
                    print("DIFF window %d (seed %d): reference %d haplotypes, product %d%s\n  ref   %s\n  reads %s\n  lowbq %s\n  reference: %s\n  product:   %s"
                          % (n, seed, len(a), len(b), " (same set, other order)" if sorted(a) == sorted(b) else "", ref, reads, lowbq, a, b))
    print("pin_dbg: %d windows (%d with more than one haplotype): %d differ (%d of them in order only)" % (n, multi, diff, order_only))
    sys.exit(1 if diff else 0)


if __name__ == "__main__":
    main()

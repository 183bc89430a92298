"""An independent second opinion on cto_dbg_consensus (csrc/debruijn.cpp): the contract of the reference's window consensus
(src/realign/debruijn_graph.cpp - KMinMaxFromReference :177-204, Build :208-232, AddKmersAndEdges :246-256, AddEdgesForRead
:262-286, CandidatePaths :288-318, HaplotypeForPath :320-329, Prune :353-385) restated in plain Python with OTHER means than the
product: dictionaries of Counters instead of vertex arrays, cycle detection by three-colour depth-first search, path enumeration by
recursion instead of a breadth-first queue.  Test infrastructure only.  STILL PARITY-UNPINNED against the reference itself (its
build needs Boost.Graph, absent here): two independent readings of the same source agreeing is a weaker statement than equality
with the compiled original, and the test that uses this says so.

The one place where no restatement can follow the original: its breadth-first queue gives up when more than 256 partial + finished
paths exist, in an order that follows heap addresses; `enumerate_consensus` returns None when the count comes near that limit and the
caller skips the case."""
import sys
from collections import Counter, defaultdict


def _min_k(ref):
    for k in range(10, min(101, len(ref) - 1) + 1):
        kmers = [ref[i:i + k] for i in range(len(ref) - k + 1)]
        if len(set(kmers)) == len(kmers):
            return k
    return None


def _graph(ref, reads, lowbq, k):
    weight, is_ref = defaultdict(Counter), set()

    def run(bases, start, end, ref_flag):
        if end <= 0:
            return
        for i in range(start, end):
            a, b = bases[i:i + k], bases[i + 1:i + 1 + k]
            weight[a][b] += 1
            if ref_flag:
                is_ref.add((a, b))
    run(ref, 0, len(ref) - k, True)
    for read, bad in zip(reads, lowbq):
        bad = set(bad)
        stop, i = len(read) - k, 0
        while i < stop:
            nb = next((p for p in range(i, len(read)) if read[p] not in "ACGT" or p in bad), len(read))
            run(read, i, nb - k, False)
            i = nb + 1
    return weight, is_ref


def _has_cycle(weight):
    colour = {}
    sys.setrecursionlimit(100000)

    def visit(v):
        colour[v] = 1
        for w in weight.get(v, ()):
            c = colour.get(w, 0)
            if c == 1 or (c == 0 and visit(w)):
                return True
        colour[v] = 2
        return False
    verts = set(weight) | {w for d in weight.values() for w in d}
    return any(colour.get(v, 0) == 0 and visit(v) for v in sorted(verts))


def enumerate_consensus(ref, reads, lowbq=None, limit_margin=40):
    lowbq = lowbq if lowbq is not None else [[] for _ in reads]
    k0 = _min_k(ref)
    if k0 is None:
        return []
    for k in range(k0, min(101, len(ref) - 1) + 1):
        weight, is_ref = _graph(ref, reads, lowbq, k)
        if _has_cycle(weight):
            continue
        succ = defaultdict(list)
        for a, d in weight.items():
            for b, n in d.items():
                if (a, b) in is_ref or n >= 2:
                    succ[a].append(b)
        source, sink = ref[:k], ref[len(ref) - k:]
        fwd, stack = {source}, [source]
        while stack:
            for w in succ.get(stack.pop(), ()):
                if w not in fwd:
                    fwd.add(w); stack.append(w)
        pred = defaultdict(list)
        for a, bs in succ.items():
            for b in bs:
                pred[b].append(a)
        back, stack = {sink}, [sink]
        while stack:
            for w in pred.get(stack.pop(), ()):
                if w not in back:
                    back.add(w); stack.append(w)
        keep = fwd & back
        succ = {a: [b for b in bs if b in keep] for a, bs in succ.items() if a in keep}
        out, budget = [], [0]

        def walk(v, text):
            budget[0] += 1
            nxt = succ.get(v, [])
            for w in nxt:
                if w == sink or not succ.get(w):
                    out.append(text + w[-1])
                else:
                    walk(w, text + w[-1])
        if source in keep:
            walk(source, source)
        if budget[0] + len(out) > 256 - limit_margin:
            return None
        return sorted(out)
    return []


def consensus_windows(seed, count):
    """The generated windows of tests/test_realign.py::test_debruijn_against_an_independent_enumerator and of tools/pin_dbg.sh: (reference,
    reads, low-quality positions per read) with SNVs, insertions, deletions, N bases, a repeated reference stretch every seventh window
    and reads that carry a tandem duplication (a cycle at small k)."""
    import numpy as np
    BASES = np.frombuffer(b"ACGT", dtype=np.uint8)
    rng = np.random.default_rng(seed)
    for it in range(count):
        n = int(rng.integers(60, 260))
        ref = bytes(rng.choice(BASES, n)).decode()
        if it % 7 == 0:                                             # a repeated stretch: the smallest k is not 10
            a = int(rng.integers(0, n - 40)); ref = ref[:a + 30] + ref[a:a + 25] + ref[a + 30:]
        haps = [ref]
        for _ in range(int(rng.integers(0, 4))):
            p = int(rng.integers(15, len(ref) - 15)); kind = int(rng.integers(0, 3))
            base = haps[int(rng.integers(0, len(haps)))]
            if kind == 0:
                haps.append(base[:p] + "ACGT"[("ACGT".index(base[p]) + 1) % 4] + base[p + 1:])
            elif kind == 1:
                haps.append(base[:p] + bytes(rng.choice(BASES, int(rng.integers(1, 9)))).decode() + base[p:])
            else:
                haps.append(base[:p] + base[p + int(rng.integers(1, 9)):])
        reads, lowbq = [], []
        for _ in range(int(rng.integers(6, 40))):
            h = haps[int(rng.integers(0, len(haps)))]
            a = int(rng.integers(0, max(1, len(h) - 40)))
            r = h[a:a + int(rng.integers(30, 151))]
            if rng.random() < 0.15:
                q = int(rng.integers(0, len(r))); r = r[:q] + "N" + r[q + 1:]
            if rng.random() < 0.1 and len(r) > 70:                 # a tandem duplication inside a read: a cycle at small k
                q = int(rng.integers(10, len(r) - 50)); r = r[:q + 30] + r[q:q + 30] + r[q + 30:]
            reads.append(r)
            lowbq.append(sorted({int(x) for x in rng.integers(0, len(r), int(rng.integers(0, 3)))}) if rng.random() < 0.3 else [])
        yield ref, reads, lowbq

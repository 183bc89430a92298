"""Drop-in counterpart of `clairs_to.py realign_reads` (reference: src/realign_reads.py; the short-read local realignment the
Illumina filter `realign_variants` runs per low-QUAL call, SURVEY.md 8f #4b).

    samtools view -h BAM ctg:(pos-1100)-(pos+1100) -q MQ   ->   realigned SAM text on stdout (read names carry a _0 / _1 strand suffix)

Same inputs, options and output text as the reference.  The two native pieces are the library's (`cto_dbg_consensus`,
`cto_realign_reads`: csrc/debruijn.cpp, csrc/realign.cpp); this module is the bookkeeping around them, restated from the
reference's behaviour (all line numbers into src/realign_reads.py):

  * rows -> reads (:255-305): dictionary by name + "_" + strand (a later row of the same name and strand replaces the earlier
    one but keeps its place), reads without a CIGAR or less than 55 % aligned are ignored;
  * evidence counter (:306-352): per reference position, the reads (MQ >= 20) that mismatch it with BQ >= 20, that carry a clean
    insertion / soft clip next to it (positions [p - len, p + len)) or delete it;
  * per 5 000-base chunk (:441-457): positions with >= --min_coverage evidence inside the chunk (+-20) and within
    --max_distance of --pos become windows (:463-499: runs no further than 160 apart, padded by 80); every read goes to the window
    it overlaps most (:180-186, :502-506);
  * per window of <= 1 000 bases (:509-615): candidate haplotypes from the de Bruijn graph of its MQ >= 14 reads (low-BQ < 15
    positions masked), nothing to do when that is empty or just the reference; otherwise the first 1 000 reads are re-aligned
    against reference-prefix + haplotype + reference-suffix, and a read keeps the re-alignment with the most indel bases seen so
    far (:153-164, `>=`: the later of equals wins);
  * output (:618-647): reads in order of their (new) start, stable; within the loop only those 1 020 bases behind the chunk.
"""
import ctypes as C
import numpy as np
import shlex
import subprocess
import sys
from argparse import ArgumentParser, SUPPRESS

from ._lib import lib, check

CHUNK = 5000                    # realign_chunk_size (:47)
MIN_DBG_MQ = MIN_DBG_BQ = 20    # :48
EXPAND = 20                     # region_expansion_in_bp = expand_align_ref_region (:49)
WINDOW_GAP = 80                 # min_windows_distance (:50)
MAX_WINDOW = MAX_READS = 1000   # max_window_size = max_region_reads_num (:51)
REF_EXPAND = 100000             # expandReferenceRegion (:52)
GRAPH_MIN_MQ = 14               # graph_min_mapping_quality (:89)
GRAPH_LOW_BQ = 15               # :526


def dbg_consensus(ref, reads, lowbq=None):
    """Candidate haplotypes of a window (cto_dbg_consensus; reference: dbg.get_consensus, :532-539)."""
    n = len(reads)
    arr = (C.c_char_p * max(n, 1))(*[r.encode() for r in reads])
    low = off = None
    if lowbq is not None:
        flat = [p for one in lowbq for p in one]
        low = (C.c_int32 * max(len(flat), 1))(*flat)
        off = (C.c_int64 * (n + 1))()
        for i, one in enumerate(lowbq):
            off[i + 1] = off[i] + len(one)
    need = C.c_size_t(0)
    cap = 1 << 16
    while True:
        buf = C.create_string_buffer(cap)
        rc = lib.cto_dbg_consensus(ref.encode(), n, arr, low, off, buf, cap, C.byref(need))
        if rc == -3 and need.value > cap:               # CTO_ENOMEM: retry with what it asked for
            cap = need.value
            continue
        check(rc)
        return [s.decode() for s in buf.raw[:need.value].split(b"\0")[:rc]]


def realign_window(seqs, positions, cigars, ref_seq, haplotypes, ref_start, prefix_len, suffix_len):
    """(positions, cigars) of cto_realign_reads (reference: realigner.realign_reads, :582-595)."""
    n = len(seqs)
    out_pos = (C.c_int32 * n)()
    cap = 64 * n + 8 * sum(len(s) for s in seqs) + sum(len(c) for c in cigars) + 64
    buf = C.create_string_buffer(cap)
    off = (C.c_int64 * (n + 1))()
    check(lib.cto_realign_reads(n, (C.c_char_p * n)(*[s.encode() for s in seqs]), (C.c_int32 * n)(*positions),
                                (C.c_char_p * n)(*[c.encode() for c in cigars]), ref_seq.encode(), " ".join(haplotypes).encode(),
                                ref_start, prefix_len, suffix_len, out_pos, buf, cap, off))
    raw = buf.raw
    return list(out_pos), [raw[off[i]:off[i + 1] - 1].decode() for i in range(n)]


def realign_windows(windows, where="device", threads=0, stats=None, statuses=None):
    """Many windows in one call (cto_realign_windows, csrc/realign_batch.hip): `windows` = list of the argument tuples of
    realign_window (seqs, positions, cigars, ref_seq, haplotypes, ref_start, prefix_len, suffix_len); returns the list of
    (positions, cigars).  where = "device": the k-mer fast pass and the striped Smith-Waterman passes of every window run as two
    HIP launches (a HIP device must be current); "host": the same windows on `threads` host workers.  Outputs are identical."""
    from ._lib import RealignJob, RealignStats
    from itertools import chain
    n = len(windows)
    jobs = (RealignJob * max(n, 1))()
    # Marshalling for the whole list at once (a run has tens of reads per window and thousands of windows: per-read ctypes objects
    # would cost more than the call): every read of every window in ONE NUL-separated buffer, the same for the CIGARs, one array of
    # positions; a job points into them (seqs_joined / cigars_joined of cto_realign_job).
    counts = np.fromiter((len(w[0]) for w in windows), dtype=np.int64, count=n)
    if n and (np.any(counts != np.fromiter((len(w[1]) for w in windows), dtype=np.int64, count=n)) or
              np.any(counts != np.fromiter((len(w[2]) for w in windows), dtype=np.int64, count=n))):
        raise ValueError("realign_windows: reads, positions and CIGARs of a window must be equally many")
    total = int(counts.sum())
    first = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(counts, out=first[1:])

    def blob(column):
        strings = list(chain.from_iterable(w[column] for w in windows))
        data = ("\0".join(strings) + "\0").encode() if strings else b"\0"
        ends = np.zeros(total + 1, dtype=np.int64)
        np.cumsum(np.fromiter(map(len, strings), dtype=np.int64, count=total) + 1, out=ends[1:])
        if total and (int(ends[-1]) != len(data) or data.count(b"\0") != total):
            raise ValueError("realign_windows: a read or CIGAR string with NUL or non-ASCII characters")
        return data, ends[first]                           # the buffer, byte offset of every window's first string (and the end)

    seq_blob, seq_at = blob(0)
    cig_blob, cig_at = blob(2)
    seq_base = C.cast(C.c_char_p(seq_blob), C.c_void_p).value
    cig_base = C.cast(C.c_char_p(cig_blob), C.c_void_p).value
    pos_all = np.fromiter(chain.from_iterable(w[1] for w in windows), dtype=np.int32, count=total)
    out_pos_all = np.empty(max(total, 1), dtype=np.int32)
    off_all = np.zeros(total + n + 1, dtype=np.int64)      # window i: m + 1 offsets from first[i] + i
    caps = 64 * counts + 8 * np.diff(seq_at) + np.diff(cig_at) + 64 if n else np.zeros(0, dtype=np.int64)
    buf_at = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(caps, out=buf_at[1:])
    buf_all = np.empty(max(int(buf_at[-1]), 1), dtype=np.uint8)      # not zeroed: the call writes what off[] then delimits
    p_pos, p_out, p_off, p_buf = pos_all.ctypes.data, out_pos_all.ctypes.data, off_all.ctypes.data, buf_all.ctypes.data
    keep = [seq_blob, cig_blob, pos_all, out_pos_all, off_all, buf_all]
    for i, (j, w) in enumerate(zip(jobs, windows)):
        f = int(first[i])
        j.n_reads = int(counts[i])
        j.seqs, j.cigars = None, None
        j.seqs_joined, j.cigars_joined = seq_base + int(seq_at[i]), cig_base + int(cig_at[i])
        j.positions = p_pos + 4 * f
        ref, haps = w[3].encode(), " ".join(w[4]).encode()
        keep.append((ref, haps))
        j.reference, j.haplotypes = ref, haps
        j.ref_start, j.ref_prefix, j.ref_suffix = w[5], w[6], w[7]
        j.out_positions, j.cigar_buf, j.cigar_cap, j.cigar_off = p_out + 4 * f, p_buf + int(buf_at[i]), int(caps[i]), p_off + 8 * (f + i)
    st = RealignStats()
    stream = None
    if where == "device":
        from ._lib import current_stream_ptr
        stream = current_stream_ptr()
    rc = lib.cto_realign_windows(n, jobs, 1 if where == "device" else 0, int(threads), stream, C.byref(st))
    if statuses is not None:          # the caller sorts the failures out window by window (a list of codes, then the first error's text)
        statuses[:] = [int(j.status) for j in jobs[:n]] + [lib.cto_last_error().decode("utf-8", "replace") if rc != 0 else ""]
        if rc != 0 and not any(statuses[:n]):
            check(rc)
    else:
        check(rc)
    if stats is not None:
        stats.update({k: getattr(st, k) for k, _ in RealignStats._fields_})
    out = []
    for i in range(n):
        f, m = int(first[i]), int(counts[i])
        at = int(buf_at[i])
        text = buf_all[at:at + int(off_all[f + i + m])].tobytes().decode() if m else ""
        out.append((out_pos_all[f:f + m].tolist(), text[:-1].split("\0") if m else []))
    return out


def _pair_codes(pairs):
    flat = [np.ascontiguousarray(x, dtype=np.int8) for p in pairs for x in p]
    codes = np.concatenate(flat) if flat else np.zeros(0, dtype=np.int8)
    lens = np.fromiter((len(x) for x in flat), dtype=np.int64, count=len(flat))
    offs = np.zeros(len(flat) + 1, dtype=np.int64)
    np.cumsum(lens, out=offs[1:])
    desc = np.stack([offs[0:-1:2], lens[0::2], offs[1::2], lens[1::2]], axis=1).astype(np.int32) if pairs else np.zeros((0, 4), dtype=np.int32)
    return codes, np.ascontiguousarray(desc)


def ssw_align_batch(pairs, where="device", threads=0):
    """Aligner::Align for many (reference, query) pairs of base codes in one call (cto_ssw_align_batch): returns (scores, ref_begins,
    list of CIGAR strings)."""
    n = len(pairs)
    codes, desc = _pair_codes(pairs)
    score, begin = np.zeros(n, dtype=np.int32), np.zeros(n, dtype=np.int32)
    cap = int(16 * n + 12 * codes.size + 64)
    buf, off = np.empty(cap, dtype=np.uint8), np.zeros(n + 1, dtype=np.int64)
    stream = None
    if where == "device":
        from ._lib import current_stream_ptr
        stream = current_stream_ptr()
    check(lib.cto_ssw_align_batch(n, codes.ctypes.data, codes.size, desc.ctypes.data, 1 if where == "device" else 0, int(threads), stream,
                                  score.ctypes.data, begin.ctypes.data, buf.ctypes.data, cap, off.ctypes.data))
    text = buf[:int(off[n])].tobytes().decode()
    return score, begin, (text[:-1].split("\0") if n else [])


def sw_ends_batch(pairs, where="device", threads=0):
    """The striped Smith-Waterman passes of many (reference, query) pairs in one call (cto_sw_ends_batch; ssw.c:781-830): pairs = list of
    (ref_codes, query_codes) int8 arrays of base codes 0..4; returns an [n, 6] int32 array {score, ref_end, read_end, ref_begin,
    read_end - read_begin, lanes}."""
    n = len(pairs)
    flat = [np.ascontiguousarray(x, dtype=np.int8) for p in pairs for x in p]
    codes = np.concatenate(flat) if flat else np.zeros(0, dtype=np.int8)
    lens = np.fromiter((len(x) for x in flat), dtype=np.int64, count=len(flat))
    offs = np.zeros(len(flat) + 1, dtype=np.int64)
    np.cumsum(lens, out=offs[1:])
    desc = np.stack([offs[0:-1:2], lens[0::2], offs[1::2], lens[1::2]], axis=1).astype(np.int32) if n else np.zeros((0, 4), dtype=np.int32)
    desc = np.ascontiguousarray(desc)
    out = np.zeros((n, 6), dtype=np.int32)
    stream = None
    if where == "device":
        from ._lib import current_stream_ptr
        stream = current_stream_ptr()
    check(lib.cto_sw_ends_batch(n, codes.ctypes.data, codes.size, desc.ctypes.data, 1 if where == "device" else 0, int(threads), stream, out.ctypes.data))
    return out


class WindowBatcher(object):
    """realign_fn of many calls at once.  The reference starts one `realign_reads` process per low-QUAL call and each hands its
    windows to the native realigner one by one (src/realign_variants.py:73-110, src/realign_reads.py:582-595).  Here the calls of
    a run are worker THREADS; a worker that needs a window realigned parks it here and sleeps, and when every live worker is
    parked (or `max_batch` windows wait) one cto_realign_windows call does them all - on the device three stages of launches per batch.
    What a call sees is what realign_window returns, so its SAM text does not depend on who shared the batch.
        with WindowBatcher("device", threads=8) as b:      # b.worker() brackets a worker thread's life
            ...
    """

    def __init__(self, where="device", threads=0, max_batch=8192, device=None, host_fallback=False):
        import threading
        self.where, self.threads, self.max_batch = where, int(threads), int(max_batch)
        # host_fallback (--realigner auto): windows a device batch gives back with an error are done again by the host form - the same
        # output - and the run goes on, with one line on stderr; an explicit `--realigner device` fails loudly instead
        self.host_fallback, self.fell_back = bool(host_fallback), 0
        self.cv = threading.Condition()
        self.queue, self.live, self.parked, self.closed = [], 0, 0, False
        self.batches, self.windows, self.stats = 0, 0, {}
        self.device = device
        self.thread = threading.Thread(target=self._run, name="cto-window-batcher", daemon=True)
        self.thread.start()

    # -- worker side
    def worker(self):
        b = self

        class _W(object):
            def __enter__(self_):
                with b.cv:
                    b.live += 1

            def __exit__(self_, *exc):
                with b.cv:
                    b.live -= 1
                    b.cv.notify_all()
        return _W()

    def __call__(self, seqs, positions, cigars, ref_seq, haplotypes, ref_start, prefix_len, suffix_len):
        slot = {"args": (seqs, positions, cigars, ref_seq, haplotypes, ref_start, prefix_len, suffix_len), "done": False, "out": None, "err": None}
        with self.cv:
            if self.closed:
                raise RuntimeError("WindowBatcher is closed")
            self.queue.append(slot)
            self.parked += 1
            self.cv.notify_all()
            while not slot["done"]:
                self.cv.wait()                                   # (the dispatcher took this slot out of `parked` when it marked it done)
        if slot["err"] is not None:
            raise RuntimeError(slot["err"])
        return slot["out"]

    # -- dispatcher
    def _run(self):
        if self.where == "device":
            import torch
            if self.device is not None:
                torch.cuda.set_device(self.device)
        while True:
            with self.cv:
                # dispatch when nobody is left to add to the batch: every live worker is parked (a caller outside worker() counts
                # as one that is), or the batch is full
                while not self.closed and not (self.queue and (self.parked >= max(self.live, 1) or len(self.queue) >= self.max_batch)):
                    self.cv.wait(0.05 if self.queue else None)
                if self.closed and not self.queue:
                    return
                batch, self.queue = self.queue, []
            st = {}
            self._realign(batch, self.where, st)
            failed = [b for b in batch if b["err"] is not None]
            if failed and self.host_fallback and self.where == "device":
                import sys
                if not self.fell_back:
                    sys.stderr.write("[WARNING] realigner: the device form gave up on %d window(s) (%s); the host form does them (--realigner auto)\n"
                                     % (len(failed), failed[0]["err"]))
                self.fell_back += len(failed)
                for b in failed:
                    b["err"] = None
                self._realign(failed, "host", {})
            with self.cv:
                self.batches += 1
                self.windows += len(batch)
                self.parked -= len(batch)                         # under the lock, together with `done`: a worker that wakes and re-queues
                                                                  # at once must not be counted twice while the others are still waking
                for k, v in st.items():
                    self.stats[k] = self.stats.get(k, 0) + v
                for b in batch:
                    b["done"] = True
                self.cv.notify_all()

    def _realign(self, batch, where, st):
        codes = []
        try:
            outs = realign_windows([b["args"] for b in batch], where=where, threads=self.threads, stats=st, statuses=codes)
            for b, o, code in zip(batch, outs, codes):
                if code == 0:
                    b["out"] = o
                else:
                    b["err"] = "cto_realign_windows: status %d (%s)" % (code, codes[-1])
        except Exception as e:          # the whole batch failed (out of device memory, ...): every caller hears of it
            for b in batch:
                b["err"] = str(e)

    def close(self):
        with self.cv:
            self.closed = True
            self.cv.notify_all()
        self.thread.join()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def _cigar_ops(cigar):
    n = 0
    for ch in cigar:
        if ch.isdigit():
            n = n * 10 + int(ch)
        else:
            yield ch, n
            n = 0


def mostly_clipped(cigar):
    """:233-248 - less than 55 % of the CIGAR's positions are not soft-clipped"""
    soft = total = 0
    for op, n in _cigar_ops(cigar):
        if op == "S":
            soft += n
        total += n
    return 1.0 - float(soft) / (total + 1) < 0.55


class Read(object):
    __slots__ = ("name", "flag", "start", "end", "mq", "cigar", "seq", "bq", "raw_bq", "rnext", "pnext", "hp", "best_cigar", "best_pos",
                 "best_score")

    def __init__(self, name, flag, start, mq, cigar, seq, raw_bq, rnext, pnext, hp):
        self.name, self.flag, self.start, self.mq, self.cigar, self.seq, self.raw_bq = name, flag, start, mq, cigar, seq, raw_bq
        self.rnext, self.pnext, self.hp = rnext, pnext, hp
        self.bq = raw_bq.encode("latin-1")            # phred + 33 per base (compare against threshold + 33)
        self.end = start + len(seq) + (sum(n for op, n in _cigar_ops(cigar) if op == "D") if "D" in cigar else 0)     # :92-98
        self.best_cigar, self.best_pos, self.best_score = cigar, start, None

    def offer(self, cigar, pos):
        """set_realignment_info (:153-164)"""
        cigar = cigar.replace("X", "M")
        if cigar == self.cigar and pos == self.start:
            return
        if self.best_score and cigar == self.best_cigar and pos == self.best_pos:
            return
        score = sum(n for op, n in _cigar_ops(cigar) if op in "ID")
        if not self.best_score or score >= self.best_score:
            self.best_cigar, self.best_pos, self.best_score = cigar, pos, score

    def sam(self, ctg):
        # TLEN repeats PNEXT (:133), and the HP field is there even when empty (:623-629)
        return "\t".join([self.name, self.flag, ctg, str(self.best_pos + 1), str(self.mq), self.best_cigar, self.rnext, self.pnext, self.pnext,
                          self.seq, self.raw_bq, "HP:i:%s" % self.hp if self.hp else ""]) + "\n"


def _hp_of(fields):
    tags = [c for c in fields if "HP:i:" in c]
    if not tags or len(tags[0]) < 6 or not tags[0][5].isdigit():
        return None
    return tags[0][5]


class RegionRealigner(object):
    """The state `reads_realignment` keeps while it streams the rows of one `samtools view -h` call (:427-652)."""

    def __init__(self, ctg_name, reference_sequence, reference_start_0_based, pos, min_coverage=2, max_distance=50,
                 consensus_fn=dbg_consensus, realign_fn=realign_window):
        self.ctg, self.ref, self.ref0, self.pos = ctg_name, reference_sequence, reference_start_0_based, pos
        self.min_coverage, self.max_distance = min_coverage, max_distance
        self.consensus_fn, self.realign_fn = consensus_fn, realign_fn
        self.header, self.header_out = [], False
        self.reads = {}
        self.evidence = {}
        self.chunk_start = self.chunk_end = None
        try:
            self._ref_b = reference_sequence.encode("ascii")
        except UnicodeEncodeError:
            self._ref_b = b""                          # the C scan then declines every read and the Python loop runs
        self._ev_buf = (C.c_int32 * 8192)()

    # ---- rows in
    def _count(self, lo, hi):
        ev = self.evidence
        for p in range(lo, hi):
            ev[p] = ev.get(p, 0) + 1

    def feed(self, row, out):
        """one row of `samtools view -h`; realigned rows that are ready go to out.write"""
        if row[0] == "@":
            self.header.append(row)
            return
        c = row.strip().split()
        if c[2] != self.ctg:
            return
        flag, start, mq, cigar, seq, raw_bq = int(c[1]), int(c[3]) - 1, int(c[4]), c[5], c[9].upper(), c[10]
        strand = int((flag & 16) == 16)
        if self.chunk_start is None:
            self.chunk_start, self.chunk_end = start, start + CHUNK
        if start >= self.chunk_end + EXPAND:
            self.flush(out)
            self.chunk_start += CHUNK
            self.chunk_end += CHUNK
        read = Read(c[0] + "_" + str(strand), str(flag), start, mq, cigar, seq, raw_bq, c[6], c[7], _hp_of(c[11:]))
        if cigar == "*" or mostly_clipped(cigar):
            return
        self.reads[read.name] = read
        if mq < MIN_DBG_MQ:
            return
        ref, ref0, bq = self.ref, self.ref0, read.bq
        lo_ok, hi_ok = self.chunk_start - EXPAND, self.chunk_end + EXPAND
        # the per-base scan in C (cto_realign_read_evidence); -1 = it met something the Python loop below would raise on
        cap = len(self._ev_buf)
        try:
            seq_b = seq.encode("ascii")
        except UnicodeEncodeError:
            seq_b = None
        k = -1 if seq_b is None else lib.cto_realign_read_evidence(seq_b, len(seq_b), bq, len(bq), cigar.encode("latin-1"), start, self._ref_b,
                                                                   len(self._ref_b), ref0, lo_ok, hi_ok, MIN_DBG_BQ, self._ev_buf, cap)
        if k >= 0:
            ev, buf = self.evidence, self._ev_buf
            for i in range(k):
                p = buf[i]
                ev[p] = ev.get(p, 0) + 1
            return
        rp, qp = start, 0
        thr = MIN_DBG_BQ + 33
        for op, n in _cigar_ops(cigar):
            if op == "=":
                rp += n
                qp += n
            elif op == "M" or op == "X":
                for _ in range(n):
                    if bq[qp] >= thr:
                        rb = ref[rp - ref0]
                        if rb in "ACGT" and seq[qp] != rb:
                            self.evidence[rp] = self.evidence.get(rp, 0) + 1
                    rp += 1
                    qp += 1
            elif op == "I" or op == "S":
                if lo_ok <= rp <= hi_ok and ref[rp - ref0 - 1] in "ACGT" and not any(q < thr for q in bq[qp:qp + n]):
                    self._count(rp - n, rp + n)
                qp += n
            elif op == "D":
                if lo_ok <= rp <= hi_ok and ref[rp - ref0 - 1] in "ACGT":
                    self._count(rp, rp + n)
                rp += n
            # N, H, P: the reference moves neither cursor (:309-352)

    # ---- a chunk boundary (or the end of the input)
    def flush(self, out):
        if self.chunk_start is None:
            return
        if not self.header_out:
            out.write("".join(self.header))
            self.header_out = True
        cs, ce = self.chunk_start, self.chunk_end
        cand = sorted(p for p, n in self.evidence.items()
                      if n >= self.min_coverage and cs - EXPAND - 1 <= p <= ce + EXPAND - 1
                      and self.pos - self.max_distance <= p < self.pos + self.max_distance)
        if not self.reads or not cand:
            return
        for idx in range((ce - cs) // MAX_WINDOW):
            lo = cs + idx * MAX_WINDOW - EXPAND - 1
            hi = lo + MAX_WINDOW + EXPAND * 2 + 1
            self._realign_split([p for p in cand if lo <= p < hi])
        behind = cs - EXPAND - MAX_WINDOW
        for name, _ in sorted(((k, r.best_pos) for k, r in self.reads.items()), key=lambda kv: kv[1]):
            if self.reads[name].best_pos < behind:
                out.write(self.reads.pop(name).sam(self.ctg))
        for p in [p for p in self.evidence if p < behind]:
            del self.evidence[p]

    def finish(self, out):
        self.flush(out)
        for name, _ in sorted(((k, r.best_pos) for k, r in self.reads.items()), key=lambda kv: kv[1]):
            out.write(self.reads.pop(name).sam(self.ctg))

    def _realign_split(self, positions):
        windows, first, last = [], None, None
        for p in positions:
            if first is None:
                first = last = p
            elif p > last + 2 * WINDOW_GAP:
                windows.append((first - WINDOW_GAP, last + WINDOW_GAP))
                first = last = p
            else:
                last = p
        if first is None:
            return
        windows.append((first - WINDOW_GAP, last + WINDOW_GAP))
        windows.sort(key=lambda w: w[0])
        reach = max(w[1] for w in windows)
        members = [[] for _ in windows]
        for name, r in self.reads.items():
            if r.start > reach:
                continue
            best, best_len = None, 0
            for i, (ws, we) in enumerate(windows):
                ov = min(r.end, we) - max(r.start, ws)
                if ov > best_len:
                    best, best_len = i, ov
            if best is not None:
                members[best].append(name)
        for (ws, we), names in zip(windows, members):
            if we - ws > MAX_WINDOW:
                continue
            self._realign_window(ws, we, names)

    def _realign_window(self, ws, we, names):
        ref, ref0 = self.ref, self.ref0
        centre = ref[ws - ref0:we - ref0]
        graph_reads, graph_lowbq = [], []
        for name in names:
            r = self.reads[name]
            if r.mq < GRAPH_MIN_MQ or r.start > we or r.end < ws:
                continue
            graph_reads.append(r.seq)
            graph_lowbq.append([i for i, q in enumerate(r.bq) if q < GRAPH_LOW_BQ + 33])
        consensus = self.consensus_fn(centre, graph_reads, graph_lowbq)
        if not consensus or (len(consensus) == 1 and consensus[0] == centre) or not names:
            return
        lo = max(0, min(min(self.reads[n].start for n in names), ws) - EXPAND)
        hi = max(max(self.reads[n].end for n in names), we) + EXPAND
        prefix, suffix = ref[lo - ref0:ws - ref0], ref[we - ref0:hi - ref0]
        take = [self.reads[n] for n in names[:MAX_READS]]
        new_pos, new_cigar = self.realign_fn([r.seq for r in take], [r.start for r in take], [r.cigar for r in take],
                                             prefix + centre + suffix, [prefix + h + suffix for h in consensus], lo, len(prefix), len(suffix))
        for r, p, cg in zip(take, new_pos, new_cigar):
            if cg == "" or (r.cigar == cg and r.start == p):
                continue
            r.offer(cg, p)


def realign_region(rows, ctg_name, reference_sequence, reference_start_0_based, pos, out, min_coverage=2, max_distance=50,
                   consensus_fn=dbg_consensus, realign_fn=realign_window):
    """rows of `samtools view -h` -> realigned SAM text written to out."""
    rr = RegionRealigner(ctg_name, reference_sequence, reference_start_0_based, pos, min_coverage, max_distance, consensus_fn, realign_fn)
    for row in rows:
        rr.feed(row, out)
    rr.finish(out)


def region_of(pos, flanking):
    """the read and reference regions `reads_realignment` fetches around --pos (:359-391): 1-based, inclusive"""
    ctg_start, ctg_end = pos - flanking, pos + flanking
    return (ctg_start - MAX_WINDOW, ctg_end + MAX_WINDOW), (max(1, ctg_start - REF_EXPAND), ctg_end + REF_EXPAND)


def faidx(samtools, ref_fn, region):
    """shared/utils.py:148-174 - the sequence of `samtools faidx`, upper-cased"""
    p = subprocess.run(shlex.split("{} faidx {} {}".format(samtools, ref_fn, region)), stdout=subprocess.PIPE, universal_newlines=True)
    if p.returncode != 0:
        return None
    return "".join(p.stdout.split("\n")[1:]).upper()


def bam_view(bam_fn, ctg_name, start, end, min_mq=0, bai_fn=None):
    """rows of `samtools view bam ctg:start-end -q min_mq` from the built-in BAM reader (cto_bam_view; PARITY UNPINNED)"""
    cap = 1 << 20
    while True:
        buf = C.create_string_buffer(cap)
        need = C.c_size_t(0)
        n = lib.cto_bam_view(str(bam_fn).encode(), str(bai_fn).encode() if bai_fn else None, ctg_name.encode(), int(start), int(end), int(min_mq),
                             buf, cap, C.byref(need))
        if n == -3 and need.value > cap:
            cap = need.value + 1024
            continue
        check(int(n))
        return buf.raw[:need.value].decode().splitlines(True)


def reads_realignment(args, out=None):
    out = out or sys.stdout
    (rd_lo, rd_hi), (ref_lo, ref_hi) = region_of(args.pos, args.realign_flanking_window)
    if getattr(args, "bam_reader", "samtools") == "native":     # no samtools: built-in BAM and FASTA readers
        from .fasta import read_region
        ref = read_region(args.ref_fn, args.ctg_name, ref_lo, ref_hi)
        if not ref:
            sys.exit("[ERROR] Failed to load reference sequence from file ({}).".format(args.ref_fn))
        rows = bam_view(args.bam_fn, args.ctg_name, max(1, rd_lo), rd_hi, args.min_mq if args.min_mq > 0 else 0)
        realign_region(rows, args.ctg_name, ref, ref_lo - 1, args.pos, out, args.min_coverage, args.max_distance,
                       realign_fn=getattr(args, "realign_fn", None) or realign_window)
        return
    ref = faidx(args.samtools, args.ref_fn, "{}:{}-{}".format(args.ctg_name, ref_lo, ref_hi))
    if not ref:
        sys.exit("[ERROR] Failed to load reference sequence from file ({}).".format(args.ref_fn))
    cmd = "{} view -h {} {}:{}-{}".format(args.samtools, args.bam_fn, args.ctg_name, rd_lo, rd_hi) + (" -q {}".format(args.min_mq) if args.min_mq > 0 else "")
    view = subprocess.Popen(shlex.split(cmd), stdout=subprocess.PIPE, universal_newlines=True, bufsize=8388608)
    realign_region(view.stdout, args.ctg_name, ref, ref_lo - 1, args.pos, out, args.min_coverage, args.max_distance,
                   realign_fn=getattr(args, "realign_fn", None) or realign_window)
    view.stdout.close()
    view.wait()


def build_parser():
    p = ArgumentParser(description="Reads realignment around one position (native consensus + realigner)")
    p.add_argument("--bam_fn", type=str, default=None)
    p.add_argument("--ref_fn", type=str, default=None)
    p.add_argument("--read_fn", type=str, default="PIPE", help="accepted for compatibility: the realigned SAM text goes to stdout")
    p.add_argument("--ctg_name", type=str, default=None)
    p.add_argument("--samtools", type=str, default="samtools")
    p.add_argument("--min_coverage", type=float, default=2)
    p.add_argument("--min_mq", type=int, default=20)                # shared/param.py:17
    p.add_argument("--realign_flanking_window", type=int, default=100)
    p.add_argument("--pos", type=int, default=None)
    p.add_argument("--max_distance", type=int, default=50, help=SUPPRESS)
    p.add_argument("--bam_reader", type=str, default="samtools", choices=["samtools", "native"],
                   help="samtools: `samtools view` / `faidx` as the reference runs them; native: the built-in BAM / FASTA readers (parity unpinned)")
    for compat in ("--ctg_start", "--ctg_end", "--bed_fn", "--extend_bed", "--test_pos"):
        p.add_argument(compat, default=None, help=SUPPRESS)
    return p


def main(argv=None):
    args = build_parser().parse_args(argv)
    if args.pos is None or args.ctg_name is None:
        sys.exit("[ERROR] clairs_to_amd realign_reads needs --pos and --ctg_name (the form `realign_variants` calls)")
    reads_realignment(args)


if __name__ == "__main__":
    main()

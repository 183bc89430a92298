"""Pin-on-arrival for the BAM -> column-pack producer ("parity unpinned").
    python tools/pin_bam.py [--samtools samtools] [--keep DIR]            (tools/pin_bam.sh is the one-command form)
Needs a real `samtools` (none in the build image, none on the GPU box).  On the BAMs the test-suite writes (tests/bamutil.py: random CIGARs
incl. N / P / leading insertions / insertions after deletions, soft and hard clips, `=`/X, IUPAC bases, missing qualities, CG-tag CIGARs,
secondary / supplementary / duplicate flags, mate pairs that overlap and disagree, a spot deeper than --max-depth) and on the SAM
specification's example assembled byte by byte (tests/test_bam_bytes_by_hand.py), it runs the reference's own command
    samtools mpileup --reverse-del --output-MQ -r CTG:S-E --min-MQ 0 --min-BQ Q [-l BED] --excl-flags 2316 [--max-depth D] BAM
(src/create_tensor_pileup_calling.py:426-446) for Q = 0 and Q = 20 and compares
  (1) the pack tokenised from the Q = 0 text (cto_pack_from_mpileup, pinned to the reference's decoder) with cto_pack_from_bam's pack of the
      same region: every array and every indel key string;
  (2) the Q = 20 text - the AFF pass - with what the kernels select from the Q = 0 pack (read-bases with BQ >= 20 in the same order;
      a column all of whose read-bases fall below 20 is still a row of depth 0 for samtools).
Exit code = verdict: 0 all equal, 1 a difference (the first rows of it printed), 2 cannot run here (no samtools)."""
import argparse
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)


def arrays(pack):
    a = {k: v.copy() for k, v in pack.numpy().items()}
    a["keys"] = [pack.key_string(k) for k in range(pack.n_keys)]
    return a


def first_difference(a, b):
    for k in ("col_pos", "col_ref", "col_off", "key_off", "entries", "key_meta", "key_group"):
        if a[k].shape != b[k].shape or not np.array_equal(a[k], b[k]):
            n = min(len(a[k]), len(b[k]))
            bad = np.nonzero(a[k][:n] != b[k][:n])[0]
            at = int(bad[0]) if len(bad) else n
            return "%s: lengths %d / %d, first difference at %d (%s / %s)" % (k, len(a[k]), len(b[k]), at, a[k][at:at + 4], b[k][at:at + 4])
    if a["keys"] != b["keys"]:
        return "key strings differ"
    return None


def bq_filtered(a, q):
    """the sub-pack the kernels read for --min-BQ q: entries with BQ >= q, per column, order kept -> {position: [entry words]}"""
    ent, off, pos = a["entries"], a["col_off"], a["col_pos"]
    bq = (ent >> 6) & 0x7f                                     # include/clairsto_amd.h: base 4 b | indel kind 2 b | BQ 7 b | MQ 8 b | key 11 b
    out = {}
    for c in range(len(pos)):
        e = ent[off[c]:off[c + 1]]
        out[int(pos[c])] = e[bq[off[c]:off[c + 1]] >= q]
    return out


def mpileup(samtools, bam, ctg, s, e, q, bed_fn, max_depth):
    cmd = [samtools, "mpileup", "--reverse-del", "--output-MQ", "-r", "%s:%d-%d" % (ctg, s, e), "--min-MQ", "0", "--min-BQ", str(q)]
    if bed_fn:
        cmd += ["-l", bed_fn]
    cmd += ["--excl-flags", "2316"]
    if max_depth is not None:
        cmd += ["--max-depth", str(max_depth)]
    p = subprocess.run(cmd + [bam], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    if p.returncode != 0:
        sys.exit("pin_bam: %s failed: %s" % (" ".join(cmd), p.stderr.decode()[-400:]))
    return p.stdout


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--samtools", default="samtools")
    ap.add_argument("--keep", default=None, help="leave the BAMs and texts in this directory")
    a = ap.parse_args()
    if shutil.which(a.samtools) is None:
        print("pin_bam: `%s` not found: cannot pin here" % a.samtools)
        sys.exit(2)
    v = subprocess.run([a.samtools, "--version"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT).stdout.decode().split("\n")[0]
    if "shim" in v.lower() or "clisim" in v.lower():
        print("pin_bam: `%s` is a stand-in (%s), not samtools: cannot pin here" % (a.samtools, v))
        sys.exit(2)
    from bamutil import write_bam
    from test_bam_reader import _random_reads
    from clairs_to_amd.pack import ColumnPack
    tmp = a.keep or tempfile.mkdtemp(prefix="pin_bam_")
    os.makedirs(tmp, exist_ok=True)
    cases = []
    for seed, paired in ((1, 0.0), (2, 0.5), (3, 0.0), (4, 0.9)):
        rng = np.random.default_rng(seed)
        ref_lens = [40000, 3000]
        refs = [("chrA", ref_lens[0]), ("chrB", ref_lens[1])]
        ref_seqs = ["".join(rng.choice(list("ACGTN"), p=[.24, .24, .24, .24, .04], size=L)) for L in ref_lens]
        reads = _random_reads(rng, 900, ref_lens, paired_frac=paired)
        for i in range(40):
            reads.append(dict(name="d%d" % i, flag=16 * (i & 1), ref=0, pos=20000 + (i % 3), mapq=60, cigar=[("M", 30)], seq="ACGT" * 7 + "AC", qual=[30] * 30))
        reads.sort(key=lambda r: (r["ref"], r["pos"]))
        bam = os.path.join(tmp, "t%d.bam" % seed)
        write_bam(bam, refs, reads, block_payload=1500 if seed != 3 else 60000)
        for ref_i, s, e, bed, md in [(0, 1, ref_lens[0], None, 8000), (0, 5000, 9000, None, 8000), (0, 16380, 16400, None, 8000), (0, 19990, 20040, None, 10),
                                     (1, 1, ref_lens[1], None, 8000), (1, 700, 2400, [(650, 720), (900, 934), (2000, 2500)], 8000), (0, 39000, 40000, [(38990, 39010)], 8000)]:
            cases.append((bam, refs[ref_i][0], s, e, bed, md, ref_seqs[ref_i]))
    try:
        import pathlib
        import test_bam_bytes_by_hand as by_hand                            # the SAM specification's example, byte by byte
        hand_dir = pathlib.Path(tmp) / "by_hand"
        hand_dir.mkdir(exist_ok=True)
        cases.append((by_hand._write(hand_dir), "ref", 1, 45, [(6, 21), (35, 45)], 8000, by_hand.REF))
    except Exception as ex:                                                 # the hand-made BAM is a bonus case
        print("pin_bam: (the by-hand BAM is not available as a function: %s)" % ex)
    bad = 0
    for k, (bam, ctg, s, e, bed, md, ref) in enumerate(cases):
        bed_fn = None
        if bed:
            bed_fn = os.path.join(tmp, "case%d.bed" % k)
            open(bed_fn, "w").write("".join("%s\t%d\t%d\n" % (ctg, b0, b1) for b0, b1 in bed))
        t0 = mpileup(a.samtools, bam, ctg, s, e, 0, bed_fn, md)
        t20 = mpileup(a.samtools, bam, ctg, s, e, 20, bed_fn, md)
        if a.keep:
            open(os.path.join(tmp, "case%d.q0.txt" % k), "wb").write(t0)
            open(os.path.join(tmp, "case%d.q20.txt" % k), "wb").write(t20)
        want = arrays(ColumnPack.from_mpileup(t0, ref, 1))
        got = arrays(ColumnPack.from_bam(bam, ctg, s, e, ref, 1, bed=bed, max_depth=md))
        d = first_difference(got, want)
        if d is None:                                                       # the AFF pass: what BQ >= 20 selects from the one pack
            sel = bq_filtered(got, 20)
            w20 = arrays(ColumnPack.from_mpileup(t20, ref, 1))
            for c in range(len(w20["col_pos"])):
                p = int(w20["col_pos"][c])
                e20 = w20["entries"][w20["col_off"][c]:w20["col_off"][c + 1]]
                mine = sel.get(p, np.zeros(0, dtype=e20.dtype))
                # key ids are per-column first-seen ids of EACH text: compare everything but the key id (low 21 bits), the key strings through (1)
                if len(mine) != len(e20) or not np.array_equal(mine & 0x1fffff, e20 & 0x1fffff):
                    d = "--min-BQ 20 pass: position %d has %d read-bases in samtools' text, %d selected from the pack" % (p, len(e20), len(mine))
                    break
        print("case %2d %s %s:%d-%d bed=%s max_depth=%s: %s" % (k, os.path.basename(bam), ctg, s, e, bool(bed), md, "equal" if d is None else "DIFFERENT - " + d))
        bad += d is not None
    print("pin_bam: %d cases, %d differ (%s)" % (len(cases), bad, v))
    if not a.keep:
        shutil.rmtree(tmp, ignore_errors=True)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()

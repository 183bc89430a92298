"""Differential check of the mpileup tokeniser (cto_pack_from_mpileup) between two builds of the library: every pack array, the key
strings, the return code and the error text must be the same for the same input - well-formed rows, rows the single-pass fast path
has to hand to the general path (extra fields, short quality strings, CR LF, '^' or an indel at the end of the field, 8-bit
characters ...) and randomly damaged text.

  python tools/tokeniser_diff.py digest [--cases N] [--seed S]      with CTO_LIB_PATH naming the build: one line per case
  python tools/tokeniser_diff.py compare OLD.so NEW.so [--cases N]  runs `digest` under both and compares the lines
"""
import argparse
import hashlib
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def base_rows(rng, n_rows, ref, start):
    """well-formed rows with everything the grammar has: both strands, '*' '#', N / n, '^x', '$', insertions, deletions, '<' '>'"""
    rows = []
    for r in range(n_rows):
        pos = start + r * int(rng.integers(1, 3))
        depth = int(rng.integers(0, 70))
        bases, nb = [], 0
        for _ in range(depth):
            u = rng.random()
            if u < 0.03:
                bases.append("^" + chr(int(rng.integers(33, 100))))
            c = "ACGTacgt*#NnACGTacgt"[int(rng.integers(0, 20))]
            bases.append(c)
            nb += 1
            u = rng.random()
            if u < 0.08:
                k = int(rng.integers(1, 70 if rng.random() < 0.05 else 6))
                seq = "".join("ACGTNacgtn"[int(x)] for x in rng.integers(0, 5, size=k) + (5 if c.islower() or c == "#" else 0))
                bases.append(("+" if rng.random() < 0.5 else "-") + str(k) + seq)
                if rng.random() < 0.1:                                   # a second indel on the same read-base replaces the first
                    bases.append("+1A")
            if rng.random() < 0.03:
                bases.append("$")
            if rng.random() < 0.01:
                bases.append("<" if rng.random() < 0.5 else ">")
        bq = "".join(chr(int(x)) for x in rng.integers(33, 127, size=nb))
        mq = "".join(chr(int(x)) for x in rng.integers(33, 94, size=nb))
        rows.append("chr1\t%d\tN\t%d\t%s\t%s\t%s" % (pos, depth, "".join(bases) if nb else "*", bq if nb else "*", mq if nb else "*"))
        start = pos
    return rows


def damage(rng, rows, fatal):
    """one unusual thing per selected row; fatal: include the ones that make the call fail (the error text is compared then)"""
    out = []
    for row in rows:
        u = rng.random()
        if not fatal and (0.89 <= u < 0.95):
            u = 0.0
        f = row.split("\t")
        if u < 0.70:
            pass
        elif u < 0.73:
            f.append("extra")
        elif u < 0.76:
            f[5] = f[5][:-1]
        elif u < 0.79:
            f[6] = f[6][: len(f[6]) // 2]
        elif u < 0.81:
            f[4] += "^"
        elif u < 0.83:
            f[4] += "+9AC"
        elif u < 0.85:
            f[6] += " "
        elif u < 0.87:
            f[5] = f[5][:1] + "\x80" + f[5][2:]
        elif u < 0.89:
            f[6] = f[6][:1] + "\x1f" + f[6][2:]
        elif u < 0.91:
            f[4] = "+2AC" + f[4]
        elif u < 0.93:
            f = f[:6]
        elif u < 0.95:
            f[1] = ""
        elif u < 0.97:
            f[4] = f[4] + "-"
        else:
            f[5] += "\t" + "x"
        row = "\t".join(f)
        if rng.random() < 0.03:
            row += "\r"
        out.append(row)
    return out


def digest(cases, seed):
    import ctypes as C
    import numpy as np
    from clairs_to_amd._lib import lib
    from clairs_to_amd.pack import ColumnPack
    rng = np.random.default_rng(seed)
    ref = bytes(rng.choice(np.frombuffer(b"ACGTacgtNn", dtype=np.uint8), size=5000, p=[.22, .22, .22, .22, .02, .02, .02, .02, .02, .02]))
    for case in range(cases):
        rows = base_rows(rng, int(rng.integers(1, 60)), ref, int(rng.integers(1, 100)))
        mode = case % 4
        if mode >= 1:
            rows = damage(rng, rows, fatal=(mode == 2 and rng.random() < 0.3))
        text = ("\n".join(rows) + ("\n" if mode != 3 or rng.random() < 0.5 else "")).encode("latin-1")
        if mode == 3:                                                   # random byte damage
            b = bytearray(text)
            for _ in range(int(rng.integers(1, 4))):
                if b:
                    b[int(rng.integers(0, len(b)))] = int(rng.integers(0, 256))
            text = bytes(b)
        try:
            p = ColumnPack.from_mpileup(text, ref, 1, 60)
        except Exception as e:
            print(case, "ERR", str(e))
            continue
        h = hashlib.sha256()
        a = p.numpy()
        for k in sorted(a):
            h.update(k.encode())
            h.update(np.ascontiguousarray(a[k]).tobytes())
        s = C.c_char_p()
        for k in range(p.n_keys):
            h.update(C.string_at(s, lib.cto_pack_key_string(p._h, k, C.byref(s))) if lib.cto_pack_key_string(p._h, k, C.byref(s)) >= 0 else b"?")
            h.update(b"|")
        print(case, p.n_cols, p.n_entries, p.n_keys, h.hexdigest()[:24])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("what", choices=["digest", "compare"])
    ap.add_argument("libs", nargs="*")
    ap.add_argument("--cases", type=int, default=2000)
    ap.add_argument("--seed", type=int, default=7)
    a = ap.parse_args()
    if a.what == "digest":
        return digest(a.cases, a.seed)
    outs = []
    for lib in a.libs:
        r = subprocess.run([sys.executable, __file__, "digest", "--cases", str(a.cases), "--seed", str(a.seed)], capture_output=True, text=True,
                           env=dict(os.environ, CTO_LIB_PATH=os.path.abspath(lib)))
        if r.returncode:
            sys.exit(r.stderr[-2000:])
        outs.append(r.stdout.split("\n"))
    bad = [(x, y) for x, y in zip(*outs) if x != y]
    n_err = sum(1 for x in outs[0] if " ERR " in x)
    print("%d cases, %d of them error returns, %d differences" % (len(outs[0]) - 1, n_err, len(bad)))
    for x, y in bad[:10]:
        print("  ", x, "\n  ", y)
    sys.exit(1 if bad or len(outs[0]) != len(outs[1]) else 0)


if __name__ == "__main__":
    main()

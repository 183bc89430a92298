"""Synthetic ONT-like pileups for benchmarks and tests.

There are no BAMs, no samtools and no network on either box, so the workload is generated:
  * `SynthChunk` - a vectorised numpy generator that writes the binary column pack directly (what
    the GPU path consumes), for one chunk of candidate sites;
  * `mpileup_text()` - the equivalent `samtools mpileup --reverse-del --output-MQ --min-BQ q` text of a
    chunk, for the CPU paths and for round-trip tests of the text tokeniser.
Distributions follow SURVEY.md 8(d): depth ~ Poisson(mean) clipped to [4, 200], strand Bernoulli(0.5), per
read-base match 0.97 / mismatch 0.01 / deletion placeholder 0.01 / insertion 0.004 / deletion 0.006 (lengths
geometric, capped at 80 so a few exceed max_indel_length), BQ ~ round(N(28, 8)) in [1, 50], MQ = 60 w.p. 0.93
else uniform 0..59, 0.1 % N in the reference, a fixed alternative allele with AF ~ U(0.05, 0.6) at each
candidate's centre column.
"""
import numpy as np

BASE_CHARS = "ACGTacgt*#Nn"

# Generator presets of SURVEY.md 8(d) + the per-platform --min_bq of the AFF pass (shared/param.py:34 min_bq_dict) and
# the indel candidate threshold (shared/param.py:23, run_clairs_to ONT override 0.1).
PLATFORMS = {
    "ont": dict(seed=20260928, min_bq=20, indel_min_af=0.1, generator=dict(depth_mean=50.0)),
    "ilmn": dict(seed=20260930, min_bq=0, indel_min_af=0.05,
                 generator=dict(depth_mean=50.0, p_ins=0.0005, p_del=0.0005,
                                bq_model=("discrete", (11, 25, 37), (0.05, 0.15, 0.80)))),
    "hifi": dict(seed=20261001, min_bq=0, indel_min_af=0.05,
                 generator=dict(depth_mean=75.0, p_ins=0.002, p_del=0.002, bq_model=("uniform", 20, 93))),
}


class SynthChunk:
    """One chunk of `n_sites` candidates on a private stretch of the contig."""

    def __init__(self, n_sites, seed=20260928, depth_mean=50.0, start=100000, spacing=250, max_indel_length=60,
                 p_mismatch=0.01, p_star=0.01, p_ins=0.004, p_del=0.006, bq_mean=28.0, bq_sd=8.0, n_rate=0.001,
                 bq_model=None):
        """bq_model: None -> round(N(bq_mean, bq_sd)) in [1, 50]; ("discrete", values, probs); ("uniform", lo, hi)."""
        rng = np.random.default_rng(seed)
        self.max_indel_length = max_indel_length
        # candidate positions: sorted, mean distance `spacing`, minimum distance 1 (windows may overlap)
        gaps = rng.geometric(1.0 / spacing, size=n_sites).astype(np.int64)
        self.site_pos = (start + np.cumsum(gaps)).astype(np.int32)
        # mpileup rows: positions x-16 .. x+17 of every candidate (the BED interval of the reference)
        win = (self.site_pos[:, None].astype(np.int64) + np.arange(-16, 18)[None, :]).ravel()
        col_pos = np.unique(win)
        col_pos = col_pos[col_pos >= 1]
        n_cols = col_pos.size
        self.col_pos = col_pos.astype(np.int32)
        # reference bases (raw characters incl. N) per column
        ref = rng.integers(0, 4, size=n_cols).astype(np.uint8)
        is_n = rng.random(n_cols) < n_rate
        self.col_ref_char = np.where(is_n, ord("N"), np.frombuffer(b"ACGT", dtype=np.uint8)[ref]).astype(np.uint8)
        self.col_ref = np.where(is_n, 0x80, ref).astype(np.uint8)   # evc_base_from: N -> A (code 0); bit 7 = raw base not ACGT
        depth = np.clip(rng.poisson(depth_mean, size=n_cols), 4, 200).astype(np.int64)
        self.col_off = np.concatenate([[0], np.cumsum(depth)]).astype(np.int64)
        n_ent = int(self.col_off[-1])
        col_of = np.repeat(np.arange(n_cols, dtype=np.int64), depth)
        # alternative allele at centre columns
        centre = np.searchsorted(col_pos, self.site_pos.astype(np.int64))
        alt_af = np.zeros(n_cols)
        alt_af[centre] = rng.uniform(0.05, 0.6, size=n_sites)
        alt_base = (((self.col_ref & 3).astype(np.int64) + rng.integers(1, 4, size=n_cols)) % 4).astype(np.uint8)
        # per read-base draws
        rev = rng.random(n_ent) < 0.5
        u = rng.random(n_ent)
        base = (self.col_ref[col_of] & 3).astype(np.int64)
        mism = u < p_mismatch
        base = np.where(mism, (base + rng.integers(1, 4, size=n_ent)) % 4, base)
        is_alt = rng.random(n_ent) < alt_af[col_of]
        base = np.where(is_alt, alt_base[col_of], base)
        star = (u >= p_mismatch) & (u < p_mismatch + p_star)
        code = np.where(star, 8 + rev, base + 4 * rev).astype(np.uint32)
        v = rng.random(n_ent)
        kind = np.where(star, 0, np.where(v < p_ins, 1, np.where(v < p_ins + p_del, 2, 0))).astype(np.uint32)
        ilen = np.minimum(rng.geometric(0.5, size=n_ent), 80).astype(np.int64)
        ivar = rng.integers(0, 4, size=n_ent).astype(np.int64)
        self._okind = kind.astype(np.uint8)               # indel kind before the over-long gate (text writer)
        gate = np.where(kind == 1, ilen, ilen + 1)
        kind = np.where((kind > 0) & (gate > max_indel_length), 3, kind).astype(np.uint32)
        if bq_model is None:
            bq = np.clip(np.rint(rng.normal(bq_mean, bq_sd, size=n_ent)), 1, 50).astype(np.uint32)
        elif bq_model[0] == "discrete":
            bq = rng.choice(np.asarray(bq_model[1]), size=n_ent, p=np.asarray(bq_model[2])).astype(np.uint32)
        elif bq_model[0] == "uniform":
            bq = rng.integers(int(bq_model[1]), int(bq_model[2]) + 1, size=n_ent).astype(np.uint32)
        else:
            raise ValueError("unknown bq_model %r" % (bq_model,))
        assert bq.max(initial=0) < 128
        mq = np.where(rng.random(n_ent) < 0.93, 60, rng.integers(0, 60, size=n_ent)).astype(np.uint32)
        # ---- distinct indel keys per column, ids in first-seen order ----
        # (over-long indels keep a key: tensor creation ignores them, candidate extraction does not)
        okind = self._okind.astype(np.int64)
        idx = np.nonzero(okind > 0)[0]
        kcode = (okind[idx] << 40) | (code[idx].astype(np.int64) << 32) | (ilen[idx] << 8) | \
            np.where(okind[idx] == 1, ivar[idx], 0)
        gkey = (col_of[idx] << 44) | kcode            # (column, key) identity; kcode < 2^43
        uniq, first_pos, inv = np.unique(gkey, return_index=True, return_inverse=True)
        first_ent = idx[first_pos]                       # entry index of each key's first occurrence
        order = np.argsort(first_ent, kind="stable")     # global key order = column, then first seen
        rank = np.empty_like(order)
        rank[order] = np.arange(order.size)
        key_col = (uniq >> 44)[order]
        n_keys = order.size
        self.key_off = np.searchsorted(key_col, np.arange(n_cols + 1)).astype(np.int32)
        kid = np.zeros(n_ent, dtype=np.uint32)
        kid[idx] = (rank[inv] - self.key_off[col_of[idx]]).astype(np.uint32)
        kc_sorted = (uniq & ((1 << 44) - 1))[order]
        k_kind = (kc_sorted >> 40).astype(np.uint8)
        k_code = ((kc_sorted >> 32) & 0xff).astype(np.uint8)
        fwd = (k_code < 4) | (k_code == 8) | (k_code == 10)
        self.key_len = ((kc_sorted >> 8) & 0xffffff).astype(np.int32)
        self.key_var = (kc_sorted & 0xff).astype(np.int32)
        k_gate = np.where(k_kind == 1, self.key_len, self.key_len + 1)
        self.key_meta = (k_kind | (fwd.astype(np.uint8) << 2) | ((k_gate > max_indel_length).astype(np.uint8) << 3)).astype(np.uint8)
        # merged allele for candidate extraction: insertions by (upper-cased anchor, sequence), deletions by length
        anchor = np.where(k_code < 8, k_code % 4, np.where(k_code == 8, 8, np.where(k_code == 9, 9, 10))).astype(np.int64)
        gcode = np.where(k_kind == 1, (np.int64(1) << 40) | (anchor << 32) | (self.key_len.astype(np.int64) << 8) | self.key_var,
                         (np.int64(2) << 40) | (self.key_len.astype(np.int64) << 8))
        gkey2 = (key_col.astype(np.int64) << 44) | gcode
        gu, gfirst, ginv = np.unique(gkey2, return_index=True, return_inverse=True)
        gorder = np.argsort(gfirst, kind="stable")          # groups in first-seen (= key) order, column-major
        grank = np.empty_like(gorder)
        grank[gorder] = np.arange(gorder.size)
        gcol = (gu >> 44)[gorder]
        goff = np.searchsorted(gcol, np.arange(n_cols + 1))
        self.key_group = (grank[ginv] - goff[key_col]).astype(np.int32)
        assert n_keys == 0 or kid.max() < 2048
        self.entries = (code | (kind << 4) | (bq << 6) | (mq << 13) | (kid << 21)).astype(np.uint32)
        # over-long indels keep their length so the text writer can print them
        self._ilen = ilen.astype(np.int32)
        self._ivar = ivar.astype(np.int32)
        self.n_sites = n_sites

    @classmethod
    def for_platform(cls, platform, n_sites, seed=None, **kw):
        """The generator presets of SURVEY.md 8(d): 'ont' (config 2/3), 'ilmn' (config 4), 'hifi' (config 5)."""
        cfg = dict(PLATFORMS[platform]["generator"])
        cfg.update(kw)
        return cls(n_sites, seed=PLATFORMS[platform]["seed"] if seed is None else seed, **cfg)

    # ---- numpy views in the cto_pack_view layout ----
    def arrays(self):
        return dict(col_pos=self.col_pos, col_ref=self.col_ref, col_off=self.col_off, key_off=self.key_off,
                    entries=self.entries, key_meta=self.key_meta, key_group=self.key_group)

    def variant(self, k, shift=0):
        """A DIFFERENT chunk of the same shape for a fraction of the generator's cost (5 s of a core per 4096-site chunk): every read-base's
        BQ moves by a per-read-base amount in -6 .. +6 (clipped to the generator's 1 .. 50: the AFF pass's --min-BQ gate and the low-BQ
        channels see other reads), every fourth one's MQ drops by 45 (the low-MQ channels, candidate extraction's --min-MQ gate), both from
        a hash of (read-base index, k), and all positions move by `shift`.  k = 0 with shift = 0 is NOT the chunk itself (its BQs move too).
        The text writers and arrays() read the same fields, so a variant is as self-consistent as a generated chunk."""
        import copy
        v = copy.copy(self)
        idx = np.arange(self.entries.size, dtype=np.uint64)
        h = (idx * np.uint64(2654435761 + 2 * int(k)) + np.uint64(40503 * int(k) + 977)) >> np.uint64(7)
        e = self.entries.astype(np.uint32)
        bq = ((e >> np.uint32(6)) & np.uint32(127)).astype(np.int64)
        mq = ((e >> np.uint32(13)) & np.uint32(255)).astype(np.int64)
        bq = np.clip(bq + (h % np.uint64(13)).astype(np.int64) - 6, 1, 50)
        mq = np.where(((h >> np.uint64(5)) & np.uint64(3)) == 3, np.maximum(mq - 45, 0), mq)
        v.entries = ((e & np.uint32(~((127 << 6) | (255 << 13)) & 0xffffffff)) | (bq.astype(np.uint32) << np.uint32(6)) |
                     (mq.astype(np.uint32) << np.uint32(13))).astype(self.entries.dtype)
        if shift:
            v.site_pos = (self.site_pos.astype(np.int64) + int(shift)).astype(self.site_pos.dtype)
            v.col_pos = (self.col_pos.astype(np.int64) + int(shift)).astype(self.col_pos.dtype)
        return v

    def ref_window(self):
        """(ref_seq, ref_start): a reference string covering every column +- 100 bp; gaps are 'A'."""
        lo = int(self.col_pos[0]) - 100
        hi = int(self.col_pos[-1]) + 100
        lo = max(lo, 1)
        seq = np.full(hi - lo + 1, ord("A"), dtype=np.uint8)
        seq[self.col_pos.astype(np.int64) - lo] = self.col_ref_char
        return seq.tobytes().decode(), lo

    def site_ref_seq(self):
        """33-character reference context of every site (raw, upper case), as the tensor text carries it."""
        ref, lo = self.ref_window()
        return [ref[p - 16 - lo: p + 17 - lo] for p in self.site_pos.tolist()]


def _ins_seq(length, var, lower):
    s = "".join("ACGT"[(var + i) % 4] for i in range(length))
    return s.lower() if lower else s


def mpileup_text(chunk, min_bq=0, ctg="chr1", col_range=None, min_mq=0, with_mq=True):
    """`samtools mpileup --reverse-del [--output-MQ] --min-MQ min_mq --min-BQ min_bq` text of a SynthChunk (Python
    loop: small n).  with_mq=False drops the MQ column (candidate extraction runs samtools without --output-MQ);
    positions whose reads are all below min_mq print no row (the read-level filter removes them from the pileup)."""
    rows = []
    c0, c1 = col_range if col_range else (0, chunk.col_pos.size)
    ent, off = chunk.entries, chunk.col_off
    for c in range(c0, c1):
        toks, bqs, mqs, n_reads = [], [], [], 0
        for e in range(int(off[c]), int(off[c + 1])):
            x = int(ent[e])
            code, kind, bq, mq = x & 15, (x >> 4) & 3, (x >> 6) & 127, (x >> 13) & 255
            if mq < min_mq:
                continue
            n_reads += 1
            if bq < min_bq:
                continue
            t = BASE_CHARS[code]
            if kind:
                ln = int(chunk._ilen[e])
                lower = code in (4, 5, 6, 7, 9, 11)
                if int(chunk._okind[e]) == 1:
                    t += "+%d%s" % (ln, _ins_seq(ln, int(chunk._ivar[e]), lower))
                else:
                    t += "-%d%s" % (ln, ("n" if lower else "N") * ln)
            toks.append(t)
            bqs.append(chr(bq + 33))
            mqs.append(chr(min(mq, 93) + 33))
        n = len(toks)
        if n_reads == 0:
            continue
        tail = ("\t" + ("".join(mqs) if n else "*")) if with_mq else ""
        if n == 0:   # samtools prints placeholders when every base was filtered (bam_plcmd.c)
            rows.append("%s\t%d\tN\t0\t*\t*%s" % (ctg, int(chunk.col_pos[c]), tail))
        else:
            rows.append("%s\t%d\tN\t%d\t%s\t%s%s" % (ctg, int(chunk.col_pos[c]), n, "".join(toks), "".join(bqs), tail))
    return "\n".join(rows) + "\n"


def likelihood_table(n_out, seed=7):
    """A synthetic likelihood_matrix_data file body (call_variants.py:655-735): n_out 10x10 matrices in (0,1)
    followed by 2*n_out rows of 10 increasing bin points (the last value of each row is dropped on load)."""
    rng = np.random.default_rng(seed)
    mats = rng.uniform(0.02, 0.98, size=(n_out * 10, 10))
    pts = np.sort(rng.uniform(0.02, 0.98, size=(2 * n_out, 10)), axis=1)
    pts[:, -1] = 1.0
    return np.vstack([mats, pts])


def lik_and_edges(table, n_out):
    """Split a loaded likelihood table into lik [K,10,10] and edges [2K,11] exactly as call_variants.py does."""
    table = np.asarray(table, dtype=np.float64)
    lik = table[: n_out * 10].reshape(n_out, 10, 10).copy()
    pts = table[n_out * 10: n_out * 12, :-1]
    edges = np.concatenate([np.zeros((2 * n_out, 1)), pts, np.ones((2 * n_out, 1))], axis=1)
    return lik, np.ascontiguousarray(edges)

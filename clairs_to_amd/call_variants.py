"""Posterior / genotype / quality: the GPU epilogue (cto_posterior) plus the host-side row assembly.

Counterpart of clairs/call_variants.py (reference):
  * likelihood table split                     call_variants.py:655-796  -> load_likelihood()
  * softmax + Bayes posterior + arg-max + QUAL call_variants.py:154-304, 79-88 -> posterior() on the device
  * alt allele / AF / GT / FILTER / INFO / row call_variants.py:135-150, 306-618 and shared/vcf.py:144-185
                                                                        -> vcf_row() on the host (strings)
"""
import ctypes as C

import numpy as np
import torch

from ._lib import lib, check, current_stream_ptr

ACGT = "ACGT"
IUPAC_TO_ACGT = dict(zip("ACGTURYSWKMBDHVN", "ACGTTACCAGACAAAA"))   # shared/utils.py:13-16


def load_likelihood(path_or_array, n_out):
    """-> (lik [K,10,10] float64, edges [2K,11] float64): 0 prepended, last point dropped, 1 appended."""
    t = np.loadtxt(path_or_array) if isinstance(path_or_array, (str, bytes)) else np.asarray(path_or_array, dtype=np.float64)
    if t.shape != (12 * n_out, 10):
        raise ValueError("likelihood table must be %d x 10, got %s" % (12 * n_out, t.shape))
    lik = np.ascontiguousarray(t[: 10 * n_out].reshape(n_out, 10, 10))
    pts = t[10 * n_out:, :-1]
    edges = np.concatenate([np.zeros((2 * n_out, 1)), pts, np.ones((2 * n_out, 1))], axis=1)
    return lik, np.ascontiguousarray(edges)


class Posterior:
    """Device-resident likelihood table + the fused epilogue launch."""

    def __init__(self, lik, edges, device="cuda"):
        self.K = int(lik.shape[0])
        self.lik = torch.from_numpy(np.ascontiguousarray(lik, dtype=np.float64)).to(device)
        self.edges = torch.from_numpy(np.ascontiguousarray(edges, dtype=np.float64)).to(device)

    def __call__(self, aff_logits, neg_logits, want_probs=True):
        """aff/neg logits float32 [K,B,2] on the device -> dict(probs [B,2K,2] f32, post [B,K] f64,
        decision [B,4] i32 (argmax, flags, posterior bits of a flagged site), qual [B] f64).  Host copies of decision / qual go
        through finalize_qual() before QUAL is read (cto_vcf_rows_batch does it itself)."""
        K, B = aff_logits.shape[0], aff_logits.shape[1]
        assert K == self.K and neg_logits.shape == aff_logits.shape
        dev = aff_logits.device
        aff_logits, neg_logits = aff_logits.contiguous(), neg_logits.contiguous()
        probs = torch.empty((B, 2 * K, 2), dtype=torch.float32, device=dev) if want_probs else None
        post = torch.empty((B, K), dtype=torch.float64, device=dev)
        dec = torch.empty((B, 4), dtype=torch.int32, device=dev)
        qual = torch.empty((B,), dtype=torch.float64, device=dev)
        check(lib.cto_posterior(aff_logits.data_ptr(), neg_logits.data_ptr(), K, B, self.lik.data_ptr(),
                                self.edges.data_ptr(), probs.data_ptr() if want_probs else None, post.data_ptr(),
                                dec.data_ptr(), qual.data_ptr(), current_stream_ptr()))
        return dict(probs=probs, post=post, decision=dec, qual=qual)


    def from_probs(self, p1):
        """p1 float64 [B,2K] on the device (8-decimal probabilities parsed from the predict text rows)."""
        B, K = p1.shape[0], self.K
        assert p1.shape[1] == 2 * K and p1.dtype == torch.float64
        p1 = p1.contiguous()
        post = torch.empty((B, K), dtype=torch.float64, device=p1.device)
        dec = torch.empty((B, 4), dtype=torch.int32, device=p1.device)
        qual = torch.empty((B,), dtype=torch.float64, device=p1.device)
        check(lib.cto_posterior_from_probs(p1.data_ptr(), K, B, self.lik.data_ptr(), self.edges.data_ptr(),
                                           post.data_ptr(), dec.data_ptr(), qual.data_ptr(), current_stream_ptr()))
        return dict(post=post, decision=dec, qual=qual)


def finalize_qual(decision, qual):
    """Host half of QUAL (call_variants.py:79-88 uses the host's math.log): decision int32 [n,4] and qual float64 [n], HOST numpy
    arrays as downloaded from the epilogue, are completed in place - the sites the device flagged as sitting on a 4-decimal
    rounding boundary (decision[:,1] bit 2, about two in a million) are re-evaluated with the host libm, flag and posterior
    bits cleared.  Returns the number of sites rewritten."""
    assert decision.dtype == np.int32 and decision.flags.c_contiguous and decision.flags.writeable and decision.shape[-1] == 4
    assert qual.dtype == np.float64 and qual.flags.c_contiguous and qual.flags.writeable and qual.size * 4 == decision.size
    return int(check(int(lib.cto_qual_finalize(decision.ctypes.data, qual.ctypes.data, int(qual.size)))))


def parse_alt_info(alt_info):
    parts = alt_info.rstrip().split("-")
    depth = int(parts[0])
    toks = parts[1].split(" ") if len(parts) > 1 else [""]
    d = dict(zip(toks[::2], [int(v) for v in toks[1::2]]))
    if depth == 0 and len(d) == 1:                       # all-indel column: depth falls back to the indel count
        (k, v), = d.items()
        if k[0] in "DI":
            depth = int(v)
    return d, depth


def vcf_rows_batch(chrom, pos, centre, alt_buf, alt_off, site_info, decision, qual, n_out, show_ref=False, qual_pass=0):
    """All VCF data rows of a chunk in ONE C call (cto_vcf_rows_batch, csrc/rows.cpp) - the same logic as vcf_row(), which is
    kept as the readable one-site form and as the cross-check.  pos int32 [n], centre bytes [n] (raw reference characters),
    alt_buf / alt_off as returned by featurize.alt_infos_packed, site_info int32 [n,12], decision int32 [n,4], qual float64 [n].
    -> (text with one '\\n'-terminated row per record, dict(rows, sites, low_coverage, clamped))."""
    n = len(pos)
    pos = np.ascontiguousarray(pos, dtype=np.int32)
    centre = np.ascontiguousarray(np.frombuffer(centre, dtype=np.uint8) if isinstance(centre, (bytes, bytearray)) else centre, dtype=np.uint8)
    site_info = np.ascontiguousarray(site_info, dtype=np.int32)
    decision = np.ascontiguousarray(decision, dtype=np.int32)
    qual = np.ascontiguousarray(qual, dtype=np.float64)
    alt_off = np.ascontiguousarray(alt_off, dtype=np.int64)
    assert len(centre) == n and site_info.shape == (n, 12) and decision.shape == (n, 4) and len(qual) == n and len(alt_off) == n + 1
    alt_keep = alt_buf if isinstance(alt_buf, (bytes, bytearray)) else bytes(alt_buf)
    alt_arr = np.frombuffer(alt_keep, dtype=np.uint8) if len(alt_keep) else np.zeros(1, dtype=np.uint8)
    cap = 512 * max(n, 1) + 2 * len(alt_keep) + 4096
    counts = np.zeros(4, dtype=np.int64)
    buf = C.create_string_buffer(cap)
    used = lib.cto_vcf_rows_batch(chrom.encode(), n, pos.ctypes.data, centre.ctypes.data, alt_arr.ctypes.data, alt_off.ctypes.data,
                                  site_info.ctypes.data, decision.ctypes.data, qual.ctypes.data, int(n_out), int(bool(show_ref)),
                                  -1.0 if qual_pass is None else float(qual_pass), C.addressof(buf), cap, counts.ctypes.data)
    check(int(used))
    return buf.raw[:used].decode(), dict(rows=int(counts[0]), sites=int(counts[1]), low_coverage=int(counts[2]), clamped=int(counts[3]))


def vcf_row(chrom, pos, ref_base, alt_info, fwd, rev, argmax, qual, n_out, show_ref=False, qual_pass=0,
            messages=None):
    """One VCF data row (without newline) or None when the reference writes nothing for the site.
    fwd/rev: the four strand counts (predict.py:626-642); argmax/qual: device epilogue outputs."""
    snv_mode = n_out == 4
    if qual != qual:      # NaN: 0/0 posterior of a site the reference raises IndexError on (decision flag bit 1); no row
        return None
    d, depth = parse_alt_info(alt_info)
    ref, alt = ref_base, ref_base
    if snv_mode:
        is_variant = ACGT[argmax] != ref_base
        is_reference = not is_variant
    else:
        is_variant = argmax >= 4
        is_reference = argmax < 4
    supported = None
    if is_variant:
        if depth <= 0:
            if messages is not None:
                messages.append("low tumor coverage")
            return None
        af = {k: c / float(depth) for k, c in d.items() if k[0] != "R" and c / float(depth) > 0}
        if not af:
            return None
        ranked = sorted(af.items(), key=lambda kv: kv[1], reverse=True)      # stable: first-seen wins ties
        best = ranked[0][0]
        supported = d[best]
        if best[0] == "X":
            alt = best[1]
            if snv_mode and ACGT[argmax] not in [k[1] for k, _ in ranked if k[0] == "X"]:
                is_variant, is_reference = False, True                      # the called base is not observed
        elif best[0] == "I":
            alt = best[1:] if best[1] != "#" else ref_base + best[2:]
        elif best[0] == "D":
            alt, ref = ref_base, ref_base + best[2:]
    if (not show_ref and is_reference) or (not is_reference and ref == alt):
        return None
    if snv_mode and (len(ref) > 1 or len(alt) > 1):
        return None
    if not snv_mode and len(ref) == 1 and len(alt) == 1 and not show_ref:
        return None
    ref_num = 0
    for k, c in d.items():
        if k[0] == "R":
            ref_num = int(c)
    if is_reference:
        supported, alt = ref_num, "."
    af_out = min((supported / depth) if depth != 0 else 0.0, 1.0)
    gt = "0/0" if is_reference else ("0/1" if af_out < 1.0 else "1/1")
    if is_reference:
        flt = "RefCall"
    elif qual_pass is None or qual >= float(qual_pass):
        flt = "PASS"
    else:
        flt = "LowQual"
    if not show_ref and gt in ("0/0", "./."):            # VcfWriter.write_row, shared/vcf.py:149
        return None
    f, r = [int(v) for v in fwd], [int(v) for v in rev]
    info = "FAU=%d;FCU=%d;FGU=%d;FTU=%d;RAU=%d;RCU=%d;RGU=%d;RTU=%d" % (f[0], f[1], f[2], f[3], r[0], r[1], r[2], r[3])
    ad = str(supported) if is_reference else "%d,%d" % (ref_num, supported)
    return "%s\t%d\t.\t%s\t%s\t%.4f\t%s\t%s\tGT:GQ:DP:AF:AD:AU:CU:GU:TU\t%s:%d:%d:%.4f:%s:%d:%d:%d:%d" % (
        chrom, int(pos), ref, alt, qual, flt, info, gt, int(float(qual)), depth, af_out, ad,
        f[0] + r[0], f[1] + r[1], f[2] + r[2], f[3] + r[3])


VCF_HEADER = """##fileformat=VCFv4.2
##source=clairs_to_amd
##FILTER=<ID=PASS,Description="All filters passed">
##FILTER=<ID=LowQual,Description="Low quality variant">
##FILTER=<ID=NonSomatic,Description="Tagged as non-somatic by a panel of normals (set downstream, kept by postprocess_vcf)">
##FILTER=<ID=RefCall,Description="Reference call">
##INFO=<ID=H,Number=0,Type=Flag,Description="Phaseable: variant seen on one haplotype only (set downstream, read by postprocess_vcf)">
##INFO=<ID=FAU,Number=1,Type=Integer,Description="Forward-strand A count in the tumor BAM">
##INFO=<ID=FCU,Number=1,Type=Integer,Description="Forward-strand C count in the tumor BAM">
##INFO=<ID=FGU,Number=1,Type=Integer,Description="Forward-strand G count in the tumor BAM">
##INFO=<ID=FTU,Number=1,Type=Integer,Description="Forward-strand T count in the tumor BAM">
##INFO=<ID=RAU,Number=1,Type=Integer,Description="Reverse-strand A count in the tumor BAM">
##INFO=<ID=RCU,Number=1,Type=Integer,Description="Reverse-strand C count in the tumor BAM">
##INFO=<ID=RGU,Number=1,Type=Integer,Description="Reverse-strand G count in the tumor BAM">
##INFO=<ID=RTU,Number=1,Type=Integer,Description="Reverse-strand T count in the tumor BAM">
##FORMAT=<ID=GT,Number=1,Type=String,Description="Genotype">
##FORMAT=<ID=GQ,Number=1,Type=Integer,Description="Genotype quality">
##FORMAT=<ID=DP,Number=1,Type=Integer,Description="Read depth">
##FORMAT=<ID=AF,Number=1,Type=Float,Description="Estimated allele frequency">
##FORMAT=<ID=AD,Number=R,Type=Integer,Description="Allelic depths (ref, alt)">
##FORMAT=<ID=AU,Number=1,Type=Integer,Description="A count in the tumor BAM">
##FORMAT=<ID=CU,Number=1,Type=Integer,Description="C count in the tumor BAM">
##FORMAT=<ID=GU,Number=1,Type=Integer,Description="G count in the tumor BAM">
##FORMAT=<ID=TU,Number=1,Type=Integer,Description="Count of T in the tumor BAM">
"""
# The TU line is byte-identical to the reference's (shared/vcf.py): its postprocess_vcf cuts the header after exactly this
# line (src/postprocess_vcf.py:165-166), so a VCF written here stays a valid input of the reference's own tail.


def call_variants_from_probability(args, device="cuda"):
    """`clairs_to.py call_variants` counterpart: probability text rows in, VCF out (call_variants.py:620-867)."""
    import ast
    import gzip
    import os
    import sys
    if not torch.cuda.is_available():
        sys.exit("[ERROR] clairs_to_amd call_variants needs a HIP device; there is no CPU fallback")
    K = 4 if args.disable_indel_calling else 6
    lik, edges = load_likelihood(args.likelihood_matrix_data, K)
    post = Posterior(lik, edges, device)
    with open(args.predict_fn, "rb") as f:
        gz = f.read(2) == b"\x1f\x8b"
    rows = []
    with (gzip.open(args.predict_fn, "rt") if gz else open(args.predict_fn)) as f:
        for row in f:
            c = row.rstrip().split("\t")
            if len(c) >= 6 + 2 * K:
                rows.append(c)
    n_rows = 0
    os.makedirs(os.path.dirname(os.path.abspath(args.call_fn)), exist_ok=True)
    with open(args.call_fn, "w") as out:
        out.write(VCF_HEADER + "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t%s\n" % args.sample_name)
        if rows:
            p1 = np.array([[float(f.split()[1]) for f in r[6:6 + 2 * K]] for r in rows], dtype=np.float64)
            o = post.from_probs(torch.from_numpy(p1).to(device))
            dec, qual = o["decision"].cpu().numpy(), o["qual"].cpu().numpy()
            finalize_qual(dec, qual)
            for i, r in enumerate(rows):
                if dec[i, 1]:
                    print("[WARNING] %s:%s probability 1.00000000 falls outside the likelihood bins (the reference "
                          "raises IndexError here); clamped" % (r[0], r[1]), file=sys.stderr)
                line = vcf_row(r[0], r[1], r[2], r[3], ast.literal_eval(r[4]), ast.literal_eval(r[5]), int(dec[i, 0]),
                               float(qual[i]), K, show_ref=args.show_ref, qual_pass=args.qual)
                if line is not None:
                    out.write(line + "\n")
                    n_rows += 1
    if n_rows == 0:
        os.remove(args.call_fn)       # the reference removes VCFs without records (call_variants.py:859-867)
    return n_rows


def main():
    from argparse import ArgumentParser
    p = ArgumentParser(description="Call variants from probability rows (GPU posterior)")
    p.add_argument("--platform", type=str, default="ont")
    p.add_argument("--call_fn", type=str, required=True)
    p.add_argument("--predict_fn", type=str, required=True)
    p.add_argument("--likelihood_matrix_data", type=str, required=True)
    p.add_argument("--ctg_name", type=str, default=None)
    p.add_argument("--sample_name", type=str, default="SAMPLE")
    p.add_argument("--qual", type=int, default=0)
    p.add_argument("--show_ref", action="store_true")
    p.add_argument("--disable_indel_calling", type=lambda v: str(v).lower() in ("yes", "true", "t", "y", "1"), default=False)
    p.add_argument("--pileup", action="store_true")
    call_variants_from_probability(p.parse_args())


if __name__ == "__main__":
    main()

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
        decision [B,4] i32 (argmax, clamped, -, -), qual [B] f64)."""
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


def vcf_row(chrom, pos, ref_base, alt_info, fwd, rev, argmax, qual, n_out, show_ref=False, qual_pass=0,
            messages=None):
    """One VCF data row (without newline) or None when the reference writes nothing for the site.
    fwd/rev: the four strand counts (predict.py:626-642); argmax/qual: device epilogue outputs."""
    snv_mode = n_out == 4
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

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


# The meta-information lines of every VCF of a run, byte for byte the reference's (shared/vcf.py:14-55: `vcf_header`): its own tools read
# them back - postprocess_vcf cuts the header after the `##FORMAT=<ID=TU` line (src/postprocess_vcf.py:165-166) - and a file written here
# must be indistinguishable from one its call_variants wrote.  Kept as tables (ID, Number, Type, Description) and rendered once.
CALLER_NAME, CALLER_VERSION = "clairs_to", "0.4.4"            # shared/param.py caller_name, version: the `##clairs_to_version=` line
_FILTERS = (
    ("PASS", "All filters passed"), ("NonSomatic", "Non-somatic variant tagged by panel of normals"), ("LowQual", "Low-quality variant"),
    ("LowAltBQ", "Average alt allele base quality <20"), ("LowAltMQ", "Average alt allele read mapping quality <20"),
    ("ReadStartEnd", ">30% of the supporting alt alleles are within 100bp of the start or end of a read"),
    ("VariantCluster", "Three or more variants clustered within 200bp"), ("NoAncestry", "Variant without an ancestral haplotype support"),
    ("MultiHap", "Alt alleles existed in multiple haplotypes"), ("StrandBias", "Strand bias p-value <0.001"),
    ("LowSeqEntropy", "Sequence entropy <0.9"),
    ("Realignment", "For short-read, both the count of supporting alt alleles and AF decreased after realignment"), ("RefCall", "Reference call"))
_INFOS = tuple(("Verdict_" + k, "0", "Flag", "Variant tagged by verdict as " + v)
               for k, v in (("Germline", "Germline"), ("Somatic", "Somatic"), ("SubclonalSomatic", "Subclonal Somatic"))) + \
    (("H", "0", "Flag", "Variant found only in one haplotype in the phased reads"),) + \
    tuple((s[0].upper() + b + "U", "1", "Integer", "Count of %s in %s strand in the tumor BAM" % (b, s)) for s in ("forward", "reverse") for b in "ACGT") + \
    (("SB", "1", "Float", "The p-value of Fisher\u2019s exact test on strand bias"),)
_FORMATS = (("GT", "1", "String", "Genotype"), ("GQ", "1", "Integer", "Genotype quality"), ("DP", "1", "Integer", "Read depth"),
            ("AF", "1", "Float", "Estimated allele frequency"),
            ("AD", "R", "Integer", "Allelic depths for the ref and alt alleles in the order listed in the ALT column")) + \
    tuple((b + "U", "1", "Integer", "Count of %s in the tumor BAM" % b) for b in "ACGT")
VCF_HEADER = "##fileformat=VCFv4.2\n##source=ClairS-TO\n##%s_version=%s\n" % (CALLER_NAME, CALLER_VERSION) + \
    "".join('##FILTER=<ID=%s,Description="%s">\n' % f for f in _FILTERS) + \
    "".join('##INFO=<ID=%s,Number=%s,Type=%s,Description="%s">\n' % i for i in _INFOS) + \
    "".join('##FORMAT=<ID=%s,Number=%s,Type=%s,Description="%s">\n' % f for f in _FORMATS)


def vcf_header(ref_fn=None, ctg_name=None, sample_name="SAMPLE", cmdline=None):
    """VcfWriter.write_header (shared/vcf.py:100-121): the meta lines, `##cmdline=` as the fourth line when given, a `##contig` line per
    row of <ref_fn>.fai (all of them, or those named in the comma-separated ctg_name), the column line."""
    import os
    out = VCF_HEADER
    if cmdline is not None and cmdline != "":
        rows = out.rstrip("\n").split("\n")
        rows.insert(3 if len(rows) >= 3 else len(rows) - 1, "##cmdline={}".format(cmdline))
        out = "\n".join(rows) + "\n"
    if ref_fn is not None:
        fai = ref_fn + ".fai"
        if not os.path.exists(fai):                       # file_path_from(..., sep='.'): ref.fa -> ref.fai
            fai = ".".join(ref_fn.split(".")[:-1]) + ".fai"
        if not os.path.exists(fai):
            raise SystemExit("[ERROR] file %s not found" % (ref_fn + ".fai"))
        names = None if ctg_name is None else (ctg_name.split(",") if "," in ctg_name else [ctg_name])
        with open(fai) as f:
            for row in f:
                c = row.strip().split("\t")
                if names is not None and c[0] not in names:
                    continue
                out += "##contig=<ID=%s,length=%s>\n" % (c[0], c[1])
    return out + "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t%s\n" % sample_name


def chunk_vcf_header(ref_fn, K, sample_name="SAMPLE"):
    """The header of a p_<chunk>.vcf as a run of run_clairs_to leaves it: its SNV call_variants command carries --ref_fn (the ##contig lines
    of the whole .fai, run_clairs_to:1300), its indel one does not (:1631-1645); neither carries --ctg_name."""
    return vcf_header(ref_fn if K == 4 else None, None, sample_name)


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
        out.write(vcf_header(getattr(args, "ref_fn", None), args.ctg_name, args.sample_name))
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


def build_parser():
    from argparse import ArgumentParser
    from ._cli import add_ignored, str2bool
    p = ArgumentParser(description="Call variants from probability rows (GPU posterior)")
    p.add_argument("--platform", type=str, default="ont")
    p.add_argument("--call_fn", type=str, required=True)
    p.add_argument("--predict_fn", type=str, required=True)
    p.add_argument("--likelihood_matrix_data", type=str, required=True)
    p.add_argument("--ref_fn", type=str, default=None, help="with its .fai: the ##contig lines of the header (shared/vcf.py:109-117)")
    p.add_argument("--ctg_name", type=str, default=None)
    p.add_argument("--sample_name", type=str, default="SAMPLE")
    p.add_argument("--qual", type=int, default=0)
    p.add_argument("--show_ref", action="store_true")
    p.add_argument("--disable_indel_calling", type=str2bool, default=False)
    p.add_argument("--pileup", action="store_true")
    add_ignored(p, samtools="str")                  # clairs/call_variants.py:890: declared, never read
    return p


def main(argv=None):
    call_variants_from_probability(build_parser().parse_args(argv))


if __name__ == "__main__":
    main()

"""Drop-in counterpart of `clairs_to.py predict --pileup` (reference: clairs/predict.py) with both networks and the
softmax on the GPU.  Same inputs (the two gzip tensor text files, two pickled checkpoints) and the same
probability text rows out (predict.py:114-152): ctg, pos, ref, alt_info, fwd counts, rev counts, K AFF "p0 p1"
fields, K NEG fields, and the trailing empty field.

The `--call_fn` branch of the reference is broken and is not reproduced."""
import gzip
import sys
from argparse import ArgumentParser

import numpy as np
import torch

from ._cli import add_ignored, add_unsupported, check_unsupported, str2bool
from ._lib import lib, check, current_stream_ptr
from .call_variants import IUPAC_TO_ACGT
from . import nn_shims

NPOS, NCHAN, FLANK = 33, 34, 16
PREDICT_BATCH = 250            # shared/param.py:85 (the GPU path takes whatever batch it is given)


def read_tensor_file(fn, min_rescale_cov):
    """tensor_generator_from (predict.py:155-242) without batching: rows whose centre reference base is not ACGT are
    dropped; the rescale is value * (min_rescale_cov / depth) in double, cast to float32, when depth > min_rescale_cov."""
    opener = gzip.open if fn.endswith(".gz") or _is_gzip(fn) else open
    meta, xs, raws = [], [], []
    with opener(fn, "rt") as f:
        for row in f:
            c = row.split("\t")[:7]
            contig, coord, seq, tensor, alt_info = c[0], c[1], c[2], c[3], c[4]
            if seq[FLANK] not in "ACGT":
                continue
            v = np.array(tensor.split(), dtype=np.float64)
            depth = float(alt_info.split("-")[0])
            raws.append(v.astype(np.float32))
            if min_rescale_cov is not None and depth > min_rescale_cov:
                v = v * (float(min_rescale_cov) / depth)
            xs.append(v.astype(np.float32))
            meta.append((contig, coord, seq, alt_info))
    X = np.stack(xs).reshape(-1, NPOS, NCHAN) if xs else np.zeros((0, NPOS, NCHAN), np.float32)
    R = np.stack(raws).reshape(-1, NPOS, NCHAN) if raws else np.zeros((0, NPOS, NCHAN), np.float32)
    return meta, X, R


def _is_gzip(fn):
    with open(fn, "rb") as f:
        return f.read(2) == b"\x1f\x8b"


def strand_counts(raw):
    """predict.py:626-642 on the un-rescaled AFF tensor; returns two float lists per site (as .tolist() yields)."""
    out = []
    for o in (0, 9):
        c = raw[:, FLANK, o:o + 4].copy()
        neg = c < 0
        sums = c.sum(axis=1, keepdims=True)
        c = np.where(neg, -sums, c)
        c = np.where(c == 0, 0.0, c)       # -0.0 -> 0.0
        out.append(c.tolist())
    return out


SPLIT_HELP = ("EXPERIMENTAL, not in the reference: run the networks' GEMMs on split 16-bit operands (hi + lo, three f16 / bf16 MFMA "
              "passes per product, fp32 accumulation) - about twice the network throughput, probabilities within ~1e-5 (f16) of the "
              "default fp32 kernels'; the default computes in fp32 like the reference")


def load_models(args, device):
    nn_shims.install_reference_aliases()      # reference pickles name clairs.model.<cls>
    aff = torch.load(args.chkpnt_fn_acgt, map_location="cpu", weights_only=False)["model_acgt"]
    neg = torch.load(args.chkpnt_fn_nacgt, map_location="cpu", weights_only=False)["model_nacgt"]
    if not args.disable_indel_calling:
        # predict.py:520-568 rebuilds the *_Indel classes and loads the pickles' state_dicts
        aff = nn_shims.from_state_dict("CvT_Indel", aff.state_dict())
        neg = nn_shims.from_state_dict("BiGRU_NACGT_Indel", neg.state_dict())
    for m in (aff, neg):
        if not isinstance(m, nn_shims._HipNet):
            raise TypeError("checkpoint does not hold a clairs.model network (got %s)" % type(m).__name__)
        m.split_operands = getattr(args, "split_operands", None)      # experimental, off by default
    return aff.eval(), neg.eval()


def predict(args, device="cuda"):
    if not torch.cuda.is_available():
        sys.exit("[ERROR] clairs_to_amd predict needs a HIP device; there is no CPU fallback")
    aff, neg = load_models(args, device)
    meta, xa, raw = read_tensor_file(args.tensor_fn_acgt, args.min_rescale_cov)
    meta_n, xn, _ = read_tensor_file(args.tensor_fn_nacgt, args.min_rescale_cov)
    if len(meta) != len(meta_n):
        sys.exit("Inconsistent number of AFF (%d) and NEG (%d) tensors" % (len(meta), len(meta_n)))
    B, K = len(meta), len(aff._heads_out)
    fwd, rev = strand_counts(raw) if B else ([], [])
    la = aff.logits(torch.from_numpy(xa).to(device))
    ln = neg.logits(torch.from_numpy(xn).to(device))
    probs = torch.empty((B, 2 * K, 2), dtype=torch.float32, device=device)
    if B:
        check(lib.cto_softmax_probs(la.data_ptr(), ln.data_ptr(), K, B, probs.data_ptr(), current_stream_ptr()))
    probs = probs.cpu().numpy()
    out = gzip.open(args.predict_fn, "wt") if args.predict_fn != "PIPE" else sys.stdout
    for i, (contig, coord, seq, alt_info) in enumerate(meta):
        ref_base = IUPAC_TO_ACGT[seq[FLANK].upper()]
        fields = [contig, coord, ref_base, alt_info, str(fwd[i]), str(rev[i])]
        fields += [" ".join("{:0.8f}".format(x) for x in probs[i, k]) for k in range(2 * K)]
        # the SNV format string carries a trailing empty field; the indel one has one placeholder too few for it
        # (predict.py:114-152), so indel rows end right after the last probability
        out.write("\t".join(fields) + ("\t\n" if K == 4 else "\n"))
    if out is not sys.stdout:
        out.close()
    print("[INFO] {} total processed positions: {}".format(args.ctg_name, B), file=sys.stderr)
    return B


def build_parser():
    p = ArgumentParser(description="Candidate variants probability prediction (HIP networks)")
    p.add_argument("--platform", type=str, default="ont")
    p.add_argument("--tensor_fn_acgt", type=str, required=True)
    p.add_argument("--tensor_fn_nacgt", type=str, required=True)
    p.add_argument("--chkpnt_fn_acgt", type=str, required=True)
    p.add_argument("--chkpnt_fn_nacgt", type=str, required=True)
    p.add_argument("--predict_fn", type=str, default="PIPE")
    p.add_argument("--ctg_name", type=str, default=None)
    p.add_argument("--min_rescale_cov", type=int, default=50)
    p.add_argument("--disable_indel_calling", type=str2bool, default=False)
    p.add_argument("--use_gpu", type=str2bool, default=True, help="accepted as the reference accepts it: the networks always run on the HIP device")
    p.add_argument("--pileup", action="store_true")
    p.add_argument("--split_operands", type=str, default=None, choices=["f16", "bf16"], help=SPLIT_HELP)
    # clairs/predict.py:736-812.  --show_ref / --qual / --sample_name / --ref_fn / --samtools only configure the VCF writer of its
    # --call_fn mode (:454-499); run_clairs_to passes --show_ref with --print_ref_calls all the same (:1287).  The one-invocation form
    # of that mode here is `pileup_call`.
    add_ignored(p, show_ref="flag", qual="int", sample_name="str", ref_fn="str", samtools="str")
    add_unsupported(p, call_fn=("str", None), is_from_tables=("bool", False), flanking=("int", None))
    return p


def main(argv=None):
    p = build_parser()
    a = p.parse_args(argv)
    check_unsupported(p, a)
    predict(a)


if __name__ == "__main__":
    main()

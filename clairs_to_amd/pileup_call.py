"""One invocation per candidate chunk: BED chunk + BAM (or mpileup text) + checkpoints + likelihood table -> `p_<chunk>.vcf`.

Replaces the four commands the reference orchestrator runs per chunk (run_clairs_to:1228-1308 / 1562-1647:
`create_tensor_pileup_calling --min_bq <platform>`, the same with `--min_bq 0`, `predict --pileup`, `call_variants`) with a
single pass that never leaves HBM between the pileup pack and the decision: no tensor text, no probability text.
The options are the union of those four commands' options under their reference names; `--predict_fn` / `--tensor_can_fn*`
remain available as debugging taps and produce exactly what the separate mirrors write.

The result is the same VCF the chained mirrors (and the reference) produce: the epilogue kernel rounds probabilities to
the 8 decimals of the text seam before the table look-up (csrc/posterior.hip), strand counts and depths come from the AFF
pass (predict.py:614, 689), rows whose centre reference base is not ACGT are dropped (predict.py:228), sites without an
mpileup row at the candidate are skipped (create_tensor_pileup_calling.py:552).
"""
import gzip
import os
import sys
from argparse import ArgumentParser

import numpy as np
import torch

from .call_variants import IUPAC_TO_ACGT, VCF_HEADER, chunk_vcf_header, load_likelihood, vcf_rows_batch
from .create_tensor_pileup_calling import EXPAND_REF, MAX_INDEL, load_pack, read_candidate_positions
from .engine import Engine
from .fasta import read_region
from .featurize import alt_infos_from_host
from .predict import load_models, str2bool, SPLIT_HELP
from .platforms import resolve_platform


def make_engine(args, device="cuda"):
    """Checkpoints + likelihood table -> Engine (done once per process; chunks re-use it)."""
    if not torch.cuda.is_available():
        sys.exit("[ERROR] clairs_to_amd pileup_call needs a HIP device; there is no CPU fallback")
    K = 4 if args.disable_indel_calling else 6
    _, family, default_bq = resolve_platform(args.platform)          # exits on a platform the reference does not know
    min_bq = args.min_bq if args.min_bq is not None else default_bq
    aff, neg = load_models(args, device)
    lik, edges = load_likelihood(args.likelihood_matrix_data, K)
    # ilmn: the reference's NEG tensors are a symlink to the AFF tensors (run_clairs_to:1248-1252), whatever --min_bq is
    return Engine(aff, neg, lik, edges, min_bq=min_bq, min_rescale_cov=args.min_rescale_cov, device=device,
                  neg_reads_aff=(family == "ilmn"))


def prepare_chunk(args, device=None, copy_stream=None):
    """Host-only half of a chunk: candidates, reference slice and the column pack (BAM decoding / mpileup tokenising; the C
    calls release the GIL, so `call_chunks` runs several of these on a thread pool while earlier chunks are on the GPU).
    With `device` the pack is also uploaded from this (producer) thread on `copy_stream`, and `uploaded` is the event the
    compute stream has to wait for - the PCIe transfer then never blocks the thread that launches kernels.  None if empty."""
    sites, ctg_start, ctg_end = read_candidate_positions(args.candidates_bed_regions, args.ctg_name)     # sorted, unique, int32
    if len(sites) == 0:
        return None
    ref_start = max(1, ctg_start - EXPAND_REF)
    ref = read_region(args.ref_fn, args.ctg_name, ref_start, ctg_end + EXPAND_REF, as_bytes=True)
    if not ref:
        sys.exit("[ERROR] Failed to load reference sequence from file ({}).".format(args.ref_fn))
    max_indel = MAX_INDEL if args.max_indel_length is None else args.max_indel_length
    pack = load_pack(args, ref, ref_start, ctg_start, ctg_end, max_indel, device=device, stream=copy_stream)
    prep = dict(sites=sites, ref=ref, ref_start=ref_start, pack=pack)
    if device is not None:
        with torch.cuda.device(device), torch.cuda.stream(copy_stream if copy_stream is not None else torch.cuda.current_stream(device)):
            prep["dev_pack"] = pack.to_device(device)
            prep["dev_sites"] = torch.from_numpy(sites).to(device)
            prep["uploaded"] = torch.cuda.Event()
            prep["uploaded"].record()
    return prep


_HOST_KEYS = ("site_info", "colvec", "sitefirst", "keycnt", "keyfirst", "decision", "qual")


def launch_chunk(eng, prep, want_probs=False, pinned=None):
    """Device half of a chunk, asynchronous: waits for the pack upload (if a producer thread did it), runs the hot path on the
    current stream and queues the device-to-host copies of everything the row formatter needs into page-locked buffers
    (`pinned`: a dict re-used from an earlier chunk, grown when too small).  Returns the host views + the event that marks
    them valid; nothing here blocks on the GPU."""
    device = eng.device
    with torch.cuda.device(device):
        main = torch.cuda.current_stream()
        if "dev_pack" in prep:
            main.wait_event(prep["uploaded"])
            dp, sp = prep["dev_pack"], prep["dev_sites"]
            for t in list(dp.t.values()) + [sp]:
                t.record_stream(main)
        else:
            dp = prep["pack"].to_device(device)
            sp = torch.from_numpy(np.ascontiguousarray(prep["sites"], dtype=np.int32)).to(device)
        res = eng.run_device(dp, sp)
        feat = res["features"]
        # only the candidate columns' vectors are needed on the host (alt_info strings): 144 B per site cross PCIe instead of 144 B
        # per pack column (19 MB per 4096-site chunk) - written by the one-kernel featurisation, gathered here after the two-stage one
        if feat.site_colvec is not None:
            site_colvec = feat.site_colvec
        else:
            centre = feat.site_info[:, 0].clamp(min=0).long()
            site_colvec = feat.colvec.view(-1, feat.colvec.shape[-1]).index_select(0, centre) if feat.colvec.numel() else feat.colvec
        src = dict(site_info=feat.site_info, colvec=site_colvec, sitefirst=feat.sitefirst, keycnt=feat.keycnt, keyfirst=feat.keyfirst,
                   decision=res["decision"], qual=res["qual"])
        if want_probs:
            src["probs"] = res["probs"]
        pinned = {} if pinned is None else pinned
        host = {}
        for k, t in src.items():
            buf = pinned.get(k)
            if buf is None or buf.dtype != t.dtype or buf.numel() < t.numel():
                buf = torch.empty(max(t.numel(), 1) * 5 // 4, dtype=t.dtype, pin_memory=True)
                pinned[k] = buf
            h = buf[: t.numel()].view(t.shape)
            h.copy_(t, non_blocking=True)
            host[k] = h
        done = torch.cuda.Event()
        done.record(main)
    return dict(host=host, done=done, pinned=pinned, keep=(res, dp, sp))


def finish_chunk(args, K, prep, launched):
    """Host tail of a chunk: waits for the copies, builds the alt_info strings and every VCF record in two C calls
    (cto_alt_info_batch, cto_vcf_rows_batch) and writes `args.call_fn` (+ the probability rows when --predict_fn is given)."""
    launched["done"].synchronize()
    h = {k: v.numpy() for k, v in launched["host"].items()}
    sites, ref, ref_start, pack = prep["sites"], prep["ref"], prep["ref_start"], prep["pack"]
    info = h["site_info"].copy()
    alt_buf, alt_off = alt_infos_from_host(pack, info, h["colvec"], h["sitefirst"], h["keycnt"], h["keyfirst"], per_site=True)
    dec, qual = h["decision"], h["qual"]
    sites_arr = np.asarray(sites, dtype=np.int64)
    centre = np.frombuffer(ref if isinstance(ref, bytes) else ref.encode("latin-1"), dtype=np.uint8)[sites_arr - ref_start]
    info[~np.isin(centre, np.frombuffer(b"ACGT", dtype=np.uint8)), 3] |= 1      # predict.py:219-228: centre not in ACGT -> no row
    # every record of the chunk in one C call (cto_vcf_rows_batch): the per-site Python formatting was a third of a chunk
    text, cnt = vcf_rows_batch(args.ctg_name, sites_arr, centre, alt_buf, alt_off, info, dec, qual, K, show_ref=args.show_ref,
                               qual_pass=args.qual)
    n_rows, n_sites = cnt["rows"], cnt["sites"]
    sink = getattr(args, "site_sink", None)
    if sink is not None:       # call_chunks --gather_outputs: this chunk's per-site outputs, for the exchange step (dist.gather_site_rows)
        alt_len = np.diff(np.asarray(alt_off, dtype=np.int64)).astype(np.int32)
        sink(dict(pos=sites_arr.copy(), centre=np.ascontiguousarray(centre), info=info, decision=dec.copy(), qual=qual.copy(),
                  probs=h["probs"].copy(), alt_buf=bytes(alt_buf), alt_len=alt_len))
    for _ in range(cnt["low_coverage"]):
        print("low tumor coverage")                                  # call_variants.py:328, one line per such site
    if cnt["clamped"]:
        for i in np.nonzero(dec[:, 1] & 3)[0]:
            print("[WARNING] %s:%d a probability printed as 1.00000000 / 0.00000000 falls outside the likelihood bins (the "
                  "reference raises IndexError here); %s" % (args.ctg_name, sites[i], "no posterior, site skipped" if dec[i, 1] & 2
                                                             else "bin clamped"), file=sys.stderr)
    os.makedirs(os.path.dirname(os.path.abspath(args.call_fn)), exist_ok=True)
    if n_rows:             # the reference removes VCFs without records (call_variants.py:859-867)
        with open(args.call_fn, "w") as out:
            out.write(chunk_vcf_header(args.ref_fn, K, args.sample_name))
            out.write(text)
    elif os.path.exists(args.call_fn):
        os.remove(args.call_fn)
    if getattr(args, "predict_fn", None):        # debugging tap: the probability rows of the predict mirror
        probs = h["probs"]
        with gzip.open(args.predict_fn, "wt") as pred:
            for i, pos in enumerate(sites):
                if info[i, 3] & 1:
                    continue
                c = chr(centre[i])
                fwd, rev = [float(v) for v in info[i, 4:8]], [float(v) for v in info[i, 8:12]]
                fields = [args.ctg_name, str(pos), IUPAC_TO_ACGT[c], alt_buf[alt_off[i]:alt_off[i + 1]].decode(), str(fwd), str(rev)]
                fields += [" ".join("{:0.8f}".format(x) for x in probs[i, k]) for k in range(2 * K)]
                pred.write("\t".join(fields) + ("\t\n" if K == 4 else "\n"))
    print("[INFO] {} total processed positions: {}".format(args.ctg_name, n_sites), file=sys.stderr)
    return n_rows


def pileup_call(args, device="cuda", engine=None, prepared=None):
    """One chunk, start to finish (the three stages above back to back)."""
    eng = engine if engine is not None else make_engine(args, device)
    prep = prepared if prepared is not None else prepare_chunk(args)
    if prep is None:
        print("[INFO] {} total processed positions: 0".format(args.ctg_name), file=sys.stderr)
        return 0
    launched = launch_chunk(eng, prep, want_probs=bool(getattr(args, "predict_fn", None)))
    return finish_chunk(args, eng.K, prep, launched)


def add_common_arguments(p):
    p.add_argument("--platform", type=str, default="ont")
    p.add_argument("--tumor_bam_fn", type=str, default=None)
    p.add_argument("--ref_fn", type=str, required=True)
    p.add_argument("--samtools", type=str, default="samtools")
    p.add_argument("--bam_reader", type=str, default="samtools", choices=["samtools", "native", "gpu"],
                   help="'native': built-in BAM + BAI reader instead of a samtools subprocess (parity unpinned, see csrc/bam.cpp)")
    p.add_argument("--min_bq", type=int, default=None, help="AFF-pass base quality gate (default: the platform's)")
    p.add_argument("--max_depth", type=int, default=None,
                   help="EXPERIMENTAL: maximum tumor depth handed to samtools mpileup / the built-in reader (the reference's default: 8000). Engine limits, reported as an error of the chunk (CTO_EUNSUPPORTED), not silently: a pileup column may hold at most 32767 read-bases and 2048 distinct indel alleles")
    p.add_argument("--max_indel_length", type=int, default=None)
    p.add_argument("--chkpnt_fn_acgt", type=str, required=True)
    p.add_argument("--chkpnt_fn_nacgt", type=str, required=True)
    p.add_argument("--min_rescale_cov", type=int, default=50)
    p.add_argument("--disable_indel_calling", type=str2bool, default=False)
    p.add_argument("--split_operands", type=str, default=None, choices=["f16", "bf16"], help=SPLIT_HELP)
    p.add_argument("--likelihood_matrix_data", type=str, required=True)
    p.add_argument("--sample_name", type=str, default="SAMPLE")
    p.add_argument("--show_ref", action="store_true")
    p.add_argument("--qual", type=int, default=0)
    p.add_argument("--pileup", action="store_true")


def main(argv=None):
    p = ArgumentParser(description="Pileup calling of one candidate chunk on the GPU: BED + BAM/mpileup -> VCF")
    add_common_arguments(p)
    p.add_argument("--mpileup_fn", type=str, default=None, help="samtools mpileup text (--min-BQ 0) instead of a BAM")
    p.add_argument("--ctg_name", type=str, required=True)
    p.add_argument("--candidates_bed_regions", type=str, required=True)
    p.add_argument("--call_fn", type=str, required=True)
    p.add_argument("--predict_fn", type=str, default=None, help="also write the probability rows (debugging tap)")
    pileup_call(p.parse_args(argv))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Generate the golden fixtures by running the REFERENCE ITSELF (HKU-BAL/ClairS-TO v0.4.4).

Runs only in the build container, where /root/reference exists; the GPU box and the tests never read the
reference - they read the files this script writes next to itself:

  columns.json.gz      decode_pileup_bases on hand-made + random mpileup columns  
  region.json.gz       create_tensor_pileup_calling end to end through a fake `samtools` shim, for the AFF
                       (--min_bq 20) and NEG (--min_bq 0) passes: inputs (mpileup text, reference, sites) and
                       the tensor text rows it wrote                                         (F9-F11)
  models_<cls>.npz     logits of the four clairs.model classes on count tensors, weights from
                       weights_recipe.py (only the (name, shape) manifest is stored)         (M1-M9)
  extract.json.gz      extract_candidates_calling through the shim: pileup text in, SNV / indel candidate positions out
  calls_<mode>.json.gz clairs_to.py predict (--predict_fn) then call_variants on those tensors: probability rows,
                       likelihood table, VCF rows                                            (H1-H6, Q1-Q6)
  calls_branches.json.gz  hand-made probability rows through the reference's call_variants: every ALT / AF / GT / FILTER
                       branch of output_vcf_from_probability (Q4, Q6), with and without --show_ref / --qual 20
  region2k.json.gz     2 000 candidates through the reference's four commands; inputs regenerated from a seed and
                       checked by SHA-256, outputs stored (tensor text as SHA-256 + per-row CRC-32)
  pickles.json.gz      genuine torch.save'd clairs.model objects (parameter values zeroed), for the checkpoint-loading seam

Usage: python tests/golden/gen_golden.py     (from the repo root)
"""
import gzip
import io
import json
import os
import stat
import subprocess
import sys
import tempfile
from argparse import Namespace

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, REF)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import torch  # noqa: E402

import src.create_tensor_pileup_calling as ct  # noqa: E402  (the reference)
import clairs.model as rm  # noqa: E402                  (the reference)
from clairs_to_amd.synth import SynthChunk, mpileup_text, likelihood_table  # noqa: E402
from weights_recipe import make_weights, CVT_CFG  # noqa: E402


def dump_json_gz(name, obj):
    raw = json.dumps(obj, separators=(",", ":")).encode()
    with open(os.path.join(HERE, name), "wb") as f:
        with gzip.GzipFile(fileobj=f, mode="wb", mtime=0) as g:
            g.write(raw)
    print("wrote", name, len(raw), "bytes raw")


# ------------------------------------------------------------------------------------------ columns
def ref_decode(bases, bq, mq, ref_base, chunk_ref, cand):
    args = Namespace(max_indel_length=60)
    pos = 1000
    tensor, base_list, _, _, _, alt_info = ct.decode_pileup_bases(
        args=args, pos=pos, pileup_bases=bases, reference_base=ref_base, minimum_snp_af_for_candidate=0,
        minimum_indel_af_for_candidate=0, has_pileup_candidates=1,
        candidates_type_dict={pos: "unknown"} if cand else {}, is_tumor=True,
        mapping_quality=[ord(c) - 33 for c in mq], base_quality=[ord(c) - 33 for c in bq],
        phasing_info=None, chunk_ref_seq=chunk_ref)
    return tensor, alt_info


def random_column(rng, depth, p_indel=0.15, p_long=0.03):
    toks, bq, mq = [], [], []
    for _ in range(depth):
        rev = rng.random() < 0.5
        r = rng.random()
        if r < 0.08:
            b = "#" if rev else "*"
        elif r < 0.12:
            b = "n" if rev else "N"
        else:
            b = "ACGT"[rng.integers(0, 4)] if rng.random() < 0.3 else "A"
            b = b.lower() if rev else b
        t = b
        if rng.random() < 0.05:
            t = "^" + chr(33 + int(rng.integers(0, 60))) + t
        if rng.random() < p_indel:
            if rng.random() < p_long:
                ln = int(rng.integers(58, 63))
            else:
                ln = int(min(rng.geometric(0.5), 8))
            if rng.random() < 0.5:
                seq = "".join("ACGT"[rng.integers(0, 2)] for _ in range(ln))
                t += "+%d%s" % (ln, seq.lower() if rev else seq)
            else:
                t += "-%d%s" % (ln, ("n" if rev else "N") * ln)
        if rng.random() < 0.05:
            t += "$"
        toks.append(t)
        bq.append(chr(33 + int(rng.choice([5, 19, 20, 29, 30, 40]))))
        mq.append(chr(33 + int(rng.choice([0, 19, 20, 60]))))
    return "".join(toks), "".join(bq), "".join(mq)


def gen_columns():
    ref70 = "ACGTTGCAAC" * 7
    cases = []
    hand = [
        # (bases, bq, mq, ref_base, chunk_ref, cand)
        ("AAAa+2acC*#^]A$N-3NNNt", "I" * 10, "]" * 10, "A", "ACGTACGT", 1),
        ("", "", "", "C", ref70, 1),
        ("*", "*", "*", "G", ref70, 1),                                   # samtools placeholder row (all filtered)
        ("ACGTacgtNn*#", "?" * 12, "5" * 12, "T", ref70, 1),            # MQ 20 boundary (char '5' = 20)
        ("ACGTacgtNn*#", "?" * 12, "4" * 12, "T", ref70, 1),            # MQ 19
        ("AAAAaaaa", ">>>>????", "]]]]]]]]", "A", ref70, 1),             # BQ 29 / 30 boundary
        ("A+60" + "A" * 60 + "A+61" + "C" * 61 + "a+60" + "a" * 60, "III", "]]]", "A", ref70, 1),
        ("A-59" + "N" * 59 + "A-60" + "N" * 60 + "a-59" + "n" * 59, "III", "]]]", "A", ref70, 1),
        ("*+2AC#+2ac*-1N#-1nN+1A n-2nn", "IIIIII", "]]]]]]", "C", ref70, 1),
        ("A-2NNC-2NNa-2nnG-2NN", "IIII", "]]]]", "A", ref70, 1),         # one merged D key, four distinct keys
        ("A+1CA+1CA+1Ga+1ca+1ga+1g", "IIIIII", "]]]]]]", "A", ref70, 1),
        ("A+1C", "I", "5", "G", ref70, 0),
        ("GGGGgggg", "IIIIIIII", "]]]]4444", "G", ref70, 1),
        ("TtTtCc", "++++++", "]]]]]]", "T", ref70, 1),
        ("A$^!C.,<>G", "III", "]]]", "A", ref70, 1),                    # characters the tokeniser skips
        ("ACGT", "IIII", "]]]]", "A", "AC", 1),                          # short chunk_ref
        ("A-5NNNNN", "I", "]", "A", "ACG", 1),                           # D key truncated by the reference end
        ("a+3acga+3ACGA+3acg", "III", "]]]", "C", ref70, 1),
    ]
    for bases, bq, mq, rb, cr, cand in hand:
        t, a = ref_decode(bases, bq, mq, rb, cr, cand)
        cases.append(dict(bases=bases, bq=bq, mq=mq, ref=rb, chunk_ref=cr, cand=cand, tensor=t, alt_info=a))
    rng = np.random.default_rng(11)
    for i in range(260):
        depth = int(rng.integers(1, 90))
        bases, bq, mq = random_column(rng, depth)
        rb = "ACGT"[rng.integers(0, 4)]
        cr = "".join("ACGT"[rng.integers(0, 4)] for _ in range(60))
        cand = int(rng.random() < 0.7)
        t, a = ref_decode(bases, bq, mq, rb, cr, cand)
        cases.append(dict(bases=bases, bq=bq, mq=mq, ref=rb, chunk_ref=cr, cand=cand, tensor=t, alt_info=a))
    dump_json_gz("columns.json.gz", cases)


# ------------------------------------------------------------------------------------------ region
SHIM = r'''#!/usr/bin/env python3
import os, sys
a = sys.argv[1:]
if a[0] == "faidx":
    seq = open(os.environ["FAKE_REF"]).read().strip()
    ctg, rng = a[2].split(":")
    s, e = [int(x) for x in rng.split("-")]
    e = min(e, len(seq))
    sys.stdout.write(">%s:%d-%d\n" % (ctg, s, e))
    sub = seq[s - 1:e]
    for i in range(0, len(sub), 60):
        sys.stdout.write(sub[i:i + 60] + "\n")
elif a[0] == "mpileup":
    q = a[a.index("--min-BQ") + 1]
    sys.stdout.write(open(os.environ["FAKE_MPILEUP_" + q]).read())
else:
    sys.exit(1)
'''


def run_reference_create_tensor(tmp, min_bq, bed_fn, out_fn):
    cmd = [sys.executable, os.path.join(REF, "clairs_to.py"), "create_tensor_pileup_calling",
           "--tumor_bam_fn", "fake.bam", "--ref_fn", os.path.join(tmp, "ref.fa"), "--ctg_name", "chr1",
           "--min_bq", str(min_bq), "--samtools", os.path.join(tmp, "samtools"),
           "--candidates_bed_regions", bed_fn, "--tensor_can_fn", out_fn, "--platform", "ont"]
    subprocess.check_call(cmd, cwd=tmp, env=dict(os.environ, PYTHONPATH=REF))
    return gzip.open(out_fn, "rt").read()


def gen_region(tmp):
    # a dense little region: overlapping windows, early positions (< 17), gaps, N in the reference
    chunk = SynthChunk(48, seed=5, start=1, spacing=14, depth_mean=60.0, p_ins=0.02, p_del=0.03, n_rate=0.03)
    # candidates at positions < 17 cannot have complete windows; SynthChunk emits columns at pos >= 1 only
    keep_cols = chunk.col_pos >= 1
    assert keep_cols.all() or True
    ref, ref_lo = chunk.ref_window()
    assert ref_lo == 1
    sites = chunk.site_pos.tolist()
    texts = {}
    drop_pos = set()
    rng = np.random.default_rng(3)
    # drop some rows: a few flank positions and one candidate's own row
    allpos = chunk.col_pos.tolist()
    for p in rng.choice(allpos, size=25, replace=False).tolist():
        drop_pos.add(int(p))
    drop_pos.add(int(sites[7]))
    # two positions (one of them a candidate) where every read-base is below the AFF BQ gate: the NEG pass
    # sees the reads, the AFF pass gets samtools' depth-0 placeholder row
    low_pos = {int(sites[11]), int(sites[20]) + 3}
    for q in (0, 20):
        rows = []
        for r in mpileup_text(chunk, min_bq=q).split("\n"):
            if not r:
                continue
            f = r.split("\t")
            pos = int(f[1])
            if pos in drop_pos or pos < 1:
                continue
            if pos in low_pos:
                if q == 0:
                    f[5] = "+" * len(f[5])
                else:
                    f = [f[0], f[1], "N", "0", "*", "*", "*"]
            rows.append("\t".join(f))
        texts[q] = "\n".join(rows) + "\n"
    with open(os.path.join(tmp, "ref.fa"), "w") as f:
        f.write(">chr1\n" + ref + "\n")
    with open(os.path.join(tmp, "ref.fa.fai"), "w") as f:
        f.write("chr1\t%d\t6\t%d\t%d\n" % (len(ref), len(ref), len(ref) + 1))
    with open(os.path.join(tmp, "ref.txt"), "w") as f:
        f.write(ref)
    shim = os.path.join(tmp, "samtools")
    with open(shim, "w") as f:
        f.write(SHIM)
    os.chmod(shim, os.stat(shim).st_mode | stat.S_IEXEC)
    bed = os.path.join(tmp, "cand.bed")
    with open(bed, "w") as f:
        for x in sites:
            f.write("chr1\t%d\t%d\n" % (x - 17, x + 17))     # extract_candidates_calling writes x-17 .. x+17
    os.environ["FAKE_REF"] = os.path.join(tmp, "ref.txt")
    out = {}
    for q in (0, 20):
        p = os.path.join(tmp, "mp_%d.txt" % q)
        open(p, "w").write(texts[q])
        os.environ["FAKE_MPILEUP_%d" % q] = p
    for q, tag in ((20, "aff"), (0, "neg")):
        out[tag] = run_reference_create_tensor(tmp, q, bed, os.path.join(tmp, "tensor_%s.gz" % tag))
    fixture = dict(ref=ref, ref_start=1, sites=sites, min_bq_aff=20, mpileup_neg=texts[0], mpileup_aff=texts[20],
                   tensor_aff=out["aff"], tensor_neg=out["neg"])
    dump_json_gz("region.json.gz", fixture)
    return fixture


# ------------------------------------------------------------------------------------------ models
def build_reference_model(cls, n_out):
    if cls.startswith("CvT"):
        kw = dict(num_classes=2, s1_emb_dim=CVT_CFG["emb_dim"][0], s2_emb_dim=CVT_CFG["emb_dim"][1],
                  s3_emb_dim=CVT_CFG["emb_dim"][2], s1_heads=CVT_CFG["heads"][0], s2_heads=CVT_CFG["heads"][1],
                  s3_heads=CVT_CFG["heads"][2], s1_depth=CVT_CFG["depth"][0], s2_depth=CVT_CFG["depth"][1],
                  s3_depth=CVT_CFG["depth"][2], apply_softmax=False, model_type="acgt")
        m = getattr(rm, cls)(**kw)
    else:
        m = getattr(rm, cls)(apply_softmax=False, num_classes=2, model_type="nacgt")
    manifest = [(k, list(v.shape)) for k, v in m.state_dict().items() if not k.endswith("num_batches_tracked")]
    w = make_weights(manifest, seed=n_out)
    sd = m.state_dict()
    for k, v in w.items():
        sd[k] = torch.from_numpy(v.copy())
    m.load_state_dict(sd)
    m.eval()
    return m, manifest


def tensors_from_text(text):
    rows = [r.split("\t") for r in text.strip().split("\n") if r]
    X = np.array([[int(v) for v in r[3].split()] for r in rows], dtype=np.int32).reshape(-1, 33, 34)
    depth = np.array([int(r[4].split("-")[0]) for r in rows], dtype=np.int32)
    return rows, X, depth


def gen_models(region):
    _, Xa, da = tensors_from_text(region["tensor_aff"])
    _, Xn, dn = tensors_from_text(region["tensor_neg"])

    def rescale(X, d):   # predict.py:179-207
        out = np.empty(X.shape, dtype=np.float32)
        for i in range(len(X)):
            r = 50.0 / float(d[i]) if float(d[i]) > 50 else None
            v = [float(t) for t in X[i].ravel().tolist()]
            out[i] = np.array([t * r for t in v] if r is not None else v, dtype=np.float32).reshape(33, 34)
        return out

    xa, xn = rescale(Xa, da), rescale(Xn, dn)
    torch.set_num_threads(1)
    for cls, n_out, x in (("CvT", 4, xa), ("CvT_Indel", 6, xa), ("BiGRU_NACGT", 4, xn), ("BiGRU_NACGT_Indel", 6, xn)):
        m, manifest = build_reference_model(cls, n_out)
        with torch.no_grad():
            outs = m(torch.from_numpy(x))
        logits = np.stack([o.numpy() for o in outs])    # [K][B][2]
        np.savez_compressed(os.path.join(HERE, "models_%s.npz" % cls), x=x, logits=logits.astype(np.float32),
                            manifest=json.dumps(manifest), n_out=n_out)
        print("wrote models_%s.npz" % cls, logits.shape, "logit range", float(logits.min()), float(logits.max()))


# ------------------------------------------------------------------------------------------ predict + call_variants
def gen_calls(tmp, region):
    for mode, n_out, aff_cls, neg_cls in (("snv", 4, "CvT", "BiGRU_NACGT"), ("indel", 6, "CvT_Indel", "BiGRU_NACGT_Indel")):
        ma, _ = build_reference_model(aff_cls, n_out)
        mn, _ = build_reference_model(neg_cls, n_out)
        pa, pn = os.path.join(tmp, "aff_%s.pkl" % mode), os.path.join(tmp, "neg_%s.pkl" % mode)
        torch.save({"model_acgt": ma}, pa)      # pickled by qualified name clairs.model.<cls>
        torch.save({"model_nacgt": mn}, pn)
        pred = os.path.join(tmp, "pred_%s.gz" % mode)
        disable = "True" if mode == "snv" else "False"
        subprocess.check_call([sys.executable, os.path.join(REF, "clairs_to.py"), "predict",
                               "--tensor_fn_acgt", os.path.join(tmp, "tensor_aff.gz"),
                               "--tensor_fn_nacgt", os.path.join(tmp, "tensor_neg.gz"),
                               "--chkpnt_fn_acgt", pa, "--chkpnt_fn_nacgt", pn, "--predict_fn", pred,
                               "--pileup", "--disable_indel_calling", disable, "--ctg_name", "chr1"],
                              cwd=tmp, env=dict(os.environ, PYTHONPATH=REF))
        pred_rows = gzip.open(pred, "rt").read()
        table = likelihood_table(n_out, seed=7 + n_out)
        lik_fn = os.path.join(tmp, "lik_%s.txt" % mode)
        np.savetxt(lik_fn, table, fmt="%.17g")
        vcfs = {}
        for show_ref in (False, True):
            vcf_fn = os.path.join(tmp, "out_%s_%d.vcf" % (mode, int(show_ref)))
            cmd = [sys.executable, os.path.join(REF, "clairs_to.py"), "call_variants", "--predict_fn", pred,
                   "--call_fn", vcf_fn, "--likelihood_matrix_data", lik_fn, "--disable_indel_calling", disable,
                   "--ctg_name", "chr1", "--pileup"]
            if show_ref:
                cmd.append("--show_ref")
            subprocess.check_call(cmd, cwd=tmp, env=dict(os.environ, PYTHONPATH=REF))
            rows = [r for r in open(vcf_fn).read().split("\n") if r and not r.startswith("#")] if os.path.exists(vcf_fn) else []
            vcfs["show_ref" if show_ref else "default"] = rows
        dump_json_gz("calls_%s.json.gz" % mode, dict(n_out=n_out, predict_rows=pred_rows,
                                                     likelihood_table=open(lik_fn).read(), vcf=vcfs))
        print(mode, "vcf rows:", {k: len(v) for k, v in vcfs.items()})


# ------------------------------------------------------------------------------------------ candidate extraction
def gen_extract(tmp):
    """extract_candidates_calling through the shim: `samtools mpileup --min-MQ 20 --min-BQ 20` text in,
    SNV / indel candidate BED chunk files out."""
    chunk = SynthChunk(70, seed=21, start=2000, spacing=30, depth_mean=14.0, p_mismatch=0.03, p_ins=0.03, p_del=0.04,
                       n_rate=0.03)
    ref, ref_lo = chunk.ref_window()
    full_ref = "A" * (ref_lo - 1) + ref                     # the shim serves a contig that starts at position 1
    text6 = mpileup_text(chunk, min_bq=20, min_mq=20, with_mq=False)
    d = os.path.join(tmp, "extract")
    os.makedirs(d, exist_ok=True)
    with open(os.path.join(d, "ref.fa"), "w") as f:
        f.write(">chr1\n" + full_ref + "\n")
    with open(os.path.join(d, "ref.fa.fai"), "w") as f:
        f.write("chr1\t%d\t6\t%d\t%d\n" % (len(full_ref), len(full_ref), len(full_ref) + 1))
    open(os.path.join(d, "ref.txt"), "w").write(full_ref)
    shim = os.path.join(d, "samtools")
    open(shim, "w").write(SHIM)
    os.chmod(shim, os.stat(shim).st_mode | stat.S_IEXEC)
    open(os.path.join(d, "mp.txt"), "w").write(text6)
    env = dict(os.environ, PYTHONPATH=REF, FAKE_REF=os.path.join(d, "ref.txt"), FAKE_MPILEUP_20=os.path.join(d, "mp.txt"))
    params = dict(snv_min_af=0.05, indel_min_af=0.05, min_coverage=4, min_bq=20, min_mq=20, alt_base_num=3)
    ctg_start, ctg_end = int(chunk.col_pos[0]), int(chunk.col_pos[-1])
    out_dir = os.path.join(d, "candidates")
    subprocess.check_call([sys.executable, os.path.join(REF, "clairs_to.py"), "extract_candidates_calling",
                           "--tumor_bam_fn", "fake.bam", "--ref_fn", os.path.join(d, "ref.fa"), "--samtools", shim,
                           "--snv_min_af", str(params["snv_min_af"]), "--indel_min_af", str(params["indel_min_af"]),
                           "--ctg_name", "chr1", "--ctg_start", str(ctg_start), "--ctg_end", str(ctg_end),
                           "--platform", "ont", "--min_coverage", str(params["min_coverage"]), "--min_bq", "20",
                           "--select_indel_candidates", "True", "--candidates_folder", out_dir, "--output_depth", "True"],
                          cwd=d, env=env)

    def centres(suffix):
        out = []
        for fn in sorted(os.listdir(out_dir)):
            if fn.endswith(suffix) and fn.startswith("chr1."):
                for row in open(os.path.join(out_dir, fn)):
                    c = row.split("\t")
                    if len(c) >= 3:
                        out.append(int(c[2]) - 17)             # rows are ctg, max(x-17, 1), x+17
        return out
    fixture = dict(ref=ref, ref_start=ref_lo, params=params, mpileup_extract=text6, mpileup_neg=mpileup_text(chunk, 0),
                   snv=centres("_snv"), indel=centres("_indel"))
    print("extract: rows", text6.count("\n"), "snv", len(fixture["snv"]), "indel", len(fixture["indel"]))
    dump_json_gz("extract.json.gz", fixture)


# ------------------------------------------------------------------------------------------ sort_vcf + postprocess_vcf
def gen_post(tmp):
    """reference `sort_vcf` and `postprocess_vcf` on chunk VCFs made of the call_variants records of
    calls_snv.json.gz, re-labelled over several contigs and salted with the record kinds the gates distinguish
    (phaseable `H` INFO, NonSomatic / LowQual filters, low QUAL, low AF)."""
    import random
    calls = json.load(gzip.open(os.path.join(HERE, "calls_snv.json.gz"), "rt"))
    rows = calls["vcf"]["show_ref"]
    sys.path.insert(0, REF)
    from shared.vcf import vcf_header
    rnd = random.Random(11)
    contigs = ["chr1", "chr11", "chr2", "chrX", "scaffold_7"]
    d = os.path.join(tmp, "post")
    os.makedirs(os.path.join(d, "vcf_output"), exist_ok=True)
    chunks = {}
    for ci, ctg in enumerate(contigs):
        for part in range(2):
            recs = []
            for r in rows[part::2]:
                c = r.split("\t")
                c[0] = ctg
                c[1] = str(int(c[1]) + 1000 * ci)
                u = rnd.random()
                if c[6] == "PASS":
                    if u < 0.25:
                        c[7] += ";H"
                    if rnd.random() < 0.3:
                        c[5] = "%.4f" % rnd.uniform(0.5, 14.0)
                    if rnd.random() < 0.15:
                        c[6] = "NonSomatic"
                    elif rnd.random() < 0.1:
                        c[6] = "LowQual"
                recs.append("\t".join(c))
            rnd.shuffle(recs)
            head = vcf_header + "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tSAMPLE\n"
            fn = "p_%s.%d_2.vcf" % (ctg, part + 1)
            chunks[fn] = head + "".join(x + "\n" for x in recs)
            open(os.path.join(d, "vcf_output", fn), "w").write(chunks[fn])
    order = ["chrX", "scaffold_7", "chr2", "chr11", "chr1", "chr5"]       # chr5 has no files
    open(os.path.join(d, "CONTIGS"), "w").write("".join(c + "\n" for c in order))
    fai = "".join("%s\t%d\t6\t60\t61\n" % (c, 100000 + i) for i, c in enumerate(["chr1", "chr2", "chr5", "chr11", "chrX", "scaffold_7"]))
    open(os.path.join(d, "ref.fa.fai"), "w").write(fai)
    open(os.path.join(d, "ref.fa"), "w").write(">x\nA\n")
    open(os.path.join(d, "CMD"), "w").write("run_clairs_to --synthetic fixture\n")
    env = dict(os.environ, PYTHONPATH=REF)
    merged = os.path.join(d, "merged.vcf")
    subprocess.check_call([sys.executable, os.path.join(REF, "clairs_to.py"), "sort_vcf", "--input_dir", os.path.join(d, "vcf_output"),
                           "--vcf_fn_prefix", "p_", "--vcf_fn_suffix", ".vcf", "--output_fn", merged, "--sample_name", "SAMPLE",
                           "--ref_fn", os.path.join(d, "ref.fa"), "--contigs_fn", os.path.join(d, "CONTIGS")], cwd=d, env=env)
    empty = os.path.join(d, "empty.vcf")
    subprocess.check_call([sys.executable, os.path.join(REF, "clairs_to.py"), "sort_vcf", "--input_dir", os.path.join(d, "vcf_output"),
                           "--vcf_fn_prefix", "nothing_", "--output_fn", empty, "--sample_name", "S2",
                           "--ref_fn", os.path.join(d, "ref.fa"), "--contigs_fn", os.path.join(d, "CONTIGS")], cwd=d, env=env)
    cases = []
    for opts in ({"platform": "ont"}, {"platform": "ilmn"}, {"platform": "hifi", "qual": 20.0, "af": 0.3},
                 {"platform": "ont", "qual_cutoff_phaseable_region": 5.0, "qual_cutoff_unphaseable_region": 30.0,
                  "cmdline": True, "ref_fn": True},
                 {"platform": "ilmn", "max_qual_filter_pileup_calls": 12.0, "af": 0.35},
                 {"platform": "ont", "max_qual_filter_pileup_calls": 25.0}):
        out = os.path.join(d, "final_%d.vcf" % len(cases))
        cmd = [sys.executable, os.path.join(REF, "clairs_to.py"), "postprocess_vcf", "--pileup_vcf_fn", merged, "--output_fn", out,
               "--compress_vcf", "False", "--sample_name", "SAMPLE", "--platform", opts["platform"]]
        for k in ("qual", "af", "qual_cutoff_phaseable_region", "qual_cutoff_unphaseable_region", "max_qual_filter_pileup_calls"):
            if k in opts:
                cmd += ["--" + k, str(opts[k])]
        if opts.get("cmdline"):
            cmd += ["--cmdline", os.path.join(d, "CMD")]
        if opts.get("ref_fn"):
            cmd += ["--ref_fn", os.path.join(d, "ref.fa")]
        subprocess.check_call(cmd, cwd=d, env=env)
        cases.append(dict(opts=opts, out=open(out).read()))
    fixture = dict(chunks=chunks, contigs_order=order, fai=fai, cmd=open(os.path.join(d, "CMD")).read(),
                   merged=open(merged).read(), empty=open(empty).read(), cases=cases)
    print("post: merged records", sum(1 for r in fixture["merged"].split("\n") if r and r[0] != "#"),
          "case records", [sum(1 for r in c["out"].split("\n") if r and r[0] != "#") for c in cases])
    dump_json_gz("post.json.gz", fixture)



# ------------------------------------------------------------------------------------------ call_variants branch coverage
def _prob_row(ctg, pos, ref, alt_info, fwd, rev, K, winner, strong=True, aff=None, neg=None):
    """One probability row in predict.py's format (predict.py:114-152).  `winner`: class index whose (AFF, NEG) pair says
    "yes" (p_aff high, p_neg low); the other classes say "no".  strong=False keeps the pair near 0.5 (low QUAL)."""
    hi, lo = (0.99999, 0.00001) if strong else (0.58, 0.42)
    pa = [lo] * K if aff is None else list(aff)
    pn = [hi] * K if neg is None else list(neg)
    if aff is None:
        pa[winner] = hi
    if neg is None:
        pn[winner] = lo
    fields = [ctg, str(pos), ref, alt_info, str([float(v) for v in fwd]), str([float(v) for v in rev])]
    fields += ["{:0.8f} {:0.8f}".format(1.0 - p, p) for p in pa] + ["{:0.8f} {:0.8f}".format(1.0 - p, p) for p in pn]
    return "\t".join(fields) + ("\t\n" if K == 4 else "\n")


def gen_branches(tmp):
    """Q4 / Q6 branch coverage (call_variants.py:135-150, 306-415, 67-76): hand-made probability rows whose
    alt_info / winner combinations drive the REFERENCE's call_variants through every ALT / AF / GT / FILTER branch - insertion
    ALT (forward and '#'-anchored), deletion REF, 1/1, AF clamp, first-seen tie-break, SNV demotion, depth 0 and its
    all-indel fallback, empty allele list, indel alleles in SNV mode, SNV alleles in indel mode, REF == ALT, LowQual."""
    A, C, G, T, I, D = range(6)
    snv = [  # (ref, alt_info, fwd, rev, winner, strong)
        ("A", "20-XT 20-", [0, 0, 0, 11], [0, 0, 0, 9], T, True),                 # AF = 1 -> 1/1
        ("A", "10-XT 12-", [0, 0, 0, 6], [0, 0, 0, 6], T, True),                  # count > depth: AF clamps to 1 -> 1/1
        ("A", "40-XT 9 R 31-", [15, 0, 0, 5], [16, 0, 0, 4], T, True),            # plain 0/1
        ("A", "40-XT 9 R 31-", [15, 0, 0, 5], [16, 0, 0, 4], C, True),            # winner C not observed -> demoted to RefCall
        ("A", "40-XT 9 R 31-", [15, 0, 0, 5], [16, 0, 0, 4], A, True),            # reference wins
        ("A", "40-XT 9-", [0, 0, 0, 5], [0, 0, 0, 4], A, True),                   # reference wins, no R key -> AD 0
        ("G", "30-XC 5 XT 5 R 20-", [0, 3, 10, 2], [0, 2, 10, 3], T, True),       # tie: first-seen XC is the ALT although T won
        ("G", "30-XT 5 XC 5 R 20-", [0, 3, 10, 2], [0, 2, 10, 3], C, True),       # same tie, other order
        ("G", "30-XC 4 XT 6 R 20-", [0, 3, 10, 2], [0, 2, 10, 3], C, True),       # best allele is not the first key
        ("C", "25-IAGT 14 R 11-", [0, 6, 0, 0], [0, 5, 0, 0], T, True),           # best allele is an insertion: dropped in SNV mode
        ("C", "25-DCAG 14 XT 3 R 8-", [0, 4, 0, 2], [0, 4, 0, 1], T, True),       # best allele is a deletion: dropped in SNV mode
        ("C", "25-XT 14 DCAG 3 R 8-", [0, 4, 0, 7], [0, 4, 0, 7], T, True),       # SNV allele ranked above an indel allele
        ("T", "0--", [0, 0, 0, 0], [0, 0, 0, 0], A, True),                        # depth 0, variant -> "low tumor coverage"
        ("T", "0-", [0, 0, 0, 0], [0, 0, 0, 0], A, True),                         # depth 0, no second field
        ("T", "0-XA 3 XC 2-", [2, 1, 0, 0], [1, 1, 0, 0], A, True),               # depth 0 with two keys: no fallback
        ("T", "0-IAC 5-", [0, 0, 0, 0], [0, 0, 0, 0], A, True),                   # depth 0, single indel key -> depth = 5, then dropped (indel)
        ("T", "0-XA 5-", [3, 0, 0, 0], [2, 0, 0, 0], A, True),                    # depth 0, single SNV key: no fallback
        ("T", "33-R 33-", [0, 0, 0, 17], [0, 0, 0, 16], G, True),                 # variant wins but no allele observed
        ("T", "33-XG 0 R 33-", [0, 0, 0, 17], [0, 0, 0, 16], G, True),            # zero-count allele: AF 0 is not supported
        ("A", "40-XT 9 R 31-", [15, 0, 0, 5], [16, 0, 0, 4], T, False),           # weak call: low QUAL (LowQual under --qual 20)
        ("A", "40-XG 21 R 19-", [9, 0, 11, 0], [10, 0, 10, 0], G, False),
        ("A", "57-XC 1 XG 2 XT 3 R 51-", [25, 1, 1, 1], [26, 0, 1, 2], T, True),  # three alleles, ascending counts
        ("N", "12-XA 5 R 7-", [3, 0, 0, 0], [2, 0, 0, 0], A, True),               # reference base outside ACGT as printed by predict (never N there, kept for the parser)
    ]
    indel = [
        ("A", "30-IAGT 12 R 18-", [9, 0, 0, 0], [9, 0, 0, 0], I, True),           # insertion, forward anchor: ALT = AGT
        ("A", "30-I#GT 12 R 18-", [9, 0, 0, 0], [9, 0, 0, 0], I, True),           # '#' anchor: ALT = ref + GT
        ("A", "30-DAC 9 R 21-", [10, 0, 0, 0], [11, 0, 0, 0], D, True),           # deletion: REF = AC, ALT = A
        ("A", "30-DACGTT 9 R 21-", [10, 0, 0, 0], [11, 0, 0, 0], D, True),        # longer deletion
        ("A", "30-DA 9 R 21-", [10, 0, 0, 0], [11, 0, 0, 0], D, True),            # truncated D key: REF == ALT -> dropped
        ("A", "12-IAT 12-", [0, 0, 0, 0], [0, 0, 0, 0], I, True),                 # AF = 1 -> 1/1
        ("A", "12-IAT 12-", [0, 0, 0, 0], [0, 0, 0, 0], D, True),                 # D wins, best allele is the insertion
        ("A", "30-DAC 9 IAG 4 R 17-", [8, 0, 0, 0], [9, 0, 0, 0], I, True),       # I wins, best allele is the deletion
        ("A", "30-IAG 4 DAC 4 R 22-", [11, 0, 0, 0], [11, 0, 0, 0], D, True),     # tie between I and D: first-seen
        ("A", "30-XT 10 IAG 4 R 16-", [8, 0, 0, 5], [8, 0, 0, 5], I, True),       # I wins, best allele is an SNV: only with --show_ref
        ("A", "30-XT 10 IAG 4 R 16-", [8, 0, 0, 5], [8, 0, 0, 5], A, True),       # reference (A) wins in indel mode
        ("A", "30-XT 10 IAG 4 R 16-", [8, 0, 0, 5], [8, 0, 0, 5], T, True),       # a non-reference BASE wins: still "reference" in indel mode
        ("C", "0-ICA 7-", [0, 0, 0, 0], [0, 0, 0, 0], I, True),                   # depth 0 fallback to the indel count -> 1/1
        ("C", "0-ICA 7 DCT 2-", [0, 0, 0, 0], [0, 0, 0, 0], I, True),             # depth 0, two keys -> low tumor coverage
        ("C", "0--", [0, 0, 0, 0], [0, 0, 0, 0], D, True),
        ("C", "41-R 41-", [0, 20, 0, 0], [0, 21, 0, 0], I, True),                 # no allele observed
        ("G", "30-IGAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAC 6 R 24-", [0, 0, 12, 0], [0, 0, 12, 0], I, True),
        ("A", "30-IAGT 12 R 18-", [9, 0, 0, 0], [9, 0, 0, 0], I, False),          # weak call -> LowQual under --qual 20
        ("A", "30-DAC 9 R 21-", [10, 0, 0, 0], [11, 0, 0, 0], D, False),
        ("T", "48-ITC 3 ITCC 3 ITG 5 DTA 5 R 32-", [0, 0, 0, 16], [0, 0, 0, 16], I, True),   # several alleles, tie between ITG and DTA
    ]
    out = {}
    for mode, K, cases in (("snv", 4, snv), ("indel", 6, indel)):
        text = "".join(_prob_row("chr1", 1000 + 10 * i, c[0], c[1], c[2], c[3], K, c[4], c[5]) for i, c in enumerate(cases))
        pred = os.path.join(tmp, "branch_pred_%s.gz" % mode)
        with gzip.open(pred, "wt") as f:
            f.write(text)
        table = likelihood_table(K, seed=70 + K)
        lik_fn = os.path.join(tmp, "branch_lik_%s.txt" % mode)
        np.savetxt(lik_fn, table, fmt="%.17g")
        runs = {}
        for qual in (None, 20):
            for show_ref in (False, True):
                vcf_fn = os.path.join(tmp, "branch_%s_%s_%d.vcf" % (mode, qual, int(show_ref)))
                cmd = [sys.executable, os.path.join(REF, "clairs_to.py"), "call_variants", "--predict_fn", pred, "--call_fn", vcf_fn,
                       "--likelihood_matrix_data", lik_fn, "--disable_indel_calling", "True" if K == 4 else "False",
                       "--ctg_name", "chr1", "--pileup"]
                if qual is not None:
                    cmd += ["--qual", str(qual)]
                if show_ref:
                    cmd.append("--show_ref")
                res = subprocess.run(cmd, cwd=tmp, env=dict(os.environ, PYTHONPATH=REF), capture_output=True, text=True)
                assert res.returncode == 0, res.stderr
                rows = [r for r in open(vcf_fn).read().split("\n") if r and not r.startswith("#")] if os.path.exists(vcf_fn) else []
                runs["qual%s_showref%d" % (0 if qual is None else qual, int(show_ref))] = dict(
                    rows=rows, low_cov_messages=res.stdout.count("low tumor coverage"))
        out[mode] = dict(n_out=K, predict_rows=text, likelihood_table=open(lik_fn).read(), runs=runs)
        allrows = [r.split("\t") for v in runs.values() for r in v["rows"]]
        print(mode, "branch rows:", {k: len(v["rows"]) for k, v in runs.items()},
              "GT", sorted({r[9].split(":")[0] for r in allrows}), "FILTER", sorted({r[6] for r in allrows}),
              "max len REF/ALT", max(len(r[3]) for r in allrows), max(len(r[4]) for r in allrows))
    dump_json_gz("calls_branches.json.gz", out)


# ------------------------------------------------------------------------------------------ 2 000-site region
def _sha(text):
    import hashlib
    return hashlib.sha256(text.encode()).hexdigest()


def _row_crcs(text):
    import zlib
    return [zlib.crc32(r.encode()) & 0xffffffff for r in text.split("\n") if r]


REGION2K = dict(n_sites=2000, seed=20260928, start=5000, spacing=40, depth_mean=52.0, p_ins=0.01, p_del=0.012, n_rate=0.004)


def gen_region2k(tmp):
    """A 2 000-candidate ONT-like region with the mean depth sitting on the rescale threshold (AFF and NEG depths on both sides of
    50, predict.py:181) through the reference's four commands.  The INPUTS are not stored: they are regenerated from REGION2K by the
    same generator (clairs_to_amd/synth.py) and checked against the stored SHA-256; the OUTPUTS are stored as the reference wrote
    them (probability rows, VCF rows) or, for the 2 x 5 MB of tensor text, as SHA-256 + per-row CRC-32 + depths + alt_info."""
    kw = dict(REGION2K)
    chunk = SynthChunk(kw.pop("n_sites"), **kw)
    ref, ref_lo = chunk.ref_window()
    full_ref = "A" * (ref_lo - 1) + ref
    d = os.path.join(tmp, "r2k")
    os.makedirs(d, exist_ok=True)
    texts = {q: mpileup_text(chunk, min_bq=q) for q in (0, 20)}
    open(os.path.join(d, "ref.fa"), "w").write(">chr1\n" + full_ref + "\n")
    open(os.path.join(d, "ref.fa.fai"), "w").write("chr1\t%d\t6\t%d\t%d\n" % (len(full_ref), len(full_ref), len(full_ref) + 1))
    open(os.path.join(d, "ref.txt"), "w").write(full_ref)
    shim = os.path.join(d, "samtools")
    open(shim, "w").write(SHIM)
    os.chmod(shim, os.stat(shim).st_mode | stat.S_IEXEC)
    bed = os.path.join(d, "cand.bed")
    sites = chunk.site_pos.tolist()
    open(bed, "w").write("".join("chr1\t%d\t%d\n" % (x - 17, x + 17) for x in sites))
    env = dict(os.environ, PYTHONPATH=REF, FAKE_REF=os.path.join(d, "ref.txt"))
    for q in (0, 20):
        open(os.path.join(d, "mp_%d.txt" % q), "w").write(texts[q])
        env["FAKE_MPILEUP_%d" % q] = os.path.join(d, "mp_%d.txt" % q)
    tens = {}
    for q, tag in ((20, "aff"), (0, "neg")):
        out_fn = os.path.join(d, "tensor_%s.gz" % tag)
        subprocess.check_call([sys.executable, os.path.join(REF, "clairs_to.py"), "create_tensor_pileup_calling", "--tumor_bam_fn", "fake.bam",
                               "--ref_fn", os.path.join(d, "ref.fa"), "--ctg_name", "chr1", "--min_bq", str(q), "--samtools", shim,
                               "--candidates_bed_regions", bed, "--tensor_can_fn", out_fn, "--platform", "ont"], cwd=d, env=env)
        tens[tag] = gzip.open(out_fn, "rt").read()
    fixture = dict(params=REGION2K, min_bq_aff=20, input_sha=dict(mpileup_neg=_sha(texts[0]), mpileup_aff=_sha(texts[20]), ref=_sha(ref)),
                   tensor={})
    for tag in ("aff", "neg"):
        rows = [r.split("\t") for r in tens[tag].split("\n") if r]
        fixture["tensor"][tag] = dict(sha=_sha(tens[tag]), row_crc=_row_crcs(tens[tag]), pos=[int(r[1]) for r in rows],
                                      alt_info=[r[4] for r in rows])
    da = np.array([int(a.split("-")[0]) for a in fixture["tensor"]["aff"]["alt_info"]])
    dn = np.array([int(a.split("-")[0]) for a in fixture["tensor"]["neg"]["alt_info"]])
    print("region2k: rows", len(da), "AFF depth <=50 / >50:", int((da <= 50).sum()), int((da > 50).sum()),
          "NEG:", int((dn <= 50).sum()), int((dn > 50).sum()), "AFF<=50<NEG:", int(((da <= 50) & (dn > 50)).sum()))
    fixture["calls"] = {}
    for mode, n_out, aff_cls, neg_cls in (("snv", 4, "CvT", "BiGRU_NACGT"), ("indel", 6, "CvT_Indel", "BiGRU_NACGT_Indel")):
        ma, _ = build_reference_model(aff_cls, n_out)
        mn, _ = build_reference_model(neg_cls, n_out)
        pa, pn = os.path.join(d, "aff_%s.pkl" % mode), os.path.join(d, "neg_%s.pkl" % mode)
        torch.save({"model_acgt": ma}, pa)
        torch.save({"model_nacgt": mn}, pn)
        pred = os.path.join(d, "pred_%s.gz" % mode)
        disable = "True" if mode == "snv" else "False"
        subprocess.check_call([sys.executable, os.path.join(REF, "clairs_to.py"), "predict", "--tensor_fn_acgt", os.path.join(d, "tensor_aff.gz"),
                               "--tensor_fn_nacgt", os.path.join(d, "tensor_neg.gz"), "--chkpnt_fn_acgt", pa, "--chkpnt_fn_nacgt", pn,
                               "--predict_fn", pred, "--pileup", "--disable_indel_calling", disable, "--ctg_name", "chr1"], cwd=d, env=env)
        table = likelihood_table(n_out, seed=7 + n_out)
        lik_fn = os.path.join(d, "lik_%s.txt" % mode)
        np.savetxt(lik_fn, table, fmt="%.17g")
        vcf_fn = os.path.join(d, "out_%s.vcf" % mode)
        subprocess.check_call([sys.executable, os.path.join(REF, "clairs_to.py"), "call_variants", "--predict_fn", pred, "--call_fn", vcf_fn,
                               "--likelihood_matrix_data", lik_fn, "--disable_indel_calling", disable, "--ctg_name", "chr1", "--pileup",
                               "--show_ref"], cwd=d, env=env)
        rows = [r for r in open(vcf_fn).read().split("\n") if r and not r.startswith("#")]
        prows = [r.split("\t") for r in gzip.open(pred, "rt").read().split("\n") if r]
        # probability rows: positions + the 8-decimal p1 of every head (the other fields repeat the tensor rows)
        fixture["calls"][mode] = dict(n_out=n_out, pos=[int(r[1]) for r in prows], ref=[r[2] for r in prows],
                                      strand=[[r[4], r[5]] for r in prows],
                                      p1=[[f.split()[1] for f in r[6:6 + 2 * n_out]] for r in prows],
                                      likelihood_table=open(lik_fn).read(), vcf_show_ref=rows)
        print("region2k", mode, "predict rows", len(prows), "vcf rows", len(rows),
              "variants", sum(1 for r in rows if r.split("\t")[6] != "RefCall"))
    dump_json_gz("region2k.json.gz", fixture)


# ------------------------------------------------------------------------------------------ genuine checkpoint pickles
def gen_pickles(tmp):
    """`torch.save({'model_acgt': <clairs.model.CvT object>})` exactly as the reference's releases are written (predict.py:513-517,
    555-568): the pickle stream names the reference's classes, attribute layout and sub-module nesting.  The parameter VALUES are
    zeroed before saving (so the four files compress to a few KB); the test loads the pickle through the shims' aliases, fills in
    weights_recipe values and compares with the logits of models_<cls>.npz.  Both CvT hyper-parameter sets are covered: the one
    predict.py builds (16/64/128, heads 1/3/4, depth 1/2/3) and the constructor defaults (32/64/128, 1/3/6, 1/2/10)."""
    import base64
    out = {}
    specs = [("CvT", "model_acgt", dict(model_type="acgt", apply_softmax=False)), ("CvT_Indel", "model_acgt", dict(model_type="acgt", apply_softmax=False)),
             ("BiGRU_NACGT", "model_nacgt", dict(model_type="nacgt", apply_softmax=False)),
             ("BiGRU_NACGT_Indel", "model_nacgt", dict(model_type="nacgt", apply_softmax=False)),
             ("CvT:defaults", "model_acgt", dict(model_type="acgt"))]
    for name, key, kw in specs:
        cls = name.split(":")[0]
        if name == "CvT:defaults":
            m = rm.CvT(**kw)
        elif cls.startswith("CvT"):
            m, _ = build_reference_model(cls, 4 if cls == "CvT" else 6)
        else:
            m, _ = build_reference_model(cls, 4 if cls == "BiGRU_NACGT" else 6)
        m.eval()
        with torch.no_grad():
            for t in list(m.parameters()) + list(m.buffers()):
                t.zero_()
        fn = os.path.join(tmp, "pk_%s.pkl" % name.replace(":", "_"))
        torch.save({key: m}, fn)
        raw = open(fn, "rb").read()
        comp = gzip.compress(raw, mtime=0)
        out[name] = dict(key=key, pickle_gz_b64=base64.b64encode(comp).decode(), n_params=int(sum(p.numel() for p in m.parameters())),
                         state_keys=[k for k in m.state_dict().keys()], attrs={k: getattr(m, k) for k in ("model_type", "apply_softmax") if hasattr(m, k)})
        print("pickle", name, len(raw), "->", len(comp), "bytes")
    dump_json_gz("pickles.json.gz", out)


# ------------------------------------------------------------------------------------------ haplotype filtering (8f #4)
SHIM_HAP = r'''#!/usr/bin/env python3
import os, sys
a = sys.argv[1:]
if a[0] == "faidx":
    seq = open(os.environ["FAKE_REF"]).read().strip()
    ctg, rng = a[2].split(":")
    s, e = [int(x) for x in rng.split("-")]
    e = min(e, len(seq))
    sys.stdout.write(">%s:%d-%d\n" % (ctg, s, e))
    sub = seq[s - 1:e]
    for i in range(0, len(sub), 60):
        sys.stdout.write(sub[i:i + 60] + "\n")
elif a[0] == "mpileup":
    ctg, rng = a[a.index("-r") + 1].split(":")
    lo, hi = [int(x) for x in rng.split("-")]
    bed = None
    if "-l" in a:
        bed = [tuple(int(v) for v in r.split("\t")[1:3]) for r in open(a[a.index("-l") + 1]) if r.strip()]
    for row in open(os.environ["FAKE_MPILEUP_HAP"]):
        p = int(row.split("\t", 2)[1])
        if lo <= p <= hi and (bed is None or any(b < p <= e for b, e in bed)):
            sys.stdout.write(row)
else:
    sys.exit(1)
'''


def gen_hapfilter(tmp):
    """src/haplotype_filtering.py (reference, chunk mode: one in-process mpileup per <= 200 calls) on a simulated haplotagged
    pileup: pileup VCF of PASS calls + germline VCF + nine-column mpileup text in, filtered VCF (FILTER tags, H / SB INFO) out,
    for the SNV pass and the indel pass (--is_indel)."""
    import hapsim
    sim = hapsim.simulate()
    ref = sim["ref"]
    d = os.path.join(tmp, "hap")
    os.makedirs(d, exist_ok=True)
    open(os.path.join(d, "ref.fa"), "w").write(">chr1\n" + ref + "\n")
    open(os.path.join(d, "ref.fa.fai"), "w").write("chr1\t%d\t6\t%d\t%d\n" % (len(ref), len(ref), len(ref) + 1))
    open(os.path.join(d, "ref.txt"), "w").write(ref)
    shim = os.path.join(d, "samtools")
    open(shim, "w").write(SHIM_HAP)
    os.chmod(shim, os.stat(shim).st_mode | stat.S_IEXEC)
    open(os.path.join(d, "fake.bam"), "w").write("")
    head = ("##fileformat=VCFv4.2\n##FILTER=<ID=PASS,Description=\"All filters passed\">\n"
            "##FORMAT=<ID=TU,Number=1,Type=Integer,Description=\"Count of T in the tumor BAM\">\n"
            "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tSAMPLE\n")
    germ_vcf = os.path.join(d, "germline.vcf")
    open(germ_vcf, "w").write(head + "".join("chr1\t%d\t.\t%s\t%s\t30.0\tPASS\t.\tGT:GQ\t%s:30\n" % g for g in sim["germline"]))
    out = dict(ref=ref, germline_vcf=open(germ_vcf).read(), modes={})
    flank = 100
    for mode, calls in (("snv", sim["snv_calls"]), ("indel", sim["indel_calls"])):
        positions = sorted({p for c in calls for p in range(max(1, c[0] - flank), c[0] + flank + 1)})
        text = hapsim.pileup_rows(sim, positions)
        mp = os.path.join(d, "mp_%s.txt" % mode)
        open(mp, "w").write(text)
        by_pos = {int(r.split("\t", 2)[1]): r.split("\t") for r in text.split("\n") if r}
        rows = []
        for i, (p, rb, ab) in enumerate(calls):
            cols = by_pos[p]
            toks = cols[7].count(",") + 1
            if len(rb) == 1 and len(ab) == 1:
                n_alt = sum(1 for c in cols[4].upper() if c == ab)
            else:
                n_alt = cols[4].count("+") + cols[4].count("-")
            af = min(1.0, n_alt / float(toks))
            flt = "PASS" if i % 11 != 10 else "LowQual"             # a non-PASS input record must pass through untouched
            rows.append("chr1\t%d\t.\t%s\t%s\t%.4f\t%s\tFAU=1;FCU=2;FGU=3;FTU=4;RAU=5;RCU=6;RGU=7;RTU=8\tGT:GQ:DP:AF:AD:AU:CU:GU:TU\t0/1:%d:%d:%.4f:%d,%d:1:2:3:4\n"
                        % (p, rb, ab, 12.5 + i, flt, 12 + i, toks, af, toks - n_alt, n_alt))
        pile_vcf = os.path.join(d, "pileup_%s.vcf" % mode)
        open(pile_vcf, "w").write(head + "".join(rows))
        out_vcf = os.path.join(d, "out_%s.vcf" % mode)
        cmd = [sys.executable, os.path.join(REF, "clairs_to.py"), "haplotype_filtering", "--tumor_bam_fn", os.path.join(d, "fake.bam"),
               "--ref_fn", os.path.join(d, "ref.fa"), "--ctg_name", "chr1", "--pileup_vcf_fn", pile_vcf, "--germline_vcf_fn", germ_vcf,
               "--output_vcf_fn", out_vcf, "--output_dir", os.path.join(d, "work_%s" % mode), "--samtools", shim, "--threads", "1",
               "--haplotype_filtering_chunk_mode", "True", "--haplotype_chunk_max_sites", "7"]
        if mode == "indel":
            cmd.append("--is_indel")
        results = set()
        for hashseed in ("0", "1", "2"):       # set / dict iteration order must not matter on this fixture
            res = subprocess.run(cmd, cwd=d, env=dict(os.environ, PYTHONPATH=REF, FAKE_REF=os.path.join(d, "ref.txt"), FAKE_MPILEUP_HAP=mp,
                                                      PYTHONHASHSEED=hashseed), capture_output=True, text=True)
            assert res.returncode == 0, res.stderr[-2000:]
            results.add(open(out_vcf).read())
        assert len(results) == 1, "reference output depends on the hash seed: pick another simulation seed"
        got = open(out_vcf).read()
        out["modes"][mode] = dict(pileup_vcf=open(pile_vcf).read(), mpileup=text, out_vcf=got, stdout=res.stdout)
        tags = {}
        for r in got.split("\n"):
            if r and not r.startswith("#"):
                for t in r.split("\t")[6].split(";"):
                    tags[t] = tags.get(t, 0) + 1
        print("hapfilter", mode, "calls", len(calls), "rows", text.count("\n"), "FILTER tags", tags,
              "H", sum(1 for r in got.split("\n") if "\tH;" in r))
    dump_json_gz("hapfilter.json.gz", out)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "hapfilter":
        with tempfile.TemporaryDirectory() as tmp:
            gen_hapfilter(tmp)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "post":
        with tempfile.TemporaryDirectory() as tmp:
            gen_post(tmp)
        return
    if len(sys.argv) > 1 and sys.argv[1] in ("branches", "region2k", "pickles"):
        with tempfile.TemporaryDirectory() as tmp:
            {"branches": gen_branches, "region2k": gen_region2k, "pickles": gen_pickles}[sys.argv[1]](tmp)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "extract":
        with tempfile.TemporaryDirectory() as tmp:
            gen_extract(tmp)
        return
    gen_columns()
    with tempfile.TemporaryDirectory() as tmp:
        region = gen_region(tmp)
        gen_models(region)
        gen_calls(tmp, region)
        gen_extract(tmp)
        gen_post(tmp)
        gen_branches(tmp)
        gen_region2k(tmp)
        gen_pickles(tmp)


if __name__ == "__main__":
    main()

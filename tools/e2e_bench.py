"""Wall time of one pileup_call invocation (BAM + BED -> VCF) by stage, on a synthetic long-read BAM.
python tools/e2e_bench.py [region_kb]"""
import cProfile
import os
import pstats
import sys
import tempfile
import time
from argparse import Namespace

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch
    from bamutil import write_bam
    from clairs_to_amd import nn_shims
    from clairs_to_amd.engine import synthetic_models
    from clairs_to_amd.pileup_call import make_engine, pileup_call
    from clairs_to_amd.synth import likelihood_table
    kb = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    L = kb * 1000
    rng = np.random.default_rng(1)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    ref = acgt[rng.integers(0, 4, size=L)]
    reads, bases, i = [], 0, 0
    while bases < 50 * L:
        n = int(np.clip(rng.lognormal(9.0, 0.5), 1000, 30000))
        pos = int(rng.integers(0, max(1, L - n)))
        n = min(n, L - pos)
        seg = ref[pos:pos + n].copy()
        mm = rng.random(n) < 0.02
        seg[mm] = acgt[rng.integers(0, 4, size=int(mm.sum()))]
        q = np.clip(np.rint(rng.normal(28, 8, size=n)), 1, 50).astype(np.uint8)
        reads.append(dict(name="r%d" % i, flag=16 * int(rng.random() < 0.5), ref=0, pos=pos, mapq=60, cigar=[("M", n)],
                          seq=seg.tobytes().decode(), qual=q.tolist()))
        bases += n
        i += 1
    reads.sort(key=lambda r: r["pos"])
    d = tempfile.mkdtemp()
    bam = os.path.join(d, "b.bam")
    write_bam(bam, [("chr1", L)], reads, block_payload=65000)
    refs = ref.tobytes().decode()
    open(os.path.join(d, "ref.fa"), "w").write(">chr1\n" + "\n".join(refs[k:k + 60] for k in range(0, L, 60)) + "\n")
    open(os.path.join(d, "ref.fa.fai"), "w").write("chr1\t%d\t6\t60\t61\n" % L)
    sites = list(range(1000, L - 1000, 250))
    bed = os.path.join(d, "chr1.1_1_snv")
    open(bed, "w").write("".join("chr1\t%d\t%d\n" % (x - 17, x + 17) for x in sites))
    models = synthetic_models(4)
    paths = {}
    nn_shims.install_reference_aliases()
    for key, tag in (("model_acgt", "aff"), ("model_nacgt", "neg")):
        paths[key] = os.path.join(d, key + ".pkl")
        torch.save({key: models[tag]}, paths[key])
    lik = os.path.join(d, "lik.txt")
    np.savetxt(lik, likelihood_table(4, seed=11), fmt="%.17g")
    args = Namespace(platform="ont", ref_fn=os.path.join(d, "ref.fa"), ctg_name="chr1", samtools="samtools", bam_reader="native",
                     tumor_bam_fn=bam, mpileup_fn=None, min_bq=None, max_depth=None, max_indel_length=None, candidates_bed_regions=bed,
                     chkpnt_fn_acgt=paths["model_acgt"], chkpnt_fn_nacgt=paths["model_nacgt"], min_rescale_cov=50,
                     disable_indel_calling=True, likelihood_matrix_data=lik, call_fn=os.path.join(d, "out.vcf"), predict_fn=None,
                     sample_name="SAMPLE", show_ref=True, qual=0, pileup=True)
    eng = make_engine(args)
    pileup_call(args, engine=eng)
    t0 = time.perf_counter()
    n = pileup_call(args, engine=eng)
    dt = time.perf_counter() - t0
    print("%d candidates -> %d VCF records in %.1f ms (%.0f sites/s per process), engine resident" % (len(sites), n, dt * 1e3, len(sites) / dt))
    # a run of several chunks through call_chunks (pack production of chunk i+1 overlapped with the GPU work of chunk i)
    from clairs_to_amd.call_chunks import call_chunks
    n_chunks = 6
    per = len(sites) // n_chunks
    names = []
    for c in range(n_chunks):
        fn = os.path.join(d, "chr1.%d_%d_snv" % (c + 1, n_chunks))
        open(fn, "w").write("".join("chr1\t%d\t%d\n" % (x - 17, x + 17) for x in sites[c * per:(c + 1) * per]))
        names.append(fn)
    open(os.path.join(d, "CANDIDATES_FILES"), "w").write("".join(n_ + "\n" for n_ in names))
    a2 = Namespace(**vars(args))
    a2.chunk_list, a2.output_dir = os.path.join(d, "CANDIDATES_FILES"), os.path.join(d, "vcf_output")
    a2.merged_vcf_fn, a2.final_vcf_fn = os.path.join(d, "merged.vcf"), None
    import clairs_to_amd.call_chunks as cc
    orig = cc.make_engine
    cc.make_engine = lambda *_a, **_k: eng               # keep the resident engine: measure the steady state, not model loading
    t0 = time.perf_counter()
    call_chunks(a2)
    dt = time.perf_counter() - t0
    cc.make_engine = orig
    print("call_chunks: %d chunks x %d candidates in %.1f ms (%.0f sites/s per process, merged VCF included)" % (
        n_chunks, per, dt * 1e3, n_chunks * per / dt))
    pr = cProfile.Profile()
    pr.enable()
    pileup_call(args, engine=eng)
    pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(14)


if __name__ == "__main__":
    main()

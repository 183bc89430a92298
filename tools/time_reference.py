#!/usr/bin/env python3
"""CPU timing of the GENUINE reference (HKU-BAL/ClairS-TO at /root/reference) on the hot path - build container only.

SURVEY.md 8(d) / BASELINE.md: N candidates of the configs[1] generator (ONT 50x synthetic pileups, SNV) through the reference's
own four commands per chunk, exactly as run_clairs_to chains them (run_clairs_to:1228-1308):
    create_tensor_pileup_calling --min_bq 20   (AFF)      \\
    create_tensor_pileup_calling --min_bq 0    (NEG)       |  one process each, gzip text between them,
    predict --pileup --disable_indel_calling True          |  torch CPU with torch.set_num_threads(1) (predict.py:475)
    call_variants                                         /
with `samtools` replaced by a shim that prints prepared mpileup text / FASTA (BAM decoding by samtools is therefore EXCLUDED),
CPython where the reference runs tensor creation under pypy3, seeded random-init weights of the predict.py architecture.
Measured with 1 process and with P concurrent processes over P chunks (GNU parallel -j P in the reference).

Writes profiles/reference_cpu_timing.json and prints the table for BASELINE.md.  The reference cannot travel to the GPU box;
bench.py quotes the figure written here as a static note (`cpu_reference_python`) next to its measured `cpu_baseline`.
Usage: python tools/time_reference.py [--sites 10000] [--procs 1,8]"""
import argparse
import gzip
import json
import os
import stat
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

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


def prepare_chunk(d, n_sites, seed):
    import numpy as np
    from clairs_to_amd.synth import SynthChunk, mpileup_text
    os.makedirs(d, exist_ok=True)
    ch = SynthChunk(n_sites, seed=seed)
    ref, lo = ch.ref_window()
    full = "A" * (lo - 1) + ref
    open(os.path.join(d, "ref.fa"), "w").write(">chr1\n" + full + "\n")
    open(os.path.join(d, "ref.fa.fai"), "w").write("chr1\t%d\t6\t%d\t%d\n" % (len(full), len(full), len(full) + 1))
    open(os.path.join(d, "ref.txt"), "w").write(full)
    shim = os.path.join(d, "samtools")
    open(shim, "w").write(SHIM)
    os.chmod(shim, os.stat(shim).st_mode | stat.S_IEXEC)
    open(os.path.join(d, "cand.bed"), "w").write("".join("chr1\t%d\t%d\n" % (x - 17, x + 17) for x in ch.site_pos.tolist()))
    for q in (0, 20):
        open(os.path.join(d, "mp_%d.txt" % q), "w").write(mpileup_text(ch, q))
    return ch.n_sites


def chunk_commands(d, ckpt, lik):
    py = sys.executable
    c2 = os.path.join(REF, "clairs_to.py")
    ct = lambda q, out: [py, c2, "create_tensor_pileup_calling", "--tumor_bam_fn", "fake.bam", "--ref_fn", os.path.join(d, "ref.fa"),
                         "--ctg_name", "chr1", "--min_bq", str(q), "--samtools", os.path.join(d, "samtools"),
                         "--candidates_bed_regions", os.path.join(d, "cand.bed"), "--tensor_can_fn", out, "--platform", "ont"]
    return [("create_tensor_aff", ct(20, os.path.join(d, "t_aff.gz"))), ("create_tensor_neg", ct(0, os.path.join(d, "t_neg.gz"))),
            ("predict", [py, c2, "predict", "--tensor_fn_acgt", os.path.join(d, "t_aff.gz"), "--tensor_fn_nacgt", os.path.join(d, "t_neg.gz"),
                         "--chkpnt_fn_acgt", ckpt[0], "--chkpnt_fn_nacgt", ckpt[1], "--predict_fn", os.path.join(d, "pred.gz"), "--pileup",
                         "--disable_indel_calling", "True", "--ctg_name", "chr1"]),
            ("call_variants", [py, c2, "call_variants", "--predict_fn", os.path.join(d, "pred.gz"), "--call_fn", os.path.join(d, "out.vcf"),
                               "--likelihood_matrix_data", lik, "--disable_indel_calling", "True", "--ctg_name", "chr1", "--pileup"])]


def env_of(d):
    return dict(os.environ, PYTHONPATH=REF, FAKE_REF=os.path.join(d, "ref.txt"), FAKE_MPILEUP_0=os.path.join(d, "mp_0.txt"),
                FAKE_MPILEUP_20=os.path.join(d, "mp_20.txt"), OMP_NUM_THREADS="1", MKL_NUM_THREADS="1")


def run_chain(d, ckpt, lik):
    """the four commands of one chunk, sequentially; returns {stage: seconds}"""
    out = {}
    for name, cmd in chunk_commands(d, ckpt, lik):
        t0 = time.perf_counter()
        subprocess.run(cmd, cwd=d, env=env_of(d), check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        out[name] = time.perf_counter() - t0
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sites", type=int, default=10000)
    ap.add_argument("--procs", default="1,8")
    a = ap.parse_args()
    if not os.path.isdir(REF):
        sys.exit("tools/time_reference.py runs in the build container only (%s is absent)" % REF)
    import numpy as np
    import torch
    sys.path.insert(0, REF)
    import clairs.model as rm
    from weights_recipe import make_weights, CVT_CFG
    from clairs_to_amd.synth import likelihood_table
    procs = [int(x) for x in a.procs.split(",")]
    pmax = max(procs)
    per = a.sites // pmax
    tmp = tempfile.mkdtemp(prefix="cto_ref_timing_")
    # checkpoints: the reference's own classes, predict.py architecture, seeded weights
    ckpt = []
    for key, m in (("model_acgt", rm.CvT(num_classes=2, s1_emb_dim=16, s2_emb_dim=64, s3_emb_dim=128, s1_heads=1, s2_heads=3, s3_heads=4,
                                         s1_depth=1, s2_depth=2, s3_depth=3, apply_softmax=False, model_type="acgt")),
                   ("model_nacgt", rm.BiGRU_NACGT(apply_softmax=False, num_classes=2, model_type="nacgt"))):
        manifest = [(k, list(v.shape)) for k, v in m.state_dict().items() if not k.endswith("num_batches_tracked")]
        sd = m.state_dict()
        for k, v in make_weights(manifest, seed=4).items():
            sd[k] = torch.from_numpy(v.copy())
        m.load_state_dict(sd)
        fn = os.path.join(tmp, key + ".pkl")
        torch.save({key: m.eval()}, fn)
        ckpt.append(fn)
    lik = os.path.join(tmp, "lik.txt")
    np.savetxt(lik, likelihood_table(4, seed=11), fmt="%.17g")
    dirs = [os.path.join(tmp, "chunk%d" % i) for i in range(pmax)]
    n_per = [prepare_chunk(d, per, 20260928 + i) for i, d in enumerate(dirs)]
    res = {"sites_per_chunk": per, "host_cpus": os.cpu_count(), "python": sys.version.split()[0], "torch": torch.__version__,
           "note": "reference v0.4.4 run from /root/reference; samtools replaced by a text shim (BAM decoding excluded); CPython "
                   "(the reference runs tensor creation under pypy3); torch CPU, 1 thread per process (predict.py:475)", "runs": []}
    for p in procs:
        t0 = time.perf_counter()
        if p == 1:
            stages = [run_chain(dirs[0], ckpt, lik)]
            n = n_per[0]
        else:
            import concurrent.futures as cf
            with cf.ThreadPoolExecutor(max_workers=p) as ex:
                stages = list(ex.map(lambda d: run_chain(d, ckpt, lik), dirs[:p]))
            n = sum(n_per[:p])
        wall = time.perf_counter() - t0
        agg = {k: round(sum(s[k] for s in stages) / len(stages), 2) for k in stages[0]}
        r = dict(processes=p, sites=n, wall_s=round(wall, 2), sites_per_s=round(n / wall, 1), sites_per_s_per_process=round(n / wall / p, 1),
                 mean_stage_seconds_per_chunk=agg)
        res["runs"].append(r)
        print(json.dumps(r), flush=True)
    n_vcf = sum(1 for r in open(os.path.join(dirs[0], "out.vcf")) if not r.startswith("#")) if os.path.exists(os.path.join(dirs[0], "out.vcf")) else 0
    res["vcf_records_chunk0"] = n_vcf
    os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
    with open(os.path.join(ROOT, "profiles", "reference_cpu_timing.json"), "w") as f:
        json.dump(res, f, indent=1)
    print("wrote profiles/reference_cpu_timing.json")


if __name__ == "__main__":
    main()

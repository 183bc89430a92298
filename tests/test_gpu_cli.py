"""GPU tests of the three drop-in sub-modules at their text seams, against the reference's own files."""
import gzip
import os

import numpy as np
import pytest

from conftest import load_json_gz, load_models_npz
from weights_recipe import make_weights

pytestmark = pytest.mark.gpu


def _write_region(tmp_path, g):
    ref = g["ref"]
    fa = tmp_path / "ref.fa"
    fa.write_text(">chr1\n" + "\n".join(ref[i:i + 60] for i in range(0, len(ref), 60)) + "\n")
    (tmp_path / "ref.fa.fai").write_text("chr1\t%d\t6\t60\t61\n" % len(ref))
    bed = tmp_path / "cand.bed"
    bed.write_text("".join("chr1\t%d\t%d\n" % (x - 17, x + 17) for x in g["sites"]))
    mp = tmp_path / "mp.txt"
    mp.write_text(g["mpileup_neg"])
    return str(fa), str(bed), str(mp)


def _pickle_models(tmp_path, aff_cls, neg_cls, K):
    """checkpoints pickled under the reference's qualified names (clairs.model.<cls>), as its releases are"""
    import torch
    from clairs_to_amd import nn_shims
    nn_shims.install_reference_aliases()
    paths = {}
    for key, cls in (("model_acgt", aff_cls), ("model_nacgt", neg_cls)):
        g = load_models_npz(cls)
        m = nn_shims.from_state_dict(cls, make_weights(g["manifest"], seed=K))
        saved = {}
        try:
            for c in vars(nn_shims).values():
                if isinstance(c, type) and c.__module__ == nn_shims.__name__:
                    saved[c] = c.__module__
                    c.__module__ = "clairs.model"
            paths[key] = str(tmp_path / (key + ".pkl"))
            torch.save({key: m}, paths[key])
        finally:
            for c, mod in saved.items():
                c.__module__ = mod
    return paths


def test_create_tensor_cli_matches_reference_text(tmp_path, golden_region):
    from argparse import Namespace
    from clairs_to_amd.create_tensor_pileup_calling import create_tensor
    fa, bed, mp = _write_region(tmp_path, golden_region)
    aff, neg = str(tmp_path / "aff.gz"), str(tmp_path / "neg.gz")
    args = Namespace(candidates_bed_regions=bed, ctg_name="chr1", ref_fn=fa, mpileup_fn=mp, samtools="samtools",
                     tumor_bam_fn=None, max_depth=None, max_indel_length=None, min_bq=20, tensor_can_fn=aff,
                     tensor_can_fn_neg=neg, platform="ont")
    create_tensor(args)
    assert gzip.open(aff, "rt").read() == golden_region["tensor_aff"]
    assert gzip.open(neg, "rt").read() == golden_region["tensor_neg"]


@pytest.mark.parametrize("mode,aff_cls,neg_cls", [("snv", "CvT", "BiGRU_NACGT"), ("indel", "CvT_Indel", "BiGRU_NACGT_Indel")])
def test_predict_and_call_variants_cli(tmp_path, golden_region, mode, aff_cls, neg_cls):
    import torch
    from argparse import Namespace
    from clairs_to_amd import nn_shims
    from clairs_to_amd.predict import predict
    from clairs_to_amd.call_variants import call_variants_from_probability
    calls = load_json_gz("calls_%s.json.gz" % mode)
    K = calls["n_out"]
    for tag in ("aff", "neg"):
        with gzip.open(tmp_path / ("t_%s.gz" % tag), "wt") as f:
            f.write(golden_region["tensor_" + tag])
    paths = _pickle_models(tmp_path, aff_cls, neg_cls, K)
    pred = str(tmp_path / "pred.gz")
    args = Namespace(tensor_fn_acgt=str(tmp_path / "t_aff.gz"), tensor_fn_nacgt=str(tmp_path / "t_neg.gz"),
                     chkpnt_fn_acgt=paths["model_acgt"], chkpnt_fn_nacgt=paths["model_nacgt"], predict_fn=pred,
                     ctg_name="chr1", min_rescale_cov=50, disable_indel_calling=(mode == "snv"), use_gpu=True, pileup=True)
    n = predict(args)
    got = [r.split("\t") for r in gzip.open(pred, "rt").read().split("\n") if r]
    want = [r.split("\t") for r in calls["predict_rows"].split("\n") if r]
    assert n == len(want) == len(got)
    for a, b in zip(got, want):
        assert a[:6] == b[:6] and len(a) == len(b)
        pa = np.array([[float(v) for v in f.split()] for f in a[6:6 + 2 * K]])
        pb = np.array([[float(v) for v in f.split()] for f in b[6:6 + 2 * K]])
        assert np.abs(pa - pb).max() < 1e-4          # north_star tolerance on probabilities
    # call_variants on the REFERENCE's probability rows: VCF records must be byte-identical
    ref_pred = str(tmp_path / "ref_pred.gz")
    with gzip.open(ref_pred, "wt") as f:
        f.write(calls["predict_rows"])
    lik = str(tmp_path / "lik.txt")
    open(lik, "w").write(calls["likelihood_table"])
    for show_ref in (False, True):
        vcf = str(tmp_path / ("out_%d.vcf" % show_ref))
        call_variants_from_probability(Namespace(call_fn=vcf, predict_fn=ref_pred, likelihood_matrix_data=lik, ctg_name="chr1",
                                                 sample_name="SAMPLE", qual=0, show_ref=show_ref,
                                                 disable_indel_calling=(mode == "snv"), pileup=True, platform="ont"))
        rows = [r for r in open(vcf).read().split("\n") if r and not r.startswith("#")]
        assert rows == calls["vcf"]["show_ref" if show_ref else "default"]


@pytest.mark.parametrize("mode,aff_cls,neg_cls", [("snv", "CvT", "BiGRU_NACGT"), ("indel", "CvT_Indel", "BiGRU_NACGT_Indel")])
def test_pileup_call_one_invocation(tmp_path, golden_region, mode, aff_cls, neg_cls):
    """The whole-chunk driver (BED + mpileup text + pickled checkpoints + likelihood table -> VCF, nothing but HBM in
    between) against the VCF the REFERENCE wrote through its four commands: same records, field for field; QUAL / GQ may
    move in the last digit because the probabilities differ by < 1e-4 before the 8-decimal rounding."""
    from argparse import Namespace
    from clairs_to_amd.pileup_call import pileup_call
    calls = load_json_gz("calls_%s.json.gz" % mode)
    K = calls["n_out"]
    fa, bed, mp = _write_region(tmp_path, golden_region)
    paths = _pickle_models(tmp_path, aff_cls, neg_cls, K)
    lik = str(tmp_path / "lik.txt")
    open(lik, "w").write(calls["likelihood_table"])
    for show_ref in (False, True):
        vcf, pred = str(tmp_path / ("one_%d.vcf" % show_ref)), str(tmp_path / ("pred_%d.gz" % show_ref))
        n = pileup_call(Namespace(platform="ont", tumor_bam_fn=None, mpileup_fn=mp, ref_fn=fa, ctg_name="chr1", samtools="samtools",
                                  min_bq=golden_region["min_bq_aff"], max_depth=None, max_indel_length=None,
                                  candidates_bed_regions=bed, chkpnt_fn_acgt=paths["model_acgt"],
                                  chkpnt_fn_nacgt=paths["model_nacgt"], min_rescale_cov=50, disable_indel_calling=(mode == "snv"),
                                  likelihood_matrix_data=lik, call_fn=vcf, predict_fn=pred, sample_name="SAMPLE",
                                  show_ref=show_ref, qual=0, pileup=True))
        want = calls["vcf"]["show_ref" if show_ref else "default"]
        got = [r for r in open(vcf).read().split("\n") if r and not r.startswith("#")] if n else []
        assert n == len(want) == len(got)
        for a, b in zip(got, want):
            a, b = a.split("\t"), b.split("\t")
            assert a[:5] == b[:5] and a[6:9] == b[6:9]
            assert abs(float(a[5]) - float(b[5])) < 0.02
            fa_, fb_ = a[9].split(":"), b[9].split(":")
            assert fa_[0] == fb_[0] and fa_[2:] == fb_[2:] and abs(int(fa_[1]) - int(fb_[1])) <= 1
        # the debugging tap writes the same rows as the predict mirror / the reference (non-probability fields identical)
        rows = [r.split("\t") for r in gzip.open(pred, "rt").read().split("\n") if r]
        ref_rows = [r.split("\t") for r in calls["predict_rows"].split("\n") if r]
        assert [r[:6] for r in rows] == [r[:6] for r in ref_rows]


@pytest.mark.parametrize("launcher", ["torchrun", "self"])
def test_bench_multi_rank_code_path(tmp_path, launcher):
    """bench.py's N > 1 control flow (rank-sharded chunks, asynchronous double-buffered all_gather of the per-site outputs,
    barrier + max-over-ranks timing, one JSON line from rank 0), exercised with two ranks on this box's single GPU over gloo
    (test hooks CTO_BENCH_BACKEND / CTO_BENCH_DEVICE); the driver's real runs use RCCL with one GPU per rank.
    launcher = "self": `python bench.py --gpus 2` with no launcher and no WORLD_SIZE must start the two ranks itself."""
    import json
    import subprocess
    import sys
    from conftest import ROOT
    env = dict(os.environ, CTO_BENCH_BACKEND="gloo", CTO_BENCH_DEVICE="0", MASTER_ADDR="127.0.0.1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    tail = [os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--pool", "2"]
    if launcher == "torchrun":
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
               "127.0.0.1", "--master-port", "29533"] + tail
    else:
        cmd = [sys.executable] + tail
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.split("\n") if l.startswith("{")]
    assert len(lines) == 1
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["steps"] == 3 and res["value"] > 0 and res["scaling"] == "weak"
    assert res["roofline"]["frac"] > 0 and "cpu_baseline" not in res
    assert res["ranks_seen"] == 2 and res["gather_verified"] is True and res["backend"].startswith("gloo")
    assert [d["rank"] for d in res["rank_devices"]] == [0, 1]
    # round 6: every N's line carries the 245-step job of every rank, the gather of each step included
    assert res["sustained"]["steps"] == 245 and res["sustained"]["sites"] == 2 * 245 * res["config"]["batch"] and res["sustained"]["sites_per_s"] > 0


def test_bench_eight_rank_dry_run(tmp_path):
    """The driver's first SCALE run must not die on set-up: `python bench.py --gpus 8` with its DEFAULT geometry (16 resident
    4096-site chunks per rank, the default steps and warm-up) as eight ranks on this box's single GPU over gloo (test hooks) - eight times the
    resident pool and workspaces on one device (a real rank has 288 GB to itself), the eight-way synthesis on the host's cores, the
    rank census, the verified gather, the watchdog armed.  Start-up time is asserted: everything before the JSON line inside 10 min."""
    import json
    import subprocess
    import sys
    import time
    from conftest import ROOT
    env = dict(os.environ, CTO_BENCH_BACKEND="gloo", CTO_BENCH_DEVICE="0", MASTER_ADDR="127.0.0.1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--watchdog", "560"], env=env, cwd=ROOT,
                       capture_output=True, text=True, timeout=600)
    took = time.time() - t0
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.split("\n") if l.startswith("{")]
    assert len(lines) == 1
    res = json.loads(lines[0])
    assert res["n_gpus"] == 8 and res["steps"] == 100 and res["warmup"] == 100 and res["value"] > 0 and res["scaling"] == "weak"
    assert res["ranks_seen"] == 8 and res["gather_verified"] is True and res["backend"].startswith("gloo")
    assert [d["rank"] for d in res["rank_devices"]] == list(range(8))
    assert res["config"]["chunks_resident_per_gpu"] == 16 and res["config"]["batch"] == 4096
    assert "cpu_baseline" not in res and "e2e" not in res and "configs" not in res      # N > 1: the line carries the scaling run only
    print("8-rank dry run: %.0f s wall, %.1f sites/s aggregate on ONE device (not a scaling figure)" % (took, res["value"]))


def test_bench_two_ranks_over_rccl(tmp_path):
    """First contact with RCCL, automatically, wherever >= 2 GPUs are visible (the build's own boxes have one: skipped there):
    `bench.py --gpus 2` must come back with one JSON line whose exchange step ran over nccl (= RCCL) between two distinct devices.
    A hang ends in bench.py's own watchdog message (stderr is shown), not in a silent time-out."""
    import json
    import subprocess
    import sys
    import torch
    from conftest import ROOT
    if torch.cuda.device_count() < 2:
        pytest.skip("needs >= 2 GPUs (RCCL over xGMI)")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "CTO_BENCH_BACKEND", "CTO_BENCH_DEVICE"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--pool", "2",
                        "--watchdog", "240"], env=env, cwd=ROOT, capture_output=True, text=True, timeout=400)
    assert r.returncode == 0, r.stderr[-3000:]
    res = json.loads([l for l in r.stdout.split("\n") if l.startswith("{")][-1])
    assert res["n_gpus"] == 2 and res["ranks_seen"] == 2 and res["gather_verified"] is True
    assert res["backend"].startswith("nccl"), res["backend"]
    pci = [d["pci"] for d in res["rank_devices"]]
    assert len(set(pci)) == 2 and [d["device"] for d in res["rank_devices"]] == ["cuda:0", "cuda:1"]


def test_call_chunks_two_ranks_over_rccl(tmp_path):
    """call_chunks on two GPUs (one rank each under torch.distributed.run; its control plane is gloo by design - chunk files shard
    with no data-path collective): same merged records as one rank; skipped on a one-GPU box (where
    test_call_chunks_matches_single_call[2] runs the same control flow on one device)."""
    import subprocess
    import sys
    import torch
    from conftest import ROOT
    from clairs_to_amd.synth import likelihood_table
    if torch.cuda.device_count() < 2:
        pytest.skip("needs >= 2 GPUs (RCCL over xGMI)")
    sc = _bam_scenario(tmp_path)
    paths = _pickle_models(tmp_path, "CvT", "BiGRU_NACGT", 4)
    lik = tmp_path / "lik.txt"
    np.savetxt(lik, likelihood_table(4, seed=11), fmt="%.17g")
    sites = sc["sites"]
    cdir = tmp_path / "candidates"
    cdir.mkdir()
    names = []
    for i in range(4):
        part = sites[i * len(sites) // 4:(i + 1) * len(sites) // 4]
        fn = cdir / ("chr1.%d_4_snv" % (i + 1))
        fn.write_text("".join("chr1\t%d\t%d\n" % (x - 17, x + 17) for x in part))
        names.append(str(fn))
    (tmp_path / "CANDIDATES_FILES").write_text("".join(n + "\n" for n in names))
    common = ["--platform", "ont", "--tumor_bam_fn", sc["bam"], "--ref_fn", sc["fa"], "--bam_reader", "native", "--chkpnt_fn_acgt",
              paths["model_acgt"], "--chkpnt_fn_nacgt", paths["model_nacgt"], "--disable_indel_calling", "True",
              "--likelihood_matrix_data", str(lik), "--show_ref"]
    outs = {}
    for world in (1, 2):
        out_dir, merged = tmp_path / ("vcf_%d" % world), tmp_path / ("merged_%d.vcf" % world)
        cmd = ["-m", "clairs_to_amd", "call_chunks", "--chunk_list", str(tmp_path / "CANDIDATES_FILES"), "--output_dir", str(out_dir),
               "--merged_vcf_fn", str(merged)] + common
        full = [sys.executable] + cmd if world == 1 else [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                                                        "--master-addr", "127.0.0.1", "--master-port", "29546"] + cmd
        env = dict(os.environ, PYTHONPATH=ROOT, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        r = subprocess.run(full, cwd=ROOT, capture_output=True, text=True, timeout=400, env=env)
        assert r.returncode == 0, r.stderr[-3000:]
        outs[world] = [l for l in open(merged).read().split("\n") if l and not l.startswith("#")]
    assert outs[1] == outs[2] and len(outs[1]) > 50
    # the exchange step in the data path: --gather_outputs all_gathers every rank's per-site outputs as DEVICE tensors - the process
    # group is "cpu:gloo,cuda:nccl", i.e. this is RCCL over xGMI - and rank 0 writes the merged VCF from the gathered buffer
    out_dir, merged = tmp_path / "vcf_gather", tmp_path / "merged_gather.vcf"
    cmd = ["-m", "clairs_to_amd", "call_chunks", "--chunk_list", str(tmp_path / "CANDIDATES_FILES"), "--output_dir", str(out_dir),
           "--merged_vcf_fn", str(merged), "--gather_outputs"] + common
    env = dict(os.environ, PYTHONPATH=ROOT, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    env.pop("CTO_GATHER_BACKEND", None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29547"] + cmd, cwd=ROOT, capture_output=True, text=True, timeout=400, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    assert [l for l in open(merged).read().split("\n") if l and not l.startswith("#")] == outs[1]
    assert "from 2 rank(s)" in r.stderr


def _bam_scenario(tmp_path):
    """A 6 kb contig, 700 synthetic long reads with substitutions and indels (BAM + BAI), candidates every 37 bp, the naive
    pileup text of their windows and a `samtools` shim that prints it."""
    import stat
    from argparse import Namespace
    from bamutil import write_bam, mpileup_rows
    from clairs_to_amd.pileup_call import pileup_call
    from clairs_to_amd.synth import likelihood_table
    rng = np.random.default_rng(17)
    L = 6000
    ref = "".join(rng.choice(list("ACGT"), size=L))
    reads = []
    for i in range(700):
        pos = int(rng.integers(0, L - 700))
        n = int(rng.integers(300, 650))
        seq, cigar, q, rp = [], [], [], pos
        while n > 0:
            m = int(min(n, rng.integers(20, 120)))
            seg = list(ref[rp:rp + m])
            for k in range(m):
                if rng.random() < 0.03:
                    seg[k] = "ACGT"[(("ACGT".index(seg[k])) + int(rng.integers(1, 4))) % 4]
            seq += seg
            cigar.append(("M", m))
            rp += m
            n -= m
            if n > 0:
                u = rng.random()
                if u < 0.3:
                    k = int(rng.integers(1, 5))
                    seq += list(rng.choice(list("ACGT"), size=k))
                    cigar.append(("I", k))
                elif u < 0.6:
                    k = int(rng.integers(1, 5))
                    cigar.append(("D", k))
                    rp += k
        reads.append(dict(name="r%d" % i, flag=16 * int(rng.random() < 0.5), ref=0, pos=pos, mapq=int(rng.choice([60, 60, 60, 10])),
                          cigar=cigar, seq="".join(seq), qual=[int(v) for v in np.clip(rng.normal(28, 8, size=len(seq)), 1, 50)]))
    reads.sort(key=lambda r: r["pos"])
    bam = str(tmp_path / "t.bam")
    write_bam(bam, [("chr1", L)], reads)
    import pickle
    pickle.dump(reads, open(tmp_path / "reads.pkl", "wb"))
    fa = tmp_path / "ref.fa"
    fa.write_text(">chr1\n" + "\n".join(ref[i:i + 60] for i in range(0, L, 60)) + "\n")
    (tmp_path / "ref.fa.fai").write_text("chr1\t%d\t6\t60\t61\n" % L)
    sites = list(range(900, 5100, 37))
    bed = tmp_path / "cand.bed"
    bed.write_text("".join("chr1\t%d\t%d\n" % (x - 17, x + 17) for x in sites))
    ext_s, ext_e = min(sites) - 16 - 33, max(sites) + 18 + 33
    text = mpileup_rows(reads, 0, "chr1", ext_s, ext_e, bed=[(x - 17, x + 17) for x in sites])
    mp = tmp_path / "mp.txt"
    mp.write_text(text)
    shim = tmp_path / "samtools"
    shim.write_text("#!/bin/sh\n# stands in for `samtools mpileup ...` (absent here): prints the prepared pileup\ncat %s\n" % mp)
    shim.chmod(shim.stat().st_mode | stat.S_IEXEC)
    return dict(bam=bam, fa=str(fa), bed=str(bed), mp=str(mp), shim=str(shim), sites=sites, L=L)


def test_pileup_call_bam_readers_agree(tmp_path):
    """One chunk called four ways - native BAM + BAI reader with the BGZF blocks inflated on the host and on the device,
    `samtools mpileup` subprocess (a shim that prints the naive pileup of the same BAM, as the real tool is absent here) and
    pre-made mpileup text - must give the same VCF bytes."""
    import numpy as np
    from argparse import Namespace
    from clairs_to_amd.pileup_call import pileup_call
    from clairs_to_amd.synth import likelihood_table
    sc = _bam_scenario(tmp_path)
    bam, fa, bed, mp, shim = sc["bam"], sc["fa"], sc["bed"], sc["mp"], sc["shim"]
    paths = _pickle_models(tmp_path, "CvT", "BiGRU_NACGT", 4)
    lik = tmp_path / "lik.txt"
    np.savetxt(lik, likelihood_table(4, seed=11), fmt="%.17g")
    out = {}
    for tag, kw in (("native", dict(bam_reader="native", tumor_bam_fn=bam, mpileup_fn=None)),
                    ("gpu", dict(bam_reader="gpu", tumor_bam_fn=bam, mpileup_fn=None)),        # BGZF blocks inflated on the device
                    ("shim", dict(bam_reader="samtools", tumor_bam_fn=bam, mpileup_fn=None)),
                    ("text", dict(bam_reader="samtools", tumor_bam_fn=None, mpileup_fn=mp))):
        vcf = str(tmp_path / (tag + ".vcf"))
        n = pileup_call(Namespace(platform="ont", ref_fn=fa, ctg_name="chr1", samtools=shim, min_bq=None, max_depth=None,
                                  max_indel_length=None, candidates_bed_regions=bed, chkpnt_fn_acgt=paths["model_acgt"],
                                  chkpnt_fn_nacgt=paths["model_nacgt"], min_rescale_cov=50, disable_indel_calling=True,
                                  likelihood_matrix_data=str(lik), call_fn=vcf, predict_fn=None, sample_name="SAMPLE", show_ref=True,
                                  qual=0, pileup=True, **kw))
        assert n > 50
        out[tag] = open(vcf).read()
    assert out["native"] == out["text"] == out["shim"] == out["gpu"]


@pytest.mark.parametrize("world", [1, 2])
def test_call_chunks_matches_single_call(tmp_path, world):
    """The chunk-list driver (one process per GPU; here 1 process, then 2 ranks sharing this box's GPU) writes the same
    records, chunk by chunk and merged, as one pileup_call over all candidates."""
    import subprocess
    import sys
    from argparse import Namespace
    from conftest import ROOT
    from clairs_to_amd.pileup_call import pileup_call
    from clairs_to_amd.synth import likelihood_table
    sc = _bam_scenario(tmp_path)
    paths = _pickle_models(tmp_path, "CvT", "BiGRU_NACGT", 4)
    lik = tmp_path / "lik.txt"
    np.savetxt(lik, likelihood_table(4, seed=11), fmt="%.17g")
    sites = sc["sites"]
    cdir = tmp_path / "candidates"
    cdir.mkdir()
    names = []
    for i in range(3):
        part = sites[i * len(sites) // 3:(i + 1) * len(sites) // 3]
        fn = cdir / ("chr1.%d_3_snv" % (i + 1))
        fn.write_text("".join("chr1\t%d\t%d\n" % (x - 17, x + 17) for x in part))
        names.append(str(fn))
    (tmp_path / "CANDIDATES_FILES").write_text("".join(n + "\n" for n in names))
    common = ["--platform", "ont", "--tumor_bam_fn", sc["bam"], "--ref_fn", sc["fa"], "--bam_reader", "native", "--chkpnt_fn_acgt",
              paths["model_acgt"], "--chkpnt_fn_nacgt", paths["model_nacgt"], "--disable_indel_calling", "True",
              "--likelihood_matrix_data", str(lik), "--show_ref"]
    out_dir, merged, final = tmp_path / "vcf_output", tmp_path / "merged.vcf", tmp_path / "final.vcf"
    cmd = ["-m", "clairs_to_amd", "call_chunks", "--chunk_list", str(tmp_path / "CANDIDATES_FILES"), "--output_dir", str(out_dir),
           "--merged_vcf_fn", str(merged), "--final_vcf_fn", str(final)] + common
    if world == 1:
        full = [sys.executable] + cmd
    else:
        full = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                "--master-port", "29544"] + cmd
    r = subprocess.run(full, cwd=ROOT, capture_output=True, text=True, timeout=280, env=dict(os.environ, PYTHONPATH=ROOT))
    assert r.returncode == 0, r.stderr[-3000:]
    one = tmp_path / "one.vcf"
    pileup_call(Namespace(platform="ont", ref_fn=sc["fa"], ctg_name="chr1", samtools="samtools", bam_reader="native", tumor_bam_fn=sc["bam"],
                          mpileup_fn=None, min_bq=None, max_depth=None, max_indel_length=None, candidates_bed_regions=sc["bed"],
                          chkpnt_fn_acgt=paths["model_acgt"], chkpnt_fn_nacgt=paths["model_nacgt"], min_rescale_cov=50,
                          disable_indel_calling=True, likelihood_matrix_data=str(lik), call_fn=str(one), predict_fn=None,
                          sample_name="SAMPLE", show_ref=True, qual=0, pileup=True))
    rec = lambda fn: [l for l in open(fn).read().split("\n") if l and not l.startswith("#")]
    assert rec(merged) == rec(one) and len(rec(one)) > 50
    assert sorted(os.listdir(out_dir)) == ["p_chr1.%d_3_snv.vcf" % (i + 1) for i in range(3)]
    assert 0 < len(rec(final)) <= len(rec(merged))      # postprocess drops PASS records under the platform's AF cut-off


@pytest.mark.parametrize("world", [1, 2])
def test_call_chunks_gather_outputs_writes_the_merged_vcf_of_the_file_merge(tmp_path, world):
    """call_chunks --gather_outputs: the per-site outputs of every rank all_gathered rank-major (dist.gather_site_rows) and the merged
    VCF written by rank 0 from the gathered buffer - byte-identical to the merge of the p_<chunk>.vcf files (the reference's way,
    run_clairs_to:1293-1317, and this driver's default).  Two ranks share this box's GPU and exchange over gloo (CTO_GATHER_BACKEND:
    on a multi-GPU box the same call goes over RCCL, test_call_chunks_two_ranks_over_rccl)."""
    import subprocess
    import sys
    from conftest import ROOT
    from clairs_to_amd.synth import likelihood_table
    sc = _bam_scenario(tmp_path)
    paths = _pickle_models(tmp_path, "CvT_Indel", "BiGRU_NACGT_Indel", 6)
    lik = tmp_path / "lik.txt"
    np.savetxt(lik, likelihood_table(6, seed=11), fmt="%.17g")
    sites = sc["sites"]
    cdir = tmp_path / "candidates"
    cdir.mkdir()
    names = []
    cuts = [0, len(sites) // 7, len(sites) // 2, len(sites) // 2, len(sites)]             # uneven chunks, one of them empty
    for i in range(4):
        fn = cdir / ("chr1.%d_4_indel" % (i + 1))
        fn.write_text("".join("chr1\t%d\t%d\n" % (x - 17, x + 17) for x in sites[cuts[i]:cuts[i + 1]]))
        names.append(str(fn))
    (tmp_path / "CANDIDATES_FILES").write_text("".join(n + "\n" for n in names))
    common = ["--platform", "ont", "--tumor_bam_fn", sc["bam"], "--ref_fn", sc["fa"], "--bam_reader", "native", "--chkpnt_fn_acgt",
              paths["model_acgt"], "--chkpnt_fn_nacgt", paths["model_nacgt"], "--disable_indel_calling", "False",
              "--likelihood_matrix_data", str(lik), "--show_ref"]

    def run(tag, extra):
        out_dir, merged = tmp_path / ("vcf_" + tag), tmp_path / ("merged_%s.vcf" % tag)
        cmd = ["-m", "clairs_to_amd", "call_chunks", "--chunk_list", str(tmp_path / "CANDIDATES_FILES"), "--output_dir", str(out_dir),
               "--merged_vcf_fn", str(merged)] + common + extra
        full = [sys.executable] + cmd if world == 1 else \
            [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
             "--master-port", "29546"] + cmd
        r = subprocess.run(full, cwd=ROOT, capture_output=True, text=True, timeout=280, env=dict(os.environ, PYTHONPATH=ROOT, CTO_GATHER_BACKEND="gloo"))
        assert r.returncode == 0, r.stderr[-3000:]
        return open(merged).read(), r.stderr
    files, _ = run("files", ["--pipeline", "python"])
    probs_fn = tmp_path / "gathered_probs.npy"
    gathered, err = run("gather", ["--gather_outputs", "--gathered_probs_fn", str(probs_fn)])
    assert gathered == files and gathered.count("\n") > 60
    assert "gathered the outputs of %d sites from %d rank(s)" % (len(sites), world) in err
    p = np.load(probs_fn)
    assert p.shape == (len(sites), 12, 2) and np.allclose(p.sum(axis=2), 1.0, atol=1e-6)
    # the C pipeline keeps the per-site outputs to itself: asking for both is an error, not a silent fallback
    r = subprocess.run([sys.executable, "-m", "clairs_to_amd", "call_chunks", "--chunk_list", str(tmp_path / "CANDIDATES_FILES"), "--output_dir",
                        str(tmp_path / "x"), "--merged_vcf_fn", str(tmp_path / "x.vcf"), "--gather_outputs", "--pipeline", "native"] + common,
                       cwd=ROOT, capture_output=True, text=True, timeout=120, env=dict(os.environ, PYTHONPATH=ROOT))
    assert r.returncode != 0 and "--gather_outputs runs on the thread-pool pipeline" in r.stderr


@pytest.mark.parametrize("source", ["bam", "text"])
@pytest.mark.parametrize("mode,aff_cls,neg_cls", [("snv", "CvT", "BiGRU_NACGT"), ("indel", "CvT_Indel", "BiGRU_NACGT_Indel")])
def test_native_pipeline_writes_the_same_files(tmp_path, source, mode, aff_cls, neg_cls):
    """cto_run_chunks (the chunk loop in C: csrc/pipeline.hip) against call_chunks.run_pipeline on the same chunk list - five
    chunks, one of them of another contig (no candidates, no file), one whose pileup is empty: same files, byte for byte."""
    from argparse import Namespace
    from bamutil import mpileup_rows
    import pickle
    from clairs_to_amd.call_chunks import native_eligible, run_pipeline, run_pipeline_native
    from clairs_to_amd.pileup_call import make_engine
    from clairs_to_amd.synth import likelihood_table
    K = 4 if mode == "snv" else 6
    sc = _bam_scenario(tmp_path)
    reads = pickle.load(open(tmp_path / "reads.pkl", "rb"))
    paths = _pickle_models(tmp_path, aff_cls, neg_cls, K)
    lik = tmp_path / "lik.txt"
    np.savetxt(lik, likelihood_table(K, seed=11), fmt="%.17g")
    sites = sc["sites"]
    parts = [sites[i * len(sites) // 3:(i + 1) * len(sites) // 3] for i in range(3)]
    parts.append([5950, 5990])               # beyond the last read's end for most reads: sparse pileup
    beds, mps = [], []
    for i, part in enumerate(parts):
        fn = tmp_path / ("chr1.%d_5_snv" % (i + 1))
        fn.write_text("".join("chr1\t%d\t%d\n" % (x - 17, x + 17) for x in part))
        beds.append(str(fn))
        mp = tmp_path / (fn.name + ".mpileup")
        mp.write_text(mpileup_rows(reads, 0, "chr1", max(1, min(part) - 16 - 33), max(part) + 18 + 33, bed=[(x - 17, x + 17) for x in part]))
        mps.append(str(mp))
    other = tmp_path / "chr2.1_1_snv"          # a chunk of a contig the job's --ctg_name does not match: no candidates
    other.write_text("chr2\t100\t134\n")
    beds.append(str(other))
    empty = tmp_path / "chr2.1_1_snv.mpileup"
    empty.write_text("")
    mps.append(str(empty))

    def chunk_args(out_dir):
        os.makedirs(out_dir, exist_ok=True)
        out = []
        for bed, mp in zip(beds, mps):
            out.append(Namespace(platform="ont", ref_fn=sc["fa"], ctg_name="chr1", samtools="samtools", bam_reader="native",
                                 tumor_bam_fn=sc["bam"], mpileup_fn=mp if source == "text" else None, min_bq=None, max_depth=None,
                                 max_indel_length=None, candidates_bed_regions=bed, chkpnt_fn_acgt=paths["model_acgt"],
                                 chkpnt_fn_nacgt=paths["model_nacgt"], min_rescale_cov=50, disable_indel_calling=(K == 4),
                                 likelihood_matrix_data=str(lik), call_fn=os.path.join(out_dir, "p_%s.vcf" % os.path.basename(bed)),
                                 predict_fn=None, sample_name="TUMOR", show_ref=True, qual=2, pileup=True))
        return out
    eng = make_engine(chunk_args(str(tmp_path / "x"))[0], "cuda:0")
    a_py, a_nat = chunk_args(str(tmp_path / "py")), chunk_args(str(tmp_path / "nat"))
    assert native_eligible(a_nat)
    st_py, st_nat = {}, {}
    n_py = run_pipeline(eng, a_py, producers=2, writers=2, stats=st_py)
    n_nat = run_pipeline_native(eng, a_nat, producers=3, writers=2, stats=st_nat, verbose=False)
    assert n_py == n_nat and n_py > 50
    assert st_nat["sites"] == st_py["sites"] == sum(len(p) for p in parts)
    names = sorted(os.listdir(tmp_path / "py"))
    assert names == sorted(os.listdir(tmp_path / "nat")) and 3 <= len(names) <= 4
    for fn in names:
        assert open(tmp_path / "py" / fn, "rb").read() == open(tmp_path / "nat" / fn, "rb").read(), fn
    # again on the same handle (slots, streams and buffers are per call), with one producer and a pipeline depth of one
    if source == "bam":
        # the default run above had some chunks' BGZF blocks inflated on the device (call_chunks.DEVICE_INFLATE); here every chunk is
        # inflated on the host cores, then two at a time on the device with the streams confined to 64 CUs - the same files each time
        assert st_nat["device_inflated"] >= 1
        for kw in (dict(inflate_cus=0), dict(inflate_cus=64, inflate_jobs=2)):
            st_dev = {}
            assert run_pipeline_native(eng, a_nat, producers=3, writers=2, stats=st_dev, verbose=False, **kw) == n_py
            assert (st_dev["device_inflated"] == 0) if kw["inflate_cus"] == 0 else (1 <= st_dev["device_inflated"] <= len(parts))
            for fn in names:
                assert open(tmp_path / "py" / fn, "rb").read() == open(tmp_path / "nat" / fn, "rb").read(), fn
    if source == "text":
        # the text tokenised on the device (cto_tokenise_device: every non-empty chunk's pack born in HBM) and on the producer threads
        # (cto_pack_from_mpileup): the same files
        for flag in (True, False):
            st_tok = {}
            assert run_pipeline_native(eng, a_nat, producers=3, writers=2, stats=st_tok, verbose=False, device_tokenise=flag) == n_py
            assert (st_tok["device_tokenised"] >= 3) if flag else (st_tok["device_tokenised"] == 0)
            for fn in names:
                assert open(tmp_path / "py" / fn, "rb").read() == open(tmp_path / "nat" / fn, "rb").read(), (flag, fn)
        for fn in names:
            assert open(tmp_path / "py" / fn, "rb").read() == open(tmp_path / "nat" / fn, "rb").read(), fn
    if source == "text":      # gzip-compressed BED and pileup text: both pipelines inflate them (gzip.open / zlib), same file again
        import gzip
        import shutil
        gz_py, gz_nat = chunk_args(str(tmp_path / "gz_py"))[:1], chunk_args(str(tmp_path / "gz_nat"))[:1]
        for a in gz_py + gz_nat:
            for attr in ("candidates_bed_regions", "mpileup_fn"):
                src = getattr(a, attr)
                if not os.path.exists(src + ".gz"):
                    with open(src, "rb") as fi, gzip.open(src + ".gz", "wb") as fo:
                        shutil.copyfileobj(fi, fo)
                setattr(a, attr, src + ".gz")
        assert native_eligible(gz_nat)
        assert run_pipeline(eng, gz_py, producers=1, writers=1) == run_pipeline_native(eng, gz_nat, producers=1, writers=1, verbose=False) > 10
        fn = os.path.basename(gz_py[0].call_fn)
        assert open(tmp_path / "gz_py" / fn, "rb").read() == open(tmp_path / "gz_nat" / fn, "rb").read() == open(tmp_path / "py" / names[0], "rb").read()
    # consecutive chunks on two compute streams (a second pair of model handles): the same files
    assert run_pipeline_native(eng, a_nat, producers=3, writers=2, verbose=False, two_streams=True) == n_py
    for fn in names:
        assert open(tmp_path / "py" / fn, "rb").read() == open(tmp_path / "nat" / fn, "rb").read(), fn
    from clairs_to_amd._lib import lib
    assert lib.cto_run_release() == 0           # the buffers kept from the first call are dropped; the next call allocates its own
    assert run_pipeline_native(eng, a_nat, producers=1, writers=1, depth=1, verbose=False) == n_py
    for fn in names:
        assert open(tmp_path / "py" / fn, "rb").read() == open(tmp_path / "nat" / fn, "rb").read(), fn


def test_native_pipeline_large_text_chunks(tmp_path):
    """22 MB of pileup text per chunk: the tokeniser runs on several threads and merges its parts straight into the pipeline's
    page-locked staging buffer (cto_pack::ext_entries) - same files as the Python pipeline, which uploads from the pack's own arrays."""
    from clairs_to_amd.call_chunks import run_pipeline, run_pipeline_native
    from clairs_to_amd.e2e import chunk_namespaces
    from clairs_to_amd.engine import Engine, synthetic_models
    from clairs_to_amd.synth import likelihood_table, lik_and_edges
    from clairs_to_amd.synth_run import make_text_run
    run = make_text_run(str(tmp_path / "run"), n_chunks=3, sites_per_chunk=4096, distinct=3)
    models = synthetic_models(4, seed=0)
    lik, edges = lik_and_edges(likelihood_table(4), 4)
    eng = Engine(models["aff"], models["neg"], lik, edges, min_bq=20, device="cuda:0")
    a_py = chunk_namespaces(run, str(tmp_path / "py"))
    a_nat = chunk_namespaces(run, str(tmp_path / "nat"))
    os.makedirs(tmp_path / "py"), os.makedirs(tmp_path / "nat")
    n_py = run_pipeline(eng, a_py, producers=2, writers=2)
    n_nat = run_pipeline_native(eng, a_nat, producers=2, writers=2, verbose=False)
    assert n_py == n_nat and n_py > 1000
    names = sorted(os.listdir(tmp_path / "py"))
    assert names == sorted(os.listdir(tmp_path / "nat")) and len(names) == 3
    for fn in names:
        assert open(tmp_path / "py" / fn, "rb").read() == open(tmp_path / "nat" / fn, "rb").read(), fn


def test_native_pipeline_tile_stream_across_chunk_seams(tmp_path):
    """cto_run_chunks feeds the networks whole rounds of 32-site tiles and carries a chunk's tail into the next chunk's launch
    (Run::launch_stream): 7 chunks of 2 500 sites = launches of 4096 / 4096 / 4096 / 4096 sites cut across the chunk seams + a
    1 116-site flush.  Sites are independent and the kernels are batch-invariant, so every file must equal the Python pipeline's,
    which launches chunk by chunk."""
    from clairs_to_amd.call_chunks import run_pipeline, run_pipeline_native
    from clairs_to_amd.e2e import chunk_namespaces
    from clairs_to_amd.engine import Engine, synthetic_models
    from clairs_to_amd.synth import likelihood_table, lik_and_edges
    from clairs_to_amd.synth_run import make_text_run
    run = make_text_run(str(tmp_path / "run"), n_chunks=7, sites_per_chunk=2500, distinct=7)
    models = synthetic_models(4, seed=0)
    lik, edges = lik_and_edges(likelihood_table(4), 4)
    eng = Engine(models["aff"], models["neg"], lik, edges, min_bq=20, device="cuda:0")
    a_py = chunk_namespaces(run, str(tmp_path / "py"))
    a_nat = chunk_namespaces(run, str(tmp_path / "nat"))
    os.makedirs(tmp_path / "py"), os.makedirs(tmp_path / "nat")
    n_py = run_pipeline(eng, a_py, producers=2, writers=2)
    for producers, writers, depth in ((2, 2, 0), (1, 1, 2), (4, 1, 0)):     # few slots: chunks waiting for rows must not starve the producers
        n_nat = run_pipeline_native(eng, a_nat, producers=producers, writers=writers, depth=depth, verbose=False)
        assert n_py == n_nat and n_py > 3000
        names = sorted(os.listdir(tmp_path / "py"))
        assert names == sorted(os.listdir(tmp_path / "nat")) and len(names) == 7
        for fn in names:
            assert open(tmp_path / "py" / fn, "rb").read() == open(tmp_path / "nat" / fn, "rb").read(), fn
            os.remove(tmp_path / "nat" / fn)


def test_native_pipeline_illumina_min_bq(tmp_path):
    """ilmn with an explicit --min_bq: the NEG network reads the AFF pass's tensors (run_clairs_to:1248-1252; cto_run_cfg.neg_reads_aff),
    the AFF gate is the option's value - the C pipeline and the Python pipeline must agree on that too (indel mode, BAM input)."""
    from argparse import Namespace
    from clairs_to_amd.call_chunks import run_pipeline, run_pipeline_native
    from clairs_to_amd.pileup_call import make_engine
    from clairs_to_amd.synth import likelihood_table
    sc = _bam_scenario(tmp_path)
    paths = _pickle_models(tmp_path, "CvT_Indel", "BiGRU_NACGT_Indel", 6)
    lik = tmp_path / "lik.txt"
    np.savetxt(lik, likelihood_table(6, seed=11), fmt="%.17g")

    def args(out):
        os.makedirs(tmp_path / out, exist_ok=True)
        return [Namespace(platform="ilmn", ref_fn=sc["fa"], ctg_name="chr1", samtools="samtools", bam_reader="native", tumor_bam_fn=sc["bam"],
                          mpileup_fn=None, min_bq=12, max_depth=None, max_indel_length=None, candidates_bed_regions=sc["bed"],
                          chkpnt_fn_acgt=paths["model_acgt"], chkpnt_fn_nacgt=paths["model_nacgt"], min_rescale_cov=50,
                          disable_indel_calling=False, likelihood_matrix_data=str(lik), call_fn=str(tmp_path / out / "p.vcf"), predict_fn=None,
                          sample_name="S", show_ref=True, qual=0, pileup=True)]
    eng = make_engine(args("x")[0], "cuda:0")
    assert eng.neg_reads_aff and eng.min_bq == 12
    n_py = run_pipeline(eng, args("py"), producers=1, writers=1)
    n_nat = run_pipeline_native(eng, args("nat"), producers=1, writers=1, verbose=False)
    assert n_py == n_nat > 50
    assert open(tmp_path / "py" / "p.vcf", "rb").read() == open(tmp_path / "nat" / "p.vcf", "rb").read()


def test_native_pipeline_runs_samtools_per_chunk(tmp_path):
    """--bam_reader samtools (the reference's producer, the default): cto_run_chunks starts `samtools mpileup ...` per chunk and
    tokenises its output - same file as the Python pipeline's subprocess path and as the built-in reader; a failing samtools is an
    error of the run.  (samtools itself is absent here: a shim that checks its arguments and prints the prepared pileup stands in.)"""
    import stat
    from argparse import Namespace
    from clairs_to_amd._lib import CtoError
    from clairs_to_amd.call_chunks import native_eligible, run_pipeline, run_pipeline_native
    from clairs_to_amd.pileup_call import make_engine
    from clairs_to_amd.synth import likelihood_table
    sc = _bam_scenario(tmp_path)
    paths = _pickle_models(tmp_path, "CvT", "BiGRU_NACGT", 4)
    lik = tmp_path / "lik.txt"
    np.savetxt(lik, likelihood_table(4, seed=11), fmt="%.17g")
    shim = tmp_path / "samtools_checked"
    shim.write_text("#!/bin/sh\n# stands in for samtools: the command line must be the reference's (with --min-BQ 0), then the prepared pileup\n"
                    "[ \"$1\" = mpileup ] && [ \"$2\" = --reverse-del ] && [ \"$3\" = --output-MQ ] && [ \"$4\" = -r ] || exit 3\n"
                    "case \"$*\" in *\"--min-MQ 0 --min-BQ 0 -l %s --excl-flags 2316 --max-depth 7000 %s\") ;; *) exit 4 ;; esac\n"
                    "cat %s\n" % (sc["bed"], sc["bam"], sc["mp"]))
    shim.chmod(shim.stat().st_mode | stat.S_IEXEC)
    bad = tmp_path / "samtools_fails"
    bad.write_text("#!/bin/sh\nexit 1\n")
    bad.chmod(bad.stat().st_mode | stat.S_IEXEC)

    def args(out, **kw):
        os.makedirs(tmp_path / out, exist_ok=True)
        base = dict(platform="ont", ref_fn=sc["fa"], ctg_name="chr1", samtools=str(shim), bam_reader="samtools", tumor_bam_fn=sc["bam"],
                    mpileup_fn=None, min_bq=None, max_depth=7000, max_indel_length=None, candidates_bed_regions=sc["bed"],
                    chkpnt_fn_acgt=paths["model_acgt"], chkpnt_fn_nacgt=paths["model_nacgt"], min_rescale_cov=50, disable_indel_calling=True,
                    likelihood_matrix_data=str(lik), call_fn=str(tmp_path / out / "p.vcf"), predict_fn=None, sample_name="S", show_ref=True,
                    qual=0, pileup=True)
        base.update(kw)
        return [Namespace(**base)]
    eng = make_engine(args("x")[0], "cuda:0")
    assert native_eligible(args("x"))
    n_py = run_pipeline(eng, args("py"), producers=1, writers=1)
    n_nat = run_pipeline_native(eng, args("nat"), producers=1, writers=1, verbose=False)
    n_own = run_pipeline_native(eng, args("own", bam_reader="native", max_depth=None), producers=1, writers=1, verbose=False, inflate_cus=0)
    assert n_py == n_nat == n_own > 50
    assert open(tmp_path / "py" / "p.vcf", "rb").read() == open(tmp_path / "nat" / "p.vcf", "rb").read() == open(tmp_path / "own" / "p.vcf", "rb").read()
    with pytest.raises(CtoError, match="mpileup failed"):
        run_pipeline_native(eng, args("bad", samtools=str(bad)), producers=1, writers=1, verbose=False)
    with pytest.raises(CtoError, match="cannot run"):
        run_pipeline_native(eng, args("bad", samtools=str(tmp_path / "no_such_samtools")), producers=1, writers=1, verbose=False)


def test_native_pipeline_reports_errors(tmp_path):
    """a missing pileup file, a contig the reference index does not hold: CtoError naming the cause, no hang, no partial state"""
    from argparse import Namespace
    from clairs_to_amd._lib import CtoError
    from clairs_to_amd.call_chunks import run_pipeline_native
    from clairs_to_amd.pileup_call import make_engine
    from clairs_to_amd.synth import likelihood_table
    sc = _bam_scenario(tmp_path)
    paths = _pickle_models(tmp_path, "CvT", "BiGRU_NACGT", 4)
    lik = tmp_path / "lik.txt"
    np.savetxt(lik, likelihood_table(4, seed=11), fmt="%.17g")
    base = dict(platform="ont", ref_fn=sc["fa"], ctg_name="chr1", samtools="samtools", bam_reader="native", tumor_bam_fn=sc["bam"],
                mpileup_fn=None, min_bq=None, max_depth=None, max_indel_length=None, candidates_bed_regions=sc["bed"],
                chkpnt_fn_acgt=paths["model_acgt"], chkpnt_fn_nacgt=paths["model_nacgt"], min_rescale_cov=50, disable_indel_calling=True,
                likelihood_matrix_data=str(lik), call_fn=str(tmp_path / "o" / "p.vcf"), predict_fn=None, sample_name="S", show_ref=False,
                qual=0, pileup=True)
    eng = make_engine(Namespace(**base), "cuda:0")
    with pytest.raises(CtoError, match="cannot open"):
        run_pipeline_native(eng, [Namespace(**dict(base, mpileup_fn=str(tmp_path / "absent.mpileup")))] * 6, producers=3, verbose=False)
    bed7 = tmp_path / "chr7.bed"
    bed7.write_text("chr7\t100\t134\n")
    with pytest.raises(CtoError, match="chr7"):
        run_pipeline_native(eng, [Namespace(**dict(base, ctg_name="chr7", candidates_bed_regions=str(bed7)))], producers=2, verbose=False)
    assert run_pipeline_native(eng, [Namespace(**base)], producers=2, verbose=False) > 10      # and the engine is still usable


def test_extract_cli_writes_reference_bed_chunks(tmp_path):
    """extract_candidates_calling at its file seam: the BED chunk files (x-17 .. x+17 windows) and the list file, from the
    golden fixture's pileup, must name exactly the candidates of the reference's own files; the native BAM reader and the
    text path agree on a synthetic BAM."""
    from argparse import Namespace
    from clairs_to_amd.extract_candidates_calling import extract_to_files
    g = load_json_gz("extract.json.gz")
    pr = g["params"]
    ref = "A" * (g["ref_start"] - 1) + g["ref"]
    fa = tmp_path / "ref.fa"
    fa.write_text(">chr1\n" + "\n".join(ref[i:i + 60] for i in range(0, len(ref), 60)) + "\n")
    (tmp_path / "ref.fa.fai").write_text("chr1\t%d\t6\t60\t61\n" % len(ref))
    mp = tmp_path / "mp.txt"
    mp.write_text(g["mpileup_neg"])
    common = dict(platform="ont", ref_fn=str(fa), ctg_name="chr1", chunk_id=None, snv_min_af=pr["snv_min_af"],
                  indel_min_af=pr["indel_min_af"], min_coverage=pr["min_coverage"], min_mq=pr["min_mq"], min_bq=pr["min_bq"],
                  alternative_base_num=pr["alt_base_num"], select_indel_candidates=True, samtools="samtools", max_depth=None)
    out = tmp_path / "cand"
    snv, indel = extract_to_files(Namespace(candidates_folder=str(out), mpileup_fn=str(mp), tumor_bam_fn=None, bam_reader="samtools",
                                            ctg_start=None, ctg_end=None, **common))
    assert snv == g["snv"] and indel == g["indel"]

    def centres(suffix):
        xs = []
        for fn in sorted(os.listdir(out)):
            if fn.endswith(suffix) and fn.startswith("chr1."):
                for row in open(out / fn):
                    c = row.split("\t")
                    xs.append(int(c[2]) - 17)
        return xs
    assert centres("_snv") == g["snv"] and centres("_indel") == g["indel"]
    # without --chunk_id the reference's file names carry `None` (extract_candidates_calling.py:182, 457, 467)
    assert open(out / "SNV_CANDIDATES_FILE_chr1_None").read().split() == [str(out / "chr1.None_0_1_snv")]
    # BAM path: native reader vs the text of the naive pileup of the same BAM
    from bamutil import mpileup_rows
    sc = _bam_scenario(tmp_path)
    L = sc["L"]
    got = extract_to_files(Namespace(candidates_folder=str(tmp_path / "c_native"), mpileup_fn=None, tumor_bam_fn=sc["bam"],
                                     bam_reader="native", ctg_start=200, ctg_end=L - 200, **dict(common, ref_fn=sc["fa"])))
    import pickle
    reads = pickle.load(open(tmp_path / "reads.pkl", "rb"))
    txt = tmp_path / "region.txt"
    txt.write_text(mpileup_rows(reads, 0, "chr1", 200 - 33, L - 200 + 33))       # --ctg_start / --ctg_end: the rows of start - 33 .. end + 33 (:289-292)
    want = extract_to_files(Namespace(candidates_folder=str(tmp_path / "c_text"), mpileup_fn=str(txt), tumor_bam_fn=None,
                                      bam_reader="samtools", ctg_start=None, ctg_end=None, **dict(common, ref_fn=sc["fa"])))
    assert got == want and len(got[0]) > 20


def _region_namespace(sc_fa, K, paths, lik, out_dir, name, **kw):
    from argparse import Namespace
    base = dict(platform="ont", ref_fn=sc_fa, ctg_name="chr1", samtools="samtools", bam_reader="native", tumor_bam_fn=None, mpileup_fn=None,
                min_bq=None, max_depth=None, max_indel_length=None, candidates_bed_regions=None, chkpnt_fn_acgt=paths["model_acgt"],
                chkpnt_fn_nacgt=paths["model_nacgt"], min_rescale_cov=50, disable_indel_calling=(K == 4), likelihood_matrix_data=str(lik),
                call_fn=os.path.join(out_dir, "p_%s.vcf" % name), predict_fn=None, sample_name="TUMOR", show_ref=True, qual=2, pileup=True)
    base.update(kw)
    os.makedirs(out_dir, exist_ok=True)
    return Namespace(**base)


@pytest.mark.parametrize("mode,aff_cls,neg_cls", [("snv", "CvT", "BiGRU_NACGT"), ("indel", "CvT_Indel", "BiGRU_NACGT_Indel")])
def test_region_job_extracts_the_references_candidates_and_calls_them(tmp_path, mode, aff_cls, neg_cls):
    """SURVEY 8(f1) delivered inside the run: a REGION job of cto_run_chunks (no candidate BED) on the golden extraction fixture's
    pileup.  (a) the candidates it extracts in HBM are exactly the lists the reference's extract_candidates_calling wrote for
    that pileup (its `_snv` / `_indel` BED chunk rows); (b) its VCF is byte-identical to the BED-driven job's on those candidates."""
    from clairs_to_amd.call_chunks import run_pipeline_native
    from clairs_to_amd.pileup_call import make_engine
    from clairs_to_amd.synth import likelihood_table
    K = 4 if mode == "snv" else 6
    g = load_json_gz("extract.json.gz")
    pr = g["params"]
    ref = "A" * (g["ref_start"] - 1) + g["ref"]
    fa = tmp_path / "ref.fa"
    fa.write_text(">chr1\n" + "\n".join(ref[i:i + 60] for i in range(0, len(ref), 60)) + "\n")
    (tmp_path / "ref.fa.fai").write_text("chr1\t%d\t6\t60\t61\n" % len(ref))
    mp = tmp_path / "region.mpileup"
    mp.write_text(g["mpileup_neg"])
    rows = [int(r.split("\t")[1]) for r in g["mpileup_neg"].split("\n") if r]
    paths = _pickle_models(tmp_path, aff_cls, neg_cls, K)
    lik = tmp_path / "lik.txt"
    np.savetxt(lik, likelihood_table(K, seed=11), fmt="%.17g")
    gates = dict(snv_min_af=pr["snv_min_af"], indel_min_af=pr["indel_min_af"], min_coverage=pr["min_coverage"], extract_min_mq=pr["min_mq"],
                 extract_min_bq=pr["min_bq"], alternative_base_num=pr["alt_base_num"])
    want = g["snv"] if K == 4 else g["indel"]
    # the region whose +-33 extension (extract_candidates_calling.py:289-292) is exactly the fixture's rows
    reg = _region_namespace(str(fa), K, paths, lik, str(tmp_path / "reg"), "region", mpileup_fn=str(mp), region=(rows[0] + 33, rows[-1] - 33),
                            candidates_out_fn=str(tmp_path / "cand.bed"), **gates)
    eng = make_engine(reg, "cuda:0")
    st = {}
    n_reg = run_pipeline_native(eng, [reg], producers=1, writers=1, stats=st, verbose=False)
    got = [int(r.split("\t")[2]) - 17 for r in open(tmp_path / "cand.bed") if r.strip()]
    assert got == want and st["sites"] == len(want)
    assert all(r.split("\t")[1] == str(max(int(r.split("\t")[2]) - 34, 1)) for r in open(tmp_path / "cand.bed") if r.strip())
    # the BED-driven job on the reference's candidate list, same pileup text
    bed = tmp_path / "chr1.1_0_1_x"
    bed.write_text("".join("chr1\t%d\t%d\n" % (max(x - 17, 1), x + 17) for x in want))
    by_bed = _region_namespace(str(fa), K, paths, lik, str(tmp_path / "bed"), "region", mpileup_fn=str(mp), candidates_bed_regions=str(bed))
    n_bed = run_pipeline_native(eng, [by_bed], producers=1, writers=1, verbose=False)
    assert n_reg == n_bed and n_reg > 0
    assert open(reg.call_fn, "rb").read() == open(by_bed.call_fn, "rb").read()
    # a narrower region: only the rows of its own +-33 range are candidates
    lo, hi = want[len(want) // 3], want[-1] - 40
    mp2 = tmp_path / "part.mpileup"        # `samtools mpileup -r` of that region's own read range (+ the window flanks at its edges)
    mp2.write_text("".join(r + "\n" for r in g["mpileup_neg"].split("\n") if r and lo - 17 <= int(r.split("\t")[1]) <= hi + 17))
    part = _region_namespace(str(fa), K, paths, lik, str(tmp_path / "part"), "part", mpileup_fn=str(mp2), region=(lo + 33, hi - 33),
                             candidates_out_fn=str(tmp_path / "part.bed"), **gates)
    run_pipeline_native(eng, [part], producers=1, writers=1, verbose=False)
    assert [int(r.split("\t")[2]) - 17 for r in open(tmp_path / "part.bed") if r.strip()] == [x for x in want if lo <= x <= hi]


@pytest.mark.parametrize("K", [4, 6])
def test_region_jobs_from_bam_equal_the_two_step_run(tmp_path, K):
    """BAM + regions, no BED: inflate -> pile-up of every position -> candidate gates -> tensors -> networks -> VCF in one call, on the
    device where a context is free (device inflate + device pile-up) and through the host reader otherwise - against the two-step run
    (extract_candidates on the same BAM's pack, then BED-driven chunks): same candidate lists, same records."""
    import torch
    from clairs_to_amd.call_chunks import run_pipeline_native
    from clairs_to_amd.extract_candidates_calling import extract_candidates, candidate_positions
    from clairs_to_amd.fasta import read_region
    from clairs_to_amd.pack import ColumnPack
    from clairs_to_amd.pileup_call import make_engine
    from clairs_to_amd.synth import likelihood_table
    sc = _bam_scenario(tmp_path)
    cls = ("CvT", "BiGRU_NACGT") if K == 4 else ("CvT_Indel", "BiGRU_NACGT_Indel")
    paths = _pickle_models(tmp_path, cls[0], cls[1], K)
    lik = tmp_path / "lik.txt"
    np.savetxt(lik, likelihood_table(K, seed=11), fmt="%.17g")
    regions = [(300, 1900), (1901, 3500), (3501, 5600)]
    # step 1 on its own (the Python mirror of extract_candidates_calling on the native reader's pack), per region with its +-33 rows
    ref = read_region(sc["fa"], "chr1", 1, sc["L"])
    want, beds = [], []
    abn = 3 if K == 4 else 1                 # the scenario's indels are private to their reads: one supporting read makes an indel candidate
    for i, (a, b) in enumerate(regions):
        lo, hi = max(1, a - 33), b + 33
        dp = ColumnPack.from_bam(sc["bam"], "chr1", lo, hi, ref, 1).to_device("cuda:0")
        flags, _ = extract_candidates(dp, 20, 20, 0.05, 0.01, 4, abn, K == 6)
        xs = candidate_positions(dp, flags, 1 if K == 4 else 2).cpu().tolist()
        want.append(xs)
        bed = tmp_path / ("chr1.%d_0_1_c" % i)
        bed.write_text("".join("chr1\t%d\t%d\n" % (max(x - 17, 1), x + 17) for x in xs))
        beds.append(str(bed))
    assert sum(len(x) for x in want) > (60 if K == 4 else 5)
    two_step = [_region_namespace(sc["fa"], K, paths, lik, str(tmp_path / "two"), "r%d" % i, tumor_bam_fn=sc["bam"], candidates_bed_regions=b)
                for i, b in enumerate(beds) if want[i]]
    eng = make_engine(two_step[0], "cuda:0")
    n_two = run_pipeline_native(eng, two_step, producers=2, writers=1, verbose=False, inflate_cus=0)
    for tag, kw in (("host", dict(inflate_cus=0)), ("device", dict(inflate_cus=64, inflate_jobs=2))):
        out = str(tmp_path / tag)
        jobs = [_region_namespace(sc["fa"], K, paths, lik, out, "r%d" % i, tumor_bam_fn=sc["bam"], region=r, indel_min_af=0.01, alternative_base_num=abn,
                                  candidates_out_fn=os.path.join(out, "cand%d.bed" % i)) for i, r in enumerate(regions)]
        st = {}
        n = run_pipeline_native(eng, jobs, producers=2, writers=1, stats=st, verbose=False, **kw)
        assert n == n_two and st["sites"] == sum(len(x) for x in want)
        if tag == "device":
            assert st["device_piled"] >= 1
        for i in range(len(regions)):
            got = [int(r.split("\t")[2]) - 17 for r in open(os.path.join(out, "cand%d.bed" % i)) if r.strip()]
            assert got == want[i], (tag, i)
            a, b = os.path.join(out, "p_r%d.vcf" % i), str(tmp_path / "two" / ("p_r%d.vcf" % i))
            assert os.path.exists(a) == os.path.exists(b)
            if os.path.exists(a):
                assert open(a, "rb").read() == open(b, "rb").read(), (tag, i)
    if K == 6:
        # --call_indels_only_in_these_regions (extract_candidates_calling.py:437-446): an indel candidate stays when [pos - 1, pos) overlaps
        # a row of the BED; rows of other contigs, comments and a zero-length row (widened by one) as bed_tree_from reads them
        keep_rows = [(900, 1500), (2999, 2999), (4000, 5000)]
        bed = tmp_path / "indel_regions.bed"
        bed.write_text("# comment\nchrX\t1\t100000\n" + "".join("chr1\t%d\t%d\n" % r for r in keep_rows))
        out = str(tmp_path / "filtered")
        jobs = [_region_namespace(sc["fa"], K, paths, lik, out, "r%d" % i, tumor_bam_fn=sc["bam"], region=r, indel_min_af=0.01, alternative_base_num=abn,
                                  candidates_out_fn=os.path.join(out, "cand%d.bed" % i), call_indels_only_in_these_regions=str(bed))
                for i, r in enumerate(regions)]
        st = {}
        run_pipeline_native(eng, jobs, producers=2, writers=1, stats=st, verbose=False, inflate_cus=64, inflate_jobs=2)
        inside = lambda x: any(a < x and (b if b > a else a + 1) > x - 1 for a, b in keep_rows)
        total = 0
        for i in range(len(regions)):
            got = [int(r.split("\t")[2]) - 17 for r in open(os.path.join(out, "cand%d.bed" % i)) if r.strip()]
            assert got == [x for x in want[i] if inside(x)], i
            total += len(got)
        assert 0 < total < sum(len(x) for x in want) and st["sites"] == total
        # the same BED gzipped, with and without a .gz name (bed_tree_from reads every BED through `gzip -fdc`, shared/interval_tree.py:43):
        # the same candidates - not "no regions, everything passes"
        import gzip as _gz
        for name in ("indel_regions.bed.gz", "indel_regions_gz_without_suffix.bed"):
            zb = tmp_path / name
            zb.write_bytes(_gz.compress(bed.read_bytes()))
            outz = str(tmp_path / ("filtered_" + name))
            jobs = [_region_namespace(sc["fa"], K, paths, lik, outz, "r%d" % i, tumor_bam_fn=sc["bam"], region=r, indel_min_af=0.01, alternative_base_num=abn,
                                      candidates_out_fn=os.path.join(outz, "cand%d.bed" % i), call_indels_only_in_these_regions=str(zb))
                    for i, r in enumerate(regions)]
            run_pipeline_native(eng, jobs, producers=2, writers=1, verbose=False, inflate_cus=64, inflate_jobs=2)
            for i in range(len(regions)):
                assert open(os.path.join(outz, "cand%d.bed" % i)).read() == open(os.path.join(out, "cand%d.bed" % i)).read(), (name, i)
        # a row that is not "name start end" ends the run (the reference's int() raises): it is not skipped
        badbed = tmp_path / "bad.bed"
        badbed.write_text("chr1\t900\t1500\nchr1 this-is-not-a-number 7\n")
        jobs = [_region_namespace(sc["fa"], K, paths, lik, str(tmp_path / "bad"), "r0", tumor_bam_fn=sc["bam"], region=regions[0], indel_min_af=0.01,
                                  alternative_base_num=abn, call_indels_only_in_these_regions=str(badbed))]
        with pytest.raises(RuntimeError) as ei:
            run_pipeline_native(eng, jobs, producers=1, writers=1, verbose=False, inflate_cus=64, inflate_jobs=2)
        assert "Invalid bed input in 2-th row" in str(ei.value)
    torch.cuda.synchronize()


def test_engine_run_region_is_extraction_plus_run_device(tmp_path):
    """Engine.run_region: candidates as an internal product on a resident pack == extract_candidates + candidate_positions + run_device"""
    import torch
    from clairs_to_amd.engine import Engine, synthetic_models
    from clairs_to_amd.extract_candidates_calling import extract_candidates, candidate_positions
    from clairs_to_amd.synth import SynthChunk, likelihood_table, lik_and_edges
    dev = torch.device("cuda:0")
    chunk = SynthChunk(300, seed=9, spacing=40, p_mismatch=0.03, p_ins=0.02, p_del=0.03)
    for K in (4, 6):
        models = synthetic_models(K)
        lik, edges = lik_and_edges(likelihood_table(K), K)
        eng = Engine(models["aff"], models["neg"], lik, edges, min_bq=20, device=dev)
        dp = eng.upload(chunk.arrays())
        lo, hi = int(chunk.col_pos[200]), int(chunk.col_pos[-200])
        sites, out = eng.run_region(dp, lo, hi, indel_min_af=0.1)
        flags, _ = extract_candidates(dp, 20, 20, 0.05, 0.1 if K == 6 else 1.0, 4, 3, K == 6)
        f, pos = flags.cpu().numpy(), chunk.col_pos
        want = pos[((f & (1 if K == 4 else 2)) != 0) & (pos >= lo) & (pos <= hi)]
        assert sites.cpu().tolist() == want.tolist() and len(want) > 5
        assert candidate_positions(dp, flags, 1 if K == 4 else 2).cpu().tolist() == pos[(f & (1 if K == 4 else 2)) != 0].tolist()
        ref = eng.run_device(dp, sites)
        for k in ("probs", "post", "decision", "qual"):
            assert torch.equal(out[k], ref[k])


@pytest.mark.parametrize("world", [1, 2])
def test_call_chunks_region_list_equals_the_two_step_run(tmp_path, world):
    """`call_chunks --region_list` from the command line (one process, then two ranks sharing this box's GPU): BAM + three adjacent regions, no
    candidate BEDs.  The merged VCF equals the merged VCF of the two-step run - candidates extracted per region with the same gates
    (written by --candidates_dir), then `call_chunks --chunk_list` over those BED files.  Adjacent regions share the candidates of their
    +-33 overlap (extract_candidates_calling.py:289-292), as the reference's chunks do; sort_vcf keeps one record per position."""
    import subprocess
    import sys
    from conftest import ROOT
    from clairs_to_amd.synth import likelihood_table
    sc = _bam_scenario(tmp_path)
    paths = _pickle_models(tmp_path, "CvT", "BiGRU_NACGT", 4)
    lik = tmp_path / "lik.txt"
    np.savetxt(lik, likelihood_table(4, seed=11), fmt="%.17g")
    regions = [(300, 1900), (1901, 3500), (3501, 5600)]
    (tmp_path / "REGIONS").write_text("# ctg start end\n" + "".join("chr1\t%d\t%d\n" % r for r in regions))
    common = ["--platform", "ont", "--tumor_bam_fn", sc["bam"], "--ref_fn", sc["fa"], "--bam_reader", "native", "--chkpnt_fn_acgt",
              paths["model_acgt"], "--chkpnt_fn_nacgt", paths["model_nacgt"], "--disable_indel_calling", "True",
              "--likelihood_matrix_data", str(lik), "--show_ref"]
    env = dict(os.environ, PYTHONPATH=ROOT)
    launcher = [sys.executable] if world == 1 else [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                                                    "--master-addr", "127.0.0.1", "--master-port", "29546"]
    merged_r, cand = tmp_path / "merged_regions.vcf", tmp_path / "cand"
    r = subprocess.run(launcher + ["-m", "clairs_to_amd", "call_chunks", "--region_list", str(tmp_path / "REGIONS"), "--output_dir", str(tmp_path / "out_r"),
                                   "--merged_vcf_fn", str(merged_r), "--candidates_dir", str(cand)] + common,
                       cwd=ROOT, capture_output=True, text=True, timeout=280, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    assert sorted(os.listdir(tmp_path / "out_r")) == sorted("p_chr1_%d_%d.vcf" % rg for rg in regions)
    beds = [str(cand / ("chr1_%d_%d.snv" % rg)) for rg in regions]
    centres = [[int(x.split("\t")[2]) - 17 for x in open(b) if x.strip()] for b in beds]
    assert all(len(c) > 10 for c in centres)
    for (a, b), c in zip(regions, centres):
        assert c == sorted(c) and a - 33 <= c[0] and c[-1] <= b + 33
    assert set(centres[0]) & set(centres[1]) or set(centres[1]) & set(centres[2])      # the overlap zones hold candidates
    chunk_files = []
    for i, b in enumerate(beds):                      # the reference's chunk-file names carry the contig in front of the first dot
        fn = tmp_path / ("chr1.%d_0_1_snv" % i)
        fn.write_text(open(b).read())
        chunk_files.append(str(fn))
    (tmp_path / "CANDIDATES_FILES").write_text("".join(n + "\n" for n in chunk_files))
    merged_b = tmp_path / "merged_beds.vcf"
    r = subprocess.run([sys.executable, "-m", "clairs_to_amd", "call_chunks", "--chunk_list", str(tmp_path / "CANDIDATES_FILES"), "--output_dir",
                        str(tmp_path / "out_b"), "--merged_vcf_fn", str(merged_b)] + common, cwd=ROOT, capture_output=True, text=True, timeout=280, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    rec = lambda fn: [l for l in open(fn).read().split("\n") if l and not l.startswith("#")]
    assert rec(merged_r) == rec(merged_b) and len(rec(merged_r)) > 50
    pos = [int(l.split("\t")[1]) for l in rec(merged_r)]
    assert pos == sorted(set(pos))                    # one record per position, in order

#!/usr/bin/env python3
"""Benchmark of the hot path: candidate sites/s for pileup-tensor creation + AFF + NEG inference + posterior.

One "step" = one pass of the whole hot path over one 4096-site chunk of the ONT 50x synthetic SNV job
(BASELINE.json configs[1]: "ONT 50x synthetic BAM, 1M candidate SNV sites, batch=4096, 1xMI355X"); the
column packs of the chunks are resident in HBM before the timed region starts.  With N > 1 ranks
(torch.distributed over RCCL, one rank per GPU) every rank processes its own chunks (sites shard with no
cross-site dependence) and the per-site probabilities are gathered over xGMI each step.

Prints ONE JSON line (rank 0).  `roofline` is measured live with HIP events around the dominant kernel
(the BiGRU layer-2 recurrent kernel), `cpu_baseline` times the CPU oracle on a bounded sample of the same
workload on the host cores of this box.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

BATCH = 4096
N_OUT = 4
PEAK_FP32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32, dense
# The genuine reference cannot travel to the GPU box; tools/time_reference.py times it in the build container (8 vCPU) on the
# same generator and writes profiles/reference_cpu_timing.json.  Quoted, not measured here.
REFERENCE_PYTHON_NOTE = ("HKU-BAL/ClairS-TO v0.4.4 itself, build container (8 vCPU), configs[1] generator, 10 000-site chunks through its four "
                         "commands per chunk (create_tensor x2, predict, call_variants; samtools decode excluded, CPython, torch 1 thread per "
                         "process): 211 sites/s with 1 process, 1 297 sites/s with 8 processes (tools/time_reference.py -> "
                         "profiles/reference_cpu_timing.json)")


# MI355X_MICROARCH.md, HBM section: on gfx950 FETCH_SIZE reports half the bytes of coalesced streaming reads (128-byte requests tallied
# as 64 B).  Calibrated on this code's own access patterns (profiles/round2_*_pmc_hbm_traffic.json): k_featurize_columns reads the
# 28.6 MB pack exactly once and shows FETCH_SIZE = 14.8 MB -> x2; WRITE_SIZE is 1.0x (GRU layer 1 writes its 138.4 MB output: 138.4 MB).
FETCH_CORRECTION = 2.0
PEAK_16BIT_MFMA_TFLOPS = 2500.0     # dense bf16 / f16 MFMA (MI355X_MICROARCH.md:42), the yardstick of the split-operand side channel only


def _pmc_file():
    """newest committed rocprofv3 PMC digest (profiles/round<N>_<tag>_pmc_hbm_traffic.json, tools/collect_profiles.sh)"""
    import glob
    fs = sorted(glob.glob(os.path.join(ROOT, "profiles", "round*_pmc_hbm_traffic.json")))
    return fs[-1] if fs else None


def pmc_traffic(batch):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes (FETCH_SIZE + WRITE_SIZE, separate
    passes, collected at batch 4096); None for any other batch."""
    fn = _pmc_file()
    if batch != 4096 or fn is None:
        return None
    k = json.load(open(fn))["kernels"]
    for name, v in k.items():
        if "k_gru_layer" in name and "<256" in name:
            return int((FETCH_CORRECTION * v["FETCH_SIZE_KB_mean_per_launch"] + v["WRITE_SIZE_KB_mean_per_launch"]) * 1024)
    return None


def live_pmc_traffic(batch, timeout_s=240):
    """HBM / fabric bytes per launch MEASURED in this run: two child runs of this file under `rocprofv3 --kernel-trace --pmc`
    (FETCH_SIZE, then WRITE_SIZE: separate passes, no other trace domain - MI355X_MICROARCH.md's recipe), three timed steps each on
    two resident chunks, legs off.  Returns {"gru_l2": bytes, "featurize": bytes} (FETCH x FETCH_CORRECTION + WRITE, mean per launch)
    or None when rocprofv3 is not on the box / a pass fails - the caller then falls back to the committed digest and says so."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if exe is None or batch != 4096:
        return None
    env = {k: v for k, v in os.environ.items() if not (k.startswith(("ROCPROF", "ROCP_", "ROCTX")) or k == "HSA_TOOLS_LIB")}
    env["TMPDIR"] = "/tmp"
    sums = {}
    with tempfile.TemporaryDirectory(dir="/tmp") as tmp:
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(tmp, ctr)
            cmd = [exe, "--kernel-trace", "--pmc", ctr, "--output-format", "csv", "-d", out, "--", sys.executable, os.path.join(ROOT, "bench.py"),
                   "--steps", "3", "--warmup", "1", "--pool", "2", "--no-cpu-baseline", "--no-e2e", "--no-sustained", "--no-split", "--no-configs",
                   "--no-postfilters", "--no-live-traffic"]
            try:
                r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=timeout_s)
            except Exception:
                return None
            fs = glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not fs:
                return None
            acc = {}
            for row in csv.DictReader(open(fs[0])):
                if row["Counter_Name"] == ctr:
                    acc.setdefault(row["Kernel_Name"], []).append(float(row["Counter_Value"]))
            for name, v in acc.items():
                key = "gru_l2" if ("k_gru_layer" in name and "<256" in name) else "featurize" if "k_featurize_sites" in name else None
                if key:
                    sums.setdefault(key, {})[ctr] = sum(v) / len(v) * 1024.0          # the counters are in KB
    out = {}
    for key, d in sums.items():
        if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
            out[key] = int(FETCH_CORRECTION * d["FETCH_SIZE"] + d["WRITE_SIZE"])
    return out or None


def pmc_traffic_featurize(batch):
    fn = _pmc_file()
    if batch != 4096 or fn is None:
        return None
    tot = 0
    for name, v in json.load(open(fn))["kernels"].items():
        if "k_featurize_sites" in name:
            tot += int((FETCH_CORRECTION * v["FETCH_SIZE_KB_mean_per_launch"] + v["WRITE_SIZE_KB_mean_per_launch"]) * 1024)
    return tot or None


def usable_cores():
    """Cores this process may really use: affinity mask, capped by the cgroup CPU quota when there is one."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = max(1, min(n, int(float(quota) / float(period) + 0.5)))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p_ = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = max(1, min(n, int(q / p_ + 0.5)))
        except Exception:
            pass
    return n


def cpu_baseline(chunks, models, lik, edges, min_bq, n_sample, budget_s=12.0):
    """CPU port of the reference path (oracle/cto_oracle.c, OpenMP over sites) on a bounded sample of the same job: the first
    n_sample sites of as many resident chunks as fit in ~budget_s seconds (at least one).  Timed on the SPEED build of that
    source (-O3 -march=native -ffast-math, compiled on this host: oracle.build_fast), which is first held to the -O2
    -ffp-contract=off checker build on 64 sites (integer tensors equal, probabilities within 1e-6); the checker build's own rate
    on a small sample is reported beside it, and so is the genuine reference's (measured in the build container, it cannot travel)."""
    import numpy as np
    import oracle
    oracle.build()
    fast_lib = oracle.build_fast()
    cores = usable_cores()
    os.environ["OMP_NUM_THREADS"] = str(cores)
    cfg = dict(emb_dim=(16, 64, 128), heads=(1, 3, 4), depth=(1, 2, 3), n_out=N_OUT)
    from concurrent.futures import ThreadPoolExecutor

    def run_sample(chunk, n, cores=cores):
        sites = chunk.site_pos[:n]
        ref, lo = chunk.ref_window()
        # tensor creation is per-site independent too: the sample is cut into one slice of sites per core, each with the
        # mpileup rows of its own windows (untimed input prep), and the slices run on a thread pool (the C calls drop the GIL)
        cuts = np.linspace(0, len(sites), min(cores, len(sites)) + 1).astype(int)
        slices = []
        for a, b in zip(cuts[:-1], cuts[1:]):
            if b > a:
                c0 = int(np.searchsorted(chunk.col_pos, int(sites[a]) - 16, side="left"))
                c1 = int(np.searchsorted(chunk.col_pos, int(sites[b - 1]) + 17, side="right"))
                slices.append((sites[a:b], {q: oracle.synth_mpileup_text(chunk, q, (c0, c1)) for q in (min_bq, 0)}))
        t0 = time.perf_counter()
        with ThreadPoolExecutor(max_workers=cores) as ex:
            parts = list(ex.map(lambda s: (oracle.create_tensor(s[1][min_bq], ref, lo, s[0])[:2], oracle.create_tensor(s[1][0], ref, lo, s[0])[:2]),
                                slices))
        ta, da = np.concatenate([p[0][0] for p in parts]), np.concatenate([p[0][1] for p in parts])
        tn, dn = np.concatenate([p[1][0] for p in parts]), np.concatenate([p[1][1] for p in parts])
        xa, xn = oracle.rescale(ta, da), oracle.rescale(tn, dn)
        la = oracle.cvt_forward(models["aff_weights"], cfg, xa)
        ln = oracle.bigru_forward(models["neg_weights"], N_OUT, xn)
        probs, post, dec, qual = oracle.posterior(la, ln, lik, edges)
        return time.perf_counter() - t0, (ta, tn, probs)

    n_chk = min(64, n_sample)
    n_slow = min(512, n_sample)
    t_slow, (ta0, tn0, p0) = run_sample(chunks[0], n_slow)            # checker build (-O2 -ffp-contract=off)
    oracle.use_library(fast_lib)
    try:
        _, (ta1, tn1, p1) = run_sample(chunks[0], n_chk)
        assert np.array_equal(ta0[:n_chk], ta1) and np.array_equal(tn0[:n_chk], tn1), "speed build of the CPU port: tensors differ from the checker build"
        dev_max = float(np.abs(p0[:n_chk] - p1).max())
        assert dev_max < 1e-6, "speed build of the CPU port differs from the checker build by %g" % dev_max
        total_sites, total_t, first_probs = 0, 0.0, None
        for chunk in chunks:
            dt, (_, _, probs) = run_sample(chunk, n_sample)
            total_t += dt
            total_sites += min(n_sample, len(chunk.site_pos))
            if first_probs is None:
                first_probs = probs
            if total_t >= budget_s:
                break
        # the same build on ONE core (OpenMP team of one, one tensor-creation slice): the per-core figure next to the reference's 211 sites/s
        single = None
        try:
            import ctypes
            gomp = ctypes.CDLL("libgomp.so.1")
            gomp.omp_set_num_threads(1)
            n1 = min(256, n_sample)
            t1, _ = run_sample(chunks[0], n1, cores=1)
            gomp.omp_set_num_threads(int(cores))
            single = {"value": round(n1 / t1, 2), "unit": "sites/s", "cores": 1, "sample": "%d sites" % n1}
        except Exception:
            pass
    finally:
        oracle.use_library(None)
    ref_py = None
    try:
        rp = json.load(open(os.path.join(ROOT, "profiles", "reference_cpu_timing.json")))
        ref_py = {"sites_per_s_1_process": rp["runs"][0]["sites_per_s"], "sites_per_s_%d_processes" % rp["runs"][-1]["processes"]: rp["runs"][-1]["sites_per_s"],
                  "host": "build container, %d vCPU (NOT this box: the reference cannot travel)" % rp["host_cpus"], "what": rp["note"],
                  "source": "profiles/reference_cpu_timing.json (tools/time_reference.py)"}
    except Exception:
        pass
    return dict(value=round(total_sites / total_t, 2), unit="sites/s", cores=cores, kind="port",
                build="gcc -O3 -march=native -ffast-math -fopenmp (this host); max |dP| vs the checker build on %d sites = %.2g" % (n_chk, dev_max),
                checker_build={"value": round(n_slow / t_slow, 2), "unit": "sites/s", "cores": cores,
                               "build": "gcc -O2 -ffp-contract=off -fopenmp (the parity checker)", "sample": "%d sites" % n_slow},
                single_core=single, reference_python=ref_py,
                sample="%d sites of the same synthetic chunks (mpileup text of both passes -> tensors -> CvT + BiGRU -> "
                       "posterior), CPU port oracle/cto_oracle.c on every usable core: tensor creation in per-core site slices, "
                       "OpenMP over sites for the networks, %.1f s" % (total_sites, total_t)), first_probs


def sustained_distinct_leg(eng, base_chunks, models, lik, edges, min_bq, batch, n_chunks=245, n_oracle=512, stream_too=True):
    """configs[1]'s 1M-site job with EVERY chunk different (after the timed region, never in `value`): the timed region and `sustained`
    cycle `--pool` resident chunks; here n_chunks distinct packs (the pool's chunks x SynthChunk.variant: other BQ / MQ of every read-base,
    other positions - other tensors, other outputs) go through ONE engine, (1) all resident in HBM, one run_device per chunk back to back,
    (2) from page-locked host arrays through Engine.run_stream (uploads behind compute, results back in pinned buffers), results of (2)
    compared bit for bit with (1)'s, and the oracle (oracle/cto_oracle.c from the chunks' mpileup text) on a sample of n_oracle sites spread
    over ALL chunks (two or three per chunk).  Also what tests/test_gpu_distinct.py asserts on."""
    import numpy as np
    import torch
    import oracle
    from concurrent.futures import ThreadPoolExecutor
    from clairs_to_amd.pack import pin_arrays
    oracle.build()
    dev, nb = eng.device, len(base_chunks)
    cores = usable_cores()

    def make(i):
        return base_chunks[i % nb].variant(i // nb, shift=(i // nb) * 40000000)
    t0 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=max(1, min(8, cores))) as ex:
        var = list(ex.map(make, range(n_chunks)))
    assert len({int(v.site_pos[0]) for v in var}) == n_chunks
    packs = [eng.upload(v.arrays()) for v in var]
    sites = [torch.from_numpy(v.site_pos).to(dev) for v in var]
    prep_s = time.perf_counter() - t0
    keys = ("probs", "decision", "qual")
    for i in range(8):                                       # workspaces and the allocator's pools for these shapes
        eng.run_device(packs[i], sites[i])
    torch.cuda.synchronize()
    dt_res = None
    for _ in range(2):                                       # the first pass also grows the allocator's pools for the 245 kept outputs
        t1 = time.perf_counter()
        res = []
        for p, sp in zip(packs, sites):
            o = eng.run_device(p, sp)
            res.append({k: o[k] for k in keys})
        torch.cuda.synchronize()
        dt = time.perf_counter() - t1
        dt_res = dt if dt_res is None or dt < dt_res else dt_res
    n_sites = sum(int(sp.numel()) for sp in sites)
    out = {"chunks": n_chunks, "distinct_sites": n_sites, "pack_bytes_resident": int(sum(p.nbytes() for p in packs)),
           "resident": {"seconds": round(dt_res, 4), "sites_per_s": round(n_sites / dt_res, 1), "ms_per_chunk": round(dt_res / n_chunks * 1e3, 4), "passes": 2},
           "input_prep_s": round(prep_s, 1),
           "how_distinct": "%d generated chunks x %d variants each (SynthChunk.variant: every read-base's BQ moved by -6..+6, every fourth MQ lowered by 45, "
                           "positions shifted by 40 Mb per variant): no two chunks share a tensor" % (nb, -(-n_chunks // nb))}
    # ---- the oracle on a sample spread over every chunk ----
    per = max(1, n_oracle // n_chunks)
    picks = []                                               # (chunk, site index)
    for i in range(n_chunks):
        for j in range(per + (1 if i < n_oracle - per * n_chunks else 0)):
            picks.append((i, (i * 131 + j * 1777 + 7) % int(sites[i].numel())))
    cfg = dict(emb_dim=(16, 64, 128), heads=(1, 3, 4), depth=(1, 2, 3), n_out=N_OUT)
    ta, da, tn, dn = [], [], [], []
    win_of = {}
    for i, j in picks:
        v = var[i]
        if win_of.get("i") != i:                              # picks are grouped by chunk: one reference window at a time
            win_of = {"i": i, "w": v.ref_window()}
        ref, lo = win_of["w"]
        x = int(v.site_pos[j])
        c0 = int(np.searchsorted(v.col_pos, x - 16, side="left"))
        c1 = int(np.searchsorted(v.col_pos, x + 17, side="right"))
        for q, (tt, dd) in ((min_bq, (ta, da)), (0, (tn, dn))):
            t_, d_ = oracle.create_tensor(oracle.synth_mpileup_text(v, q, (c0, c1)), ref, lo, v.site_pos[j:j + 1])[:2]
            tt.append(t_)
            dd.append(d_)
    xa, xn = oracle.rescale(np.concatenate(ta), np.concatenate(da)), oracle.rescale(np.concatenate(tn), np.concatenate(dn))
    os.environ.setdefault("OMP_NUM_THREADS", str(cores))
    la = oracle.cvt_forward(models["aff_weights"], cfg, xa)
    ln = oracle.bigru_forward(models["neg_weights"], N_OUT, xn)
    probs_o, _, dec_o, qual_o = oracle.posterior(la, ln, lik, edges)
    got_p = np.stack([res[i]["probs"][j].cpu().numpy() for i, j in picks])
    got_d = np.stack([res[i]["decision"][j].cpu().numpy() for i, j in picks])
    out["oracle"] = {"sites": len(picks), "chunks_sampled": len({i for i, _ in picks}), "max_abs_dP": float(np.abs(got_p - probs_o).max()),
                     "decisions_equal_frac": float((got_d == np.asarray(dec_o).reshape(got_d.shape)).all(axis=1).mean())}
    # ---- the same chunks from the host through Engine.run_stream ----
    if stream_too:
        with ThreadPoolExecutor(max_workers=max(1, min(8, cores))) as ex:
            pinned = list(ex.map(lambda v: pin_arrays(v.arrays()), var))
        host_sites = [v.site_pos for v in var]
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        got = list(eng.run_stream(zip(pinned, host_sites), depth=2))
        dt_st = time.perf_counter() - t2
        same = all(np.array_equal(g[k], r[k].cpu().numpy()) for g, r in zip(got, res) for k in keys)
        out["run_stream"] = {"seconds": round(dt_st, 4), "sites_per_s": round(n_sites / dt_st, 1), "ms_per_chunk": round(dt_st / n_chunks * 1e3, 4),
                             "bit_equal_to_resident_pass": bool(same),
                             "includes": "PCIe both ways (28.6 MB of pack up, 140 B per site down per chunk), uploads on a copy stream behind the previous chunk's kernels"}
    return out


def split_mfma_leg(dev, packs, sites, batch, lik, edges, min_bq, ref_probs, ref_dec, probs_cpu, steps=20, warm=40):
    """EXPERIMENT, never in `value` (whose arithmetic stays f32): the step with both BiGRU recurrences + fc1 and the CvT's block
    GEMMs (64- and 128-channel stages) on split 16-bit operands (csrc/split_mfma.h, gru_split_kernel.h, cvt_gemm.h;
    CTO_GRU_SPLIT / CTO_CVT_SPLIT = f16|bf16 at model creation: a = hi + lo, three f16 / bf16 MFMA passes per product, fp32
    accumulation; states, gates, LayerNorm, softmax, residual stream fp32).  Per kind: the step's rate, the network kernels' times,
    and how far its probabilities are from the f32 path's on a whole chunk and (when the cpu_baseline leg ran) from the CPU port's
    on its sample."""
    import ctypes as C
    import numpy as np
    import torch
    from clairs_to_amd._lib import lib, check
    from clairs_to_amd.engine import Engine, synthetic_models
    out = {"note": "side channel: split-operand MFMA for the two BiGRU recurrences + fc1 and the CvT block GEMMs; embedding, classifier, "
                   "stage-1 block and everything outside the networks run the f32 product kernels; tests/test_gpu_split.py holds both kinds "
                   "to the oracle within the 1e-4 tolerance"}
    pool = len(packs)
    for kind in ("f16", "bf16"):
        models = synthetic_models(N_OUT, seed=0)          # fresh module objects: the arithmetic is chosen when a module creates its handle
        models["aff"].split_operands = kind               # (cto_*_create_ex: an argument, not the process environment)
        models["neg"].split_operands = kind
        eng = Engine(models["aff"], models["neg"], lik, edges, min_bq=min_bq, device=dev)
        for i in range(warm):
            eng.run_device(packs[i % pool], sites[i % pool])
        check(lib.cto_model_profile(eng.h_neg, 2))
        check(lib.cto_model_profile(eng.h_aff, 1))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for i in range(steps):
            eng.run_device(packs[i % pool], sites[i % pool])
        e1.record()
        torch.cuda.synchronize()
        check(lib.cto_model_profile(eng.h_neg, 0))
        check(lib.cto_model_profile(eng.h_aff, 0))
        ms = e0.elapsed_time(e1) / steps
        l2_ms, l2_macs, l1_ms, l1_macs = C.c_double(0.0), C.c_int64(0), C.c_double(0.0), C.c_int64(0)
        cvt_ms, cvt_macs = C.c_double(0.0), C.c_int64(0)
        check(lib.cto_model_profile_read(eng.h_aff, C.byref(cvt_ms), C.byref(cvt_macs)))
        check(lib.cto_model_profile_read(eng.h_neg, C.byref(l2_ms), C.byref(l2_macs)))
        check(lib.cto_model_profile_read_stage(eng.h_neg, 1, C.byref(l1_ms), C.byref(l1_macs)))
        got = eng.run_device(packs[0], sites[0])
        probs = got["probs"].cpu().numpy()
        dec = got["decision"].cpu().numpy()
        tf = 2.0 * l2_macs.value * batch / (l2_ms.value * 1e-3) / 1e12
        o = {"sites_per_s": round(batch / (ms * 1e-3), 1), "ms_per_step": round(ms, 4), "steps": steps,
             "gru_l2_ms": round(l2_ms.value, 4), "gru_l2_algorithmic_tflops": round(tf, 1), "gru_l1_ms": round(l1_ms.value, 4), "cvt_ms": round(cvt_ms.value, 4),
             "max_abs_dP_vs_f32_path": float(np.abs(probs - ref_probs).max()), "sites_compared": int(probs.shape[0]),
             "decisions_differing_from_f32_path": int(((dec[:, 0] != ref_dec[:, 0]) | ((dec[:, 1] & 3) != (ref_dec[:, 1] & 3))).sum())}
        if probs_cpu is not None:
            o["max_abs_dP_vs_cpu_sample"] = float(np.abs(probs[: probs_cpu.shape[0]] - probs_cpu).max())
        # its own roofline, against the 16-bit dense MFMA peak: three MFMA passes execute per algorithmic product
        exe_tf = 3.0 * tf
        o["roofline"] = {"bound": "mfma", "kernel": "k_gru_split<256,256,192,2,%s,true> (BiGRU layer 2 + fused fc1 on hi+lo %s operands)" % ("true" if kind == "f16" else "false", kind),
                         "achieved": round(exe_tf, 1), "peak": PEAK_16BIT_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": round(exe_tf / PEAK_16BIT_MFMA_TFLOPS, 4),
                         "algorithmic_tflops": round(tf, 1), "launch_ms": round(l2_ms.value, 4), "traffic": None,
                         "note": "achieved = 3 x algorithmic FLOPs / launch time (hi.hi + hi.lo + lo.hi); peak = dense bf16 / f16 MFMA (MI355X_MICROARCH.md:42)"}
        out[kind] = o
        del eng, models
    return out


def config_legs(dev, batch, steps=20, warm=40, pool=4):
    """After the timed region, never in `value`: the other single-GPU workloads BASELINE.json names, each as `steps` passes of the
    whole hot path over `pool` resident chunks of its generator preset with its model pair - sites/s by HIP events on
    the launch stream, per-stage kernel times from cto_model_profile - plus the clustered-candidate case of the ONT workload."""
    import ctypes as C
    import numpy as np
    import torch
    from concurrent.futures import ThreadPoolExecutor
    from clairs_to_amd._lib import lib, check, current_stream_ptr
    from clairs_to_amd.engine import Engine, synthetic_models, CVT_CONSTRUCTOR_CFG
    from clairs_to_amd.featurize import featurize
    from clairs_to_amd.synth import SynthChunk, PLATFORMS, likelihood_table, lik_and_edges

    def frac(ms, macs_):
        tf = 2.0 * macs_ * batch / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
        return {"ms": round(ms, 4), "tflops": round(tf, 2), "frac_of_fp32_mfma_peak": round(tf / PEAK_FP32_MFMA_TFLOPS, 4)}

    def leg(platform, K, cvt_cfg=None, spacing=None, what=""):
        models = synthetic_models(K, seed=0, cvt_cfg=cvt_cfg)
        lik, edges = lik_and_edges(likelihood_table(K), K)
        pf = PLATFORMS[platform]
        # Illumina: the NEG tensor files are symlinks to the AFF ones (run_clairs_to:1248-1252)
        eng = Engine(models["aff"], models["neg"], lik, edges, min_bq=pf["min_bq"], device=dev, neg_reads_aff=(platform == "ilmn"))
        kw = {} if spacing is None else {"spacing": spacing}
        with ThreadPoolExecutor(max_workers=max(1, min(pool, usable_cores()))) as ex:
            chunks = list(ex.map(lambda i: SynthChunk.for_platform(platform, batch, seed=pf["seed"] + 7 * i, start=100000 + i * 3000000, **kw), range(pool)))
        packs = [eng.upload(ch.arrays()) for ch in chunks]
        sites = [torch.from_numpy(ch.site_pos).to(dev) for ch in chunks]
        for i in range(warm):              # ~80 ms of work: the seconds of host-side synthesis before it let the shader clock drop
            eng.run_device(packs[i % pool], sites[i % pool])
        check(lib.cto_model_profile(eng.h_aff, 1))
        check(lib.cto_model_profile(eng.h_neg, 2))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for i in range(steps):
            eng.run_device(packs[i % pool], sites[i % pool])
        e1.record()
        torch.cuda.synchronize()
        check(lib.cto_model_profile(eng.h_aff, 0))
        check(lib.cto_model_profile(eng.h_neg, 0))
        ms = e0.elapsed_time(e1) / steps
        cvt_ms, cvt_macs, l2_ms, l2_macs, l1_ms, l1_macs = C.c_double(0.0), C.c_int64(0), C.c_double(0.0), C.c_int64(0), C.c_double(0.0), C.c_int64(0)
        check(lib.cto_model_profile_read(eng.h_aff, C.byref(cvt_ms), C.byref(cvt_macs)))
        check(lib.cto_model_profile_read(eng.h_neg, C.byref(l2_ms), C.byref(l2_macs)))
        check(lib.cto_model_profile_read_stage(eng.h_neg, 1, C.byref(l1_ms), C.byref(l1_macs)))
        depth = float(np.mean([np.diff(ch.col_off).mean() for ch in chunks]))
        out = {"workload": what, "platform": platform, "K": K, "min_bq_aff": pf["min_bq"], "neg_reads_aff": platform == "ilmn",
               "mean_depth": round(depth, 1), "pack_bytes_per_chunk": int(sum(p_.nbytes() for p_ in packs) / pool),
               "pack_columns_per_candidate": round(float(np.mean([ch.col_pos.size / batch for ch in chunks])), 2),
               "sites_per_s": round(batch / (ms * 1e-3), 1), "ms_per_step": round(ms, 4), "steps": steps,
               "mflop_per_site": round(2.0 * eng.macs_per_site / 1e6, 2),
               "end_to_end_tflops": round(2.0 * eng.macs_per_site * batch / (ms * 1e-3) / 1e12, 2),
               "stage_fracs": {"gru_l2": frac(l2_ms.value, l2_macs.value), "gru_l1": frac(l1_ms.value, l1_macs.value),
                               "cvt": frac(cvt_ms.value, cvt_macs.value)}}
        return out, eng, packs, sites

    res = {}
    plan = [("configs3_illumina_snv", "ilmn", 4, None, None, "BASELINE configs[3]: Illumina 50x, SNV model pair (K=4), NEG network reads the AFF tensor"),
            ("configs3_illumina_indel", "ilmn", 6, None, None, "BASELINE configs[3]: Illumina 50x, indel model pair (K=6)"),
            ("configs4_hifi_snv", "hifi", 4, None, None, "BASELINE configs[4], one GPU's share: PacBio HiFi 75x, SNV pair (no third model exists in the reference)"),
            ("configs4_hifi_indel", "hifi", 6, None, None, "BASELINE configs[4], one GPU's share: PacBio HiFi 75x, indel pair (K=6)"),
            ("configs1_ont_indel", "ont", 6, None, None, "ONT 50x with the indel model pair (K=6)"),
            ("ont_snv_constructor_default_cvt", "ont", 4, CVT_CONSTRUCTOR_CFG, None,
             "ONT 50x, SNV, AFF network with clairs/model.py:153-184's constructor defaults (emb 32/64/128, heads 1/3/6, depth 1/2/10) - "
             "what a pickled SNV module may carry instead of the predict.py:520-553 configuration"),
            ("ont_snv_clustered_candidates", "ont", 4, None, 3, "ONT 50x, SNV, candidates 3 bp apart on average (30 of 33 window columns shared)")]
    for key, platform, K, cvt_cfg, spacing, what in plan:
        out, eng, packs, sites = leg(platform, K, cvt_cfg, spacing, what)
        if spacing is not None:
            # tensor creation on clustered candidates, both forms, on preallocated outputs (one event pair per call): the one-kernel
            # form re-reads shared columns once per candidate (out of L2), the two-stage form histograms every column once and gathers
            # (what run_device picks below 8 pack columns per candidate).  Algorithmic bytes: the pack once + two fp32 tensors per site.
            pool = len(packs)
            f1 = featurize(packs[0], sites[0], eng.min_bq, 50, fused=True)
            f2 = featurize(packs[0], sites[0], eng.min_bq, 50, fused=False)
            nk = max(max(p_.n_keys for p_ in packs), 1)
            nc = max(max(p_.n_cols for p_ in packs), 1)
            kc = torch.empty((nk,), dtype=torch.int32, device=dev)
            kf = torch.empty((nk, 2), dtype=torch.int32, device=dev)
            colvec = torch.empty((nc, 72), dtype=torch.int16, device=dev)
            coldepth = torch.empty((nc, 2), dtype=torch.int32, device=dev)
            sp = current_stream_ptr()

            def one(j):
                check(lib.cto_featurize_sites(C.byref(packs[j].view), sites[j].data_ptr(), batch, int(eng.min_bq), 50, f1.x_aff.data_ptr(),
                                              f1.x_neg.data_ptr(), None, None, f1.site_info.data_ptr(), f1.site_colvec.data_ptr(),
                                              f1.sitefirst.data_ptr(), kc.data_ptr(), kf.data_ptr(), sp))

            def two(j):
                check(lib.cto_featurize_columns(C.byref(packs[j].view), int(eng.min_bq), colvec.data_ptr(), coldepth.data_ptr(), kc.data_ptr(), sp))
                check(lib.cto_gather_windows(C.byref(packs[j].view), colvec.data_ptr(), coldepth.data_ptr(), sites[j].data_ptr(), batch,
                                             int(eng.min_bq), 50, f2.x_aff.data_ptr(), f2.x_neg.data_ptr(), None, None, f2.site_info.data_ptr(),
                                             f2.sitefirst.data_ptr(), kf.data_ptr(), sp))
            tc = {}
            nbytes = out["pack_bytes_per_chunk"] + 2 * 33 * 34 * 4 * batch
            for name, fn in (("one_kernel", one), ("two_stage", two)):
                evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(12)]
                for i in range(-2, 12):
                    if i >= 0:
                        evs[i][0].record()
                    fn(i % pool)
                    if i >= 0:
                        evs[i][1].record()
                torch.cuda.synchronize()
                t = sum(a.elapsed_time(b) for a, b in evs) / 12
                tc[name] = {"ms": round(t, 4), "gb_per_s": round(nbytes / (t * 1e-3) / 1e9, 1), "frac_of_hbm_peak": round(nbytes / (t * 1e-3) / 1e9 / 8000.0, 4)}
            tc["bytes_per_launch"] = int(nbytes)
            tc["equal"] = bool(torch.equal(f1.x_aff, f2.x_aff) and torch.equal(f1.x_neg, f2.x_neg))
            tc["run_device_uses"] = "two_stage" if packs[0].n_cols < 8 * batch else "one_kernel"
            out["tensor_creation"] = tc
        res[key] = out
        del eng, packs, sites
        torch.cuda.empty_cache()
    return res


def realign_leg(dev, n_windows=1500, seed=20260930):
    """configs[3]'s `realign_reads` leg (SURVEY.md 8f #4b), after the timed region and never in `value`: the native half of the
    Illumina realignment filter (reference: realign_reads(...) of src/realign/realigner.cpp:782-857, one call per window from one
    process per low-QUAL call) on synthetic windows (clairs_to_amd/synth_realign.py: 1-60 reads of 20-250 bases, 1-18 haplotypes),
    the whole set handed to cto_realign_windows in ONE call: host SSE2 on 1 thread and on every usable core, and the device form
    (k_fast_pass + k_sw_ends, traceback / composition on the cores).  The three outputs are compared for equality here; the tests hold
    them to the reference compiled into oracle/_ref."""
    import numpy as np
    from clairs_to_amd.synth_realign import gen_window
    from clairs_to_amd.realign_reads import realign_windows
    rng = np.random.default_rng(seed)
    ws = [gen_window(rng) for _ in range(n_windows)]
    args = [(w["seqs"], w["positions"], w["cigars"], w["reference"], w["haplotypes"], w["ref_start"], w["ref_prefix"], w["ref_suffix"]) for w in ws]
    reads = sum(len(w["seqs"]) for w in ws)
    cores = usable_cores()
    out, legs = {}, {}

    def run(name, where, threads, reps):
        best, got, st = None, None, {}
        for _ in range(reps):
            st_i = {}
            t = time.perf_counter()
            got = realign_windows(args, where=where, threads=threads, stats=st_i)
            dt = time.perf_counter() - t
            if best is None or dt < best:
                best, st = dt, st_i                                                   # every figure of a leg is of its fastest call
        c_s = (st.get("device_stage_ms", 0.0) + st.get("host_ms", 0.0)) * 1e-3         # inside cto_realign_windows
        legs[name] = {"seconds": round(best, 4), "windows_per_s": round(n_windows / best, 1), "reads_per_s": round(reads / best, 1), "host_threads": threads,
                      "c_call_seconds": round(c_s, 4), "reads_per_s_c_call": round(reads / c_s, 1) if c_s > 0 else None}
        return got, st
    sub = 300
    t = time.perf_counter()
    one = realign_windows(args[:sub], where="host", threads=1)
    dt1 = time.perf_counter() - t
    r1 = sum(len(w["seqs"]) for w in ws[:sub])
    legs["host_sse2_1_thread"] = {"seconds": round(dt1, 4), "windows_per_s": round(sub / dt1, 1), "reads_per_s": round(r1 / dt1, 1),
                                  "host_threads": 1, "sample": "first %d windows" % sub}
    host, _ = run("host_sse2_all_cores", "host", cores, 2)
    devo, st = run("device", "device", cores, 5)
    legs["device"].update({
        "kernel_fast_pass_ms": round(st["fast_pass_ms"], 3), "kernel_sw_ms": round(st["sw_ms"], 3),
        "fast_pass_pairs": int(st["fast_pairs"]), "sw_alignments": int(st["sw_pairs"]), "sw_cells": int(st["sw_cells"]),
        "sw_gcups": round(st["sw_cells"] / (st["sw_ms"] * 1e-3) / 1e9, 2) if st["sw_ms"] > 0 else None,
        "device_stage_wall_ms": round(st["device_stage_ms"], 2), "host_stage_wall_ms": round(st["host_ms"], 2),
        "kernel_traceback_ms": round(st["traceback_ms"], 3), "tracebacks": int(st["tracebacks"]), "tracebacks_left_to_host": int(st["tracebacks_declined"]),
        "windows_on_host": int(st["host_windows"])})
    out = {"workload": "%d synthetic Illumina realignment windows, %d reads (BASELINE configs[3]: realign_reads path)" % (n_windows, reads),
           "cores": cores, "outputs_equal": bool(host == devo and host[:sub] == one), **legs,
           "note": "one cto_realign_windows call per figure (the fastest of 2 host / 5 device calls, all its figures from that call); seconds / reads_per_s include the Python side (every read and CIGAR of the list joined into one buffer each, the outputs split again: seconds - c_call_seconds), c_call_seconds / reads_per_s_c_call are the C call alone (what a C or C++ orchestrator pays); the reference's own library on one core runs "
                   "this generator's windows at ~3.8 k reads/s (tools/realign_bench.py, build container); kernel times are HIP events, "
                   "sw_gcups = reference x query cells of every alignment / k_sw time (both passes of an alignment counted once); tracebacks = banded tracebacks run by k_banded (every haplotype against the reference + the pair each unplaced read picks)"}
    return out


def hapfilter_leg(repeats=8):
    """the long-read post-calling filter (src/haplotype_filtering.py:882; SURVEY.md 8f #4a) on the committed reference-generated
    fixture (tests/golden/hapfilter.json.gz: SNV and indel pass of a simulated haplotagged BAM), file to file through
    clairs_to_amd.haplotype_filtering.haplotype_filter (C evaluation of every read-level rule of a job in one call) - calls/s on the
    host cores; host code by nature (strings and per-read sets), no device form."""
    import contextlib
    import gzip
    import io
    import tempfile
    from argparse import Namespace
    from clairs_to_amd.haplotype_filtering import haplotype_filter
    fx = os.path.join(ROOT, "tests", "golden", "hapfilter.json.gz")
    if not os.path.exists(fx):
        return {"error": "tests/golden/hapfilter.json.gz not present"}
    with gzip.open(fx, "rt") as f:
        g = json.load(f)
    res = {}
    with tempfile.TemporaryDirectory() as tmp:
        ref = g["ref"]
        open(os.path.join(tmp, "ref.fa"), "w").write(">chr1\n" + ref + "\n")
        open(os.path.join(tmp, "ref.fa.fai"), "w").write("chr1\t%d\t6\t%d\t%d\n" % (len(ref), len(ref), len(ref) + 1))
        open(os.path.join(tmp, "germline.vcf"), "w").write(g["germline_vcf"])
        for mode in ("snv", "indel"):
            m = g["modes"][mode]
            open(os.path.join(tmp, "pileup_%s.vcf" % mode), "w").write(m["pileup_vcf"])
            open(os.path.join(tmp, "mp_%s.txt" % mode), "w").write(m["mpileup"])
            a = Namespace(tumor_bam_fn="unused.bam", ref_fn=os.path.join(tmp, "ref.fa"), ctg_name="chr1",
                          pileup_vcf_fn=os.path.join(tmp, "pileup_%s.vcf" % mode), output_vcf_fn=os.path.join(tmp, "out_%s.vcf" % mode),
                          germline_vcf_fn=os.path.join(tmp, "germline.vcf"), output_dir=os.path.join(tmp, "work_" + mode), threads=usable_cores(),
                          input_filter_tag=None, show_ref=False, samtools="samtools", mpileup_fn=os.path.join(tmp, "mp_%s.txt" % mode),
                          apply_haplotype_filtering=True, min_mq=20, min_bq=0, min_alt_coverage=2, is_indel=(mode == "indel"), test_pos=None,
                          flanking=100, haplotype_chunk_max_sites=7, haplotype_chunk_max_span=5000000, disable_read_start_end_filtering=False)
            best, n_calls = None, 0
            for _ in range(repeats):
                t = time.perf_counter()
                with contextlib.redirect_stdout(io.StringIO()):        # the stage prints its tallies; the bench prints ONE line
                    r = haplotype_filter(a)
                dt = time.perf_counter() - t
                n_calls = len(r)
                best = dt if best is None or dt < best else best
            same = open(a.output_vcf_fn).read() == m["out_vcf"]
            res[mode] = {"calls": n_calls, "seconds": round(best, 4), "calls_per_s": round(n_calls / best, 1), "vcf_equals_reference": bool(same),
                         "mpileup_bytes": len(m["mpileup"])}
    res["cores"] = usable_cores()
    res["note"] = "pileup VCF + germline VCF + nine-column mpileup text in, tagged VCF out (best of %d passes); the reference starts one `samtools mpileup` per call" % repeats
    try:
        res["many_calls"] = hapfilter_many_calls()
    except Exception as e:                                    # a leg after the timed region must not cost the line
        res["many_calls"] = {"error": repr(e)[:300]}
    return res


def _hapsim_job(job):
    """worker of hapfilter_many_calls (spawned process: input synthesis only): one simulated contig's inputs written under d"""
    seed, ctg, d, modes = job
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import gen_hapfilter_wide as gw
    import hapsim
    sim = hapsim.simulate(seed=seed)
    os.makedirs(d, exist_ok=True)
    ref = sim["ref"]
    open(os.path.join(d, "ref.fa"), "w").write(">%s\n%s\n" % (ctg, ref))
    open(os.path.join(d, "ref.fa.fai"), "w").write("%s\t%d\t%d\t%d\t%d\n" % (ctg, len(ref), len(ctg) + 2, len(ref), len(ref) + 1))
    open(os.path.join(d, "germline.vcf"), "w").write(gw.germline_vcf(sim, ctg))
    n = {}
    for mode in modes:
        v, t = gw.inputs_for(sim, ctg, mode)
        open(os.path.join(d, "pileup_%s.vcf" % mode), "w").write(v)
        open(os.path.join(d, "mp_%s.txt" % mode), "w").write(t)
        n[mode] = (sum(1 for r in v.split("\n") if r and r[0] != "#" and "\tPASS\t" in r), len(t))
    return ctg, d, n


def hapfilter_many_calls(n_contigs=216, indel_every=4):
    """a throughput figure for the long-read filter: n_contigs simulated contigs (tests/golden/hapsim.py, one seed each: the generator of the
    reference-made fixtures), >= 5 000 PASS calls in the SNV pass, every contig one haplotype_filter job as a real run has one per contig,
    all usable cores inside each job; beside it the reference's own cost per call (profiles/reference_hapfilter_timing.json, build container)."""
    import contextlib
    import io
    import multiprocessing as mp
    import shutil
    import tempfile
    from argparse import Namespace
    from concurrent.futures import ProcessPoolExecutor
    from clairs_to_amd.haplotype_filtering import haplotype_filter
    cores = usable_cores()
    tmp = tempfile.mkdtemp(prefix="cto_hapfilter_")
    try:
        t0 = time.perf_counter()
        jobs = [(5000 + k, "ctg%d" % (k + 1), os.path.join(tmp, "c%d" % k), ("snv", "indel") if k % indel_every == 0 else ("snv",)) for k in range(n_contigs)]
        with ProcessPoolExecutor(max_workers=max(1, cores), mp_context=mp.get_context("spawn")) as ex:
            made = list(ex.map(_hapsim_job, jobs, chunksize=2))
        prep_s = time.perf_counter() - t0
        out = {}
        for mode in ("snv", "indel"):
            sel = made if mode == "snv" else made[::indel_every]
            calls = sum(n[mode][0] for _, _, n in sel)
            text_bytes = sum(n[mode][1] for _, _, n in sel)
            best = None
            for _ in range(2):
                t1 = time.perf_counter()
                evaluated = 0
                for ctg, d, n in sel:
                    a = Namespace(tumor_bam_fn="unused.bam", ref_fn=os.path.join(d, "ref.fa"), ctg_name=ctg, pileup_vcf_fn=os.path.join(d, "pileup_%s.vcf" % mode),
                                  output_vcf_fn=os.path.join(d, "out_%s.vcf" % mode), germline_vcf_fn=os.path.join(d, "germline.vcf"),
                                  output_dir=os.path.join(d, "work_" + mode), threads=cores, input_filter_tag=None, show_ref=False, samtools="samtools",
                                  mpileup_fn=os.path.join(d, "mp_%s.txt" % mode), apply_haplotype_filtering=True, min_mq=20, min_bq=0, min_alt_coverage=2,
                                  is_indel=(mode == "indel"), test_pos=None, flanking=100, haplotype_chunk_max_sites=200, haplotype_chunk_max_span=5000000,
                                  disable_read_start_end_filtering=False)
                    with contextlib.redirect_stdout(io.StringIO()):
                        evaluated += len(haplotype_filter(a))
                dt = time.perf_counter() - t1
                best = dt if best is None or dt < best else best
            out[mode] = {"contig_jobs": len(sel), "pass_calls": calls, "calls_evaluated": evaluated, "seconds": round(best, 4), "calls_per_s": round(evaluated / best, 1),
                         "mpileup_text_mb": round(text_bytes / 1e6, 1), "text_mb_per_s": round(text_bytes / 1e6 / best, 1)}
        out["cores"] = cores
        out["input_synthesis_s"] = round(prep_s, 1)
        try:
            rp = json.load(open(os.path.join(ROOT, "profiles", "reference_hapfilter_timing.json")))
            r1 = rp["runs"]["chunk_mode_threads_1"]
            out["reference_python"] = {"calls_per_s_chunk_mode": {m: r1[m]["calls_per_s"] for m in r1}, "calls_per_s_per_call_mode": {m: v["calls_per_s"] for m, v in rp["runs"]["percall_mode_threads_1"].items()},
                                       "host": "build container, %d vCPU (NOT this box)" % rp["host_cpus"], "what": rp["what"], "source": "profiles/reference_hapfilter_timing.json (tools/time_reference_hapfilter.py)"}
        except Exception:
            pass
        out["note"] = "one haplotype_filter job per contig, one after the other, all cores inside a job (best of 2 passes over all contigs); host code by nature - no device form; mpileup text pre-made on both sides (BAM decoding excluded)"
        return out
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def self_launch(n):
    """Re-exec this command line under torch.distributed.run with n ranks on this node (rendezvous on 127.0.0.1, a free
    port); the ranks' stdout is ours, so rank 0's JSON line is the only line printed.  Returns the launcher's exit code."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"),
               OMP_NUM_THREADS=os.environ.get("OMP_NUM_THREADS", "1"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=100,
                    help="untimed steps before the timed region; the default (0.2 s of work) is what the shader clock needs to settle after the "
                         "seconds of host-side input synthesis in front of it (the first ~60 steps run the kernels ~2-3 %% slower)")
    ap.add_argument("--pool", type=int, default=16,
                    help="distinct synthetic chunks resident in HBM per rank (16 x 28.6 MB of packs = 458 MB: more than the 256 MB "
                         "Infinity Cache, so the tensor-creation stage really reads HBM)")
    ap.add_argument("--batch", type=int, default=BATCH)
    ap.add_argument("--watchdog", type=int, default=1500, help="N > 1: seconds after which a rank that is still running gives up with a message")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=4096)
    ap.add_argument("--no-sustained", action="store_true", help="skip the 245-step run and the per-stage kernel times after the timed region")
    ap.add_argument("--no-split", action="store_true", help="skip the split-operand experiment leg (`split_mfma`; never part of `value`)")
    ap.add_argument("--no-configs", action="store_true", help="skip the legs on the other BASELINE configs' single-GPU workloads (Illumina, HiFi, "
                    "K = 6, the constructor-default CvT, clustered candidates)")
    ap.add_argument("--no-live-traffic", action="store_true", help="`roofline.traffic` from the newest committed PMC digest instead of two "
                    "rocprofv3 --pmc child runs of this file (FETCH_SIZE, WRITE_SIZE) after the timed region")
    ap.add_argument("--no-postfilters", action="store_true", help="skip the post-calling filter legs (`configs3_realign`: the Illumina realigner, host and "
                    "device form; `hapfilter`: the long-read haplotype filter) - after the timed region, never part of `value`")
    ap.add_argument("--no-e2e", action="store_true", help="skip the file-to-file legs (mpileup text -> VCF, BAM -> VCF)")
    ap.add_argument("--e2e-chunks", type=int, default=96, help="chunk files of the mpileup-text leg (the BAM leg uses a third as many)")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    if args.gpus > 1 and int(os.environ.get("WORLD_SIZE", "1")) == 1:
        # `python bench.py --gpus N` with no launcher around it: become the launcher (one rank per GPU, RCCL over xGMI)
        sys.exit(self_launch(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.exit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks" % (args.gpus, world))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # test hooks (never set by the driver): run several ranks on ONE GPU over gloo to exercise the multi-rank code path
    backend = os.environ.get("CTO_BENCH_BACKEND", "nccl")
    if "CTO_BENCH_DEVICE" in os.environ:
        local_rank = int(os.environ["CTO_BENCH_DEVICE"])
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        # a rendezvous or a first collective that hangs (IPC mode, a dead peer, a GPU another rank also took) must end in a message,
        # not in the driver's time-out: the process group times out on its own, and a watchdog thread ends the rank if the whole
        # run overstays --watchdog seconds, naming the stage it was in
        import datetime
        import threading
        stage = {"name": "init_process_group(%s)" % backend}

        def _watchdog():
            time.sleep(args.watchdog)
            sys.stderr.write("[bench.py rank %d/%d on cuda:%d] watchdog: still in stage '%s' after %d s - giving up\n"
                             % (rank, world, local_rank, stage["name"], args.watchdog))
            sys.stderr.flush()
            os._exit(3)
        threading.Thread(target=_watchdog, daemon=True).start()
        tmo = datetime.timedelta(seconds=min(600, args.watchdog))
        try:
            dist.init_process_group(backend, device_id=dev, timeout=tmo) if backend == "nccl" else dist.init_process_group(backend, timeout=tmo)
        except Exception as e:
            sys.exit("[bench.py rank %d/%d on cuda:%d] init_process_group(%s) failed: %r (MASTER_ADDR=%s MASTER_PORT=%s HSA_ENABLE_IPC_MODE_LEGACY=%s)"
                     % (rank, world, local_rank, backend, e, os.environ.get("MASTER_ADDR"), os.environ.get("MASTER_PORT"),
                        os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY")))
    else:
        stage = {"name": "single"}

    from clairs_to_amd._lib import lib, check
    from clairs_to_amd.engine import Engine, synthetic_models
    from clairs_to_amd.synth import SynthChunk, likelihood_table, lik_and_edges
    import ctypes as C

    min_bq = 20                                              # ONT sup models: shared/param.py min_bq_dict
    models = synthetic_models(N_OUT, seed=0)
    lik, edges = lik_and_edges(likelihood_table(N_OUT), N_OUT)
    eng = Engine(models["aff"], models["neg"], lik, edges, min_bq=min_bq, device=dev)

    # ---- synthetic job: `pool` chunks per rank, packs resident in HBM ----
    stage["name"] = "synthesising %d chunks" % args.pool
    chunks, packs, sites = [], [], []
    # the generator is vectorised numpy whose random draws release the GIL: a few threads per rank cut the set-up (16 chunks x ~5 s
    # of one core each) to a fraction - 8 ranks on a 16-core allotment must still start well inside the driver's patience
    from concurrent.futures import ThreadPoolExecutor
    gen_threads = max(1, min(8, usable_cores() // max(1, world)))
    with ThreadPoolExecutor(max_workers=gen_threads) as ex:
        chunks = list(ex.map(lambda i: SynthChunk(args.batch, seed=20260928 + 1000 * rank + i, start=100000 + (rank * args.pool + i) * 2000000),
                             range(args.pool)))
    for ch in chunks:
        packs.append(eng.upload(ch.arrays()))
        sites.append(torch.from_numpy(ch.site_pos).to(dev))
    pack_bytes = sum(p.nbytes() for p in packs) / len(packs)
    # per-site outputs go to every rank over xGMI (262 KB per rank and step).  The gather is asynchronous and double-buffered:
    # it runs on RCCL's stream under the next step's kernels and is only waited for when its buffer comes up for reuse.
    gather_buf = [torch.empty((world * args.batch, 2 * N_OUT, 2), dtype=torch.float32, device=dev) for _ in range(2)] if world > 1 else None   # rank-major = genomic order
    pending = [None, None]

    def step(i):
        out = eng.run_device(packs[i % args.pool], sites[i % args.pool])
        if world > 1:
            if pending[i & 1] is not None:
                pending[i & 1][0].wait()
            pending[i & 1] = (dist.all_gather_into_tensor(gather_buf[i & 1], out["probs"], async_op=True), out["probs"])
        return out

    def sync():
        if world > 1:
            for k in (0, 1):
                if pending[k] is not None:
                    pending[k][0].wait()
                    pending[k] = None
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # set-up, not a step: the first pass sizes the per-model workspaces (hipMalloc) and the caching allocator's pools
    eng.run_device(packs[0], sites[0])
    torch.cuda.synchronize()
    stage["name"] = "first collective (communicator set-up)"
    if world > 1:
        # communicator set-up (seconds on the first collective) never lands in the timed region, whatever --warmup is
        dist.all_gather_into_tensor(gather_buf[0], torch.zeros((args.batch, 2 * N_OUT, 2), dtype=torch.float32, device=dev))
        dist.barrier()
    stage["name"] = "warm-up + timed steps"
    for i in range(args.warmup):
        step(i)
    sync()
    check(lib.cto_model_profile(eng.h_neg, 1))
    t0 = time.perf_counter()
    for i in range(args.steps):
        out = step(i)
    sync()
    dt = time.perf_counter() - t0
    check(lib.cto_model_profile(eng.h_neg, 0))
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # ---- proof that the exchange step ran between `world` distinct ranks: every rank's last per-site block must sit in its
    # rank-major slot of the gathered buffer, bit for bit, and every rank reports the device it computed on ----
    census, gather_ok = None, None
    if world > 1:
        last = (args.steps - 1) & 1
        mine = gather_buf[last][rank * args.batch:(rank + 1) * args.batch]
        ok_local = bool(torch.equal(mine, out["probs"]))
        sums = torch.stack([gather_buf[last][r * args.batch:(r + 1) * args.batch].double().sum() for r in range(world)])
        own = torch.zeros(world, dtype=torch.float64, device=dev)
        own[rank] = out["probs"].double().sum()
        dist.all_reduce(own)                                  # own[r] = checksum rank r computed of its own block
        flag = torch.tensor([1 if (ok_local and torch.equal(sums, own)) else 0], device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        gather_ok = bool(flag.item())
        props = torch.cuda.get_device_properties(dev)
        me = {"rank": rank, "device": "cuda:%d" % local_rank, "name": props.name,
              "pci": getattr(props, "pci_bus_id", None), "uuid": str(getattr(props, "uuid", ""))}
        census = [None] * world
        dist.all_gather_object(census, me)

    # ---- roofline of the dominant kernel (BiGRU layer-2 recurrent kernel), live HIP-event timing ----
    mean_ms, macs = C.c_double(0.0), C.c_int64(0)
    n_meas = check(lib.cto_model_profile_read(eng.h_neg, C.byref(mean_ms), C.byref(macs)))
    flops_per_launch = 2.0 * macs.value * args.batch
    achieved = flops_per_launch / (mean_ms.value * 1e-3) / 1e12 if n_meas > 0 and mean_ms.value > 0 else 0.0

    # ---- after the timed region, never in `value`: (1) the whole job configs[1] names, 245 steps = 1 003 520 sites, timed in 20-step
    # windows by events on the launch stream (no host sync inside); (2) every stage's kernel time from live HIP events ----
    sustained, stage_fracs = None, None
    if not args.no_sustained:
        # every rank runs the whole job on its shard (weak scaling: 245 steps each), the exchange step included when there are several;
        # bracketed like the timed region (barrier + synchronize on both sides, MAX over ranks)
        stage["name"] = "sustained run"
        n_sus, win = 245, 20
        marks = [torch.cuda.Event(enable_timing=True) for _ in range(n_sus // win + 2)]
        sync()
        t1 = time.perf_counter()
        marks[0].record()
        k = 1
        for i in range(n_sus):
            step(i)
            if (i + 1) % win == 0:
                marks[k].record()
                k += 1
        marks[k].record()
        sync()
        dt_sus = time.perf_counter() - t1
        if world > 1:
            t = torch.tensor([dt_sus], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt_sus = float(t.item())
        wins = [marks[j].elapsed_time(marks[j + 1]) / win for j in range(n_sus // win)]
        sustained = {"steps": n_sus, "sites": world * n_sus * args.batch, "seconds": round(dt_sus, 4), "ms_per_step": round(dt_sus / n_sus * 1e3, 4),
                     "sites_per_s": round(world * n_sus * args.batch / dt_sus, 1), "window_steps": win,
                     "ms_per_step_min_window": round(min(wins), 4), "ms_per_step_max_window": round(max(wins), 4),
                     "note": ("configs[1]'s whole job (1M sites in %d-site steps) back to back on the resident packs" % args.batch) +
                             (" of every rank, the all_gather of each step included; seconds = MAX over ranks, windows = rank 0's" if world > 1 else "") +
                             "; windows timed by events on the launch stream"}
    if world == 1 and not args.no_sustained:
        l1_ms, l1_macs = C.c_double(0.0), C.c_int64(0)
        check(lib.cto_model_profile(eng.h_aff, 1))
        check(lib.cto_model_profile(eng.h_neg, 2))            # layer 1 bracketed too
        for i in range(20):
            step(i)
        torch.cuda.synchronize()
        check(lib.cto_model_profile(eng.h_aff, 0))
        check(lib.cto_model_profile(eng.h_neg, 0))
        cvt_ms, cvt_macs, l2_ms, l2_macs = C.c_double(0.0), C.c_int64(0), C.c_double(0.0), C.c_int64(0)
        check(lib.cto_model_profile_read(eng.h_aff, C.byref(cvt_ms), C.byref(cvt_macs)))
        check(lib.cto_model_profile_read(eng.h_neg, C.byref(l2_ms), C.byref(l2_macs)))
        check(lib.cto_model_profile_read_stage(eng.h_neg, 1, C.byref(l1_ms), C.byref(l1_macs)))

        def frac(ms, macs_):
            tf = 2.0 * macs_ * args.batch / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
            return {"ms": round(ms, 4), "tflops": round(tf, 2), "frac_of_fp32_mfma_peak": round(tf / PEAK_FP32_MFMA_TFLOPS, 4)}
        stage_fracs = {"gru_l2": frac(l2_ms.value, l2_macs.value), "gru_l1": frac(l1_ms.value, l1_macs.value),
                       "cvt": frac(cvt_ms.value, cvt_macs.value),
                       "note": "mean of 20 launches each, HIP events on the launch stream (cto_model_profile); cvt = its three launches (one per stage, all blocks of the stage inside)"}

    # ---- secondary roofline: pileup-tensor creation (HBM-bound stage), timed on its own after the timed region ----
    # (the C entry on preallocated outputs, one event pair per launch: through featurize() the eight allocations of a call take as
    # long as the kernel, and a slow host core would be what is measured)
    from clairs_to_amd.featurize import featurize
    from clairs_to_amd._lib import current_stream_ptr
    f0 = featurize(packs[0], sites[0], min_bq, 50, fused=True)
    nk_max = max(max(p.n_keys for p in packs), 1)
    kc = torch.empty((nk_max,), dtype=torch.int32, device=dev)
    kf = torch.empty((nk_max, 2), dtype=torch.int32, device=dev)
    n_feat = 16
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n_feat)]
    sptr = current_stream_ptr()
    for i in range(-2, n_feat):
        j = i % args.pool
        if i >= 0:
            evs[i][0].record()
        check(lib.cto_featurize_sites(C.byref(packs[j].view), sites[j].data_ptr(), args.batch, int(min_bq), 50, f0.x_aff.data_ptr(),
                                      f0.x_neg.data_ptr(), None, None, f0.site_info.data_ptr(), f0.site_colvec.data_ptr(),
                                      f0.sitefirst.data_ptr(), kc.data_ptr(), kf.data_ptr(), sptr))
        if i >= 0:
            evs[i][1].record()
    torch.cuda.synchronize()
    feat_ms = sum(a.elapsed_time(b) for a, b in evs) / n_feat
    # algorithmic bytes: the pack once (4 B per read-base + column tables) + two fp32 [33][34] tensors per site
    feat_bytes = pack_bytes + 2 * 33 * 34 * 4 * args.batch
    # candidate extraction (the front of a REGION job of cto_run_chunks) on the same resident packs: the gates of
    # extract_candidates_calling + the compaction of the candidate list, one event pair per call.  Algorithmic bytes: the pack once + 5 B
    # per column (flag + depth) out.
    nc_max = max(p.n_cols for p in packs)
    xflags = torch.empty((nc_max,), dtype=torch.uint8, device=dev)
    xdepth = torch.empty((nc_max,), dtype=torch.int32, device=dev)
    xout = torch.empty((nc_max,), dtype=torch.int32, device=dev)
    xscr = torch.empty(((nc_max + 255) // 256 + 2,), dtype=torch.int32, device=dev)
    xn = torch.empty((1,), dtype=torch.int32, device=dev)
    evx = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n_feat)]
    for i in range(-2, n_feat):
        j = i % args.pool
        if i >= 0:
            evx[i][0].record()
        check(lib.cto_extract_candidates(C.byref(packs[j].view), 20, int(min_bq), 0.05, 1.0, 4.0, 3, 0, xflags.data_ptr(), xdepth.data_ptr(), sptr))
        if i >= 0:
            evx[i][1].record()
        check(lib.cto_candidate_positions(C.byref(packs[j].view), xflags.data_ptr(), 1, 1, 2 ** 31 - 1, xout.data_ptr(), nc_max, xscr.data_ptr(),
                                          xn.data_ptr(), sptr))
        if i >= 0:
            evx[i][2].record()
    torch.cuda.synchronize()
    ext_ms = sum(a.elapsed_time(b) for a, b, _ in evx) / n_feat
    cmp_ms = sum(b.elapsed_time(c) for _, b, c in evx) / n_feat
    ext_bytes = pack_bytes + 5 * sum(p.n_cols for p in packs) / len(packs)
    ext_cands = int(xn.item())

    live_traffic = None
    if rank == 0 and world == 1 and not args.no_live_traffic:
        live_traffic = live_pmc_traffic(args.batch)
    if rank == 0:
        sites_total = world * args.steps * args.batch
        res = {
            "metric": "candidate sites/sec (pileup-tensor + AFF/NEG inference) at 1/2/4/8 MI355X",
            "value": round(sites_total / dt, 1), "unit": "sites/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "ONT 50x synthetic pileups, SNV (K=4), 1M-site job cut into %d-site chunks "
                                   "(BASELINE.json configs[1]); step = featurize(AFF+NEG) + CvT + BiGRU + posterior on one chunk"
                                   % args.batch,
                       "batch": args.batch, "chunks_resident_per_gpu": args.pool, "min_bq_aff": min_bq,
                       "pack_bytes_per_chunk": int(pack_bytes),
                       "parallelism": "sites sharded, 1 rank/GPU" + (", all_gather of per-site probabilities (RCCL)" if world > 1 else ""),
                       "weights": "seeded random init (no pretrained weights offline)"},
            "roofline": {"bound": "mfma", "kernel": "k_gru_layer_rot<256,256,192,2,true> (BiGRU layer 2 + fused fc1, both directions)",
                         "achieved": round(achieved, 3), "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(achieved / PEAK_FP32_MFMA_TFLOPS, 4),
                         "traffic": (live_traffic or {}).get("gru_l2") or pmc_traffic(args.batch),
                         "traffic_source": "measured in this run (two rocprofv3 --pmc child passes of this file after the timed region)" if (live_traffic or {}).get("gru_l2")
                                           else "from_digest (%s)" % (os.path.relpath(_pmc_file(), ROOT) if _pmc_file() else "none committed"),
                         "traffic_note": "fabric (HBM + Infinity Cache) bytes per launch = 2 x FETCH_SIZE (gfx950 halves coalesced reads; calibrated on the pack read of "
                                         "k_featurize_columns) + WRITE_SIZE, separate rocprofv3 --pmc passes (%s); algorithmic 151 MB: the 138 MB layer-1 output is "
                                         "fetched once per DIRECTION (two workgroups per site tile, on different XCDs), the second time from the Infinity Cache"
                                         % (os.path.relpath(_pmc_file(), ROOT) if _pmc_file() else "no digest committed"),
                         "launch_ms": round(mean_ms.value, 4), "launches_measured": int(n_meas),
                         "flops_per_launch": flops_per_launch},
            "roofline_tensor_creation": {"bound": "hbm", "kernel": "k_featurize_sites (one workgroup per candidate: histograms of its 33 columns in LDS, both passes, rescale fused)",
                                         "achieved": round(feat_bytes / (feat_ms * 1e-3) / 1e9, 1), "peak": 8000.0, "unit": "GB/s",
                                         "frac": round(feat_bytes / (feat_ms * 1e-3) / 1e9 / 8000.0, 4), "launch_ms": round(feat_ms, 4),
                                         "bytes_per_launch": int(feat_bytes), "traffic": (live_traffic or {}).get("featurize") or pmc_traffic_featurize(args.batch),
                                         "traffic_source": "measured in this run" if (live_traffic or {}).get("featurize") else "from_digest",
                                         "note": "latency / issue bound, not bandwidth bound (a chain of ~8 dependent global accesses per candidate; scalar unit, VALU and LDS each about half busy: profiles/round3_fused_featurize.md); ~2 % of the step"},
            "roofline_candidate_extraction": {"bound": "hbm", "kernel": "k_extract_candidates (the gates of extract_candidates_calling on the resident pack: one lane per column, the 64 columns of a wave staged in LDS, counters in registers - no atomics)",
                                              "achieved": round(ext_bytes / (ext_ms * 1e-3) / 1e9, 1), "peak": 8000.0, "unit": "GB/s",
                                              "frac": round(ext_bytes / (ext_ms * 1e-3) / 1e9 / 8000.0, 4), "launch_ms": round(ext_ms, 4),
                                              "bytes_per_launch": int(ext_bytes), "compaction_ms": round(cmp_ms, 4), "candidates_last_pack": ext_cands,
                                              "note": "bound by dependent-instruction latency at 7-10 waves per CU (16 KB of LDS each), not by bandwidth (0.25 of the HBM peak on a region's 225 MB pack); what a REGION job of cto_run_chunks runs between pile-up and tensor creation (e2e.bam_to_vcf_with_extraction); not part of `value`"},
            "end_to_end_tflops": round(2.0 * eng.macs_per_site * sites_total / dt / 1e12, 3),
            "ranks_seen": dist.get_world_size() if world > 1 else 1,
            "backend": (dist.get_backend() + (" (RCCL over xGMI)" if backend == "nccl" else " (test hook)")) if world > 1 else None,
            "rank_devices": census, "gather_verified": gather_ok,
        }
        if sustained is not None:
            res["sustained"] = sustained
        if stage_fracs is not None:
            stage_fracs["tensor_creation"] = {"ms": round(feat_ms, 4), "gb_per_s": round(feat_bytes / (feat_ms * 1e-3) / 1e9, 1),
                                              "frac_of_hbm_peak": round(feat_bytes / (feat_ms * 1e-3) / 1e9 / 8000.0, 4)}
            res["stage_fracs"] = stage_fracs
        probs_cpu = None
        if world == 1 and not args.no_cpu_baseline:
            cb, probs_cpu = cpu_baseline(chunks, models, lik, edges, min_bq, min(args.cpu_sample, args.batch))
            res["cpu_baseline"] = cb
            got = eng.run_device(packs[0], sites[0])["probs"][: probs_cpu.shape[0]].cpu().numpy()
            res["parity_max_abs_dP_vs_cpu_sample"] = float(np.abs(got - probs_cpu).max())
        if world == 1 and not args.no_sustained and not args.no_cpu_baseline:
            res["sustained_distinct"] = sustained_distinct_leg(eng, chunks, models, lik, edges, min_bq, args.batch)
        if world == 1 and not args.no_split:
            ref = eng.run_device(packs[0], sites[0])
            res["split_mfma"] = split_mfma_leg(dev, packs[:4], sites[:4], args.batch, lik, edges, min_bq, ref["probs"].cpu().numpy(),
                                               ref["decision"].cpu().numpy(), probs_cpu)
        if world == 1 and not args.no_configs:
            res["configs"] = config_legs(dev, args.batch)
        if world == 1 and not args.no_postfilters:
            res.setdefault("configs", {})["configs3_realign"] = realign_leg(dev)
            res["hapfilter"] = hapfilter_leg()
        if world == 1 and not args.no_e2e:
            # ---- file-to-file legs (never `value`): chunk files + pileup source on disk -> p_<chunk>.vcf through the call_chunks
            # pipeline, everything a real run pays included; the rate is set by the host (cores stated), not by the GPU ----
            # run as a child process with any attached profiler's hooks stripped from its environment: the legs launch the same
            # kernels in a pipelined context (overlapping PCIe copies), which must not mix into this process's kernel statistics
            import subprocess
            env = {k: v for k, v in os.environ.items()
                   if not (k.startswith(("ROCPROF", "ROCP_", "ROCTX", "ROCPROFILER")) or k in ("HSA_TOOLS_LIB", "HSA_TOOLS_REPORT_LOAD_FAILURE"))}
            if "LD_PRELOAD" in env:
                env["LD_PRELOAD"] = ":".join(x for x in env["LD_PRELOAD"].split(":") if "rocprof" not in x and "roctracer" not in x)
            child = subprocess.run([sys.executable, "-m", "clairs_to_amd.e2e", "--chunks", str(args.e2e_chunks), "--bam-chunks",
                                    str(max(2, args.e2e_chunks // 3)), "--batch", str(args.batch), "--reference-chunk-sites", "10000"],
                                   cwd=ROOT, env=env, capture_output=True, text=True)
            lines = [ln for ln in child.stdout.split("\n") if ln.startswith("{")]
            e2e = json.loads(lines[-1]) if child.returncode == 0 and lines else {"error": child.stderr[-500:]}
            e2e["cpu_reference_python"] = REFERENCE_PYTHON_NOTE
            res["e2e"] = e2e
        print(json.dumps(res))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

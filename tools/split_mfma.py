#!/usr/bin/env python3
"""The split-operand experiment (csrc/gru_split_kernel.h) beside the fp32 product kernel: the step on one batch with the default
NEG model and with models created under CTO_GRU_SPLIT=f16 / bf16 - HIP-event times of layer 2 (cto_model_profile) and of the
whole step, max |d logit| of the NEG network against the fp32 kernel, and on a small sample the max |dP| and the NEG network's
max |d logit| of each against the oracle (the CPU restatement; checker use only).
python tools/split_mfma.py [--batch 4096] [--reps 40] [--oracle-sites 96]"""
import argparse
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def neg_engine(n_out, lik, edges, dev, split):
    """an engine over its own module objects: a module creates its C-ABI handle once, and the switch is read at creation"""
    from clairs_to_amd.engine import Engine, synthetic_models
    models = synthetic_models(n_out)
    if split:
        os.environ["CTO_GRU_SPLIT"] = split
        os.environ["CTO_CVT_SPLIT"] = split
    try:
        return Engine(models["aff"], models["neg"], lik, edges, min_bq=20, device=dev)
    finally:
        os.environ.pop("CTO_GRU_SPLIT", None)
        os.environ.pop("CTO_CVT_SPLIT", None)


def measure(batch=4096, reps=40, oracle_sites=96, n_out=4):
    import numpy as np
    import torch
    from clairs_to_amd._lib import lib, check
    from clairs_to_amd.engine import synthetic_models
    from clairs_to_amd.featurize import featurize
    from clairs_to_amd.synth import SynthChunk, mpileup_text, likelihood_table, lik_and_edges
    dev = torch.device("cuda:0")
    models = synthetic_models(n_out)
    lik, edges = lik_and_edges(likelihood_table(n_out), n_out)
    engs = {"f32": neg_engine(n_out, lik, edges, dev, None), "split_f16": neg_engine(n_out, lik, edges, dev, "f16"),
            "split_bf16": neg_engine(n_out, lik, edges, dev, "bf16")}
    ch = SynthChunk(batch, seed=1)
    dp = engs["f32"].upload(ch.arrays())
    sp = torch.from_numpy(ch.site_pos).to(dev)
    feat = featurize(dp, sp, 20, 50)
    s = int(torch.cuda.current_stream().cuda_stream)
    out, out_aff, res = {}, {}, {"batch": batch, "reps": reps}
    for name, eng in engs.items():
        ln = torch.empty((n_out, batch, 2), device=dev)
        fn = lambda: check(lib.cto_model_forward(eng.h_neg, feat.x_neg.data_ptr(), batch, ln.data_ptr(), s))
        for _ in range(5):
            fn()
        check(lib.cto_model_profile(eng.h_neg, 2))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        check(lib.cto_model_profile(eng.h_neg, 0))
        ms, macs = C.c_double(0.0), C.c_int64(0)
        check(lib.cto_model_profile_read(eng.h_neg, C.byref(ms), C.byref(macs)))
        ms1, macs1 = C.c_double(0.0), C.c_int64(0)
        check(lib.cto_model_profile_read_stage(eng.h_neg, 1, C.byref(ms1), C.byref(macs1)))
        step = eng.run_device(dp, sp)
        e0.record()
        for _ in range(reps):
            eng.run_device(dp, sp)
        e1.record()
        torch.cuda.synchronize()
        step_ms = e0.elapsed_time(e1) / reps
        out[name] = ln.cpu().numpy()
        la = torch.empty((n_out, batch, 2), device=dev)
        fa = lambda: check(lib.cto_model_forward(eng.h_aff, feat.x_aff.data_ptr(), batch, la.data_ptr(), s))
        for _ in range(5):
            fa()
        e0.record()
        for _ in range(reps):
            fa()
        e1.record()
        torch.cuda.synchronize()
        aff_ms = e0.elapsed_time(e1) / reps
        out_aff[name] = la.cpu().numpy()
        res[name] = {"cvt_ms": aff_ms, "gru_l2_ms": ms.value, "gru_l1_ms": ms1.value, "step_ms": step_ms,
                     "sites_per_s": batch / step_ms * 1e3}
        del step
    for name in engs:
        if name != "f32":
            res[name]["neg_max_abs_dlogit_vs_f32_kernel"] = float(np.abs(out[name] - out["f32"]).max())
            res[name]["aff_max_abs_dlogit_vs_f32_kernel"] = float(np.abs(out_aff[name] - out_aff["f32"]).max())
            res[name]["l2_speedup"] = res["f32"]["gru_l2_ms"] / res[name]["gru_l2_ms"]
    if oracle_sites:
        import oracle
        small = SynthChunk(oracle_sites, seed=1)
        ref, lo = small.ref_window()
        ta, da, _, _ = oracle.create_tensor(mpileup_text(small, 20), ref, lo, small.site_pos)
        tn, dn, _, _ = oracle.create_tensor(mpileup_text(small, 0), ref, lo, small.site_pos)
        cfg = dict(emb_dim=(16, 64, 128), heads=(1, 3, 4), depth=(1, 2, 3), n_out=n_out)
        la = oracle.cvt_forward(models["aff_weights"], cfg, oracle.rescale(ta, da))
        ln = oracle.bigru_forward(models["neg_weights"], n_out, oracle.rescale(tn, dn))
        probs, post, dec, qual = oracle.posterior(la, ln, lik, edges)
        sdp = engs["f32"].upload(small.arrays())
        sfeat = featurize(sdp, torch.from_numpy(small.site_pos).to(dev), 20, 50)
        for name, eng in engs.items():
            got = eng.run_chunk(small.arrays(), small.site_pos)
            torch.cuda.synchronize()
            res[name]["max_abs_dP_vs_oracle"] = float(np.abs(got["probs"].cpu().numpy() - probs).max())
            res[name]["max_abs_dposterior_vs_oracle"] = float(np.abs(got["post"].cpu().numpy() - post).max())
            res[name]["decisions_equal_oracle"] = bool((got["decision"].cpu().numpy()[:, :2] & 3 == (np.asarray(dec)[:, :2] & 3)).all())
            lg = torch.empty((n_out, oracle_sites, 2), device=dev)
            check(lib.cto_model_forward(eng.h_neg, sfeat.x_neg.data_ptr(), oracle_sites, lg.data_ptr(), s))
            torch.cuda.synchronize()
            res[name]["neg_max_abs_dlogit_vs_oracle"] = float(np.abs(lg.cpu().numpy() - np.asarray(ln).reshape(lg.shape)).max())
            check(lib.cto_model_forward(eng.h_aff, sfeat.x_aff.data_ptr(), oracle_sites, lg.data_ptr(), s))
            torch.cuda.synchronize()
            res[name]["aff_max_abs_dlogit_vs_oracle"] = float(np.abs(lg.cpu().numpy() - np.asarray(la).reshape(lg.shape)).max())
        res["oracle_sites"] = oracle_sites
    return res


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--reps", type=int, default=40)
    ap.add_argument("--oracle-sites", type=int, default=96)
    a = ap.parse_args()
    print(json.dumps(measure(a.batch, a.reps, a.oracle_sites), indent=1))

#!/usr/bin/env python3
"""The split-operand experiment (csrc/gru_split_kernel.h) beside the fp32 product kernel: the step on one batch with the default
NEG model and with models created under CTO_GRU_SPLIT=f16 / bf16 - HIP-event times of layer 2 (cto_model_profile) and of the
whole step, max |d logit| of both networks against the fp32 kernels.  (How far each is from the oracle is measured where the oracle may be
used: tests/test_gpu_split.py, `-s` prints the figures.)
python tools/split_mfma.py [--batch 4096] [--reps 40]"""
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


def measure(batch=4096, reps=40, n_out=4):
    import numpy as np
    import torch
    from clairs_to_amd._lib import lib, check
    from clairs_to_amd.engine import synthetic_models
    from clairs_to_amd.featurize import featurize
    from clairs_to_amd.synth import SynthChunk, likelihood_table, lik_and_edges
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
    return res


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--reps", type=int, default=40)
    a = ap.parse_args()
    print(json.dumps(measure(a.batch, a.reps), indent=1))

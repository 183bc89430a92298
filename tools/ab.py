#!/usr/bin/env python3
"""A/B timing of kernel variants: every `lib*.so` given (built with `make -C clairs_to_amd/csrc OBJDIR=build_x TARGET=../libx.so
EXTRA=-D...`) runs the AFF and NEG networks on the same 4096-site batch in its own process (CTO_LIB_PATH) and reports
HIP-event times: NEG total, its layer-2 kernel (cto_model_profile), the remainder (layer 1 + tail), AFF total.
python tools/ab.py clairs_to_amd/libclairsto_amd.so clairs_to_amd/libx.so ... [--reps 40] [--batch 4096]"""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(a):
    import torch
    from clairs_to_amd._lib import lib, check
    from clairs_to_amd.engine import Engine, synthetic_models
    from clairs_to_amd.featurize import featurize
    from clairs_to_amd.synth import SynthChunk, likelihood_table, lik_and_edges
    dev = torch.device("cuda:0")
    K = a.n_out
    models = synthetic_models(K)
    lik, edges = lik_and_edges(likelihood_table(K), K)
    eng = Engine(models["aff"], models["neg"], lik, edges, min_bq=20, device=dev)
    ch = SynthChunk(a.batch, seed=1)
    dp = eng.upload(ch.arrays())
    sp = torch.from_numpy(ch.site_pos).to(dev)
    feat = featurize(dp, sp, 20, 50)
    B = a.batch
    la = torch.empty((K, B, 2), device=dev)
    ln = torch.empty((K, B, 2), device=dev)
    s = int(torch.cuda.current_stream().cuda_stream)

    def timed(fn):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / a.reps
    res = {}
    res["aff_ms"] = timed(lambda: check(lib.cto_model_forward(eng.h_aff, feat.x_aff.data_ptr(), B, la.data_ptr(), s)))
    check(lib.cto_model_profile(eng.h_neg, 1))
    res["neg_ms"] = timed(lambda: check(lib.cto_model_forward(eng.h_neg, feat.x_neg.data_ptr(), B, ln.data_ptr(), s)))
    check(lib.cto_model_profile(eng.h_neg, 0))
    ms, macs = C.c_double(0.0), C.c_int64(0)
    check(lib.cto_model_profile_read(eng.h_neg, C.byref(ms), C.byref(macs)))
    res["gru_l2_ms"] = ms.value
    res["gru_l1_plus_tail_ms"] = res["neg_ms"] - ms.value
    res["step_ms"] = timed(lambda: eng.run_device(dp, sp))
    res["logit_checksum"] = float(la.double().sum().item() + ln.double().sum().item())
    print(json.dumps(res))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("libs", nargs="*")
    ap.add_argument("--reps", type=int, default=40)
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--n-out", type=int, default=4)
    ap.add_argument("--child", action="store_true")
    a = ap.parse_args()
    if a.child:
        return child(a)
    for lp in a.libs or [os.path.join(ROOT, "clairs_to_amd", "libclairsto_amd.so")]:
        env = dict(os.environ, CTO_LIB_PATH=os.path.abspath(lp))
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", "--reps", str(a.reps), "--batch", str(a.batch),
                            "--n-out", str(a.n_out)], env=env, capture_output=True, text=True)
        line = [l for l in r.stdout.split("\n") if l.startswith("{")]
        if r.returncode != 0 or not line:
            print("%-40s FAILED: %s" % (os.path.basename(lp), r.stderr[-400:]))
            continue
        d = json.loads(line[-1])
        print("%-40s aff %.4f  neg %.4f  (l2 %.4f, l1+tail %.4f)  step %.4f  checksum %.6f" % (
            os.path.basename(lp), d["aff_ms"], d["neg_ms"], d["gru_l2_ms"], d["gru_l1_plus_tail_ms"], d["step_ms"], d["logit_checksum"]), flush=True)


if __name__ == "__main__":
    main()

"""BAM -> VCF with every chunk through the device (CTO_CTX_WAIT_MS: a BED chunk waits for an inflate context instead of decoding on the host)
against the hybrid default, by producers and contexts.  python tools/experiments/bam_alldevice_sweep.py   (GPU box)"""
import os, sys, tempfile, shutil, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
CHILD = r'''
import os, sys, json
sys.path.insert(0, %r)
import torch
from clairs_to_amd.e2e import build_run, time_run
from clairs_to_amd.engine import Engine, synthetic_models
from clairs_to_amd.synth import likelihood_table, lik_and_edges
d, prod, jobs, cus = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
dev = torch.device("cuda", 0)
models = synthetic_models(4, seed=0)
lik, edges = lik_and_edges(likelihood_table(4), 4)
eng = Engine(models["aff"], models["neg"], lik, edges, min_bq=20, device=dev)
import pickle
run = pickle.load(open(os.path.join(d, "run.pkl"), "rb"))
r = time_run(eng, run, "bam", os.path.join(d, "o_%%d_%%d_%%d_%%s" %% (prod, jobs, cus, os.environ.get("CTO_CTX_WAIT_MS", "0"))), prod, 4, 3, pipeline="native", times=3, inflate_cus=cus, inflate_jobs=jobs)
print(json.dumps({"sites_per_s": r["sites_per_s"], "device_inflated": r.get("device_inflated"), "cpu_ms": r["host_process"]["user_cpu_ms_per_chunk"], "device_ms": r["stage_thread_time"]["device_ms_per_chunk"]}))
''' % ROOT
def main():
    import pickle
    from clairs_to_amd.e2e import build_run
    d = tempfile.mkdtemp(prefix="cto_bs_")
    run, _ = build_run(d, "bam", 32, 4096, 3)
    pickle.dump(run, open(os.path.join(d, "run.pkl"), "wb"))
    open(os.path.join(d, "child.py"), "w").write(CHILD)
    for wait in ("0", "2000"):
        for prod, jobs, cus in ((20, 10, 144), (12, 10, 144), (24, 16, 144), (24, 16, 176), (32, 24, 176), (16, 12, 144)):
            env = dict(os.environ, CTO_CTX_WAIT_MS=wait)
            p = subprocess.run([sys.executable, os.path.join(d, "child.py"), d, str(prod), str(jobs), str(cus)], env=env, capture_output=True, text=True)
            line = [l for l in p.stdout.split("\n") if l.startswith("{")]
            print("wait %5s  producers %2d  contexts %2d  CUs %3d : %s" % (wait, prod, jobs, cus, line[-1] if line else p.stderr[-300:]), flush=True)
    shutil.rmtree(d, ignore_errors=True)
if __name__ == "__main__":
    main()

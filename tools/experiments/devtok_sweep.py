import os, sys, json, tempfile, shutil, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from clairs_to_amd import e2e
from clairs_to_amd.engine import Engine, synthetic_models
from clairs_to_amd.synth import likelihood_table, lik_and_edges
dev = torch.device("cuda", 0)
models = synthetic_models(4, seed=0)
lik, edges = lik_and_edges(likelihood_table(4), 4)
eng = Engine(models["aff"], models["neg"], lik, edges, min_bq=20, device=dev)
d = tempfile.mkdtemp(prefix="cto_e2e_")
run, source = e2e.build_run(d, "text", 48, 4096, 3, None)
for p in (1, 2, 3, 4, 6):
    for dt in (True, False):
        r = e2e.time_run(eng, run, "text", os.path.join(d, "o%d%d" % (p, dt)), p, 2, 4, pipeline="native", device_tokenise=dt)
        print(p, "device" if dt else "host", r["sites_per_s"], r["stage_thread_time"]["produce_ms_per_chunk"], r["stage_thread_time"]["device_ms_per_chunk"], r["host_process"]["user_cpu_ms_per_chunk"])
shutil.rmtree(d)

import json, os, sys, tempfile, shutil
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from clairs_to_amd.e2e import build_run, time_run
from clairs_to_amd.engine import Engine, synthetic_models
from clairs_to_amd.synth import likelihood_table, lik_and_edges
dev = torch.device("cuda", 0)
models = synthetic_models(4, seed=0)
lik, edges = lik_and_edges(likelihood_table(4), 4)
eng = Engine(models["aff"], models["neg"], lik, edges, min_bq=20, device=dev)
d = tempfile.mkdtemp(prefix="cto_ds_")
run, src = build_run(d, "text", 96, 4096, 3)
for depth in (0, 16, 24, 40):
    os.environ["CTO_PIPELINE_DEPTH"] = str(depth)
    for prod, wr, two in ((6, 4, False), (8, 6, False), (6, 4, True)):
        r = time_run(eng, run, "text", os.path.join(d, "o_%d_%d_%d_%d" % (depth, prod, wr, two)), prod, wr, 3, pipeline="native", times=2, two_streams=two)
        print("depth %2d producers %d writers %d two_streams %d: %.0f sites/s  %s" % (depth, prod, wr, two, r["sites_per_s"], {k: r["stage_thread_time"][k] for k in ("produce_ms_per_chunk", "finish_ms_per_chunk", "device_ms_per_chunk", "launcher_waits_for_producer_ms_per_chunk")}), flush=True)
shutil.rmtree(d, ignore_errors=True)

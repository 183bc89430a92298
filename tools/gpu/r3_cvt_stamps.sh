cd /root/repo
CTO_BLOCK_PROF=1 timeout 300 python - <<'PY' 2>&1 | grep "stamps" | tail -12
import torch, ctypes as C
from clairs_to_amd._lib import lib, check
from clairs_to_amd.engine import Engine, synthetic_models
from clairs_to_amd.featurize import featurize
from clairs_to_amd.synth import SynthChunk, likelihood_table, lik_and_edges
dev = torch.device("cuda:0")
models = synthetic_models(4)
lik, edges = lik_and_edges(likelihood_table(4), 4)
eng = Engine(models["aff"], models["neg"], lik, edges, min_bq=20, device=dev)
ch = SynthChunk(4096, seed=1)
dp = eng.upload(ch.arrays()); sp = torch.from_numpy(ch.site_pos).to(dev)
feat = featurize(dp, sp, 20, 50)
la = torch.empty((4, 4096, 2), device=dev)
s = int(torch.cuda.current_stream().cuda_stream)
for _ in range(3):
    check(lib.cto_model_forward(eng.h_aff, feat.x_aff.data_ptr(), 4096, la.data_ptr(), s))
torch.cuda.synchronize()
PY

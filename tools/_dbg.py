import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests"); sys.path.insert(0, "tests/golden")
import oracle
from clairs_to_amd.engine import synthetic_models
from weights_recipe import CVT_CFG
models = synthetic_models(4)
rng = np.random.default_rng(0)
B = int(sys.argv[1])
x = rng.integers(-40, 40, size=(B, 33, 34)).astype(np.float32)
ref = oracle.cvt_forward(models["aff_weights"], dict(CVT_CFG, n_out=4), x)
m = models["aff"].to("cuda")
got = m.logits(torch.from_numpy(x).cuda()).cpu().numpy()
err = np.abs(got - ref).max(axis=(0, 2))
bad = np.nonzero(err > 1e-4)[0]
print(B, os.environ.get("CTO_CVT_NO_EMBED_FUSE"), os.environ.get("CTO_CVT_NO_HEAD_FUSE"), err.max(), len(bad), bad[:40])

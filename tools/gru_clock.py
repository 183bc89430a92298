"""Average shader clock during the BiGRU kernels (s_memtime / s_memrealtime).  Needs a library built with
   make -C clairs_to_amd/csrc clean all CXXFLAGS="... -DCTO_GRU_CLOCKS" (debug probe, compiled out by default)."""
import os, sys, ctypes as C
sys.path.insert(0, os.getcwd())
import torch, numpy as np
from clairs_to_amd._lib import lib
from clairs_to_amd.engine import synthetic_models
m = synthetic_models(4)["neg"].to("cuda")
x = torch.randn(4096, 33, 34, device="cuda")
for _ in range(5): m.logits(x)
torch.cuda.synchronize()
out = (C.c_longlong * 16)()
raw = C.CDLL(lib._name)
raw.cto_debug_gru_clocks(out)
v = list(out)
for name, o in (("L1", 0), ("L2", 4)):
    print(name, "clock64 delta", v[o], "wall(100MHz) delta", v[o+1], "=> %.0f MHz, %.1f us" % (v[o] / (v[o+1] / 100.0), v[o+1] / 100.0))
    ph = v[8 + o: 12 + o]
    print("   cycles of wave 0 over 33 steps: barrier wait %d, h part %d, x part + gates %d, step tail %d (sum %d of %d)" % (*ph, sum(ph), v[o]))

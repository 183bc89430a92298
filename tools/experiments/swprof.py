import ctypes as C, os, sys, json
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from clairs_to_amd._lib import lib
from clairs_to_amd.synth_realign import gen_window
from clairs_to_amd.realign_reads import realign_windows
torch.cuda.set_device(0)
rng = np.random.default_rng(20260930)
ws = [gen_window(rng) for _ in range(1500)]
args = [(w["seqs"], w["positions"], w["cigars"], w["reference"], w["haplotypes"], w["ref_start"], w["ref_prefix"], w["ref_suffix"]) for w in ws]
out = (C.c_longlong * 8)()
lib.cto_debug_sw_prof(out, 1)
st = {}
realign_windows(args, where="device", threads=16, stats=st)
lib.cto_debug_sw_prof(out, 1)
print("all four launches, block 0 thread 0 of each (summed): main %d lazy %d scan %d cycles; columns %d lazy chunks %d" % tuple(out[:5]))
print(st)

"""`nn.Module` front-ends of the HIP networks, state_dict- and pickle-compatible with the reference.

The reference's checkpoints are pickled module OBJECTS (clairs/predict.py:513-517 loads
`torch.load(...)['model_acgt']` and calls it), so a drop-in needs importable classes with the reference's
qualified names and parameter tree (`clairs.model.CvT`, `CvT_Indel`, `BiGRU_NACGT`, `BiGRU_NACGT_Indel` and the
helper modules they contain).  The classes below only HOLD parameters under those names; `forward` hands the
state_dict to the C ABI once (cto_cvt_create / cto_bigru_create) and then runs cto_model_forward, returning
the same tuple of K float32 [B,2] logit tensors that clairs/predict.py:646-658 unpacks.

`install_reference_aliases()` registers this module as `clairs.model` so reference pickles unpickle onto it.
"""
import ctypes as C
import os
import sys
import types

import numpy as np
import torch
from torch import nn

from ._lib import lib, check, CvtCfg, c_vp, current_stream_ptr

NPOS, NCHAN = 33, 34
_SNV, _INDEL = ("a", "c", "g", "t"), ("a", "c", "g", "t", "i", "d")


# ---- parameter holders (names and shapes follow clairs/model.py:57-147; no compute here) ----
class LayerNorm(nn.Module):
    def __init__(self, dim, eps=1e-5):
        super().__init__()
        self.eps = eps
        self.g = nn.Parameter(torch.ones(1, dim, 1, 1))
        self.b = nn.Parameter(torch.zeros(1, dim, 1, 1))


class PreNorm(nn.Module):
    def __init__(self, dim, fn):
        super().__init__()
        self.norm, self.fn = LayerNorm(dim), fn


class FeedForward(nn.Module):
    def __init__(self, dim, mult=4, dropout=0.0):
        super().__init__()
        self.net = nn.Sequential(nn.Conv2d(dim, dim * mult, 1), nn.GELU(), nn.Dropout(dropout),
                                 nn.Conv2d(dim * mult, dim, 1), nn.Dropout(dropout))


class DepthWiseConv2d(nn.Module):
    def __init__(self, dim_in, dim_out, kernel_size, padding, stride, bias=True):
        super().__init__()
        self.net = nn.Sequential(
            nn.Conv2d(dim_in, dim_in, kernel_size, stride=stride, padding=padding, groups=dim_in, bias=bias),
            nn.BatchNorm2d(dim_in), nn.Conv2d(dim_in, dim_out, 1, bias=bias))


class Attention(nn.Module):
    def __init__(self, dim, proj_kernel, kv_proj_stride, heads=8, dim_head=64, dropout=0.0):
        super().__init__()
        inner = dim_head * heads
        self.heads, self.scale = heads, dim_head ** -0.5
        self.attend = nn.Softmax(dim=-1)
        self.to_q = DepthWiseConv2d(dim, inner, proj_kernel, proj_kernel // 2, 1, bias=False)
        self.to_kv = DepthWiseConv2d(dim, inner * 2, proj_kernel, proj_kernel // 2, kv_proj_stride, bias=False)
        self.to_out = nn.Sequential(nn.Conv2d(inner, dim, 1), nn.Dropout(dropout))


class Transformer(nn.Module):
    def __init__(self, dim, proj_kernel, kv_proj_stride, depth, heads, dim_head=64, mlp_mult=4, dropout=0.0):
        super().__init__()
        self.layers = nn.ModuleList(
            nn.ModuleList([PreNorm(dim, Attention(dim, proj_kernel, kv_proj_stride, heads, dim_head, dropout)),
                           PreNorm(dim, FeedForward(dim, mlp_mult, dropout))]) for _ in range(depth))


# ---- engine plumbing shared by the four public classes ----
class _HipNet(nn.Module):
    _kind = None          # "cvt" | "bigru"
    _heads_out = _SNV

    def _cfg(self):
        raise NotImplementedError

    # EXPERIMENTAL, off by default: "f16" or "bf16" runs this network's GEMMs on split 16-bit operands (three MFMA passes per
    # product, fp32 accumulation).  Set it before the first forward, or any time: the handle is rebuilt.
    split_operands = None

    def _weights_version(self):
        return (self.split_operands,) + tuple((k, int(v._version), v.data_ptr()) for k, v in self.state_dict().items())

    def _handle(self, slot="_cto_state"):
        """the C-ABI model handle of this module's current weights (rebuilt when they change).  A handle's activation workspace
        belongs to one stream at a time: `_handle2()` is a second, independent handle for a caller that runs two streams."""
        ver = self._weights_version()
        st = self.__dict__.get(slot)
        if st is not None and st[0] == ver:
            return st[1]
        if st is not None:
            lib.cto_model_destroy(st[1])
        w = c_vp(lib.cto_weights_new())
        try:
            for k, v in self.state_dict().items():
                if k.endswith("num_batches_tracked"):
                    continue
                a = np.ascontiguousarray(v.detach().to("cpu", torch.float32).numpy())
                check(lib.cto_weights_add(w, k.encode(), a.ctypes.data, a.size))
            out = c_vp()
            # the arithmetic travels as an argument (cto_*_create_ex), never through os.environ: a handle built at the same
            # moment on another thread must not see this module's choice.  None = the documented process-wide default
            # (CTO_CVT_SPLIT / CTO_GRU_SPLIT), "f32" pins fp32 whatever the environment says.
            modes = {None: -1, "f32": 0, "f16": 1, "bf16": 2}
            if self.split_operands not in modes:
                raise ValueError("split_operands must be None, 'f32', 'f16' or 'bf16', not %r" % (self.split_operands,))
            mode = modes[self.split_operands]
            if self._kind == "cvt":
                cfg = self._cfg()
                check(lib.cto_cvt_create_ex(w, C.byref(cfg), mode, C.byref(out)))
            else:
                check(lib.cto_bigru_create_ex(w, len(self._heads_out), mode, C.byref(out)))
        finally:
            lib.cto_weights_free(w)
        self.__dict__[slot] = (ver, c_vp(out.value))
        return self.__dict__[slot][1]

    def _handle2(self):
        return self._handle("_cto_state2")

    def __del__(self):
        for slot in ("_cto_state", "_cto_state2"):
            st = self.__dict__.get(slot)
            if st is not None:
                self.__dict__[slot] = None
                lib.cto_model_destroy(st[1])

    def __getstate__(self):
        d = dict(self.__dict__)
        d.pop("_cto_state", None)     # the device handle never travels in a pickle
        d.pop("_cto_state2", None)
        d.pop("_cto_packed", None)    # ... nor does the packed copy of the weights
        return d

    def logits(self, x):
        """x: float32 [B,33,34] on a HIP device -> float32 [K,B,2] (post-SELU logits)."""
        if x.device.type != "cuda":
            raise RuntimeError("clairs_to_amd: the networks run on the HIP device only; move the input to 'cuda' "
                               "(there is no CPU fallback)")
        if x.dim() != 3 or x.shape[1] != NPOS or x.shape[2] != NCHAN:
            raise ValueError("expected [B,%d,%d], got %s" % (NPOS, NCHAN, tuple(x.shape)))
        x = x.to(torch.float32).contiguous()
        # through PyTorch's dispatcher (csrc/torch_ops.cpp): the operator keeps a device-side handle per packed-weights tensor
        if self._kind == "cvt":
            cfg = self._cfg()
            return torch.ops.clairsto.cvt_forward(x, self._packed(x.device), list(cfg.emb_dim) + list(cfg.heads) + list(cfg.depth) + [cfg.n_out])
        return torch.ops.clairsto.bigru_forward(x, self._packed(x.device), len(self._heads_out))

    def _packed(self, device):
        """The `packed_weights` operand of the custom ops: every floating-point tensor of state_dict() flattened and
        concatenated in state_dict order (cto_cvt_create_packed / cto_model_manifest), resident on `device`; rebuilt when a
        parameter changes."""
        ver = (self._weights_version(), str(device))
        st = self.__dict__.get("_cto_packed")
        if st is None or st[0] != ver:
            flat = torch.cat([v.detach().reshape(-1).to("cpu", torch.float32) for v in self.state_dict().values() if v.is_floating_point()])
            st = (ver, flat.to(device))
            self.__dict__["_cto_packed"] = st
        return st[1]

    def macs_per_site(self):
        return int(lib.cto_model_macs_per_site(self._handle()))

    def _tuple(self, x):
        out = self.logits(x)
        if getattr(self, "apply_softmax", False):        # clairs/model.py:255-259 / 461-465: nn.Softmax(dim=1) per head, on the device
            out = torch.ops.clairsto.softmax2(out)       # (cto_softmax_pairs: no torch operator computes on the path)
        return tuple(out[k] for k in range(out.shape[0]))


def _stage(dim_in, cfg, dropout):
    return nn.Sequential(
        nn.Conv2d(dim_in, cfg["emb_dim"], cfg["emb_kernel"], stride=cfg["emb_stride"], padding=cfg["emb_kernel"] // 2),
        LayerNorm(cfg["emb_dim"]),
        Transformer(cfg["emb_dim"], cfg["proj_kernel"], cfg["kv_proj_stride"], cfg["depth"], cfg["heads"],
                    mlp_mult=cfg["mlp_mult"], dropout=dropout))


class _CvTBase(_HipNet):
    _kind = "cvt"

    def __init__(self, num_classes=2, dropout=0.0, dropout_fc=0.3, depth=1, width=NPOS, dim=NCHAN, apply_softmax=False,
                 model_type="acgt", **stage_kw):
        super().__init__()
        # constructor defaults of clairs/model.py:153-177 (the shipped pickles may use other values; the
        # forward derives everything from tensor shapes)
        dflt = dict(emb_kernel=3, emb_stride=2, proj_kernel=3, kv_proj_stride=2, mlp_mult=4)
        per = {"s1": dict(emb_dim=32, heads=1, depth=1), "s2": dict(emb_dim=64, heads=3, depth=2),
               "s3": dict(emb_dim=128, heads=6, depth=10)}
        self.model_type, self.layers_prefix, self.apply_softmax = model_type, ("s1", "s2", "s3"), apply_softmax
        d_in, w = dim, width
        for i, p in enumerate(self.layers_prefix):
            cfg = dict(dflt, **per[p])
            cfg.update({k[len(p) + 1:]: v for k, v in stage_kw.items() if k.startswith(p + "_")})
            setattr(self, "layer%d" % (i + 1), _stage(d_in, cfg, dropout))
            d_in, w = cfg["emb_dim"], -(-w // 2)
        self.dropout_fc1, self.dropout_fc2, self.flatten = nn.Dropout(dropout_fc), nn.Dropout(dropout_fc), nn.Flatten()
        self.fc1 = nn.Linear(d_in * w, 128)
        self.fc2 = nn.Linear(128, num_classes)          # present in the state_dict, unused by forward
        for h in self._heads_out:
            setattr(self, h + "_fc2", nn.Linear(128, 128))
        for h in self._heads_out:
            setattr(self, h + "_fc3", nn.Linear(128, num_classes))
        self.selu = nn.SELU()
        if apply_softmax:
            self.softmax = nn.Softmax(dim=-1)

    def _cfg(self):
        cfg = CvtCfg()
        for i in range(3):
            st = getattr(self, "layer%d" % (i + 1))
            cfg.emb_dim[i] = st[0].weight.shape[0]
            blk = st[2].layers[0][0].fn
            cfg.heads[i] = blk.to_q.net[2].weight.shape[0] // 64
            cfg.depth[i] = len(st[2].layers)
        cfg.n_out = len(self._heads_out)
        return cfg

    def forward(self, x):
        if self.model_type == "acgt":                   # clairs/model.py:244 - any other value returns None
            return self._tuple(x)


class CvT(_CvTBase):
    """clairs.model.CvT (clairs/model.py:150-261): heads a, c, g, t."""
    _heads_out = _SNV


class CvT_Indel(_CvTBase):
    """clairs.model.CvT_Indel (clairs/model.py:263-384): heads a, c, g, t, i, d."""
    _heads_out = _INDEL


class _BiGRUBase(_HipNet):
    _kind = "bigru"

    def __init__(self, num_classes=2, width=NPOS, batch_first=True, apply_softmax=False, channel_size=NCHAN,
                 model_type="acgt"):
        super().__init__()
        self.model_type, self.apply_softmax = model_type, apply_softmax
        self.num_layers, self.flatten = 2, nn.Flatten()
        self.lstm_hidden_size, self.lstm_hidden_size2, self.dim = 128, 192, channel_size
        self.input_shape = [width, 2 * self.lstm_hidden_size2]
        # the reference names its GRUs `lstm` / `lstm_2` (clairs/model.py:412-417); the state_dict keys depend on it
        self.lstm = nn.GRU(channel_size, 128, num_layers=1, batch_first=batch_first, bidirectional=True)
        self.lstm_2 = nn.GRU(256, 192, num_layers=1, batch_first=batch_first, bidirectional=True)
        self.dropout_fc1, self.dropout_fc2 = nn.Dropout(0.3), nn.Dropout(0.3)
        self.fc1 = nn.Linear(width * 384, 128)
        self.fc2 = nn.Linear(128, 128)                 # unused by forward, kept for the state_dict
        for h in self._heads_out:
            setattr(self, "n" + h + "_fc2", nn.Linear(128, 128))
        for h in self._heads_out:
            setattr(self, "n" + h + "_fc3", nn.Linear(128, num_classes))
        self.selu = nn.SELU()
        if apply_softmax:
            self.softmax = nn.Softmax(dim=-1)

    def forward(self, x):
        if self.model_type == "nacgt":                  # clairs/model.py:450 (the constructor default returns None)
            return self._tuple(x)


class BiGRU_NACGT(_BiGRUBase):
    """clairs.model.BiGRU_NACGT (clairs/model.py:387-467): heads na, nc, ng, nt."""
    _heads_out = _SNV


class BiGRU_NACGT_Indel(_BiGRUBase):
    """clairs.model.BiGRU_NACGT_Indel (clairs/model.py:470-560): heads na, nc, ng, nt, ni, nd."""
    _heads_out = _INDEL


def install_reference_aliases():
    """Make `clairs.model` resolve to this module so pickled reference checkpoints load onto the HIP classes."""
    me = sys.modules[__name__]
    pkg = sys.modules.get("clairs")
    if pkg is None:
        pkg = types.ModuleType("clairs")
        pkg.__path__ = []
        sys.modules["clairs"] = pkg
    pkg.model = me
    sys.modules["clairs.model"] = me
    return me


def from_state_dict(cls_name, state_dict, **kw):
    """Build one of the four classes and load a reference state_dict (CPU tensors / numpy arrays)."""
    cls = {"CvT": CvT, "CvT_Indel": CvT_Indel, "BiGRU_NACGT": BiGRU_NACGT, "BiGRU_NACGT_Indel": BiGRU_NACGT_Indel}[cls_name]
    sd = {k: torch.as_tensor(np.asarray(v)) for k, v in state_dict.items()}
    if cls_name.startswith("CvT"):
        emb = [sd["layer%d.0.weight" % i].shape[0] for i in (1, 2, 3)]
        heads = [sd["layer%d.2.layers.0.0.fn.to_q.net.2.weight" % i].shape[0] // 64 for i in (1, 2, 3)]
        depth = [len({k.split(".")[3] for k in sd if k.startswith("layer%d.2.layers." % i)}) for i in (1, 2, 3)]
        for i in range(3):
            kw.setdefault("s%d_emb_dim" % (i + 1), emb[i])
            kw.setdefault("s%d_heads" % (i + 1), heads[i])
            kw.setdefault("s%d_depth" % (i + 1), depth[i])
        kw.setdefault("model_type", "acgt")
    else:
        kw.setdefault("model_type", "nacgt")
    m = cls(**kw)
    full = m.state_dict()
    for k, v in sd.items():
        full[k] = v.reshape(full[k].shape).to(full[k].dtype)
    m.load_state_dict(full)
    return m.eval()

"""Pileup-tensor creation on the GPU: host wrapper of cto_featurize_sites (one kernel, the default) and of the two-stage path
cto_featurize_columns / cto_gather_windows (every column's vector in HBM: candidate extraction, A/B runs).

Mirrors what src/create_tensor_pileup_calling.py (reference) produces for ONE chunk of candidates, for the
AFF pass (--min_bq <platform>) and the NEG pass (--min_bq 0) at once, plus the rescale and strand counts that
clairs/predict.py derives from the tensor text (predict.py:172-207, 626-642)."""
import ctypes as C
import os

import numpy as np
import torch

from ._lib import lib, check, current_stream_ptr

NPOS, NCHAN, COLVEC_STRIDE = 33, 34, 72


class Features:
    """What tensor creation leaves in HBM for one batch of candidates.  The network inputs exist as the un-rescaled int16 tensors
    (raw_aff / raw_neg - the reference's own tensor text, create_tensor_pileup_calling.py) and / or as the rescaled fp32 tensors
    (x_aff / x_neg - what clairs/predict.py:172-207 makes of that text); a record that was made without the fp32 form derives it on
    first use from the int16 one: float(double(v) * min_rescale_cov / depth), the tensor kernel's own expression, same bits."""

    def __init__(self, x_aff, x_neg, raw_aff, raw_neg, site_info, colvec, coldepth, sitefirst, keycnt, keyfirst, site_colvec=None,
                 min_rescale_cov=0):
        self._x = [x_aff, x_neg]           # [n,33,34] float32, rescaled AFF / NEG tensors (network inputs) or None
        self.raw_aff = raw_aff             # [n,33,34] int16 or None
        self.raw_neg = raw_neg
        self.site_info = site_info         # [n,12] int32: centre col, depth_aff, depth_neg, flags, fwd ACGT, rev ACGT
        self.colvec = colvec               # [n_cols,72] int16 (two-stage path only; None from the one-kernel path)
        self.coldepth = coldepth           # [n_cols,2] int32 (two-stage path only)
        self.sitefirst = sitefirst         # [n,8] int32 ([pass][A,C,G,T]) first-seen entry index within the candidate column
        self.keycnt = keycnt               # [n_keys] int32 (uint32 bits: low16 AFF count, high16 NEG count)
        self.keyfirst = keyfirst           # [n_keys,2] int32 (per pass; defined for the keys of candidate columns only)
        self.site_colvec = site_colvec     # [n,72] int16: the candidate column's own vector (one-kernel path; keycnt is then
        self.min_rescale_cov = int(min_rescale_cov or 0)      # defined for the keys of candidate columns only, like keyfirst)

    def _expanded(self, which):
        if self._x[which] is None:
            raw = self.raw_aff if which == 0 else self.raw_neg
            if raw is None:
                return None
            depth = self.site_info[:, 1 + which].to(torch.float64)
            cov = float(self.min_rescale_cov)
            scale = torch.where(depth > cov, cov / depth, torch.ones_like(depth)) if cov > 0 else torch.ones_like(depth)
            self._x[which] = (raw.to(torch.float64) * scale.view(-1, 1, 1)).to(torch.float32)
        return self._x[which]

    x_aff = property(lambda self: self._expanded(0), lambda self, v: self._x.__setitem__(0, v))
    x_neg = property(lambda self: self._expanded(1), lambda self, v: self._x.__setitem__(1, v))


def fused_default():
    """CTO_FUSED_FEATURIZE=0 selects the two-stage path everywhere (pipeline.hip reads the same variable)."""
    return os.environ.get("CTO_FUSED_FEATURIZE", "1") != "0"


def featurize(dev_pack, site_pos, min_bq, min_rescale_cov=50, want_raw=False, want_x=True, fused=None):
    """dev_pack: DevicePack; site_pos: int32 tensor on the same device (1-based candidate positions).
    fused (default: on unless CTO_FUSED_FEATURIZE=0): one kernel, per-candidate outputs only (Features.site_colvec instead of
    .colvec / .coldepth); the same tensors, strand counts, alt_info inputs either way."""
    dev = dev_pack.device
    if site_pos.device != dev or site_pos.dtype != torch.int32:
        site_pos = site_pos.to(device=dev, dtype=torch.int32)
    site_pos = site_pos.contiguous()
    n = site_pos.numel()
    nc, nk = dev_pack.n_cols, dev_pack.n_keys
    # default: one kernel unless the windows overlap so much (< 8 pack columns per candidate) that histogramming every column once
    # and gathering is cheaper - the same rule as csrc/pipeline.hip; results are identical
    if (fused_default() and nc >= 8 * n) if fused is None else fused:
        keycnt = torch.empty((max(nk, 1),), dtype=torch.int32, device=dev)
        keyfirst = torch.empty((max(nk, 1), 2), dtype=torch.int32, device=dev)
        x_aff = torch.empty((n, NPOS, NCHAN), dtype=torch.float32, device=dev) if want_x else None
        x_neg = torch.empty((n, NPOS, NCHAN), dtype=torch.float32, device=dev) if want_x else None
        raw_aff = torch.empty((n, NPOS, NCHAN), dtype=torch.int16, device=dev) if want_raw else None
        raw_neg = torch.empty((n, NPOS, NCHAN), dtype=torch.int16, device=dev) if want_raw else None
        site_info = torch.empty((n, 12), dtype=torch.int32, device=dev)
        sitefirst = torch.empty((max(n, 1), 8), dtype=torch.int32, device=dev)
        site_colvec = torch.empty((max(n, 1), COLVEC_STRIDE), dtype=torch.int16, device=dev)
        ptr = lambda t: t.data_ptr() if t is not None else None
        check(lib.cto_featurize_sites(C.byref(dev_pack.view), site_pos.data_ptr(), n, int(min_bq), int(min_rescale_cov) if min_rescale_cov else 0,
                                      ptr(x_aff), ptr(x_neg), ptr(raw_aff), ptr(raw_neg), site_info.data_ptr(), site_colvec.data_ptr(),
                                      sitefirst.data_ptr(), keycnt.data_ptr(), keyfirst.data_ptr(), current_stream_ptr()))
        return Features(x_aff, x_neg, raw_aff, raw_neg, site_info, None, None, sitefirst[:n], keycnt[:nk], keyfirst[:nk], site_colvec[:n],
                        min_rescale_cov=min_rescale_cov)
    colvec = torch.empty((max(nc, 1), COLVEC_STRIDE), dtype=torch.int16, device=dev)   # never a null pointer
    coldepth = torch.empty((max(nc, 1), 2), dtype=torch.int32, device=dev)
    keycnt = torch.empty((max(nk, 1),), dtype=torch.int32, device=dev)
    keyfirst = torch.empty((max(nk, 1), 2), dtype=torch.int32, device=dev)
    s = current_stream_ptr()
    check(lib.cto_featurize_columns(C.byref(dev_pack.view), int(min_bq), colvec.data_ptr(), coldepth.data_ptr(),
                                    keycnt.data_ptr(), s))
    x_aff = torch.empty((n, NPOS, NCHAN), dtype=torch.float32, device=dev) if want_x else None
    x_neg = torch.empty((n, NPOS, NCHAN), dtype=torch.float32, device=dev) if want_x else None
    raw_aff = torch.empty((n, NPOS, NCHAN), dtype=torch.int16, device=dev) if want_raw else None
    raw_neg = torch.empty((n, NPOS, NCHAN), dtype=torch.int16, device=dev) if want_raw else None
    site_info = torch.empty((n, 12), dtype=torch.int32, device=dev)
    sitefirst = torch.empty((max(n, 1), 8), dtype=torch.int32, device=dev)
    ptr = lambda t: t.data_ptr() if t is not None else None
    check(lib.cto_gather_windows(C.byref(dev_pack.view), colvec.data_ptr(), coldepth.data_ptr(), site_pos.data_ptr(), n,
                                 int(min_bq), int(min_rescale_cov) if min_rescale_cov else 0, ptr(x_aff), ptr(x_neg), ptr(raw_aff),
                                 ptr(raw_neg), site_info.data_ptr(), sitefirst.data_ptr(), keyfirst.data_ptr(), s))
    return Features(x_aff, x_neg, raw_aff, raw_neg, site_info, colvec[:nc], coldepth[:nc], sitefirst[:n], keycnt[:nk], keyfirst[:nk],
                    min_rescale_cov=min_rescale_cov)


def featurize_op(dev_pack, site_pos, min_bq, min_rescale_cov=50):
    """The same through PyTorch's dispatcher: torch.ops.clairsto.pileup_featurize (csrc/torch_ops.cpp) on the pack's device
    tensors.  Network inputs only (no raw int16 tensors); returns the same Features record as featurize()."""
    t = dev_pack.t
    site_pos = site_pos.to(device=dev_pack.device, dtype=torch.int32).contiguous()
    x_aff, x_neg, site_info, colvec, coldepth, keycnt, sitefirst, keyfirst = torch.ops.clairsto.pileup_featurize(
        t["entries"], t["col_off"], t["col_pos"], t["col_ref"], t["key_off"], t["key_meta"], t["key_group"], site_pos, int(min_bq),
        int(min_rescale_cov) if min_rescale_cov else 0)
    return Features(x_aff, x_neg, None, None, site_info, colvec, coldepth, sitefirst, keycnt, keyfirst, min_rescale_cov=min_rescale_cov)


def alt_infos(feat, host_pack, site_info_host=None, pass_idx=0):
    """The reference's alt_info strings of pass `pass_idx` (0 = AFF, 1 = NEG) for every site with a centre column
    (create_tensor_pileup_calling.py:158-209); '' for sites without one.  Host work on device results."""
    raw, offsets = alt_infos_packed(feat, host_pack, site_info_host, pass_idx)
    return [raw[offsets[i]:offsets[i + 1]].decode() for i in range(len(offsets) - 1)]


def alt_infos_packed(feat, host_pack, site_info_host=None, pass_idx=0):
    """alt_infos() without the per-site Python strings: (bytes with the strings back to back, int64 offsets [n + 1]) - the form
    cto_vcf_rows_batch consumes."""
    info = feat.site_info.cpu().numpy() if site_info_host is None else site_info_host
    per_site = feat.site_colvec is not None
    colvec = feat.site_colvec if per_site else feat.colvec
    return alt_infos_from_host(host_pack, info, colvec.cpu().numpy(), feat.sitefirst.cpu().numpy(), feat.keycnt.cpu().numpy(),
                               feat.keyfirst.cpu().numpy(), pass_idx, per_site=per_site)


def alt_infos_from_host(host_pack, info, colvec, sitefirst, keycnt, keyfirst, pass_idx=0, per_site=False):
    """alt_infos_packed() on host copies (numpy arrays) of the featurisation outputs - what a pipelined caller has after its own
    asynchronous device-to-host copies (call_chunks).  per_site: `colvec` holds one row per candidate (its candidate column's
    vector, gathered on the device: 144 B per site cross PCIe instead of 144 B per pack column)."""
    sitefirst = np.ascontiguousarray(sitefirst)
    keycnt = np.ascontiguousarray(np.asarray(keycnt).view(np.uint32))
    keyfirst = np.ascontiguousarray(keyfirst)
    if keycnt.size == 0:
        keycnt = np.zeros(1, dtype=np.uint32)
        keyfirst = np.zeros((1, 2), dtype=np.int32)
    info = np.ascontiguousarray(info, dtype=np.int32)
    n = info.shape[0]
    if n == 0:
        return b"", np.zeros(1, dtype=np.int64)
    colvec = np.ascontiguousarray(colvec)
    cap = 256 * n + (1 << 16)
    offsets = np.zeros(n + 1, dtype=np.int64)
    while True:
        buf = C.create_string_buffer(cap)
        fn = lib.cto_alt_info_batch_sites if per_site else lib.cto_alt_info_batch
        used = fn(host_pack._h, n, info.ctypes.data, int(pass_idx), colvec.ctypes.data, sitefirst.ctypes.data,
                  keycnt.ctypes.data, keyfirst.ctypes.data, C.addressof(buf), cap, offsets.ctypes.data)
        if used >= 0:
            break
        if "buffer too small" not in lib.cto_last_error().decode():
            check(int(used))
        cap *= 4
    return buf.raw[:int(used)], offsets

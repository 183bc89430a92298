"""Column packs on the host and in HBM (see include/clairsto_amd.h for the binary layout)."""
import ctypes as C

import numpy as np
import torch

from ._lib import lib, check, PackView, c_vp

_FIELDS = (("col_pos", np.int32), ("col_ref", np.uint8), ("col_off", np.int64), ("key_off", np.int32),
           ("entries", np.uint32), ("key_meta", np.uint8), ("key_group", np.int32))


class ColumnPack:
    """Host-side pack (owns a cto_pack). Build it from mpileup text or from arrays."""

    def __init__(self, handle):
        self._h = c_vp(handle)
        v = PackView()
        check(lib.cto_pack_view_of(self._h, C.byref(v)))
        self.view = v
        self.n_cols, self.n_entries, self.n_keys = int(v.n_cols), int(v.n_entries), int(v.n_keys)

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            lib.cto_pack_free(h)

    @classmethod
    def from_mpileup(cls, text, ref_seq, ref_start, max_indel_length=60):
        """text: `samtools mpileup --reverse-del --output-MQ --min-BQ 0` rows of ONE contig, increasing position.
        Replaces the tokeniser of src/create_tensor_pileup_calling.py:120-144 (reference)."""
        if isinstance(text, np.ndarray):                      # e.g. np.memmap of a text file: tokenised straight from the page cache
            arr = text if text.dtype == np.uint8 else text.view(np.uint8)
        else:
            tb = text.encode() if isinstance(text, str) else bytes(text)
            arr = np.frombuffer(tb, dtype=np.uint8)
        n = int(arr.size)
        if n == 0:
            arr = np.zeros(1, dtype=np.uint8)
        rb = ref_seq.encode() if isinstance(ref_seq, str) else bytes(ref_seq)
        out = c_vp()
        check(lib.cto_pack_from_mpileup(arr.ctypes.data, n, rb, int(ref_start), len(rb), int(max_indel_length), C.byref(out)))
        return cls(out.value)

    @classmethod
    def from_bam(cls, bam_fn, ctg_name, start, end, ref_seq, ref_start, bed=None, bai_fn=None, excl_flags=2316, min_mq=0,
                 max_depth=8000, max_indel_length=60, inflated=None):
        """The pack of `samtools mpileup -r ctg:start-end --min-BQ 0 ...` on `bam_fn`, without samtools (cto_pack_from_bam;
        PARITY UNPINNED, see csrc/bam.cpp).  bed: iterable of 0-based [begin, end) intervals restricting the positions.
        inflated = (host uint8 tensor, block table) from bgzf.inflate_span: the BGZF blocks were inflated on the device."""
        rb = ref_seq.encode() if isinstance(ref_seq, str) else bytes(ref_seq)
        iv = None
        if bed is not None:
            iv = sorted((int(a), int(b)) for a, b in bed)
            merged = []
            for a, b in iv:
                if merged and a <= merged[-1][1]:
                    merged[-1][1] = max(merged[-1][1], b)
                else:
                    merged.append([a, b])
            iv = np.ascontiguousarray(np.array(merged, dtype=np.int64).reshape(-1, 2))
        out = c_vp()
        if inflated is not None:
            h_out, blocks = inflated
            blocks = np.ascontiguousarray(blocks)
            check(lib.cto_pack_from_bam_inflated(str(bam_fn).encode(), str(bai_fn).encode() if bai_fn else None, ctg_name.encode(),
                                                 int(start), int(end), iv.ctypes.data if iv is not None and len(iv) else None,
                                                 len(iv) if iv is not None else 0, rb, int(ref_start), len(rb), int(excl_flags),
                                                 int(min_mq), int(max_depth), int(max_indel_length), h_out.data_ptr(),
                                                 int(h_out.numel()), blocks.ctypes.data, len(blocks), C.byref(out)))
            return cls(out.value)
        check(lib.cto_pack_from_bam(str(bam_fn).encode(), str(bai_fn).encode() if bai_fn else None, ctg_name.encode(), int(start),
                                    int(end), iv.ctypes.data if iv is not None and len(iv) else None, len(iv) if iv is not None else 0,
                                    rb, int(ref_start), len(rb), int(excl_flags), int(min_mq), int(max_depth), int(max_indel_length),
                                    C.byref(out)))
        return cls(out.value)

    @classmethod
    def from_arrays(cls, col_pos, col_ref, col_off, key_off, entries, key_meta, key_group=None, key_str_off=None,
                    key_str=None):
        if key_group is None:
            key_group = np.zeros(len(key_meta), dtype=np.int32)
        arrs = dict(col_pos=col_pos, col_ref=col_ref, col_off=col_off, key_off=key_off, entries=entries, key_meta=key_meta,
                    key_group=key_group)
        keep = {k: np.ascontiguousarray(arrs[k], dtype=dt) for k, dt in _FIELDS}
        v = PackView()
        v.n_cols, v.n_entries, v.n_keys = len(keep["col_pos"]), len(keep["entries"]), len(keep["key_meta"])
        assert len(keep["col_off"]) == v.n_cols + 1 and len(keep["key_off"]) == v.n_cols + 1
        for k, _ in _FIELDS:
            setattr(v, k, keep[k].ctypes.data)
        kso = ks = None
        if key_str_off is not None:
            kso = np.ascontiguousarray(key_str_off, dtype=np.int64)
            ks = bytes(key_str)
        out = c_vp()
        check(lib.cto_pack_from_arrays(C.byref(v), kso.ctypes.data if kso is not None else None, ks, C.byref(out)))
        return cls(out.value)

    def numpy(self):
        """Zero-copy numpy views of the pack arrays (valid while this object lives)."""
        v = self.view
        n = dict(col_pos=v.n_cols, col_ref=v.n_cols, col_off=v.n_cols + 1, key_off=v.n_cols + 1,
                 entries=v.n_entries, key_meta=v.n_keys, key_group=v.n_keys)
        out = {}
        for k, dt in _FIELDS:
            ptr = getattr(v, k)
            cnt = int(n[k])
            if cnt == 0 or not ptr:
                out[k] = np.zeros(0, dtype=dt)
            else:
                buf = (C.c_char * (cnt * np.dtype(dt).itemsize)).from_address(ptr)
                out[k] = np.frombuffer(buf, dtype=dt, count=cnt)
        return out

    def key_string(self, k):
        s = C.c_char_p()
        n = check(lib.cto_pack_key_string(self._h, int(k), C.byref(s)))
        return C.string_at(s, n).decode()

    def to_device(self, device="cuda"):
        return DevicePack(self.numpy(), device, host=self)


class DeviceTokeniser:
    """mpileup text -> pack in HBM (cto_tokenise_device, csrc/tokenise.hip).  One context per thread / stream; the arrays of a result
    live in the context until its next call."""

    def __init__(self):
        h = c_vp()
        check(lib.cto_dev_tokeniser_create(C.byref(h)))
        self._h = h

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            lib.cto_dev_tokeniser_destroy(h)

    def __call__(self, text, ref_seq, ref_start, max_indel_length=60):
        """-> (PackView of device pointers, ColumnPack `lite` without entries) or None when the text has to go through
        ColumnPack.from_mpileup (fallback)."""
        from ._lib import current_stream_ptr
        tb = text.encode() if isinstance(text, str) else bytes(text)
        rb = ref_seq.encode() if isinstance(ref_seq, str) else bytes(ref_seq)
        arr = np.frombuffer(tb, dtype=np.uint8) if tb else np.zeros(1, dtype=np.uint8)
        v, lite, fb = PackView(), c_vp(), C.c_int(0)
        check(lib.cto_tokenise_device(self._h, arr.ctypes.data, len(tb), rb, int(ref_start), len(rb), int(max_indel_length),
                                      current_stream_ptr(), C.byref(v), C.byref(lite), C.byref(fb)))
        if fb.value:
            return None
        return v, ColumnPack(lite.value)

    @staticmethod
    def download(view):
        """the device arrays of a PackView as numpy arrays (tests)"""
        n = dict(col_pos=view.n_cols, col_ref=view.n_cols, col_off=view.n_cols + 1, key_off=view.n_cols + 1, entries=view.n_entries,
                 key_meta=view.n_keys, key_group=view.n_keys)
        out = {}
        for k, dt in _FIELDS:
            a = np.zeros(int(n[k]), dtype=dt)
            if a.size:
                check(lib.cto_device_read(getattr(view, k), a.ctypes.data, a.nbytes))
            out[k] = a
        return out


def pin_arrays(arrays):
    """Copies of the pack arrays in page-locked host memory (torch tensors), for asynchronous uploads.  A producer that
    fills such buffers directly (instead of copying into them) gets the PCIe transfer fully off the critical path."""
    out = {}
    for k, dt in _FIELDS:
        a = np.ascontiguousarray(arrays[k], dtype=dt)
        t = torch.from_numpy(a.view(np.int32) if dt == np.uint32 else a.copy() if a.size == 0 else a)
        out[k] = t.pin_memory()
    return out


class DevicePack:
    """The pack arrays resident in HBM (torch owns the memory) plus the cto_pack_view of device pointers.
    arrays: numpy arrays (synchronous upload) or the pinned tensors of `pin_arrays` (asynchronous, on the current stream)."""

    def __init__(self, arrays, device="cuda", host=None):
        self.host = host
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("clairs_to_amd: the featurisation kernels need a HIP device; there is no CPU fallback")
        self.t = {}
        for k, dt in _FIELDS:
            if isinstance(arrays[k], torch.Tensor):
                self.t[k] = arrays[k].to(self.device, non_blocking=True)
                continue
            a = np.ascontiguousarray(arrays[k], dtype=dt)
            # torch has no uint32: carry the bits in int32
            t = torch.from_numpy(a.view(np.int32) if dt == np.uint32 else a.copy() if a.size == 0 else a)
            self.t[k] = t.to(self.device, non_blocking=False)
        v = PackView()
        v.n_cols = len(arrays["col_pos"])
        v.n_entries = len(arrays["entries"])
        v.n_keys = len(arrays["key_meta"])
        for k, _ in _FIELDS:
            setattr(v, k, self.t[k].data_ptr())
        self.view = v
        self.n_cols, self.n_entries, self.n_keys = int(v.n_cols), int(v.n_entries), int(v.n_keys)

    def nbytes(self):
        return sum(t.numel() * t.element_size() for t in self.t.values())

"""BGZF blocks of a BAM inflated on the device (csrc/inflate.hip: one wavefront per block) for the native BAM -> pack producer.
On the host, inflate is what bounds that producer; here the host only reads the compressed byte range the
index names for a chunk, finds the block boundaries, and gets the inflated records back over PCIe."""
import ctypes as C
import threading

import numpy as np

from ._lib import CtoError, check, lib

BGZF_PAD = 1024            # include/clairsto_amd.h: CTO_BGZF_PAD
BLOCK_DTYPE = np.dtype([("file_off", "<u8"), ("in_off", "<u8"), ("out_off", "<u8"), ("csize", "<u4"), ("isize", "<u4"),
                        ("bsize", "<u4"), ("crc32", "<u4")])
STATUS = {1: "reserved block type", 2: "stored block length check", 3: "bad code-length table", 4: "invalid literal / length code",
          5: "invalid distance", 6: "more output than ISIZE", 7: "ran past the compressed data", 8: "less output than ISIZE",
          9: "output slot closer than CTO_BGZF_SLOT_PAD to the next one (block table not laid out by cto_bgzf_scan)"}

_tls = threading.local()


def _pinned(name, nbytes):
    """a page-locked uint8 host buffer of at least nbytes, kept per thread (pinning tens of MB per chunk would cost more than the copy)"""
    import torch
    cur = getattr(_tls, name, None)
    if cur is None or cur.numel() < nbytes:
        cur = torch.empty(int(nbytes * 1.25) + 4096, dtype=torch.uint8, pin_memory=True)
        setattr(_tls, name, cur)
    return cur


def scan(buf, nbytes, file_begin=0):
    """block table (numpy structured array) of the whole BGZF blocks in buf[:nbytes]; -> (blocks, out_bytes)"""
    arr = np.frombuffer(buf, dtype=np.uint8) if not isinstance(buf, np.ndarray) else buf
    cap = max(64, nbytes // 512 + 64)
    while True:
        blocks = np.zeros(cap, dtype=BLOCK_DTYPE)
        out_bytes = C.c_int64(0)
        n = lib.cto_bgzf_scan(arr.ctypes.data, C.c_size_t(int(nbytes)), C.c_int64(int(file_begin)), blocks.ctypes.data, C.c_int64(cap),
                              C.byref(out_bytes))
        if n == -3 and cap < (1 << 24):       # CTO_ENOMEM: many tiny blocks
            cap *= 8
            continue
        if n < 0:
            check(int(n))
        return blocks[:int(n)], int(out_bytes.value)


def inflate_device(d_bytes, blocks, out_bytes, device, stream=None):
    """d_bytes: uint8 device tensor holding the byte range (+ BGZF_PAD); -> (uint8 device tensor of out_bytes, int32 device status)"""
    import torch
    n = len(blocks)
    d_blocks = torch.from_numpy(np.ascontiguousarray(blocks).view(np.uint8).reshape(-1)).to(device, non_blocking=True) if n else None
    d_out = torch.empty(max(out_bytes, 256), dtype=torch.uint8, device=device)
    d_status = torch.zeros(max(n, 1), dtype=torch.int32, device=device)
    s = stream if stream is not None else torch.cuda.current_stream(device)
    if n:
        check(lib.cto_bgzf_inflate(d_bytes.data_ptr(), d_blocks.data_ptr(), n, d_out.data_ptr(), d_status.data_ptr(), C.c_void_p(s.cuda_stream)))
    return d_out, d_status


def check_status(status, blocks):
    bad = np.nonzero(status[:len(blocks)])[0]
    if len(bad):
        b = int(bad[0])
        raise CtoError("BGZF block at file offset %d does not inflate: %s" % (int(blocks["file_off"][b]), STATUS.get(int(status[b]), "error")))


def inflate_bytes(raw, device):
    """every whole BGZF block of `raw` (bytes) inflated on `device` -> list of bytes objects (tests, tools)"""
    import torch
    n = len(raw)
    host = np.zeros(n + BGZF_PAD, dtype=np.uint8)
    host[:n] = np.frombuffer(raw, dtype=np.uint8)
    blocks, out_bytes = scan(host, n)
    d_in = torch.from_numpy(host).to(device)
    d_out, d_status = inflate_device(d_in, blocks, out_bytes, device)
    torch.cuda.synchronize(device)
    check_status(d_status.cpu().numpy(), blocks)
    out = d_out.cpu().numpy()
    res = [out[int(b["out_off"]):int(b["out_off"]) + int(b["isize"])].tobytes() for b in blocks]
    import zlib
    for b, r in zip(blocks, res):          # the gzip trailer's CRC-32, as cto_pack_from_bam_inflated checks it for every block it reads
        if zlib.crc32(r) != int(b["crc32"]):
            raise CtoError("BGZF block at file offset %d fails its CRC-32" % int(b["file_off"]))
    return res


def inflate_span(bam_fn, bai_fn, ctg_name, start, end, device, stream=None):
    """The BGZF blocks holding the alignments that overlap ctg:start-end, inflated on the device and copied back:
    -> (page-locked uint8 host tensor with the inflated blocks, block table).  Runs on `stream` and waits for it."""
    import torch
    fb, fe = C.c_int64(0), C.c_int64(0)
    check(lib.cto_bam_chunk_span(str(bam_fn).encode(), str(bai_fn).encode() if bai_fn else None, ctg_name.encode(), int(start), int(end),
                                 C.byref(fb), C.byref(fe)))
    nbytes = max(0, fe.value - fb.value)
    h_in = _pinned("h_in", nbytes + BGZF_PAD)
    view = h_in.numpy()
    if nbytes:
        with open(bam_fn, "rb", buffering=0) as f:
            f.seek(fb.value)
            got = f.readinto(memoryview(view)[:nbytes])
            while got is not None and 0 < got < nbytes:
                more = f.readinto(memoryview(view)[got:nbytes])
                if not more:
                    break
                got += more
            nbytes = got or 0
    view[nbytes:nbytes + BGZF_PAD] = 0
    blocks, out_bytes = scan(view, nbytes, fb.value)
    s = stream if stream is not None else torch.cuda.current_stream(device)
    with torch.cuda.stream(s):
        d_in = h_in[:nbytes + BGZF_PAD].to(device, non_blocking=True)
        d_out, d_status = inflate_device(d_in, blocks, out_bytes, device, s)
        h_out = _pinned("h_out", max(out_bytes, 256))
        h_out[:max(out_bytes, 256)].copy_(d_out, non_blocking=True)
        h_status = d_status.to("cpu", non_blocking=False)
    s.synchronize()
    check_status(h_status.numpy(), blocks)
    return h_out, blocks


class DevicePileup(object):
    """csrc/pileup.hip: the column pack of a chunk built on the device from the blocks inflated there (one reusable context).
    pileup(...) -> (PackView of device arrays owned by the context - valid until the next call -, host pack handle without entries,
    fallback flag); with fallback the chunk needs the host reader (cto_pack_from_bam)."""

    def __init__(self):
        self.h = C.c_void_p()
        check(lib.cto_dev_pileup_create(C.byref(self.h)))

    def __del__(self):
        if getattr(self, "h", None) and self.h.value:
            lib.cto_dev_pileup_destroy(self.h)
            self.h = C.c_void_p()

    def pileup(self, bam_fn, bai_fn, ctg_name, start, end, ref_seq, ref_start, device, bed=None, excl_flags=2316, min_mq=0, max_depth=8000,
               max_indel_length=60, stream=None):
        import torch
        from ._lib import PackView
        fb, fe = C.c_int64(0), C.c_int64(0)
        bai = str(bai_fn).encode() if bai_fn else None
        check(lib.cto_bam_chunk_span(str(bam_fn).encode(), bai, ctg_name.encode(), int(start), int(end), C.byref(fb), C.byref(fe)))
        nbytes = max(0, fe.value - fb.value)
        h_in = _pinned("h_in", nbytes + BGZF_PAD)
        view = h_in.numpy()
        if nbytes:
            with open(bam_fn, "rb", buffering=0) as f:
                f.seek(fb.value)
                got = 0
                while got < nbytes:
                    more = f.readinto(memoryview(view)[got:nbytes])
                    if not more:
                        break
                    got += more
                nbytes = got
        view[nbytes:nbytes + BGZF_PAD] = 0
        blocks, out_bytes = scan(view, nbytes, fb.value)
        pv, lite, fallback = PackView(), C.c_void_p(), C.c_int(0)
        if len(blocks) == 0:
            return None, None, True
        cap = 4096 + int((end - start) >> 14) + 64
        voffs = np.zeros(cap, dtype=np.uint64)
        tid = C.c_int32(-1)
        n_st = -3
        for _ in range(4):                                   # CTO_ENOMEM (-3): the index names more offsets than the table holds
            n_st = int(lib.cto_bam_record_starts(str(bam_fn).encode(), bai, ctg_name.encode(), int(start), int(end), fb.value, fe.value,
                                                 voffs.ctypes.data, cap, C.byref(tid)))
            if n_st != -3:
                break
            cap *= 8
            voffs = np.zeros(cap, dtype=np.uint64)
        if n_st == -3:
            return None, None, True                          # still too many: the host reader takes the region
        check(n_st)
        if n_st == 0:
            return None, None, True
        s = stream if stream is not None else torch.cuda.current_stream(device)
        with torch.cuda.stream(s):
            d_in = h_in[:nbytes + BGZF_PAD].to(device, non_blocking=True)
            d_out, d_status = inflate_device(d_in, blocks, out_bytes, device, s)
            h_status = d_status.to("cpu", non_blocking=False)
            s.synchronize()
            check_status(h_status.numpy(), blocks)
            bed_arr = None if bed is None else np.ascontiguousarray(np.asarray(bed, dtype=np.int64).reshape(-1))
            hb = np.ascontiguousarray(blocks)
            ref_b = ref_seq.encode() if isinstance(ref_seq, str) else ref_seq
            check(lib.cto_pileup_device(self.h, d_out.data_ptr(), hb.ctypes.data, len(hb), voffs.ctypes.data, int(n_st), tid.value, int(start), int(end),
                                        None if bed_arr is None else bed_arr.ctypes.data, 0 if bed_arr is None else len(bed_arr) // 2, ref_b,
                                        int(ref_start), len(ref_b), int(excl_flags), int(min_mq), int(max_depth), int(max_indel_length),
                                        C.c_void_p(s.cuda_stream), C.byref(pv), C.byref(lite), C.byref(fallback)))
        self._keep = (d_in, d_out)
        return pv, lite, bool(fallback.value)

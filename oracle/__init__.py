"""ctypes wrapper of the CPU oracle (oracle/cto_oracle.c).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
Nothing under clairs_to_amd/ may import this package (tests/test_layout.py enforces it).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "liboracle.so")


_SSW_LIB = os.path.join(_HERE, "libsswmodel.so")


def build(force=False):
    stale = False
    for lib_, src in ((os.path.join(_HERE, "liboracle.so"), "cto_oracle.c"), (_SSW_LIB, "ssw_model.cpp")):
        src = os.path.join(_HERE, src)
        stale = stale or not os.path.exists(lib_) or os.path.getmtime(lib_) < os.path.getmtime(src)
    if force or stale:
        subprocess.check_call(["make", "-s", "-C", _HERE] + (["-B"] if force else []))
    return os.path.join(_HERE, "liboracle.so")


def build_fast():
    """The same source built for speed, for bench.py's cpu_baseline leg only (never the checker): -O3 -march=native -ffast-math
    on THIS host - the file name carries a hash of the host's CPU flags, so a library built on another machine is never loaded."""
    import hashlib
    flags = ""
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("flags"):
                flags = ln
                break
    except OSError:
        pass
    out = os.path.join(_HERE, "liboracle_fast.%s.so" % hashlib.sha1(flags.encode()).hexdigest()[:10])
    src = os.path.join(_HERE, "cto_oracle.c")
    if not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(src):
        # two steps: -ffast-math on the LINK line would pull in crtfastmath.o, whose constructor switches the whole process to
        # flush-to-zero / denormals-are-zero the moment the library is loaded
        obj = out[:-3] + ".o"
        subprocess.check_call(["gcc", "-O3", "-march=native", "-ffast-math", "-fPIC", "-fopenmp", "-std=c11", "-D_GNU_SOURCE", "-c", "-o", obj, src])
        subprocess.check_call(["gcc", "-shared", "-fopenmp", "-o", out, obj, "-lm"])
        os.remove(obj)
    return out


def use_library(path=None):
    """Switch the module to another build of cto_oracle.c (build_fast()'s) or back to the checker (None)."""
    global _lib, _LIB
    _lib = None
    _LIB = path if path else os.path.join(_HERE, "liboracle.so")


_ssw = None


def ssw_pass(ref_codes, read_codes, lanes, reverse=False, terminate=None):
    """Scalar model of one striped Smith-Waterman pass (oracle/ssw_model.cpp): (score, ref_end, read_end, overflow).
    ref_codes / read_codes: int8 arrays of base codes 0..4."""
    global _ssw
    if _ssw is None:
        build()
        _ssw = C.CDLL(_SSW_LIB)
        _ssw.orc_ssw_pass.restype = None
    ref_codes = np.ascontiguousarray(ref_codes, dtype=np.int8)
    read_codes = np.ascontiguousarray(read_codes, dtype=np.int8)
    out = np.zeros(4, dtype=np.int32)
    if terminate is None:
        terminate = 255 if lanes == 16 else 65535
    _ssw.orc_ssw_pass(_p(ref_codes), C.c_int(len(ref_codes)), C.c_int(int(bool(reverse))), _p(read_codes), C.c_int(len(read_codes)),
                      C.c_int(lanes), C.c_int(int(terminate)), _p(out))
    return int(out[0]), int(out[1]), int(out[2]), bool(out[3])


def ssw_pass_ex(ref_codes, read_codes, lanes, reverse=False, terminate=None, closed_form=False):
    """ssw_pass with the lazy-F loops as the reference has them or as their closed form; returns ((score, ref_end, read_end, overflow),
    hash of every H column after its lazy-F step)."""
    global _ssw
    if _ssw is None:
        build()
        _ssw = C.CDLL(_SSW_LIB)
        _ssw.orc_ssw_pass.restype = None
    ref_codes = np.ascontiguousarray(ref_codes, dtype=np.int8)
    read_codes = np.ascontiguousarray(read_codes, dtype=np.int8)
    out = np.zeros(4, dtype=np.int32)
    h = C.c_uint64(0)
    if terminate is None:
        terminate = 255 if lanes == 16 else 65535
    _ssw.orc_ssw_pass_ex.restype = None
    _ssw.orc_ssw_pass_ex(_p(ref_codes), C.c_int(len(ref_codes)), C.c_int(int(bool(reverse))), _p(read_codes), C.c_int(len(read_codes)),
                         C.c_int(lanes), C.c_int(int(terminate)), C.c_int(int(bool(closed_form))), _p(out), C.byref(h))
    return (int(out[0]), int(out[1]), int(out[2]), bool(out[3])), int(h.value)


_lib = None


class _Tensor(C.Structure):
    _fields_ = [("name", C.c_char_p), ("data", C.c_void_p), ("numel", C.c_int64)]


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            build()
        _lib = C.CDLL(_LIB)
        _lib.orc_decode_column.restype = C.c_int
        _lib.orc_ref_base.restype = C.c_char
        _lib.orc_ref_base.argtypes = [C.c_char]
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def ref_base(c):
    return lib().orc_ref_base(c.encode()).decode()


def decode_column(bases, bq, mq, ref_base_char, chunk_ref, is_candidate, max_indel_length=60):
    """decode_pileup_bases restatement: returns (tensor[34] list, depth, alt_info)."""
    out = np.zeros(34, dtype=np.int32)
    buf = C.create_string_buffer(1 << 16)
    b, q, m, cr = bases.encode(), bq.encode(), mq.encode(), chunk_ref.encode()
    depth = lib().orc_decode_column(b, len(b), q, len(q), m, len(m), C.c_char(ref_base_char.encode()), cr, len(cr),
                                    int(max_indel_length), int(bool(is_candidate)), _p(out), buf, len(buf))
    return out.tolist(), depth, buf.value.decode()


def create_tensor(text, ref, ref_start, sites, max_indel_length=60, alt_stride=4096):
    """create_tensor restatement on (already BQ-filtered) mpileup text.
    Returns tensor int32 [n,33,34], depth int32 [n], alt_info list[str], flags uint8 [n]."""
    sites = np.ascontiguousarray(sites, dtype=np.int32)
    n = len(sites)
    tensor = np.zeros((n, 33, 34), dtype=np.int32)
    depth = np.zeros(n, dtype=np.int32)
    flags = np.zeros(n, dtype=np.uint8)
    alts = np.zeros((n, alt_stride), dtype=np.uint8)
    tb = text.encode() if isinstance(text, str) else text
    rb = ref.encode() if isinstance(ref, str) else ref
    lib().orc_create_tensor(tb, C.c_size_t(len(tb)), rb, C.c_int64(ref_start), C.c_int64(len(rb)), _p(sites), n,
                            int(max_indel_length), _p(tensor), _p(depth), _p(alts), alt_stride, _p(flags))
    alt_list = [bytes(alts[i]).split(b"\0", 1)[0].decode() for i in range(n)]
    return tensor, depth, alt_list, flags


def rescale(tensor, depth, min_rescale_cov=50):
    tensor = np.ascontiguousarray(tensor, dtype=np.int32)
    out = np.zeros(tensor.shape, dtype=np.float32)
    for i in range(tensor.shape[0]):
        lib().orc_rescale(_p(tensor[i]), int(depth[i]), int(min_rescale_cov), _p(out[i]))
    return out


def strand_counts(tensor):
    tensor = np.ascontiguousarray(tensor, dtype=np.int32)
    f = np.zeros((tensor.shape[0], 4), dtype=np.int32)
    r = np.zeros((tensor.shape[0], 4), dtype=np.int32)
    for i in range(tensor.shape[0]):
        lib().orc_strand_counts(_p(tensor[i]), _p(f[i]), _p(r[i]))
    return f, r


def _table(weights):
    keep = []
    arr = (_Tensor * len(weights))()
    for i, (k, v) in enumerate(weights.items()):
        a = np.ascontiguousarray(v, dtype=np.float32)
        keep.append(a)
        arr[i].name = k.encode()
        arr[i].data = a.ctypes.data
        arr[i].numel = a.size
    return arr, keep


def cvt_forward(weights, cfg, x):
    """weights: dict name -> array (state_dict), cfg = dict(emb_dim, heads, depth, n_out); x [B,33,34] float32.
    Returns logits [K,B,2] float32."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    B, K = x.shape[0], int(cfg["n_out"])
    out = np.zeros((K, B, 2), dtype=np.float32)
    arr, keep = _table(weights)
    i3 = lambda v: (C.c_int * 3)(*[int(t) for t in v])
    lib().orc_cvt_forward(arr, len(weights), i3(cfg["emb_dim"]), i3(cfg["heads"]), i3(cfg["depth"]), K, _p(x),
                          C.c_int64(B), _p(out))
    return out


def bigru_forward(weights, n_out, x):
    x = np.ascontiguousarray(x, dtype=np.float32)
    B, K = x.shape[0], int(n_out)
    out = np.zeros((K, B, 2), dtype=np.float32)
    arr, keep = _table(weights)
    lib().orc_bigru_forward(arr, len(weights), K, _p(x), C.c_int64(B), _p(out))
    return out


def posterior(aff, neg, lik, edges):
    """aff/neg logits [K,B,2] float32; lik [K,10,10] float64; edges [2K,11] float64.
    Returns probs [B,2K,2] f32, post [B,K] f64, decision [B,4] i32, qual [B] f64."""
    aff = np.ascontiguousarray(aff, dtype=np.float32)
    neg = np.ascontiguousarray(neg, dtype=np.float32)
    lik = np.ascontiguousarray(lik, dtype=np.float64)
    edges = np.ascontiguousarray(edges, dtype=np.float64)
    K, B = aff.shape[0], aff.shape[1]
    probs = np.zeros((B, 2 * K, 2), dtype=np.float32)
    post = np.zeros((B, K), dtype=np.float64)
    dec = np.zeros((B, 4), dtype=np.int32)
    qual = np.zeros(B, dtype=np.float64)
    lib().orc_posterior(_p(aff), _p(neg), K, C.c_int64(B), _p(lik), _p(edges), _p(probs), _p(post), _p(dec), _p(qual))
    return probs, post, dec, qual


def posterior_from_probs(p1, lik, edges):
    """p1 [B,2K] float64 (8-decimal probabilities of the text seam). Returns post, decision, qual."""
    p1 = np.ascontiguousarray(p1, dtype=np.float64)
    lik = np.ascontiguousarray(lik, dtype=np.float64)
    edges = np.ascontiguousarray(edges, dtype=np.float64)
    B, K = p1.shape[0], p1.shape[1] // 2
    post = np.zeros((B, K), dtype=np.float64)
    dec = np.zeros((B, 4), dtype=np.int32)
    qual = np.zeros(B, dtype=np.float64)
    lib().orc_posterior_from_probs(_p(p1), K, C.c_int64(B), _p(lik), _p(edges), _p(post), _p(dec), _p(qual))
    return post, dec, qual


def synth_mpileup_text(chunk, min_bq=0, col_range=None, min_mq=0, with_mq=True):
    """Fast (C) equivalent of clairs_to_amd.synth.mpileup_text for a SynthChunk; returns bytes."""
    c0, c1 = col_range if col_range else (0, chunk.col_pos.size)
    n_ent = int(chunk.col_off[c1] - chunk.col_off[c0])
    cap = n_ent * 90 + (c1 - c0) * 64 + 1024
    buf = np.empty(cap, dtype=np.uint8)
    a = lambda v, dt: np.ascontiguousarray(v, dtype=dt)
    arrs = [a(chunk.col_pos, np.int32), a(chunk.col_off, np.int64), a(chunk.entries, np.uint32), a(chunk._okind, np.uint8),
            a(chunk._ilen, np.int32), a(chunk._ivar, np.int32)]
    f = lib().orc_pack_to_mpileup
    f.restype = C.c_int64
    n = f(*[_p(x) for x in arrs], C.c_int64(c0), C.c_int64(c1), int(min_bq), int(min_mq), int(bool(with_mq)), _p(buf),
          C.c_int64(cap))
    assert n >= 0
    return buf[:n].tobytes()


def extract_candidates(text, ref, ref_start, snv_min_af=0.05, indel_min_af=0.05, min_coverage=4, alt_base_num=3,
                       select_indel=True):
    """extract_candidates_calling restatement on `samtools mpileup --min-MQ 20 --min-BQ q` text (6 columns).
    Returns pos int32 [n_rows], flags uint8 [n_rows] (bit0 SNV, bit1 indel, bit2 pass_af), depth int32 [n_rows]."""
    tb = text.encode() if isinstance(text, str) else text
    rb = ref.encode() if isinstance(ref, str) else ref
    cap = tb.count(b"\n") + 1
    pos = np.zeros(cap, dtype=np.int32)
    flags = np.zeros(cap, dtype=np.uint8)
    depth = np.zeros(cap, dtype=np.int32)
    f = lib().orc_extract_candidates
    f.restype = C.c_int64
    n = f(tb, C.c_size_t(len(tb)), rb, C.c_int64(ref_start), C.c_int64(len(rb)), C.c_double(min_coverage),
          C.c_double(snv_min_af), C.c_double(indel_min_af), int(alt_base_num), int(bool(select_indel)), _p(pos), _p(flags),
          _p(depth), C.c_int64(cap))
    return pos[:n], flags[:n], depth[:n]


def hybrid_info_rows(text, ref, ref_start, positions, ctg, snv_min_af=0.05, indel_min_af=0.05, min_coverage=4, alt_base_num=3,
                     select_indel=False):
    """The rows of `<ctg>.<chunk>_hybrid_info` (extract_candidates_calling.py:352-354, 490-497) for the listed positions that have a row in
    the extraction pileup `text` (6 columns) and an A/C/G/T reference base."""
    pos, flags, _ = extract_candidates(text, ref, ref_start, snv_min_af, indel_min_af, min_coverage, alt_base_num, select_indel)
    at = {int(p): i for i, p in enumerate(pos)}
    wanted = set(int(p) for p in positions)
    f = lib().orc_hybrid_alt_info
    out = []
    buf = C.create_string_buffer(1 << 20)
    tb = text.encode() if isinstance(text, str) else text
    for row in tb.split(b"\n"):
        c = row.split(b"\t")
        if len(c) < 5 or int(c[1]) not in wanted:
            continue
        p = int(c[1])
        if not flags[at[p]] & 32:
            continue
        n = f(c[4], len(c[4]), int(bool(select_indel)), int(bool(flags[at[p]] & 4)), buf, len(buf))
        assert n >= 0
        out.append("%s\t%d\t%s\t%s\n" % (ctg, p, ref[p - ref_start].upper(), buf.raw[:n].decode()))
    return "".join(out)

"""ctypes binding of libclairsto_amd.so (the C ABI declared in include/clairsto_amd.h).

The library is the product: there is no Python/CPU fallback.  Importing this module loads it and fails
loudly when it has not been built (`python -c "import __graft_entry__ as g; g.build()"` or
`make -C clairs_to_amd/csrc`).  torch is imported first so that the process ends up with ONE HIP runtime:
torch's bundled libamdhip64.so and /opt/rocm's share the SONAME libamdhip64.so.7, so the copy torch loaded
is the one our library binds to and torch streams / device pointers are valid inside it.
"""
import ctypes as C
import os

import torch  # noqa: F401  (must precede the dlopen, see above)

_HERE = os.path.dirname(os.path.abspath(__file__))
# CTO_LIB_PATH: A/B testing of kernel variants built next to the product library (tools/ only; never set by the product)
LIB_PATH = os.environ.get("CTO_LIB_PATH") or os.path.join(_HERE, "libclairsto_amd.so")

if not os.path.exists(LIB_PATH):
    raise ImportError(
        "clairs_to_amd: %s is missing - build the HIP extension first (__graft_entry__.build() or "
        "`make -C clairs_to_amd/csrc`). There is no CPU fallback." % LIB_PATH)

lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)

c_i64 = C.c_int64
c_i32 = C.c_int32
c_vp = C.c_void_p


class PackView(C.Structure):
    _fields_ = [("n_cols", c_i64), ("n_entries", c_i64), ("n_keys", c_i64),
                ("col_pos", c_vp), ("col_ref", c_vp), ("col_off", c_vp), ("key_off", c_vp),
                ("entries", c_vp), ("key_meta", c_vp), ("key_group", c_vp)]


class CvtCfg(C.Structure):
    _fields_ = [("emb_dim", C.c_int * 3), ("heads", C.c_int * 3), ("depth", C.c_int * 3), ("n_out", C.c_int)]


class ChunkJob(C.Structure):
    _fields_ = [("ctg_name", C.c_char_p), ("bed_path", C.c_char_p), ("mpileup_path", C.c_char_p), ("bam_path", C.c_char_p),
                ("vcf_path", C.c_char_p), ("region_start", c_i64), ("region_end", c_i64), ("candidates_path", C.c_char_p),
                ("confident_intervals", C.c_void_p), ("n_confident_intervals", c_i64), ("restrict_to_confident", C.c_int),
                ("known_pos", C.c_void_p), ("n_known_pos", c_i64), ("hybrid_info_path", C.c_char_p)]


class RunCfg(C.Structure):
    _fields_ = [("aff", c_vp), ("neg", c_vp), ("d_lik", c_vp), ("d_edges", c_vp), ("K", C.c_int), ("min_bq", C.c_int),
                ("min_rescale_cov", C.c_int), ("max_indel_length", C.c_int), ("max_depth", C.c_int), ("neg_reads_aff", C.c_int),
                ("show_ref", C.c_int), ("verbose", C.c_int), ("qual_pass", C.c_double), ("ref_fa", C.c_char_p),
                ("vcf_header", C.c_char_p), ("producers", C.c_int), ("writers", C.c_int), ("depth", C.c_int),
                ("inflate_cus", C.c_int), ("inflate_jobs", C.c_int), ("pack_threads", C.c_int), ("samtools", C.c_char_p),
                ("samtools_max_depth", C.c_int), ("aff2", c_vp), ("neg2", c_vp), ("device_pileup", C.c_int),
                ("extract_min_mq", C.c_int), ("extract_min_bq", C.c_int), ("alt_base_num", C.c_int), ("snv_min_af", C.c_double),
                ("indel_min_af", C.c_double), ("min_coverage", C.c_double), ("indel_regions_bed", C.c_char_p), ("device_tokenise", C.c_int),
                ("indel_bed_superseded", C.c_int)]


class RealignJob(C.Structure):
    _fields_ = [("n_reads", c_i32), ("seqs", c_vp), ("positions", c_vp), ("cigars", c_vp), ("reference", C.c_char_p),
                ("haplotypes", C.c_char_p), ("ref_start", c_i32), ("ref_prefix", c_i32), ("ref_suffix", c_i32),
                ("out_positions", c_vp), ("cigar_buf", c_vp), ("cigar_cap", C.c_size_t), ("cigar_off", c_vp), ("status", c_i32),
                ("seqs_joined", c_vp), ("cigars_joined", c_vp)]


class RealignStats(C.Structure):
    _fields_ = [("windows", c_i64), ("host_windows", c_i64), ("reads", c_i64), ("haplotypes", c_i64), ("fast_pairs", c_i64),
                ("sw_pairs", c_i64), ("sw_cells", c_i64), ("fast_pass_ms", C.c_double), ("sw_ms", C.c_double),
                ("device_stage_ms", C.c_double), ("host_ms", C.c_double),
                ("tracebacks", c_i64), ("tracebacks_declined", c_i64), ("traceback_ms", C.c_double)]


class RunStats(C.Structure):
    _fields_ = [("candidates", c_i64), ("sites", c_i64), ("rows", c_i64), ("low_coverage", c_i64), ("clamped", c_i64), ("seconds", C.c_double),
                ("produce_s", C.c_double), ("finish_s", C.c_double), ("launch_s", C.c_double), ("launcher_wait_s", C.c_double),
                ("pack_s", C.c_double), ("upload_s", C.c_double), ("device_s", C.c_double), ("device_piled", c_i64), ("device_inflated", c_i64), ("device_tokenised", c_i64)]


# every symbol include/clairsto_amd.h declares: (restype, argtypes)
SYMBOLS = {
    "cto_last_error": (C.c_char_p, []),
    "cto_version": (C.c_int, []),
    "cto_device_count": (C.c_int, []),
    "cto_debug_poison_lds": (C.c_int, [c_vp]),
    "cto_pack_from_mpileup": (C.c_int, [c_vp, C.c_size_t, C.c_char_p, c_i64, C.c_size_t, C.c_int, C.POINTER(c_vp)]),
    "cto_set_pack_threads": (None, [C.c_int]),
    "cto_pack_from_bam": (C.c_int, [C.c_char_p, C.c_char_p, C.c_char_p, c_i64, c_i64, c_vp, c_i64, C.c_char_p, c_i64, C.c_size_t,
                                    C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(c_vp)]),
    "cto_pack_from_bam_inflated": (C.c_int, [C.c_char_p, C.c_char_p, C.c_char_p, c_i64, c_i64, c_vp, c_i64, C.c_char_p, c_i64, C.c_size_t,
                                             C.c_int, C.c_int, C.c_int, C.c_int, c_vp, C.c_size_t, c_vp, c_i64, C.POINTER(c_vp)]),
    "cto_bam_chunk_span": (C.c_int, [C.c_char_p, C.c_char_p, C.c_char_p, c_i64, c_i64, C.POINTER(c_i64), C.POINTER(c_i64)]),
    "cto_bgzf_scan": (c_i64, [c_vp, C.c_size_t, c_i64, c_vp, c_i64, C.POINTER(c_i64)]),
    "cto_bgzf_inflate": (C.c_int, [c_vp, c_vp, C.c_int, c_vp, c_vp, c_vp]),
    "cto_bam_view": (c_i64, [C.c_char_p, C.c_char_p, C.c_char_p, c_i64, c_i64, C.c_int, c_vp, C.c_size_t, C.POINTER(C.c_size_t)]),
    "cto_dev_pileup_create": (C.c_int, [C.POINTER(c_vp)]),
    "cto_dev_pileup_destroy": (None, [c_vp]),
    "cto_bam_record_starts": (c_i64, [C.c_char_p, C.c_char_p, C.c_char_p, c_i64, c_i64, c_i64, c_i64, c_vp, c_i64, C.POINTER(c_i32)]),
    "cto_pileup_device": (C.c_int, [c_vp, c_vp, c_vp, c_i64, c_vp, c_i64, c_i32, c_i64, c_i64, c_vp, c_i64, C.c_char_p, c_i64, C.c_size_t,
                                    C.c_int, C.c_int, C.c_int, C.c_int, c_vp, C.POINTER(PackView), C.POINTER(c_vp), C.POINTER(C.c_int)]),
    "cto_device_read": (C.c_int, [c_vp, c_vp, C.c_size_t]),
    "cto_pack_from_arrays": (C.c_int, [C.POINTER(PackView), c_vp, c_vp, C.POINTER(c_vp)]),
    "cto_pack_view_of": (C.c_int, [c_vp, C.POINTER(PackView)]),
    "cto_pack_key_string": (C.c_int, [c_vp, c_i64, C.POINTER(C.c_char_p)]),
    "cto_pack_free": (None, [c_vp]),
    "cto_featurize_columns": (C.c_int, [C.POINTER(PackView), C.c_int, c_vp, c_vp, c_vp, c_vp]),
    "cto_gather_windows": (C.c_int, [C.POINTER(PackView), c_vp, c_vp, c_vp, c_i64, C.c_int, C.c_int, c_vp, c_vp, c_vp, c_vp, c_vp,
                                     c_vp, c_vp, c_vp]),
    "cto_featurize_sites": (C.c_int, [C.POINTER(PackView), c_vp, c_i64, C.c_int, C.c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "cto_extract_candidates": (C.c_int, [C.POINTER(PackView), C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, C.c_int,
                                         C.c_int, c_vp, c_vp, c_vp]),
    "cto_extract_restrict": (C.c_int, [C.POINTER(PackView), c_vp, c_vp, c_vp, C.c_int, C.c_int, c_vp]),
    "cto_extract_mark": (C.c_int, [C.POINTER(PackView), c_vp, c_vp, C.c_int, C.c_int, c_vp]),
    "cto_hybrid_info": (C.c_int, [C.POINTER(PackView), c_vp, c_vp, C.c_int, C.c_int, C.c_int, C.c_int, c_vp, c_vp, c_vp, c_vp]),
    "cto_hybrid_info_rows": (c_i64, [c_vp, C.c_char_p, c_i64, c_vp, c_vp, C.c_int, c_vp, c_vp, c_vp, C.c_size_t]),
    "cto_alt_info": (C.c_int, [c_vp, c_i64, C.c_int, c_vp, c_i32, c_vp, c_vp, c_vp, C.c_char_p, C.c_size_t]),
    "cto_alt_info_batch": (c_i64, [c_vp, c_i64, c_vp, C.c_int, c_vp, c_vp, c_vp, c_vp, c_vp, C.c_size_t, c_vp]),
    "cto_alt_info_batch_sites": (c_i64, [c_vp, c_i64, c_vp, C.c_int, c_vp, c_vp, c_vp, c_vp, c_vp, C.c_size_t, c_vp]),
    "cto_weights_new": (c_vp, []),
    "cto_weights_add": (C.c_int, [c_vp, C.c_char_p, c_vp, c_i64]),
    "cto_weights_free": (None, [c_vp]),
    "cto_cvt_create": (C.c_int, [c_vp, C.POINTER(CvtCfg), C.POINTER(c_vp)]),
    "cto_bigru_create": (C.c_int, [c_vp, C.c_int, C.POINTER(c_vp)]),
    "cto_cvt_create_ex": (C.c_int, [c_vp, C.POINTER(CvtCfg), C.c_int, C.POINTER(c_vp)]),
    "cto_bigru_create_ex": (C.c_int, [c_vp, C.c_int, C.c_int, C.POINTER(c_vp)]),
    "cto_cvt_create_packed": (C.c_int, [c_vp, c_i64, C.POINTER(CvtCfg), C.POINTER(c_vp)]),
    "cto_bigru_create_packed": (C.c_int, [c_vp, c_i64, C.c_int, C.POINTER(c_vp)]),
    "cto_model_manifest": (c_i64, [C.c_int, C.POINTER(CvtCfg), C.c_int, c_vp, C.c_size_t]),
    "cto_bed_centres": (c_i64, [c_vp, C.c_size_t, C.c_char_p, c_vp, c_i64, c_vp, c_vp]),
    "cto_run_chunks": (C.c_int, [C.POINTER(RunCfg), C.POINTER(ChunkJob), c_i64, c_vp, C.POINTER(RunStats)]),
    "cto_run_release": (C.c_int, []),
    "cto_haplotype_filter": (C.c_int, [c_vp, C.c_size_t, C.c_char_p, c_i64, C.c_size_t, c_i64, c_vp, c_vp, c_vp, c_vp, C.c_int, C.c_int,
                                       C.c_int, c_vp, c_vp]),
    "cto_realign_reads": (C.c_int, [C.c_int, c_vp, c_vp, c_vp, C.c_char_p, C.c_char_p, c_i32, c_i32, c_i32, c_vp, c_vp, C.c_size_t, c_vp]),
    "cto_dev_tokeniser_create": (C.c_int, [C.POINTER(c_vp)]),
    "cto_dev_tokeniser_destroy": (None, [c_vp]),
    "cto_dev_tokeniser_buffer": (c_vp, [c_vp, C.c_size_t]),
    "cto_tokenise_device": (C.c_int, [c_vp, c_vp, C.c_size_t, C.c_char_p, c_i64, C.c_size_t, C.c_int, c_vp, C.POINTER(PackView), C.POINTER(c_vp),
                                      C.POINTER(C.c_int)]),
    "cto_softmax_pairs": (C.c_int, [c_vp, c_i64, c_vp, c_vp]),
    "cto_qual_pending": (C.c_int, [c_vp, c_i64, c_vp, c_vp]),
    "cto_realign_windows": (C.c_int, [C.c_int, C.POINTER(RealignJob), C.c_int, C.c_int, c_vp, C.POINTER(RealignStats)]),
    "cto_sw_ends_batch": (C.c_int, [C.c_int, c_vp, C.c_size_t, c_vp, C.c_int, C.c_int, c_vp, c_vp]),
    "cto_ssw_align_batch": (C.c_int, [C.c_int, c_vp, C.c_size_t, c_vp, C.c_int, C.c_int, c_vp, c_vp, c_vp, c_vp, C.c_size_t, c_vp]),
    "cto_ssw_align": (C.c_int, [C.c_char_p, C.c_char_p, C.POINTER(c_i32), C.POINTER(c_i32), c_vp, C.c_size_t]),
    "cto_ssw_pass": (C.c_int, [c_vp, C.c_int, C.c_int, c_vp, C.c_int, C.c_int, C.c_int, c_vp]),
    "cto_set_realign_threads": (C.c_int, [C.c_int]),
    "cto_realign_read_evidence": (c_i64, [C.c_char_p, c_i64, C.c_char_p, c_i64, C.c_char_p, c_i64, C.c_char_p, c_i64, c_i64, c_i64, c_i64, C.c_int,
                                          c_vp, c_i64]),
    "cto_dbg_consensus": (C.c_int, [C.c_char_p, C.c_int, c_vp, c_vp, c_vp, c_vp, C.c_size_t, C.POINTER(C.c_size_t)]),
    "cto_vcf_rows_batch": (c_i64, [C.c_char_p, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, C.c_int, C.c_int, C.c_double,
                                   c_vp, C.c_size_t, c_vp]),
    "cto_model_forward": (C.c_int, [c_vp, c_vp, c_i64, c_vp, c_vp]),
    "cto_model_forward_raw": (C.c_int, [c_vp, c_vp, c_vp, C.c_int, C.c_int, c_i64, c_vp, c_vp]),
    "cto_model_macs_per_site": (c_i64, [c_vp]),
    "cto_model_n_out": (C.c_int, [c_vp]),
    "cto_model_destroy": (None, [c_vp]),
    "cto_model_profile": (C.c_int, [c_vp, C.c_int]),
    "cto_model_profile_read": (C.c_int, [c_vp, C.POINTER(C.c_double), C.POINTER(c_i64)]),
    "cto_model_profile_read_stage": (C.c_int, [c_vp, C.c_int, C.POINTER(C.c_double), C.POINTER(c_i64)]),
    "cto_posterior": (C.c_int, [c_vp, c_vp, C.c_int, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "cto_qual_finalize": (c_i64, [c_vp, c_vp, c_i64]),
    "cto_candidate_positions": (C.c_int, [c_vp, c_vp, C.c_int, c_i32, c_i32, c_vp, c_i64, c_vp, c_vp, c_vp]),
    "cto_softmax_probs": (C.c_int, [c_vp, c_vp, C.c_int, c_i64, c_vp, c_vp]),
    "cto_posterior_from_probs": (C.c_int, [c_vp, C.c_int, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
}

for _name, (_res, _args) in SYMBOLS.items():
    _fn = getattr(lib, _name)   # AttributeError here = the built library is stale; rebuild it
    _fn.restype = _res
    _fn.argtypes = _args


# torch.ops.clairsto.* (csrc/torch_ops.cpp): the same entry points registered with PyTorch's dispatcher.  Loaded after the
# C-ABI library, which it links against; missing = not built, and the custom ops (hence the nn.Module shims) fail loudly.
TORCH_LIB_PATH = os.path.join(_HERE, "libclairsto_torch.so")
if not os.path.exists(TORCH_LIB_PATH):
    raise ImportError("clairs_to_amd: %s is missing - build it with `make -C clairs_to_amd/csrc` (torch custom ops)" % TORCH_LIB_PATH)
torch.ops.load_library(TORCH_LIB_PATH)


def model_manifest(kind, cfg=None, n_out=4):
    """[(state_dict name, numel)] in packed-weights order (cto_model_manifest); kind 0 = CvT (cfg: CvtCfg), 1 = BiGRU."""
    need = lib.cto_model_manifest(kind, C.byref(cfg) if cfg is not None else None, n_out, None, 0)
    check(int(need))
    buf = C.create_string_buffer(int(need))
    check(int(lib.cto_model_manifest(kind, C.byref(cfg) if cfg is not None else None, n_out, C.addressof(buf), int(need))))
    return [(ln.split("\t")[0], int(ln.split("\t")[1])) for ln in buf.value.decode().split("\n") if ln]


class CtoError(RuntimeError):
    pass


def check(rc):
    """Raise CtoError for a negative return code of the C ABI."""
    if rc is not None and rc < 0:
        raise CtoError("clairsto_amd error %d: %s" % (rc, lib.cto_last_error().decode()))
    return rc


def current_stream_ptr():
    """hipStream_t of torch's current stream on the current device, as an integer for the void* argument."""
    return int(torch.cuda.current_stream().cuda_stream)

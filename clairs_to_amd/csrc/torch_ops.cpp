// torch.ops.clairsto.* - the hot path as PyTorch custom operators (BASELINE.json north_star: "invoked from Python via
// PyTorch-ROCm custom ops so clairs/predict.py and clairs/call_variants.py see the same tensor shapes and posterior
// outputs").  Each operator is a thin binding: it checks shapes / dtypes / devices, allocates the outputs
// with torch's allocator and calls the C ABI of include/clairsto_amd.h on torch's CURRENT stream of the input's device.
// No arithmetic happens here and no torch operator computes anything on the path.
//
//   clairsto::cvt_forward(x, packed_weights, cfg)   model_aff(x)  of clairs/predict.py:646-651   -> logits [K][B][2]
//   clairsto::bigru_forward(x, packed_weights, K)   model_neg(x)  of clairs/predict.py:653-658   -> logits [K][B][2]
//   clairsto::posterior(aff, neg, lik, edges)       softmax (predict.py:659-684) + call_variants.py:154-304, 79-88
//   clairsto::pileup_featurize(pack..., site_pos, min_bq, min_rescale_cov)
//                                                   create_tensor_pileup_calling.py:95-233, 536-570 + predict.py:172-207, 626-642
//
// Host-only translation unit (g++): it needs torch's headers, not hipcc.  Meta kernels give the output shapes so the
// operators work under FakeTensor / torch.compile tracing; there is no CPU kernel (the product has no CPU fallback).
#include <ATen/ATen.h>
#include <c10/hip/HIPGuard.h>
#include <c10/hip/HIPStream.h>
#include <torch/library.h>
#include <hip/hip_runtime_api.h>

#include <mutex>
#include <tuple>
#include <vector>

#include "../../include/clairsto_amd.h"

namespace {

using at::Tensor;

void check_rc(int64_t rc, const char* what) {
    TORCH_CHECK(rc >= 0, "clairsto::", what, ": error ", rc, ": ", cto_last_error());
}

void* stream_of(const Tensor& t) { return c10::hip::getCurrentHIPStream(t.get_device()).stream(); }

void check_x(const Tensor& x, const char* op) {
    TORCH_CHECK(x.is_cuda(), "clairsto::", op, ": x must live on the HIP device (there is no CPU fallback)");
    TORCH_CHECK(x.scalar_type() == at::kFloat && x.dim() == 3 && x.size(1) == CTO_NPOS && x.size(2) == CTO_NCHAN,
                "clairsto::", op, ": x must be float32 [B,", CTO_NPOS, ",", CTO_NCHAN, "], got ", x.sizes());
}

// ---- model handles, cached per packed-weights tensor -------------------------------------------------------------------
// A handle (device-resident repacked weights + workspace) is built the first time a packed_weights tensor is seen and
// reused while that tensor is alive and unmodified: the key is its storage (held weakly), data pointer, numel, version
// counter, device and configuration.  An in-place update of the weights bumps the version and rebuilds the handle.
// A handle's activation workspace belongs to one stream at a time (include/clairsto_amd.h): the operator may be called with the same
// weights from several streams or threads, so every forward records an event behind its kernels and a forward on ANOTHER stream
// waits for it first; the look-up and the launch happen under one lock (a launch is microseconds of host time).
struct Entry {
    c10::weak_intrusive_ptr<c10::StorageImpl> storage;
    const void* ptr;
    int64_t numel;
    uint32_t version;
    int device, kind;
    std::vector<int64_t> cfg;
    cto_model* model;
    hipEvent_t done;            // behind the last forward's kernels
    void* last_stream;
    bool used;                  // last_stream is meaningful (the default stream is a null pointer)
};
std::mutex g_mu;
std::vector<Entry> g_cache;

uint32_t version_of(const Tensor& t) {
    return t.is_inference() ? 0u : uint32_t(t.unsafeGetTensorImpl()->version_counter().current_version());
}

void drop(Entry& e) {
    cto_model_destroy(e.model);
    if (e.done) (void)hipEventDestroy(e.done);
}

// g_mu held by the caller
Entry& model_for(const Tensor& packed, int kind, const std::vector<int64_t>& cfg, const char* op) {
    TORCH_CHECK(packed.is_cuda() && packed.scalar_type() == at::kFloat && packed.dim() == 1 && packed.is_contiguous(),
                "clairsto::", op, ": packed_weights must be a contiguous 1-D float32 tensor on the HIP device");
    const auto* st = packed.storage().unsafeGetStorageImpl();
    const uint32_t ver = version_of(packed);
    for (size_t i = 0; i < g_cache.size();) {
        Entry& e = g_cache[i];
        auto alive = e.storage.lock();
        if (!alive) {                                       // the weights tensor is gone: so is its handle
            drop(e);
            g_cache.erase(g_cache.begin() + i);
            continue;
        }
        if (alive.get() == st && e.ptr == packed.data_ptr() && e.numel == packed.numel() && e.device == packed.get_device() &&
            e.kind == kind && e.cfg == cfg) {
            if (e.version == ver) return e;
            drop(e);                                        // same tensor, modified in place
            g_cache.erase(g_cache.begin() + i);
            continue;
        }
        ++i;
    }
    const Tensor host = packed.to(at::kCPU);                // once per weights version
    cto_model* m = nullptr;
    if (kind == 0) {
        TORCH_CHECK(cfg.size() == 10, "clairsto::cvt_forward: cfg = [emb_dim x3, heads x3, depth x3, n_out]");
        cto_cvt_cfg c;
        for (int i = 0; i < 3; ++i) { c.emb_dim[i] = int(cfg[i]); c.heads[i] = int(cfg[3 + i]); c.depth[i] = int(cfg[6 + i]); }
        c.n_out = int(cfg[9]);
        check_rc(cto_cvt_create_packed(host.data_ptr<float>(), host.numel(), &c, &m), op);
    } else {
        check_rc(cto_bigru_create_packed(host.data_ptr<float>(), host.numel(), int(cfg[0]), &m), op);
    }
    hipEvent_t ev = nullptr;
    TORCH_CHECK(hipEventCreateWithFlags(&ev, hipEventDisableTiming) == hipSuccess, "clairsto::", op, ": hipEventCreate failed");
    g_cache.push_back(Entry{c10::weak_intrusive_ptr<c10::StorageImpl>(packed.storage().getWeakStorageImpl()), packed.data_ptr(),
                            packed.numel(), ver, int(packed.get_device()), kind, cfg, m, ev, nullptr, false});
    return g_cache.back();
}

Tensor run_model(const Tensor& x, const Tensor& packed, int kind, const std::vector<int64_t>& cfg, int64_t K, const char* op) {
    check_x(x, op);
    TORCH_CHECK(packed.get_device() == x.get_device(), "clairsto::", op, ": x and packed_weights live on different devices");
    const c10::hip::HIPGuard guard(x.get_device());
    const Tensor xc = x.contiguous();
    Tensor out = at::empty({K, xc.size(0), 2}, xc.options());
    if (xc.size(0) == 0) return out;
    std::lock_guard<std::mutex> lock(g_mu);
    Entry& e = model_for(packed, kind, cfg, op);
    void* s = stream_of(xc);
    if (e.used && e.last_stream != s)       // the workspace's previous user ran on another stream: order behind it
        TORCH_CHECK(hipStreamWaitEvent(static_cast<hipStream_t>(s), e.done, 0) == hipSuccess, "clairsto::", op, ": hipStreamWaitEvent failed");
    check_rc(cto_model_forward(e.model, xc.data_ptr<float>(), xc.size(0), out.data_ptr<float>(), s), op);
    TORCH_CHECK(hipEventRecord(e.done, static_cast<hipStream_t>(s)) == hipSuccess, "clairsto::", op, ": hipEventRecord failed");
    e.last_stream = s;
    e.used = true;
    return out;
}

Tensor cvt_forward(const Tensor& x, const Tensor& packed, at::IntArrayRef cfg) {
    TORCH_CHECK(cfg.size() == 10 && (cfg[9] == 4 || cfg[9] == 6), "clairsto::cvt_forward: cfg = [emb_dim x3, heads x3, depth x3, n_out in {4, 6}]");
    return run_model(x, packed, 0, cfg.vec(), cfg[9], "cvt_forward");
}
Tensor bigru_forward(const Tensor& x, const Tensor& packed, int64_t n_out) {
    TORCH_CHECK(n_out == 4 || n_out == 6, "clairsto::bigru_forward: n_out must be 4 or 6");
    return run_model(x, packed, 1, {n_out}, n_out, "bigru_forward");
}
Tensor cvt_forward_meta(const Tensor& x, const Tensor&, at::IntArrayRef cfg) {
    TORCH_CHECK(cfg.size() == 10, "clairsto::cvt_forward: cfg = [emb_dim x3, heads x3, depth x3, n_out]");
    return at::empty({cfg[9], x.size(0), 2}, x.options());
}
Tensor bigru_forward_meta(const Tensor& x, const Tensor&, int64_t n_out) { return at::empty({n_out, x.size(0), 2}, x.options()); }

// ---- posterior ---------------------------------------------------------------------------------------------------------
std::tuple<Tensor, Tensor, Tensor, Tensor> posterior(const Tensor& aff, const Tensor& neg, const Tensor& lik, const Tensor& edges) {
    TORCH_CHECK(aff.is_cuda() && neg.is_cuda() && lik.is_cuda() && edges.is_cuda(), "clairsto::posterior: all inputs must live on the HIP device");
    TORCH_CHECK(aff.scalar_type() == at::kFloat && aff.dim() == 3 && aff.size(2) == 2 && neg.sizes() == aff.sizes() &&
                    neg.scalar_type() == at::kFloat,
                "clairsto::posterior: aff_logits / neg_logits must be float32 [K,B,2]");
    const int64_t K = aff.size(0), B = aff.size(1);
    TORCH_CHECK(K == 4 || K == 6, "clairsto::posterior: K must be 4 or 6");
    TORCH_CHECK(lik.scalar_type() == at::kDouble && lik.numel() == K * 100 && edges.scalar_type() == at::kDouble && edges.numel() == 2 * K * 11,
                "clairsto::posterior: lik must be float64 [K,10,10] and edges float64 [2K,11]");
    const c10::hip::HIPGuard guard(aff.get_device());
    const Tensor a = aff.contiguous(), n = neg.contiguous(), l = lik.contiguous(), e = edges.contiguous();
    Tensor probs = at::empty({B, 2 * K, 2}, a.options());
    Tensor post = at::empty({B, K}, a.options().dtype(at::kDouble));
    Tensor dec = at::empty({B, 4}, a.options().dtype(at::kInt));
    Tensor qual = at::empty({B}, a.options().dtype(at::kDouble));
    if (B > 0)
        check_rc(cto_posterior(a.data_ptr<float>(), n.data_ptr<float>(), int(K), B, l.data_ptr<double>(), e.data_ptr<double>(),
                               probs.data_ptr<float>(), post.data_ptr<double>(), dec.data_ptr<int32_t>(), qual.data_ptr<double>(),
                               stream_of(a)),
                 "posterior");
    if (B > 0) {
        // host half of QUAL (cto_qual_finalize): the op hands back final values.  The flags are counted on the device
        // (cto_qual_pending) and only those four bytes cross PCIe per call; the 16 B/site download and the host pass happen on the
        // ~1 call in a hundred that holds a site on a rounding boundary
        Tensor pending = at::empty({1}, a.options().dtype(at::kInt));
        check_rc(cto_qual_pending(dec.data_ptr<int32_t>(), B, pending.data_ptr<int32_t>(), stream_of(a)), "qual_pending");
        int32_t n_pending = 0;
        C10_HIP_CHECK(hipMemcpyAsync(&n_pending, pending.data_ptr<int32_t>(), sizeof(int32_t), hipMemcpyDeviceToHost,
                                     static_cast<hipStream_t>(stream_of(a))));
        C10_HIP_CHECK(hipStreamSynchronize(static_cast<hipStream_t>(stream_of(a))));
        if (n_pending > 0) {
            Tensor dec_h = dec.cpu();
            Tensor qual_h = qual.cpu();
            check_rc(int(cto_qual_finalize(dec_h.data_ptr<int32_t>(), qual_h.data_ptr<double>(), B)) < 0 ? -1 : 0, "qual_finalize");
            dec.copy_(dec_h);
            qual.copy_(qual_h);
        }
    }
    return {probs, post, dec, qual};
}
std::tuple<Tensor, Tensor, Tensor, Tensor> posterior_meta(const Tensor& aff, const Tensor&, const Tensor&, const Tensor&) {
    const int64_t K = aff.size(0), B = aff.size(1);
    return {at::empty({B, 2 * K, 2}, aff.options()), at::empty({B, K}, aff.options().dtype(at::kDouble)),
            at::empty({B, 4}, aff.options().dtype(at::kInt)), at::empty({B}, aff.options().dtype(at::kDouble))};
}

// ---- the modules' own Softmax(dim=1) (apply_softmax=True) ---------------------------------------------------------------
Tensor softmax2(const Tensor& logits) {
    TORCH_CHECK(logits.is_cuda() && logits.scalar_type() == at::kFloat && logits.dim() >= 1 && logits.size(-1) == 2,
                "clairsto::softmax2: float32 [..., 2] on the HIP device");
    const c10::hip::HIPGuard guard(logits.get_device());
    const Tensor a = logits.contiguous();
    Tensor out = at::empty_like(a);
    check_rc(cto_softmax_pairs(a.data_ptr<float>(), a.numel() / 2, out.data_ptr<float>(), stream_of(a)), "softmax2");
    return out;
}
Tensor softmax2_meta(const Tensor& logits) { return at::empty_like(logits); }

// ---- pileup featurisation ----------------------------------------------------------------------------------------------
using Feat = std::tuple<Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor>;

Feat pileup_featurize(const Tensor& entries, const Tensor& col_off, const Tensor& col_pos, const Tensor& col_ref,
                      const Tensor& key_off, const Tensor& key_meta, const Tensor& key_group, const Tensor& site_pos, int64_t min_bq,
                      int64_t min_rescale_cov) {
    for (const Tensor* t : {&entries, &col_off, &col_pos, &col_ref, &key_off, &key_meta, &key_group, &site_pos})
        TORCH_CHECK(t->is_cuda() && t->is_contiguous() && t->get_device() == site_pos.get_device(),
                    "clairsto::pileup_featurize: every pack array and site_pos must be contiguous on one HIP device");
    TORCH_CHECK(entries.scalar_type() == at::kInt && col_off.scalar_type() == at::kLong && col_pos.scalar_type() == at::kInt &&
                    col_ref.scalar_type() == at::kByte && key_off.scalar_type() == at::kInt && key_meta.scalar_type() == at::kByte &&
                    key_group.scalar_type() == at::kInt && site_pos.scalar_type() == at::kInt,
                "clairsto::pileup_featurize: dtypes are entries i32 (uint32 bits), col_off i64, col_pos i32, col_ref u8, key_off i32, "
                "key_meta u8, key_group i32, site_pos i32");
    const int64_t nc = col_pos.numel(), nk = key_meta.numel(), n = site_pos.numel();
    TORCH_CHECK(col_off.numel() == nc + 1 && key_off.numel() == nc + 1 && col_ref.numel() == nc && key_group.numel() == nk,
                "clairsto::pileup_featurize: inconsistent pack array lengths");
    const c10::hip::HIPGuard guard(site_pos.get_device());
    cto_pack_view v;
    v.n_cols = nc; v.n_entries = entries.numel(); v.n_keys = nk;
    v.col_pos = col_pos.data_ptr<int32_t>(); v.col_ref = col_ref.data_ptr<uint8_t>(); v.col_off = col_off.data_ptr<int64_t>();
    v.key_off = key_off.data_ptr<int32_t>(); v.entries = reinterpret_cast<const uint32_t*>(entries.data_ptr<int32_t>());
    v.key_meta = key_meta.data_ptr<uint8_t>(); v.key_group = key_group.data_ptr<int32_t>();
    const auto o32 = site_pos.options();
    Tensor colvec = at::empty({std::max<int64_t>(nc, 1), 72}, o32.dtype(at::kShort));
    Tensor coldepth = at::empty({std::max<int64_t>(nc, 1), 2}, o32);
    Tensor keycnt = at::empty({std::max<int64_t>(nk, 1)}, o32);
    Tensor keyfirst = at::empty({std::max<int64_t>(nk, 1), 2}, o32);
    Tensor x_aff = at::empty({n, CTO_NPOS, CTO_NCHAN}, o32.dtype(at::kFloat));
    Tensor x_neg = at::empty({n, CTO_NPOS, CTO_NCHAN}, o32.dtype(at::kFloat));
    Tensor site_info = at::empty({n, 12}, o32);
    Tensor sitefirst = at::empty({std::max<int64_t>(n, 1), 8}, o32);
    void* s = stream_of(site_pos);
    check_rc(cto_featurize_columns(&v, int(min_bq), colvec.data_ptr<int16_t>(), coldepth.data_ptr<int32_t>(),
                                   reinterpret_cast<uint32_t*>(keycnt.data_ptr<int32_t>()), s),
             "pileup_featurize");
    check_rc(cto_gather_windows(&v, colvec.data_ptr<int16_t>(), coldepth.data_ptr<int32_t>(), site_pos.data_ptr<int32_t>(), n, int(min_bq),
                                int(min_rescale_cov), x_aff.data_ptr<float>(), x_neg.data_ptr<float>(), nullptr, nullptr,
                                site_info.data_ptr<int32_t>(), sitefirst.data_ptr<int32_t>(), keyfirst.data_ptr<int32_t>(), s),
             "pileup_featurize");
    return {x_aff, x_neg, site_info, colvec.narrow(0, 0, nc), coldepth.narrow(0, 0, nc), keycnt.narrow(0, 0, nk), sitefirst.narrow(0, 0, n),
            keyfirst.narrow(0, 0, nk)};
}
Feat pileup_featurize_meta(const Tensor&, const Tensor&, const Tensor& col_pos, const Tensor&, const Tensor&, const Tensor& key_meta,
                           const Tensor&, const Tensor& site_pos, int64_t, int64_t) {
    const int64_t nc = col_pos.numel(), nk = key_meta.numel(), n = site_pos.numel();
    const auto o32 = site_pos.options();
    return {at::empty({n, CTO_NPOS, CTO_NCHAN}, o32.dtype(at::kFloat)), at::empty({n, CTO_NPOS, CTO_NCHAN}, o32.dtype(at::kFloat)),
            at::empty({n, 12}, o32), at::empty({nc, 72}, o32.dtype(at::kShort)), at::empty({nc, 2}, o32), at::empty({nk}, o32),
            at::empty({n, 8}, o32), at::empty({nk, 2}, o32)};
}

}  // namespace

TORCH_LIBRARY(clairsto, m) {
    m.def("cvt_forward(Tensor x, Tensor packed_weights, int[] cfg) -> Tensor");
    m.def("bigru_forward(Tensor x, Tensor packed_weights, int n_out) -> Tensor");
    m.def("softmax2(Tensor logits) -> Tensor");
    m.def("posterior(Tensor aff_logits, Tensor neg_logits, Tensor lik, Tensor edges) -> (Tensor probs, Tensor post, Tensor decision, Tensor qual)");
    m.def("pileup_featurize(Tensor entries, Tensor col_off, Tensor col_pos, Tensor col_ref, Tensor key_off, Tensor key_meta, "
          "Tensor key_group, Tensor site_pos, int min_bq, int min_rescale_cov) -> (Tensor x_aff, Tensor x_neg, Tensor site_info, "
          "Tensor colvec, Tensor coldepth, Tensor keycnt, Tensor sitefirst, Tensor keyfirst)");
}

TORCH_LIBRARY_IMPL(clairsto, CUDA, m) {      // "CUDA" is the dispatch key of HIP devices in PyTorch-ROCm
    m.impl("cvt_forward", &cvt_forward);
    m.impl("bigru_forward", &bigru_forward);
    m.impl("softmax2", &softmax2);
    m.impl("posterior", &posterior);
    m.impl("pileup_featurize", &pileup_featurize);
}

TORCH_LIBRARY_IMPL(clairsto, Meta, m) {
    m.impl("cvt_forward", &cvt_forward_meta);
    m.impl("bigru_forward", &bigru_forward_meta);
    m.impl("softmax2", &softmax2_meta);
    m.impl("posterior", &posterior_meta);
    m.impl("pileup_featurize", &pileup_featurize_meta);
}

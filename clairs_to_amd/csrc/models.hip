// Host side of the two networks: weight hand-over by state_dict name, packing into the layouts the
// kernels consume, and the launch sequences of the eval-mode forward passes.
//   CvT / CvT_Indel            clairs/model.py:150-384   (AFF network)
//   BiGRU_NACGT / .._Indel     clairs/model.py:387-560   (NEG network; the `lstm*` attributes are nn.GRU)
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <vector>
#include "common.h"
#include "nn_kernels.h"
#include "cvt_block.h"

using namespace cto;

// The recurrent kernels live in their own translation unit (gru.hip): co-compiling them with the CvT kernels changed
// their register allocation and cost up to 4 % from one unrelated edit to the next.
int launch_gru_layer1(hipStream_t s, const float* x, const float* W, const float* Wf, const float* bias, float* out, int64_t B);
bool gru_layer1_takes_raw();
int launch_gru_layer1_raw(hipStream_t s, const int16_t* x_raw, const int32_t* site_info, int which, int min_rescale_cov, const float* Wf,
                          const float* bias, float* out, int64_t B);
int launch_gru_layer2_fc1_split(hipStream_t s, const float* x, const void* Wp, const float* bias, const void* Fp, float* fc1_part,
                                int64_t B, bool f16, const float* scale5);
int launch_gru_layer1_split(hipStream_t s, const float* x, const void* Wp, const float* bias, float* out, int64_t B, bool f16,
                            const float* scale5);
int launch_gru_layer2_fc1(hipStream_t s, const float* x, const float* W, const float* Wf, const float* bias, const float* fc1w,
                          const float* fc1f, float* fc1_part, int64_t B);


struct cto_weights {
    std::map<std::string, std::vector<float>> t;
};

extern "C" cto_weights* cto_weights_new(void) { return new cto_weights(); }
extern "C" void cto_weights_free(cto_weights* w) { delete w; }
extern "C" int cto_weights_add(cto_weights* w, const char* name, const float* data, int64_t numel) {
    CTO_REQUIRE(w && name && data && numel >= 0, CTO_EINVAL, "cto_weights_add: bad argument");
    w->t[name].assign(data, data + numel);
    return CTO_OK;
}

namespace {

struct Arena {
    std::vector<void*> ptrs;
    ~Arena() { for (void* p : ptrs) (void)hipFree(p); }
    int alloc(void** out, size_t bytes) {
        CTO_HIP(hipMalloc(out, bytes ? bytes : 16));
        ptrs.push_back(*out);
        return CTO_OK;
    }
    int upload(const std::vector<float>& v, float** out) {
        void* p = nullptr;
        int rc = alloc(&p, v.size() * sizeof(float));
        if (rc != CTO_OK) return rc;
        CTO_HIP(hipMemcpy(p, v.data(), v.size() * sizeof(float), hipMemcpyHostToDevice));
        *out = static_cast<float*>(p);
        return CTO_OK;
    }
};

struct HeadDev {          // fc1 -> SELU -> K x (fc2 -> SELU -> fc3 -> SELU)
    float *w1 = nullptr, *b1 = nullptr;   // [128][K1]
    float* w1p = nullptr;                 // CvT: fc1 over the LDS image of the last block's tile ([128][KCH1*16]), or null
    float *w2 = nullptr, *b2 = nullptr;   // [K*128][128]
    float *w3 = nullptr, *b3 = nullptr;   // [K][2][128]
    int k1 = 0;
};

struct BlockDev {
    float *n0g, *n0b, *dwq, *bnq, *wq, *dwkv, *bnkv, *wkv, *wo, *bo, *n1g, *n1b, *w1, *b1, *w2, *b2;
    // the five GEMM weights in fragment order for the fused block kernel (cvt_gemm.h); the row-major ones serve the unfused path
    float *wq_f = nullptr, *wkv_f = nullptr, *wo_f = nullptr, *w1_f = nullptr, *w2_f = nullptr;
    // split-operand experiment (CTO_CVT_SPLIT): (hi, lo) 16-bit fragments per GEMM weight, or null
    float *wq_s = nullptr, *wkv_s = nullptr, *wo_s = nullptr, *w1_s = nullptr, *w2_s = nullptr;
};

struct StageDev {
    int cin, c, win, w, wkv, heads, inner;
    float *wemb, *bemb, *lng, *lnb;
    float* wembp = nullptr;       // wemb re-laid for the in-block embedding: [C][emb_kch(cin)*16], positions padded to emb_ps(cin)
    std::vector<BlockDev> blocks;
};

}  // namespace

struct cto_model {
    int device = -1;   // HIP device the weights and workspaces live on (the device that was current at creation)
    int kind = 0;      // 0 = CvT, 1 = BiGRU
    int n_out = 4;
    Arena arena;
    // CvT
    StageDev st[3];
    // BiGRU
    float *gw1 = nullptr, *gb1 = nullptr, *gw2 = nullptr, *gb2 = nullptr;
    float *gw1f = nullptr, *gw2f = nullptr, *f1f = nullptr;     // the recurrent weights and the fused fc1 in fragment order (rotated kernels)
    float *gw1_split = nullptr, *gw2_split = nullptr, *f1_split = nullptr;     // layer 2 / fc1 as (hi, lo) 16-bit fragments: CTO_GRU_SPLIT=f16|bf16 (experiment)
    bool split_f16 = false;
    float split_sc1[5] = {1.f, 1.f, 1.f, 1.f, 1.f}, split_sc2[5] = {1.f, 1.f, 1.f, 1.f, 1.f};   // GruSplitScale of layer 1 / layer 2 (gru_split_kernel.h)
    int cvt_split = 0;          // CvT block GEMMs on split operands: 0 = fp32 kernels, 1 = f16, 2 = bf16 (CTO_CVT_SPLIT, experiment)
    HeadDev head;
    int64_t macs = 0;
    // workspace
    int64_t ws_B = 0;
    std::vector<void*> ws_ptrs;
    float* b_xexp = nullptr;          // cto_model_forward_raw's fp32 copy of the int16 tensor where the first layer cannot read it
    int64_t xexp_B = 0;
    // CvT fusion levels, read from the environment when the model is created (debugging / A-B testing):
    bool fuse_blocks = true;   // fused transformer-block kernel where the stage geometry allows (CTO_CVT_UNFUSED=1 disables)
    bool fuse_embed = true;    // stage embedding + LayerNorm inside the stage's first block (CTO_CVT_NO_EMBED_FUSE=1 disables)
    bool fuse_head = true;     // fc1 + classifier tail inside the network's last block (CTO_CVT_NO_HEAD_FUSE=1 disables)
    // live kernel timing (cto_model_profile)
    bool prof = false;          // the dominant kernel is bracketed by events
    bool prof_all = false;      // ... and BiGRU layer 1 too (cto_model_profile(m, 2))
    std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_ev;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_ev1;     // BiGRU layer 1 (cto_model_profile_read_stage, stage 1)
    int64_t prof_macs = 0;
    float *b_h = nullptr, *b_t = nullptr, *b_yq = nullptr, *b_ykv = nullptr, *b_q = nullptr, *b_kv = nullptr,
          *b_o = nullptr, *b_u = nullptr, *b_slab = nullptr, *b_h2 = nullptr;
    ~cto_model() {
        if (b_xexp) (void)hipFree(b_xexp);
        for (void* p : ws_ptrs) (void)hipFree(p);
        for (auto& e : prof_ev) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
        for (auto& e : prof_ev1) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
    }
};

namespace {

const std::vector<float>* find(const cto_weights* w, const std::string& name, int64_t numel, int* rc) {
    auto it = w->t.find(name);
    if (it == w->t.end()) { set_error("weight '%s' missing", name.c_str()); *rc = CTO_EMISSING; return nullptr; }
    if (int64_t(it->second.size()) != numel) {
        set_error("weight '%s' has %zu elements, expected %lld", name.c_str(), it->second.size(), (long long)numel);
        *rc = CTO_EMISSING;
        return nullptr;
    }
    return &it->second;
}

#define GETW(var, name, numel)                                   \
    const std::vector<float>* var = find(w, (name), (numel), &rc); \
    if (!var) return rc;

// a row-major panel W[ntiles * 16][kch * 16] in the order the MFMA lanes read it: [n-tile][16-wide k chunk][lane = (kg << 4) | j][4],
// lane (j, kg) holding W[tile * 16 + j][16 c + 4 kg .. + 3] - a wave's request becomes one contiguous 1 KB (cvt_gemm.h: load_group)
std::vector<float> pack_fragments(const float* W, int ntiles, int kch) {
    std::vector<float> f(size_t(ntiles) * kch * 256);
    for (int t = 0; t < ntiles; ++t)
        for (int c = 0; c < kch; ++c)
            for (int lane = 0; lane < 64; ++lane) {
                const int j = lane & 15, kg = lane >> 4;
                for (int e = 0; e < 4; ++e)
                    f[((size_t(t) * kch + c) * 64 + lane) * 4 + e] = W[size_t(t * 16 + j) * (kch * 16) + c * 16 + 4 * kg + e];
            }
    return f;
}

int build_head(const cto_weights* w, const char* const* names, int K, int k1, const std::vector<float>& w1perm,
               Arena& a, HeadDev& h) {
    int rc = CTO_OK;
    h.k1 = k1;
    GETW(b1, "fc1.bias", 128);
    if ((rc = a.upload(w1perm, &h.w1)) != CTO_OK) return rc;
    if ((rc = a.upload(*b1, &h.b1)) != CTO_OK) return rc;
    std::vector<float> w2(size_t(K) * 128 * 128), b2(size_t(K) * 128), w3(size_t(K) * 2 * 128), b3(size_t(K) * 2);
    for (int k = 0; k < K; ++k) {
        const std::string p = names[k];
        GETW(f2w, p + "_fc2.weight", 128 * 128);
        GETW(f2b, p + "_fc2.bias", 128);
        GETW(f3w, p + "_fc3.weight", 2 * 128);
        GETW(f3b, p + "_fc3.bias", 2);
        const std::vector<float> frag = pack_fragments(f2w->data(), 8, 8);       // the classifier tail reads fc2 in fragment order
        std::copy(frag.begin(), frag.end(), w2.begin() + size_t(k) * 128 * 128);
        std::copy(f2b->begin(), f2b->end(), b2.begin() + size_t(k) * 128);
        std::copy(f3w->begin(), f3w->end(), w3.begin() + size_t(k) * 256);
        std::copy(f3b->begin(), f3b->end(), b3.begin() + size_t(k) * 2);
    }
    if ((rc = a.upload(w2, &h.w2)) != CTO_OK) return rc;
    if ((rc = a.upload(b2, &h.b2)) != CTO_OK) return rc;
    if ((rc = a.upload(w3, &h.w3)) != CTO_OK) return rc;
    if ((rc = a.upload(b3, &h.b3)) != CTO_OK) return rc;
    return CTO_OK;
}

int pack_bn(const cto_weights* w, const std::string& p, int C, Arena& a, float** out) {
    int rc = CTO_OK;
    GETW(wt, p + ".weight", C);
    GETW(bs, p + ".bias", C);
    GETW(mu, p + ".running_mean", C);
    GETW(var, p + ".running_var", C);
    std::vector<float> v(size_t(4) * C);
    for (int c = 0; c < C; ++c) {
        v[size_t(c)] = (*mu)[size_t(c)];
        v[size_t(C + c)] = float(1.0 / std::sqrt(double((*var)[size_t(c)]) + 1e-5));   // BatchNorm2d eps (eval)
        v[size_t(2 * C + c)] = (*wt)[size_t(c)];
        v[size_t(3 * C + c)] = (*bs)[size_t(c)];
    }
    return a.upload(v, out);
}

int pack_dw(const cto_weights* w, const std::string& name, int C, Arena& a, float** out) {
    int rc = CTO_OK;
    GETW(k, name, int64_t(C) * 9);
    std::vector<float> v(size_t(C) * 3);
    for (int c = 0; c < C; ++c)
        for (int t = 0; t < 3; ++t) v[size_t(c * 3 + t)] = (*k)[size_t((c * 3 + 1) * 3 + t)];   // middle kernel row
    return a.upload(v, out);
}

int upload_named(const cto_weights* w, const std::string& name, int64_t numel, Arena& a, float** out) {
    int rc = CTO_OK;
    GETW(v, name, numel);
    return a.upload(*v, out);
}

// ---------------------------------------------------------------- launch helpers
int launch_gemm(hipStream_t s, const float* A, int64_t lda, const float* W, int64_t ldw, const float* bias,
                const float* R, int64_t ldr, float* C, int64_t ldc, int M, int N, int K, int act,
                int conv_win = 0, int conv_wout = 0, int conv_cin = 0, int splitk = 1, int64_t slab = 0) {
    GemmArgs g;
    g.A = A; g.lda = lda; g.W = W; g.ldw = ldw; g.bias = bias; g.R = R; g.ldr = ldr; g.C = C; g.ldc = ldc;
    g.M = M; g.N = N; g.K = K; g.act = act;
    g.conv_win = conv_win; g.conv_wout = conv_wout; g.conv_cin = conv_cin;
    g.kslice = splitk > 1 ? int(cdiv(cdiv(K, splitk), 16) * 16) : K;
    g.slab = slab;
    g.vecA = (conv_wout == 0 && (lda % 4) == 0 && (K % 4) == 0 && (reinterpret_cast<uintptr_t>(A) % 16) == 0) ? 1 : 0;
    if (M <= 0) return CTO_OK;
    if (N <= 16) {
        dim3 grid(unsigned(cdiv(M, 128)), unsigned(cdiv(N, 16)), unsigned(splitk));
        hipLaunchKernelGGL((k_gemm<4, 1, 2, 1>), grid, dim3(256), 0, s, g);
    } else if (N <= 64) {
        dim3 grid(unsigned(cdiv(M, 64)), unsigned(cdiv(N, 64)), unsigned(splitk));
        hipLaunchKernelGGL((k_gemm<2, 2, 2, 2>), grid, dim3(256), 0, s, g);
    } else {
        dim3 grid(unsigned(cdiv(M, 64)), unsigned(cdiv(N, 128)), unsigned(splitk));
        hipLaunchKernelGGL((k_gemm<2, 2, 2, 4>), grid, dim3(256), 0, s, g);
    }
    CTO_HIP(hipGetLastError());
    return CTO_OK;
}

int ensure_ws(cto_model* m, int64_t B) {
    if (B <= m->ws_B) return CTO_OK;
    for (void* p : m->ws_ptrs) (void)hipFree(p);
    m->ws_ptrs.clear();
    m->ws_B = 0;
    auto get = [&](float** out, int64_t per_site) -> int {
        void* p = nullptr;
        CTO_HIP(hipMalloc(&p, size_t(B) * size_t(per_site) * sizeof(float)));
        m->ws_ptrs.push_back(p);
        *out = static_cast<float*>(p);
        return CTO_OK;
    };
    int rc;
    if (m->kind == 0) {
        int64_t act = 0, qn = 0, kvn = 0, un = 0;
        for (const StageDev& s : m->st) {
            act = std::max<int64_t>(act, int64_t(s.w) * s.c);
            qn = std::max<int64_t>(qn, int64_t(s.w) * s.inner);
            kvn = std::max<int64_t>(kvn, int64_t(s.wkv) * 2 * s.inner);
            un = std::max<int64_t>(un, int64_t(s.w) * 4 * s.c);
        }
        if ((rc = get(&m->b_h, act)) || (rc = get(&m->b_h2, act)) || (rc = get(&m->b_t, act)) || (rc = get(&m->b_yq, act)) ||
            (rc = get(&m->b_ykv, act)) || (rc = get(&m->b_q, qn)) || (rc = get(&m->b_kv, kvn)) ||
            (rc = get(&m->b_o, qn)) || (rc = get(&m->b_u, un)) || (rc = get(&m->b_slab, 4 * 128)))
            return rc;
    } else {
        if ((rc = get(&m->b_h, 33 * 256)) || (rc = get(&m->b_slab, 2 * 128))) return rc;
    }
    m->ws_B = B;
    return CTO_OK;
}

// fc1 (split-K when the reduction is long) -> SELU -> fc2 heads -> SELU -> fc3 -> SELU
int launch_head(cto_model* m, hipStream_t s, const float* slabs, int S, int64_t B, float* logits) {
    const HeadDev& h = m->head;
    const int K = m->n_out;
    const size_t smem = size_t(head_lds_floats(K)) * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        CTO_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_head), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    int(head_lds_floats(6) * sizeof(float))));
        attr_set = true;
    }
    HeadTailParams hp{h.w2, h.b2, h.w3, h.b3, logits, K};
    hipLaunchKernelGGL(k_head, dim3(unsigned(cdiv(B, 16))), dim3(512), smem, s, slabs, S, B * 128, h.b1, hp, B);
    CTO_HIP(hipGetLastError());
    return CTO_OK;
}

int run_head(cto_model* m, hipStream_t s, const float* feat, int64_t B, float* logits) {
    const HeadDev& h = m->head;
    int rc;
    if (feat == nullptr)   // BiGRU: fc1 was accumulated inside the layer-2 recurrent kernel, one slab per direction
        return launch_head(m, s, m->b_slab, 2, B, logits);
    // CvT (unfused tail): fc1 has M = B rows and only N = 128 columns; split K four ways so that the launch fills the chip
    const int S = 4;
    if ((rc = launch_gemm(s, feat, h.k1, h.w1, h.k1, nullptr, nullptr, 0, m->b_slab, 128, int(B), 128, h.k1, ACT_NONE,
                          0, 0, 0, S, B * 128)))
        return rc;
    return launch_head(m, s, m->b_slab, S, B, logits);
}

struct RawIn { const int16_t* x; const int32_t* site_info; int which, min_rescale_cov; };     // the int16 tensor + what rescales it

struct BlockExtra {            // what the first / last block of the network additionally needs
    const float* xin = nullptr;          // stage input when the embedding runs in the block
    const RawIn* raw = nullptr;          // ... as the int16 tensor instead
    const StageDev* st = nullptr;
    const HeadDev* head = nullptr;       // classifier when the tail runs in the block
    float* logits = nullptr;
    int n_out = 0;
};

template <int C, int W, int WKV, int TS, int CIN, bool HEAD, int SPLIT = 0>
int launch_cvt_block(hipStream_t s, float* h, const BlockDev* blocks, int nblk, int heads, int64_t B, const BlockExtra& ex) {
    using G = CvtBlockGeom<C, W, WKV, TS>;
    static_assert(G::LDS_BYTES <= 160 * 1024, "fused CvT block does not fit the 160 KB LDS of a gfx950 CU");
    static bool attr_set = false;
    if (!attr_set) {
        CTO_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_cvt_block<C, W, WKV, TS, CIN, HEAD, SPLIT>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, int(G::LDS_BYTES)));
        attr_set = true;
    }
    static long long* prof_buf = nullptr;
    static const bool prof_on = [] { const char* e = getenv("CTO_BLOCK_PROF"); return e && e[0] == '1'; }();
    if (prof_on && !prof_buf) CTO_HIP(hipMalloc(reinterpret_cast<void**>(&prof_buf), 256 * sizeof(long long)));
    if (prof_on) CTO_HIP(hipMemsetAsync(prof_buf, 0, 256 * sizeof(long long), s));
    CvtStageParams sp{};
    sp.nblk = nblk;
    for (int i = 0; i < nblk; ++i) {
        const BlockDev& b = blocks[i];
        sp.blk[i] = CvtBlockParams{b.n0g, b.n0b, b.dwq, b.bnq, b.wq_f, b.dwkv, b.bnkv, b.wkv_f, b.wo_f, b.bo, b.n1g, b.n1b, b.w1_f, b.b1, b.w2_f, b.b2,
                                   prof_on ? prof_buf : nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                                   nullptr, nullptr, nullptr, nullptr, nullptr};
        if (SPLIT != 0) {
            CTO_REQUIRE(b.wq_s && b.wkv_s && b.wo_s && b.w1_s && b.w2_s, CTO_EINVAL, "split CvT block without split weights");
            CvtBlockParams& p = sp.blk[i];
            p.wq_s = reinterpret_cast<const unsigned short*>(b.wq_s); p.wkv_s = reinterpret_cast<const unsigned short*>(b.wkv_s);
            p.wo_s = reinterpret_cast<const unsigned short*>(b.wo_s); p.w1_s = reinterpret_cast<const unsigned short*>(b.w1_s);
            p.w2_s = reinterpret_cast<const unsigned short*>(b.w2_s);
        }
    }
    HeadTailParams hp{};
    if (CIN > 0) {
        CvtBlockParams& p = sp.blk[0];
        p.xin = ex.xin; p.wembp = ex.st->wembp; p.bemb = ex.st->bemb; p.lng = ex.st->lng; p.lnb = ex.st->lnb;
        if (ex.raw) { p.xraw = reinterpret_cast<const short*>(ex.raw->x); p.xinfo = ex.raw->site_info; p.xwhich = ex.raw->which; p.xcov = ex.raw->min_rescale_cov; }
    }
    if (HEAD) {
        CvtBlockParams& p = sp.blk[nblk - 1];
        p.w1p = ex.head->w1p; p.b1h = ex.head->b1;
        hp = HeadTailParams{ex.head->w2, ex.head->b2, ex.head->w3, ex.head->b3, ex.logits, ex.n_out};
    }
    hipLaunchKernelGGL((k_cvt_block<C, W, WKV, TS, CIN, HEAD, SPLIT>), dim3(unsigned(cdiv(B, TS))), dim3(CVT_BLOCK_THREADS), G::LDS_BYTES,
                       s, h, sp, hp, heads, int(B));
    CTO_HIP(hipGetLastError());
    if (prof_on) {   // debug aid: phase time stamps of workgroup 0 (cycles since the kernel's first stamp)
        long long hst[256];
        CTO_HIP(hipStreamSynchronize(s));
        CTO_HIP(hipMemcpy(hst, prof_buf, sizeof(hst), hipMemcpyDeviceToHost));
        fprintf(stderr, "k_cvt_block<%d,%d,%d,%d,%d,%d> x%d stamps:", C, W, WKV, TS, CIN, int(HEAD), nblk);
        for (int i = 1; i < 256 && hst[i]; ++i) fprintf(stderr, " %lld", hst[i] - hst[i - 1]);
        fprintf(stderr, "\n");
    }
    return CTO_OK;
}

// The instantiated geometries: (C, W, WKV) -> sites per workgroup and the stage-input channel count whose embedding
// can run inside the stage's first block.
template <int C, int W, int WKV, int TS, int CIN, int SPLIT>
int dispatch_block_kind(hipStream_t s, float* h, const BlockDev* blocks, int nblk, int heads, int64_t B, const BlockExtra& ex, bool embed, bool head) {
    if constexpr (C == 128 && CvtBlockGeom<C, W, WKV, TS>::HEAD_OK) {
        if (head) return embed ? launch_cvt_block<C, W, WKV, TS, CIN, true, SPLIT>(s, h, blocks, nblk, heads, B, ex)
                               : launch_cvt_block<C, W, WKV, TS, 0, true, SPLIT>(s, h, blocks, nblk, heads, B, ex);
    }
    return embed ? launch_cvt_block<C, W, WKV, TS, CIN, false, SPLIT>(s, h, blocks, nblk, heads, B, ex)
                 : launch_cvt_block<C, W, WKV, TS, 0, false, SPLIT>(s, h, blocks, nblk, heads, B, ex);
}
// split (0 = fp32; 1 = f16, 2 = bf16: experiment) exists for the 64- and 128-channel stages with 16-site tiles
template <int C, int W, int WKV, int TS, int CIN>
int dispatch_block(hipStream_t s, float* h, const BlockDev* blocks, int nblk, int heads, int64_t B, const BlockExtra& ex, bool embed, bool head,
                   int split) {
    if constexpr (C % 64 == 0 && MT_EXACT<C, W, WKV, TS>()) {
        if (split == 1) return dispatch_block_kind<C, W, WKV, TS, CIN, 1>(s, h, blocks, nblk, heads, B, ex, embed, head);
        if (split == 2) return dispatch_block_kind<C, W, WKV, TS, CIN, 2>(s, h, blocks, nblk, heads, B, ex, embed, head);
    }
    return dispatch_block_kind<C, W, WKV, TS, CIN, 0>(s, h, blocks, nblk, heads, B, ex, embed, head);
}
// Sites per workgroup of the stage-1 / stage-2 blocks.  What they trade is LDS per workgroup (q / k / v tiles dominate) against
// workgroups resident per CU: the blocks of these stages do little matrix work per phase, so a second and third resident
// workgroup - whose GEMM phases run under this one's LayerNorm / softmax / barrier phases - is worth more than a taller tile.
#ifndef CTO_CVT_TS1
#define CTO_CVT_TS1 8
#endif
#ifndef CTO_CVT_TS2
#define CTO_CVT_TS2 16
#endif
#ifndef CTO_CVT_TS3
#define CTO_CVT_TS3 16      // stage 3; 8 (tools/ builds, with CTO_CVT_NO_HEAD_FUSE=1: the classifier needs the 16-site tile) prices half-height tiles
#endif
struct FusedGeom { int c, w, wkv, ts, cin; };
const FusedGeom* fused_geom(const StageDev& st) {
    static const FusedGeom G[4] = {{128, 5, 3, CTO_CVT_TS3, 64}, {64, 9, 5, CTO_CVT_TS2, 16}, {16, 17, 9, CTO_CVT_TS1, 34}, {32, 17, 9, CTO_CVT_TS1, 34}};
    for (const FusedGeom& g : G)
        if (st.c == g.c && st.w == g.w && st.wkv == g.wkv) return &g;
    return nullptr;
}
bool can_fuse_embed(const StageDev& st) { const FusedGeom* g = fused_geom(st); return g && g->cin == st.cin && st.wembp; }
bool can_fuse_head(const StageDev& st, const HeadDev& hd) { const FusedGeom* g = fused_geom(st); return g && g->ts == 16 && g->c == 128 && hd.w1p; }

// nblk consecutive fused transformer blocks of a stage in one launch, when the stage geometry has an instantiation; returns 1 if it
// ran, 0 if not, < 0 on error
int try_fused_blocks(hipStream_t s, const StageDev& st, const BlockDev* b, int nblk, float* h, int64_t B, const BlockExtra& ex, bool embed,
                     bool head, int split) {
    int rc = CTO_OK;
    if (!b->wq_s) split = 0;
    if (st.c == 128 && st.w == 5 && st.wkv == 3) rc = dispatch_block<128, 5, 3, CTO_CVT_TS3, 64>(s, h, b, nblk, st.heads, B, ex, embed, head, split);
    else if (st.c == 64 && st.w == 9 && st.wkv == 5) rc = dispatch_block<64, 9, 5, CTO_CVT_TS2, 16>(s, h, b, nblk, st.heads, B, ex, embed, head, split);
    else if (st.c == 16 && st.w == 17 && st.wkv == 9) rc = dispatch_block<16, 17, 9, CTO_CVT_TS1, 34>(s, h, b, nblk, st.heads, B, ex, embed, head, split);
    else if (st.c == 32 && st.w == 17 && st.wkv == 9) rc = dispatch_block<32, 17, 9, CTO_CVT_TS1, 34>(s, h, b, nblk, st.heads, B, ex, embed, head, split);
    else return 0;
    return rc == CTO_OK ? 1 : rc;
}

int cvt_forward(cto_model* m, const float* x, int64_t B, float* logits, hipStream_t s, const RawIn* raw = nullptr) {
    int rc;
    const float* in = x;
    for (int si = 0; si < 3; ++si) {
        const StageDev& st = m->st[si];
        const int M = int(B) * st.w, Mkv = int(B) * st.wkv, C = st.c;
        // The stage's residual stream alternates between two buffers: with the embedding inside the first block, that block
        // reads the previous stage's output (another per-site layout) while other workgroups already write this stage's.
        float* hbuf = (si & 1) ? m->b_h2 : m->b_h;
        // conv embedding (3-tap stride 2) + channel LayerNorm: inside the stage's first block when that is fused
        const bool embed_in_block = m->fuse_blocks && m->fuse_embed && can_fuse_embed(st);
        if (!embed_in_block) {
            if ((rc = launch_gemm(s, in, 0, st.wemb, 3 * st.cin, st.bemb, nullptr, 0, m->b_t, C, M, C, 3 * st.cin, ACT_NONE,
                                  st.win, st.w, st.cin)))
                return rc;
            hipLaunchKernelGGL(k_layernorm, dim3(unsigned(cdiv(M, 4))), dim3(256), 0, s, m->b_t, hbuf, st.lng, st.lnb, M, C);
            CTO_HIP(hipGetLastError());
        }
        // consecutive blocks of the stage share a launch (CTO_CVT_BLOCKS_PER_LAUNCH=1: one launch per block, as before round 3)
        static const int group_max = [] {
            const char* e = getenv("CTO_CVT_BLOCKS_PER_LAUNCH");
            const int v = e ? atoi(e) : CVT_MAX_BLK;
            return v < 1 ? 1 : (v > CVT_MAX_BLK ? CVT_MAX_BLK : v);
        }();
        for (size_t bi = 0; bi < st.blocks.size(); ++bi) {
            const BlockDev& b = st.blocks[bi];
            if (m->fuse_blocks) {
                BlockExtra ex;
                const int nblk = int(std::min<size_t>(size_t(group_max), st.blocks.size() - bi));
                const bool embed = embed_in_block && bi == 0;
                const bool head = m->fuse_head && si == 2 && bi + size_t(nblk) == st.blocks.size() && can_fuse_head(st, m->head);
                ex.xin = in; ex.st = &st; ex.head = &m->head; ex.logits = logits; ex.n_out = m->n_out;
                ex.raw = (si == 0 && embed) ? raw : nullptr;
                const int fr = try_fused_blocks(s, st, &b, nblk, hbuf, B, ex, embed, head, m->cvt_split);
                if (fr < 0) return fr;
                if (fr == 1) {
                    if (head) return CTO_OK;
                    bi += size_t(nblk) - 1;
                    continue;
                }
            }
            float* h = hbuf;
            hipLaunchKernelGGL(k_ln_dw, dim3(unsigned(B)), dim3(256), 0, s, h, b.n0g, b.n0b, b.dwq, b.bnq, b.dwkv,
                               b.bnkv, m->b_yq, m->b_ykv, st.w, st.wkv, C);
            CTO_HIP(hipGetLastError());
            if ((rc = launch_gemm(s, m->b_yq, C, b.wq, C, nullptr, nullptr, 0, m->b_q, st.inner, M, st.inner, C, ACT_NONE)))
                return rc;
            if ((rc = launch_gemm(s, m->b_ykv, C, b.wkv, C, nullptr, nullptr, 0, m->b_kv, 2 * st.inner, Mkv, 2 * st.inner,
                                  C, ACT_NONE)))
                return rc;
            const size_t smem = size_t(st.w * st.inner + st.wkv * 2 * st.inner + st.heads * st.w * st.wkv) * sizeof(float);
            hipLaunchKernelGGL(k_attention, dim3(unsigned(B)), dim3(256), smem, s, m->b_q, m->b_kv, m->b_o, st.w, st.wkv,
                               st.heads);
            CTO_HIP(hipGetLastError());
            // h = h + to_out(o)
            if ((rc = launch_gemm(s, m->b_o, st.inner, b.wo, st.inner, b.bo, h, C, h, C, M, C, st.inner, ACT_NONE)))
                return rc;
            hipLaunchKernelGGL(k_layernorm, dim3(unsigned(cdiv(M, 4))), dim3(256), 0, s, h, m->b_t, b.n1g, b.n1b, M, C);
            CTO_HIP(hipGetLastError());
            if ((rc = launch_gemm(s, m->b_t, C, b.w1, C, b.b1, nullptr, 0, m->b_u, 4 * C, M, 4 * C, C, ACT_GELU))) return rc;
            // h = h + ff2(u)
            if ((rc = launch_gemm(s, m->b_u, 4 * C, b.w2, 4 * C, b.b2, h, C, h, C, M, C, 4 * C, ACT_NONE))) return rc;
        }
        in = hbuf;
    }
    return run_head(m, s, in, B, logits);
}

int prof_begin(cto_model* m, hipStream_t s, hipEvent_t* e0, hipEvent_t* e1) {
    CTO_HIP(hipEventCreate(e0));
    CTO_HIP(hipEventCreate(e1));
    CTO_HIP(hipEventRecord(*e0, s));
    return CTO_OK;
}

int bigru_forward(cto_model* m, const float* x, int64_t B, float* logits, hipStream_t s, const RawIn* raw = nullptr) {
    int rc;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (m->prof_all && (rc = prof_begin(m, s, &e0, &e1))) return rc;
    if (raw) rc = launch_gru_layer1_raw(s, raw->x, raw->site_info, raw->which, raw->min_rescale_cov, m->gw1f, m->gb1, m->b_h, B);
    else if (m->gw1_split) rc = launch_gru_layer1_split(s, x, m->gw1_split, m->gb1, m->b_h, B, m->split_f16, m->split_sc1);
    else rc = launch_gru_layer1(s, x, m->gw1, m->gw1f, m->gb1, m->b_h, B);
    if (rc) return rc;
    if (m->prof_all) {
        CTO_HIP(hipEventRecord(e1, s));
        m->prof_ev1.emplace_back(e0, e1);
    }
    if (m->prof && (rc = prof_begin(m, s, &e0, &e1))) return rc;
    // layer 2 with the head's fc1 folded in: writes one partial [B][128] slab per direction into b_slab
    if (m->gw2_split) rc = launch_gru_layer2_fc1_split(s, m->b_h, m->gw2_split, m->gb2, m->f1_split, m->b_slab, B, m->split_f16, m->split_sc2);
    else rc = launch_gru_layer2_fc1(s, m->b_h, m->gw2, m->gw2f, m->gb2, m->head.w1, m->f1f, m->b_slab, B);
    if (rc) return rc;
    if (m->prof) {
        CTO_HIP(hipEventRecord(e1, s));
        m->prof_ev.emplace_back(e0, e1);
    }
    return run_head(m, s, nullptr, B, logits);
}

// [W_ih | W_hh] per gate row, W_ih zero-padded to KP; bias rows: r (b_ir + b_hr), z (b_iz + b_hz), b_in, b_hn
// Wfout: the same weights in the order the rotated kernel's lanes read them, [dir][wave][16-wide chunk][nb][gate][lane][4] with lane
// (j, kg) holding row  gate * H + (wave * NB + nb) * 16 + j,  k = 16 c + 4 kg .. + 3  of [W_ih (zero-padded to kp) | W_hh]; when the
// last x chunk holds at most four real channels (layer 1: 34 -> channels 32, 33) it is ONE k-step whose lane group kg holds
// k = 16 c + kg in element 0 (gru_kernel.h: TAIL1)
int pack_gru(const cto_weights* w, const std::string& base, int kin, int kp, int H, Arena& a, float** Wout, float** bout, float** Wfout) {
    int rc = CTO_OK;
    const int KT = kp + H + GRU_WPAD;
    const int NB = H / 64, NX = kp / 16, NC = NX + H / 16;
    const bool tail1 = (kin % 16 != 0) && (kin - 16 * (NX - 1) <= 4);
    std::vector<float> W(size_t(2) * 3 * H * KT, 0.f), bv(size_t(2) * 4 * H), Wf(size_t(2) * 4 * NC * NB * 3 * 256, 0.f);
    for (int d = 0; d < 2; ++d) {
        const std::string sfx = d == 0 ? "" : "_reverse";
        GETW(wih, base + ".weight_ih_l0" + sfx, int64_t(3) * H * kin);
        GETW(whh, base + ".weight_hh_l0" + sfx, int64_t(3) * H * H);
        GETW(bih, base + ".bias_ih_l0" + sfx, 3 * H);
        GETW(bhh, base + ".bias_hh_l0" + sfx, 3 * H);
        for (int n = 0; n < 3 * H; ++n) {
            float* row = W.data() + (size_t(d) * 3 * H + n) * KT;
            for (int k = 0; k < kin; ++k) row[k] = (*wih)[size_t(n) * kin + k];
            for (int k = 0; k < H; ++k) row[kp + k] = (*whh)[size_t(n) * H + k];
        }
        for (int wv = 0; wv < 4; ++wv)
            for (int c = 0; c < NC; ++c)
                for (int nb = 0; nb < NB; ++nb)
                    for (int q = 0; q < 3; ++q)
                        for (int lane = 0; lane < 64; ++lane) {
                            const int jj = lane & 15, kg = lane >> 4;
                            const float* row = W.data() + (size_t(d) * 3 * H + q * H + (wv * NB + nb) * 16 + jj) * KT;
                            float* f = &Wf[(((((size_t(d) * 4 + wv) * NC + c) * NB + nb) * 3 + q) * 64 + lane) * 4];
                            if (tail1 && c == NX - 1) f[0] = row[16 * c + kg];
                            else for (int e = 0; e < 4; ++e) f[e] = row[16 * c + 4 * kg + e];      // c >= NX: kp + 16 (c - NX) = 16 c
                        }
        float* b = bv.data() + size_t(d) * 4 * H;
        for (int j = 0; j < H; ++j) {
            b[0 * H + j] = (*bih)[size_t(j)] + (*bhh)[size_t(j)];
            b[1 * H + j] = (*bih)[size_t(H + j)] + (*bhh)[size_t(H + j)];
            b[2 * H + j] = (*bih)[size_t(2 * H + j)];
            b[3 * H + j] = (*bhh)[size_t(2 * H + j)];
        }
    }
    if ((rc = a.upload(W, Wout)) != CTO_OK) return rc;
    if ((rc = a.upload(Wf, Wfout)) != CTO_OK) return rc;
    return a.upload(bv, bout);
}

// fc1.weight [128][33 * 2H] for the fused layer-2 kernel: [dir][t][wave][kh][nt][lane][4], lane (j, kg) holding row
// wave * 32 + nt * 16 + j,  k = t 2H + dir H + 16 kh + 4 kg .. + 3
std::vector<float> pack_fc1_fragments(const std::vector<float>& fc1, int H) {
    const int T = 33, NH = H / 16;
    std::vector<float> f(size_t(2) * T * 4 * NH * 2 * 256);
    for (int d = 0; d < 2; ++d)
        for (int t = 0; t < T; ++t)
            for (int wv = 0; wv < 4; ++wv)
                for (int kh = 0; kh < NH; ++kh)
                    for (int nt = 0; nt < 2; ++nt)
                        for (int lane = 0; lane < 64; ++lane) {
                            const int jj = lane & 15, kg = lane >> 4;
                            const float* src = fc1.data() + size_t(wv * 32 + nt * 16 + jj) * (T * 2 * H) + size_t(t) * 2 * H + d * H + 16 * kh + 4 * kg;
                            float* dst = &f[((((((size_t(d) * T + t) * 4 + wv) * NH + kh) * 2 + nt) * 64) + lane) * 4];
                            for (int e = 0; e < 4; ++e) dst[e] = src[e];
                        }
    return f;
}

// ---- split 16-bit operands of the layer-2 recurrence (gru_split_kernel.h; experiment behind CTO_GRU_SPLIT=f16|bf16) ----
inline uint16_t bf16_rne(float v) {
    uint32_t u;
    memcpy(&u, &v, 4);
    return uint16_t((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}
inline float bf16_value(uint16_t h) {
    const uint32_t u = uint32_t(h) << 16;
    float v;
    memcpy(&v, &u, 4);
    return v;
}
inline uint16_t f16_rne(float v) {
    const _Float16 h = _Float16(v);
    uint16_t u;
    memcpy(&u, &h, 2);
    return u;
}
inline float f16_value(uint16_t u) {
    _Float16 h;
    memcpy(&h, &u, 2);
    return float(h);
}
// eight consecutive k of one row -> the 16 bytes a lane holds, hi then (64 lanes later) lo
inline void put_split8(uint16_t* hi, uint16_t* lo, const float* src, int n_valid, bool f16, float scale = 1.f) {
    for (int e = 0; e < 8; ++e) {
        const float v = e < n_valid ? src[e] * scale : 0.f;
        if (f16) { hi[e] = f16_rne(v); lo[e] = f16_rne(v - f16_value(hi[e])); }
        else { hi[e] = bf16_rne(v); lo[e] = bf16_rne(v - bf16_value(hi[e])); }
    }
}
int upload_halves(const std::vector<uint16_t>& v, Arena& a, float** out) {
    std::vector<float> f(v.size() / 2);
    memcpy(f.data(), v.data(), v.size() * 2);
    return a.upload(f, out);
}
// a row-major weight matrix W[ntiles * 16][kch32 * 32] as (hi, lo) 16-bit fragments: [n-tile][32-wide k chunk][hi, lo][lane][8],
// lane (j, kg) holding W[tile * 16 + j][32 c + 8 kg .. + 7] (cvt_gemm.h: gemm_lds_split)
int upload_split_fragments(const cto_weights* w, const std::string& name, int ntiles, int kch32, bool f16, Arena& a, float** out) {
    int rc = CTO_OK;
    const int K = kch32 * 32;
    GETW(v, name, int64_t(ntiles) * 16 * K);
    std::vector<uint16_t> fr(size_t(ntiles) * kch32 * 2 * 64 * 8);
    for (int t = 0; t < ntiles; ++t)
        for (int c = 0; c < kch32; ++c)
            for (int lane = 0; lane < 64; ++lane) {
                const int j = lane & 15, kg = lane >> 4;
                const float* src = v->data() + size_t(t * 16 + j) * K + c * 32 + kg * 8;
                if (f16)
                    for (int e = 0; e < 8; ++e)
                        CTO_REQUIRE(std::fabs(src[e]) < 60000.f, CTO_EUNSUPPORTED, "CTO_CVT_SPLIT=f16: weight %s holds %g, outside the f16 range",
                                    name.c_str(), double(src[e]));
                const size_t u = ((size_t(t) * kch32 + c) * 2) * 64;
                put_split8(&fr[(u + lane) * 8], &fr[(u + 64 + lane) * 8], src, 8, f16);
            }
    return upload_halves(fr, a, out);
}
int upload_fragments(const cto_weights* w, const std::string& name, int ntiles, int kch, Arena& a, float** out) {
    int rc = CTO_OK;
    GETW(v, name, int64_t(ntiles) * 16 * kch * 16);
    return a.upload(pack_fragments(v->data(), ntiles, kch), out);
}

// Wp[dir][wave][chunk][nb][gate][hi,lo][lane][8], Fp[dir][t][wave][kh][nt][hi,lo][lane][8] (layouts in gru_split_kernel.h)
// fc1 == nullptr: a layer without the fused head (layer 1); kp = kin rounded up to whole 32-wide chunks, zero weights in the padding
// The f16 form lifts every operand class towards the top of the f16 range by a power of two (GruSplitScale in gru_split_kernel.h):
// log2_sx is the scale of the layer's input as the kernel stages it (0 for layer 1, whose input are counts up to 32 767; 14 for
// layer 2, whose input is layer 1's |h| <= 1), h is staged times 2^14; the weights take what is left of a common accumulator unit
// 2^S = min over the two parts of (activation scale x the largest power of two that keeps the part's largest weight below 32 768).
int pack_gru_split(const cto_weights* w, const std::string& base, int kin, int kp, int H, const std::vector<float>* fc1, bool f16,
                   int log2_sx, Arena& a, float** Wout, float** Fout, float* scale5) {
    const int NB = H / 64, NX = kp / 32, NH = H / 32, NC = NX + NH, T = 33;
    int rc = CTO_OK;
    std::vector<uint16_t> W(size_t(2) * 4 * NC * NB * 6 * 64 * 8), F(fc1 ? size_t(2) * T * 4 * NH * 4 * 64 * 8 : 0);
    float w_ih_scale = 1.f, w_hh_scale = 1.f, f_scale = 1.f;
    scale5[0] = scale5[1] = scale5[2] = scale5[3] = scale5[4] = 1.f;
    if (f16) {
        float mx_ih = 0.f, mx_hh = 0.f, mx_f = 0.f;
        for (int d = 0; d < 2; ++d) {
            const std::string sfx = d == 0 ? "" : "_reverse";
            GETW(wih, base + ".weight_ih_l0" + sfx, int64_t(3) * H * kin);
            GETW(whh, base + ".weight_hh_l0" + sfx, int64_t(3) * H * H);
            for (float v : *wih) mx_ih = std::max(mx_ih, std::fabs(v));
            for (float v : *whh) mx_hh = std::max(mx_hh, std::fabs(v));
        }
        if (fc1) for (float v : *fc1) mx_f = std::max(mx_f, std::fabs(v));
        CTO_REQUIRE(std::isfinite(mx_ih) && std::isfinite(mx_hh) && std::isfinite(mx_f), CTO_EUNSUPPORTED,
                    "split f16: %s holds a non-finite weight", base.c_str());
        auto room = [](float mx) { return mx > 0.f ? std::max(-60, std::min(60, int(std::floor(std::log2(32768.0 / double(mx)))))) : 60; };
        const int log2_sh = 14;
        const int S = std::min(log2_sx + room(mx_ih), log2_sh + room(mx_hh));
        w_ih_scale = std::ldexp(1.f, S - log2_sx);
        w_hh_scale = std::ldexp(1.f, S - log2_sh);
        const int Sf = room(mx_f);
        f_scale = std::ldexp(1.f, Sf);
        scale5[0] = std::ldexp(1.f, log2_sx); scale5[1] = std::ldexp(1.f, log2_sh); scale5[2] = std::ldexp(1.f, S);
        scale5[3] = std::ldexp(1.f, -S); scale5[4] = std::ldexp(1.f, -(log2_sh + Sf));
    }
    for (int d = 0; d < 2; ++d) {
        const std::string sfx = d == 0 ? "" : "_reverse";
        GETW(wih, base + ".weight_ih_l0" + sfx, int64_t(3) * H * kin);
        GETW(whh, base + ".weight_hh_l0" + sfx, int64_t(3) * H * H);
        for (int wv = 0; wv < 4; ++wv)
            for (int c = 0; c < NC; ++c)
                for (int nb = 0; nb < NB; ++nb)
                    for (int q = 0; q < 3; ++q)
                        for (int lane = 0; lane < 64; ++lane) {
                            const int j = lane & 15, kg = lane >> 4;
                            const int n = q * H + (wv * NB + nb) * 16 + j;
                            const int k0 = (c < NX ? c : c - NX) * 32 + kg * 8;
                            const float* src = c < NX ? wih->data() + size_t(n) * kin + k0 : whh->data() + size_t(n) * H + k0;
                            const int n_valid = c < NX ? std::max(0, std::min(8, kin - k0)) : 8;
                            const size_t u = ((((size_t(d) * 4 + wv) * NC + c) * NB + nb) * 6 + q * 2) * 64;
                            put_split8(&W[(u + lane) * 8], &W[(u + 64 + lane) * 8], n_valid > 0 ? src : whh->data(), n_valid, f16,
                                       c < NX ? w_ih_scale : w_hh_scale);
                        }
        for (int t = 0; fc1 && t < T; ++t)
            for (int wv = 0; wv < 4; ++wv)
                for (int kh = 0; kh < NH; ++kh)
                    for (int nt = 0; nt < 2; ++nt)
                        for (int lane = 0; lane < 64; ++lane) {
                            const int j = lane & 15, kg = lane >> 4;
                            const int n = wv * 32 + nt * 16 + j;
                            const float* src = fc1->data() + size_t(n) * (T * 2 * H) + size_t(t) * 2 * H + d * H + kh * 32 + kg * 8;
                            const size_t u = ((((size_t(d) * T + t) * 4 + wv) * NH + kh) * 4 + nt * 2) * 64;
                            put_split8(&F[(u + lane) * 8], &F[(u + 64 + lane) * 8], src, 8, f16, f_scale);
                        }
    }
    rc = upload_halves(W, a, Wout);
    return rc != CTO_OK || !fc1 ? rc : upload_halves(F, a, Fout);
}

}  // namespace

extern "C" int cto_cvt_create(const cto_weights* w, const cto_cvt_cfg* cfg, cto_model** out) {
    return cto_cvt_create_ex(w, cfg, CTO_SPLIT_ENV, out);
}

extern "C" int cto_cvt_create_ex(const cto_weights* w, const cto_cvt_cfg* cfg, int split_mode, cto_model** out) {
    CTO_REQUIRE(w && cfg && out, CTO_EINVAL, "cto_cvt_create: null argument");
    CTO_REQUIRE(split_mode >= CTO_SPLIT_ENV && split_mode <= CTO_SPLIT_BF16, CTO_EINVAL, "cto_cvt_create_ex: split_mode %d", split_mode);
    CTO_REQUIRE(cfg->n_out == 4 || cfg->n_out == 6, CTO_EINVAL, "n_out must be 4 or 6");
    for (int i = 0; i < 3; ++i)
        CTO_REQUIRE(cfg->emb_dim[i] >= 4 && cfg->emb_dim[i] <= 128 && cfg->emb_dim[i] % 4 == 0 && cfg->heads[i] >= 1 &&
                        cfg->heads[i] <= 8 && cfg->depth[i] >= 1,
                    CTO_EUNSUPPORTED, "CvT stage %d config out of range (emb_dim <= 128, multiple of 4)", i + 1);
    std::unique_ptr<cto_model> m(new cto_model());
    CTO_HIP(hipGetDevice(&m->device));
    m->kind = 0;
    m->n_out = cfg->n_out;
    {   // experiment (side channel, never the default): the block GEMMs of the 64- / 128-channel stages on split 16-bit operands
        const char* e = split_mode == CTO_SPLIT_ENV ? getenv("CTO_CVT_SPLIT") : nullptr;
        if (e && e[0]) {
            const std::string kind(e);
            CTO_REQUIRE(kind == "f16" || kind == "bf16", CTO_EINVAL, "CTO_CVT_SPLIT must be f16 or bf16, not '%s'", e);
            m->cvt_split = kind == "f16" ? 1 : 2;
        }
        if (split_mode > 0) m->cvt_split = split_mode;
    }
    Arena& a = m->arena;
    int rc = CTO_OK;
    int cin = CTO_NCHAN, win = CTO_NPOS;
    int64_t macs = 0;
    auto fail = [&](int code) { return code; };
    for (int si = 0; si < 3; ++si) {
        StageDev& st = m->st[si];
        st.cin = cin; st.c = cfg->emb_dim[si]; st.win = win; st.w = (win + 1) / 2; st.wkv = (st.w + 1) / 2;
        st.heads = cfg->heads[si]; st.inner = 64 * st.heads;
        const int C = st.c;
        const std::string L = "layer" + std::to_string(si + 1);
        {
            GETW(cw, L + ".0.weight", int64_t(C) * cin * 9);
            std::vector<float> v(size_t(C) * 3 * cin);
            for (int n = 0; n < C; ++n)
                for (int t = 0; t < 3; ++t)
                    for (int ci = 0; ci < cin; ++ci)
                        v[(size_t(n) * 3 + t) * cin + ci] = (*cw)[((size_t(n) * cin + ci) * 3 + 1) * 3 + t];
            if ((rc = a.upload(v, &st.wemb))) return fail(rc);
            // the same taps re-laid for the in-block embedding: k = t * PS + ci, zero elsewhere
            const int PS = emb_ps(cin), KP = emb_kch(cin) * 16;
            std::vector<float> vp(size_t(C) * KP, 0.f);
            for (int n = 0; n < C; ++n)
                for (int t = 0; t < 3; ++t)
                    for (int ci = 0; ci < cin; ++ci) vp[size_t(n) * KP + t * PS + ci] = v[(size_t(n) * 3 + t) * cin + ci];
            if (C % 16 == 0) { if ((rc = a.upload(pack_fragments(vp.data(), C / 16, KP / 16), &st.wembp))) return fail(rc); }   // fragment order
        }
        if ((rc = upload_named(w, L + ".0.bias", C, a, &st.bemb))) return fail(rc);
        if ((rc = upload_named(w, L + ".1.g", C, a, &st.lng))) return fail(rc);
        if ((rc = upload_named(w, L + ".1.b", C, a, &st.lnb))) return fail(rc);
        macs += int64_t(st.w) * 3 * cin * C;
        for (int d = 0; d < cfg->depth[si]; ++d) {
            BlockDev b;
            const std::string P = L + ".2.layers." + std::to_string(d);
            if ((rc = upload_named(w, P + ".0.norm.g", C, a, &b.n0g)) || (rc = upload_named(w, P + ".0.norm.b", C, a, &b.n0b)) ||
                (rc = pack_dw(w, P + ".0.fn.to_q.net.0.weight", C, a, &b.dwq)) ||
                (rc = pack_bn(w, P + ".0.fn.to_q.net.1", C, a, &b.bnq)) ||
                (rc = upload_named(w, P + ".0.fn.to_q.net.2.weight", int64_t(st.inner) * C, a, &b.wq)) ||
                (rc = pack_dw(w, P + ".0.fn.to_kv.net.0.weight", C, a, &b.dwkv)) ||
                (rc = pack_bn(w, P + ".0.fn.to_kv.net.1", C, a, &b.bnkv)) ||
                (rc = upload_named(w, P + ".0.fn.to_kv.net.2.weight", int64_t(2) * st.inner * C, a, &b.wkv)) ||
                (rc = upload_named(w, P + ".0.fn.to_out.0.weight", int64_t(C) * st.inner, a, &b.wo)) ||
                (rc = upload_named(w, P + ".0.fn.to_out.0.bias", C, a, &b.bo)) ||
                (rc = upload_named(w, P + ".1.norm.g", C, a, &b.n1g)) || (rc = upload_named(w, P + ".1.norm.b", C, a, &b.n1b)) ||
                (rc = upload_named(w, P + ".1.fn.net.0.weight", int64_t(4) * C * C, a, &b.w1)) ||
                (rc = upload_named(w, P + ".1.fn.net.0.bias", 4 * C, a, &b.b1)) ||
                (rc = upload_named(w, P + ".1.fn.net.3.weight", int64_t(4) * C * C, a, &b.w2)) ||
                (rc = upload_named(w, P + ".1.fn.net.3.bias", C, a, &b.b2)))
                return fail(rc);
            const int NI = st.inner / 16, NC16 = C / 16;
            if (C % 16 == 0 &&      // the fused block kernel exists for 16 / 32 / 64 / 128 channels; other widths run the unfused path
                ((rc = upload_fragments(w, P + ".0.fn.to_q.net.2.weight", NI, NC16, a, &b.wq_f)) ||
                (rc = upload_fragments(w, P + ".0.fn.to_kv.net.2.weight", 2 * NI, NC16, a, &b.wkv_f)) ||
                (rc = upload_fragments(w, P + ".0.fn.to_out.0.weight", NC16, NI, a, &b.wo_f)) ||
                (rc = upload_fragments(w, P + ".1.fn.net.0.weight", 4 * NC16, NC16, a, &b.w1_f)) ||
                 (rc = upload_fragments(w, P + ".1.fn.net.3.weight", NC16, 4 * NC16, a, &b.w2_f))))
                return fail(rc);
            if (m->cvt_split && C % 64 == 0) {
                const bool f16 = m->cvt_split == 1;
                if ((rc = upload_split_fragments(w, P + ".0.fn.to_q.net.2.weight", NI, C / 32, f16, a, &b.wq_s)) ||
                    (rc = upload_split_fragments(w, P + ".0.fn.to_kv.net.2.weight", 2 * NI, C / 32, f16, a, &b.wkv_s)) ||
                    (rc = upload_split_fragments(w, P + ".0.fn.to_out.0.weight", NC16, st.inner / 32, f16, a, &b.wo_s)) ||
                    (rc = upload_split_fragments(w, P + ".1.fn.net.0.weight", 4 * NC16, C / 32, f16, a, &b.w1_s)) ||
                    (rc = upload_split_fragments(w, P + ".1.fn.net.3.weight", NC16, 4 * C / 32, f16, a, &b.w2_s)))
                    return fail(rc);
            }
            st.blocks.push_back(b);
            macs += int64_t(st.w) * C * st.inner + int64_t(st.wkv) * C * 2 * st.inner + int64_t(st.w) * st.inner * C +
                    int64_t(st.w) * 8 * C * C + int64_t(2) * st.heads * st.w * st.wkv * 64 +
                    int64_t(3) * C * (st.w + st.wkv);
        }
        cin = C;
        win = st.w;
    }
    // fc1: torch flattens [C][1][W] as c*W + w; our activations are channels-last (w*C + c)
    const int C3 = m->st[2].c, W3 = m->st[2].w, k1 = C3 * W3;
    {
        GETW(f1, "fc1.weight", int64_t(128) * k1);
        std::vector<float> v(size_t(128) * k1);
        for (int n = 0; n < 128; ++n)
            for (int c = 0; c < C3; ++c)
                for (int ww = 0; ww < W3; ++ww) v[size_t(n) * k1 + ww * C3 + c] = (*f1)[size_t(n) * k1 + c * W3 + ww];
        static const char* const names[6] = {"a", "c", "g", "t", "i", "d"};
        if ((rc = build_head(w, names, m->n_out, k1, v, a, m->head))) return fail(rc);
        if (C3 == 128 && W3 == 5) {      // fc1 over the LDS image of the last block's tile: k = w * RS + c
            using G = CvtBlockGeom<128, 5, 3, 16>;
            const int KP = G::KCH1 * 16;
            std::vector<float> vp(size_t(128) * KP, 0.f);
            for (int n = 0; n < 128; ++n)
                for (int ww = 0; ww < W3; ++ww)
                    for (int c = 0; c < C3; ++c) vp[size_t(n) * KP + ww * G::RS + c] = v[size_t(n) * k1 + ww * C3 + c];
            if ((rc = a.upload(pack_fragments(vp.data(), 8, G::KCH1), &m->head.w1p))) return fail(rc);     // fragment order
        }
    }
    macs += int64_t(k1) * 128 + int64_t(m->n_out) * (128 * 128 + 256);
    m->macs = macs;
    if (const char* e = getenv("CTO_CVT_UNFUSED")) m->fuse_blocks = !(e[0] == '1');
    if (const char* e = getenv("CTO_CVT_NO_EMBED_FUSE")) m->fuse_embed = !(e[0] == '1');
    if (const char* e = getenv("CTO_CVT_NO_HEAD_FUSE")) m->fuse_head = !(e[0] == '1');
    *out = m.release();
    return CTO_OK;
}

extern "C" int cto_bigru_create(const cto_weights* w, int n_out, cto_model** out) {
    return cto_bigru_create_ex(w, n_out, CTO_SPLIT_ENV, out);
}

extern "C" int cto_bigru_create_ex(const cto_weights* w, int n_out, int split_mode, cto_model** out) {
    CTO_REQUIRE(w && out, CTO_EINVAL, "cto_bigru_create: null argument");
    CTO_REQUIRE(split_mode >= CTO_SPLIT_ENV && split_mode <= CTO_SPLIT_BF16, CTO_EINVAL, "cto_bigru_create_ex: split_mode %d", split_mode);
    CTO_REQUIRE(n_out == 4 || n_out == 6, CTO_EINVAL, "n_out must be 4 or 6");
    std::unique_ptr<cto_model> m(new cto_model());
    CTO_HIP(hipGetDevice(&m->device));
    m->kind = 1;
    m->n_out = n_out;
    int rc = CTO_OK;
    auto fail = [&](int code) { return code; };
    if ((rc = pack_gru(w, "lstm", 34, 48, 128, m->arena, &m->gw1, &m->gb1, &m->gw1f))) return fail(rc);
    if ((rc = pack_gru(w, "lstm_2", 256, 256, 192, m->arena, &m->gw2, &m->gb2, &m->gw2f))) return fail(rc);
    const int k1 = 33 * 384;
    {
        GETW(f1, "fc1.weight", int64_t(128) * k1);
        static const char* const names[6] = {"na", "nc", "ng", "nt", "ni", "nd"};
        if ((rc = build_head(w, names, n_out, k1, *f1, m->arena, m->head))) return fail(rc);
        if ((rc = m->arena.upload(pack_fc1_fragments(*f1, 192), &m->f1f))) return fail(rc);
        // experiment (side channel, never the default): layer 2 + fc1 on split 16-bit operands, three f16 / bf16 MFMA passes per product
        const char* e = split_mode == CTO_SPLIT_ENV ? getenv("CTO_GRU_SPLIT") : split_mode == CTO_SPLIT_F16 ? "f16" : split_mode == CTO_SPLIT_BF16 ? "bf16" : nullptr;
        if (e && e[0]) {
            const std::string kind(e);
            CTO_REQUIRE(kind == "f16" || kind == "bf16", CTO_EINVAL, "CTO_GRU_SPLIT must be f16 or bf16, not '%s'", e);
            m->split_f16 = kind == "f16";
            if ((rc = pack_gru_split(w, "lstm_2", 256, 256, 192, f1, m->split_f16, 14, m->arena, &m->gw2_split, &m->f1_split, m->split_sc2))) return fail(rc);
            // CTO_GRU_SPLIT_LAYERS=2 keeps layer 1 on the fp32 kernel (default: both recurrent layers on split operands)
            const char* l = getenv("CTO_GRU_SPLIT_LAYERS");
            if (!(l && l[0] == '2') && (rc = pack_gru_split(w, "lstm", 34, 64, 128, nullptr, m->split_f16, 0, m->arena, &m->gw1_split, nullptr, m->split_sc1)))
                return fail(rc);
        }
    }
    m->macs = int64_t(33) * 2 * 3 * 128 * (34 + 128) + int64_t(33) * 2 * 3 * 192 * (256 + 192) + int64_t(k1) * 128 +
              int64_t(n_out) * (128 * 128 + 256);
    *out = m.release();
    return CTO_OK;
}

extern "C" int cto_model_forward(cto_model* m, const float* x, int64_t B, float* logits, void* stream) {
    CTO_REQUIRE(m && x && logits && B >= 0, CTO_EINVAL, "cto_model_forward: bad argument");
    CTO_REQUIRE(B * 33 < (int64_t(1) << 31) / 640, CTO_EUNSUPPORTED, "batch too large for 32-bit row indices; split it");
    if (B == 0) return CTO_OK;
    {
        int dev = -1;
        CTO_HIP(hipGetDevice(&dev));
        CTO_REQUIRE(dev == m->device, CTO_EINVAL, "cto_model_forward: model lives on device %d but device %d is current (one process per GPU)",
                    m->device, dev);
    }
    int rc = ensure_ws(m, B);
    if (rc != CTO_OK) return rc;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (m->kind == 1) return bigru_forward(m, x, B, logits, s);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (m->prof && (rc = prof_begin(m, s, &e0, &e1))) return rc;
    rc = cvt_forward(m, x, B, logits, s);
    if (m->prof && rc == CTO_OK) {
        CTO_HIP(hipEventRecord(e1, s));
        m->prof_ev.emplace_back(e0, e1);
    }
    return rc;
}

namespace {
// the fp32 tensor of the int16 one, for handles whose first layer has no int16 loader (split operands, the plain GRU schedule, an
// embedding outside the block): float(double(v) * scale), the tensor kernel's own expression
__global__ __launch_bounds__(256) void k_expand_raw(const int16_t* __restrict__ raw, const int32_t* __restrict__ site_info, int which, int cov,
                                                    int64_t n, float* __restrict__ out) {
    const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int depth = site_info[(i / (33 * 34)) * 12 + 1 + which];
    const double sc = (cov > 0 && depth > cov) ? double(cov) / double(depth) : 1.0;
    out[i] = float(double(int(raw[i])) * sc);
}
}  // namespace

extern "C" int cto_model_forward_raw(cto_model* m, const int16_t* x_raw, const int32_t* site_info, int which, int min_rescale_cov, int64_t B,
                                     float* logits, void* stream) {
    CTO_REQUIRE(m && x_raw && site_info && logits && B >= 0 && (which == 0 || which == 1), CTO_EINVAL, "cto_model_forward_raw: bad argument");
    CTO_REQUIRE(B * 33 < (int64_t(1) << 31) / 640, CTO_EUNSUPPORTED, "batch too large for 32-bit row indices; split it");
    if (B == 0) return CTO_OK;
    {
        int dev = -1;
        CTO_HIP(hipGetDevice(&dev));
        CTO_REQUIRE(dev == m->device, CTO_EINVAL, "cto_model_forward_raw: model lives on device %d but device %d is current (one process per GPU)",
                    m->device, dev);
    }
    int rc = ensure_ws(m, B);
    if (rc != CTO_OK) return rc;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const bool native = m->kind == 1 ? (!m->gw1_split && gru_layer1_takes_raw())
                                     : (m->fuse_blocks && m->fuse_embed && can_fuse_embed(m->st[0]));
    if (!native) {
        if (m->xexp_B < B) {
            if (m->b_xexp) (void)hipFree(m->b_xexp);
            m->b_xexp = nullptr; m->xexp_B = 0;
            CTO_HIP(hipMalloc(reinterpret_cast<void**>(&m->b_xexp), size_t(B) * 33 * 34 * sizeof(float)));
            m->xexp_B = B;
        }
        const int64_t n = B * 33 * 34;
        hipLaunchKernelGGL(k_expand_raw, dim3(unsigned(cdiv(n, 256))), dim3(256), 0, s, x_raw, site_info, which, min_rescale_cov, n, m->b_xexp);
        CTO_HIP(hipGetLastError());
        return cto_model_forward(m, m->b_xexp, B, logits, stream);
    }
    const RawIn raw{x_raw, site_info, which, min_rescale_cov};
    if (m->kind == 1) return bigru_forward(m, nullptr, B, logits, s, &raw);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (m->prof && (rc = prof_begin(m, s, &e0, &e1))) return rc;
    rc = cvt_forward(m, nullptr, B, logits, s, &raw);
    if (m->prof && rc == CTO_OK) {
        CTO_HIP(hipEventRecord(e1, s));
        m->prof_ev.emplace_back(e0, e1);
    }
    return rc;
}

extern "C" int cto_model_profile(cto_model* m, int enable) {
    CTO_REQUIRE(m, CTO_EINVAL, "cto_model_profile: null model");
    m->prof = enable != 0;
    m->prof_all = enable == 2;
    // per-site MACs of the measured kernel: BiGRU layer 2 (both directions) incl. the fused fc1, or the whole CvT
    m->prof_macs = m->kind == 1 ? int64_t(33) * 2 * 3 * 192 * (256 + 192) + int64_t(33) * 384 * 128 : m->macs;
    return CTO_OK;
}

extern "C" int cto_model_profile_read(cto_model* m, double* mean_ms, int64_t* macs_per_site);
static int profile_drain(std::vector<std::pair<hipEvent_t, hipEvent_t>>& evs, double* mean_ms) {
    double sum = 0.0;
    int n = 0;
    for (auto& e : evs) {
        CTO_HIP(hipEventSynchronize(e.second));
        float ms = 0.f;
        CTO_HIP(hipEventElapsedTime(&ms, e.first, e.second));
        sum += ms;
        ++n;
        (void)hipEventDestroy(e.first);
        (void)hipEventDestroy(e.second);
    }
    evs.clear();
    *mean_ms = n ? sum / n : 0.0;
    return n;
}

extern "C" int cto_model_profile_read_stage(cto_model* m, int stage, double* mean_ms, int64_t* macs_per_site) {
    CTO_REQUIRE(m && mean_ms && macs_per_site, CTO_EINVAL, "cto_model_profile_read_stage: null argument");
    if (stage == 0) return cto_model_profile_read(m, mean_ms, macs_per_site);
    CTO_REQUIRE(stage == 1 && m->kind == 1, CTO_EINVAL, "cto_model_profile_read_stage: stage 1 exists for the BiGRU only");
    *macs_per_site = int64_t(33) * 2 * 3 * 128 * (34 + 128);      // layer 1, both directions
    return profile_drain(m->prof_ev1, mean_ms);
}

extern "C" int cto_model_profile_read(cto_model* m, double* mean_ms, int64_t* macs_per_site) {
    CTO_REQUIRE(m && mean_ms && macs_per_site, CTO_EINVAL, "cto_model_profile_read: null argument");
    *macs_per_site = m->prof_macs;
    return profile_drain(m->prof_ev, mean_ms);
}

extern "C" int64_t cto_model_macs_per_site(const cto_model* m) { return m ? m->macs : 0; }
extern "C" int cto_model_n_out(const cto_model* m) { return m ? m->n_out : 0; }
extern "C" void cto_model_destroy(cto_model* m) { delete m; }

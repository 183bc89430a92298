// Posterior / decision / quality epilogue (clairs/call_variants.py of the reference):
//   2-way softmax            clairs/predict.py:659-684 (torch Softmax(dim=1), fp32)
//   "{:0.8f}" text round trip predict.py:121-132 -> call_variants.py:803-829 (the posterior is computed
//                            on the 8-decimal values; p * 1e8 is exact in double for an fp32 p in [0,1],
//                            so rint(p * 1e8) / 1e8 reproduces format() + float() bit for bit)
//   bin lookup + Bayes       call_variants.py:181-209 / 246-288 (np.digitize, right-open bins)
//   arg-max                  call_variants.py:213-214 / 292-293 (first maximum)
//   QUAL                     call_variants.py:79-88
// All of it is fp64 with contraction off so the operation order equals the Python expression's.
#include "common.h"

namespace {

#pragma clang fp contract(off)

__device__ inline double round8(float p) { return rint(double(p) * 1e8) / 1e8; }

__device__ inline float softmax_p1(const float* o) {
    const float m = fmaxf(o[0], o[1]);
    const float e0 = expf(o[0] - m), e1 = expf(o[1] - m);
    return e1 / (e0 + e1);
}
__device__ inline float softmax_p0(const float* o) {
    const float m = fmaxf(o[0], o[1]);
    const float e0 = expf(o[0] - m), e1 = expf(o[1] - m);
    return e0 / (e0 + e1);
}

__device__ inline int digitize(double x, const double* edges) {   // np.digitize(x, edges) - 1, 11 increasing edges
    int c = 0;
#pragma unroll
    for (int i = 0; i < 11; ++i) c += (edges[i] <= x) ? 1 : 0;
    return c - 1;
}

// Python's round(q, 4) (call_variants.py:88): the EXACT binary value of q rounded to 4 decimals, ties to even, then the
// double nearest to that decimal.  rint(q * 1e4) alone rounds the already-rounded product and can land on the other side of
// a ...5 boundary; the product's rounding error is recovered exactly with one fma (q * 1e4 = hi + lo as real numbers).
//
// *near_tie: q * 1e4 lies within 1e-6 of a ...5 boundary.  The device's log() may differ from the host libm's in its last bit
// (|dq| ~ 1e-14), which can only change round(q, 4) for such a q: those sites (about two in a million) are flagged in
// decision[.][1] bit 2 and carry the winning posterior's bits in decision[.][2..3], and the host re-evaluates them with ITS
// libm - the one the reference's math.log would use on this machine (cto_qual_finalize, cto_vcf_rows_batch).
__device__ __forceinline__ double round4(double q, bool* near_tie) {
    *near_tie = false;
    if (!(q == q)) return q;
    const double hi = q * 1e4, lo = fma(q, 1e4, -hi);
    double r = rint(hi);
    const double d = (hi - r) + lo;                 // exact: |hi - r| <= 0.5 and lo is tiny
    *near_tie = fabs(fabs(d) - 0.5) < 1e-6;
    const bool odd = fmod(r, 2.0) != 0.0;
    if (d > 0.5 || (d == 0.5 && odd)) r += 1.0;
    else if (d < -0.5 || (d == -0.5 && odd)) r -= 1.0;
    return r / 1e4;
}

template <bool FROM_PROBS>
__global__ __launch_bounds__(256) void k_posterior(const float* __restrict__ aff, const float* __restrict__ neg,
                                                   const double* __restrict__ p1, int K,
                                                   int64_t B, const double* __restrict__ lik,
                                                   const double* __restrict__ edges, float* __restrict__ probs,
                                                   double* __restrict__ post, int32_t* __restrict__ decision,
                                                   double* __restrict__ qual) {
    const int64_t b = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (b >= B) return;
    int best = 0, clamped = 0;
    double bestv = 0.0;
    for (int k = 0; k < K; ++k) {
        double pa, pn;
        if constexpr (FROM_PROBS) {
            pa = p1[b * (2 * K) + k];
            pn = p1[b * (2 * K) + K + k];
        } else {
            const float* oa = aff + (int64_t(k) * B + b) * 2;
            const float* on = neg + (int64_t(k) * B + b) * 2;
            const float pa_f = softmax_p1(oa), pn_f = softmax_p1(on);
            if (probs) {
                float* pr = probs + b * (4 * K);
                pr[2 * k + 0] = softmax_p0(oa);
                pr[2 * k + 1] = pa_f;
                pr[2 * (K + k) + 0] = softmax_p0(on);
                pr[2 * (K + k) + 1] = pn_f;
            }
            pa = round8(pa_f);
            pn = round8(pn_f);
        }
        int i = digitize(pa, edges + (2 * k) * 11);
        int j = digitize(1 - pn, edges + (2 * k + 1) * 11);
        if (i < 0 || i > 9 || j < 0 || j > 9) {   // the reference raises IndexError here; clamp and flag
            clamped = 1;
            i = i < 0 ? 0 : (i > 9 ? 9 : i);
            j = j < 0 ? 0 : (j > 9 ? 9 : j);
        }
        const double w = lik[(k * 10 + i) * 10 + j] + 2.220446049250313e-16;
        const double num = pa * (1 - pn) * w;
        const double v = num / (num + ((1 - pa) * pn * (1 - w)));
        post[b * K + k] = v;
        // np.argmax (call_variants.py:213 / 292): first maximum, and a NaN (0/0: both heads saturated at 0.00000000) beats
        // every number - the first NaN wins
        if (k == 0 || (!(bestv != bestv) && (v > bestv || v != v))) { bestv = v; best = k; }
    }
    decision[b * 4 + 0] = best;
    // bit 0: a bin index was clamped (the reference raises IndexError on this site); bit 1: the winning posterior is NaN
    // (only possible together with bit 0) - the host must not format a row from it
    decision[b * 4 + 1] = clamped | ((bestv != bestv) ? 2 : 0);
    const double phred = -10.0 * (1.0 / 2.302585092994046);   // -10 * log(e, 10)
    double q = phred * log(((1.0 - bestv) + 1e-10) / (bestv + 1e-10)) + 2.0;
    q = q > 0.0 ? q : 0.0;
    bool near_tie;
    qual[b] = round4(q, &near_tie);
    const long long bits = near_tie ? __double_as_longlong(bestv) : 0ll;
    if (near_tie) decision[b * 4 + 1] |= 4;
    decision[b * 4 + 2] = int32_t(bits & 0xffffffffll);
    decision[b * 4 + 3] = int32_t((bits >> 32) & 0xffffffffll);
}

__global__ __launch_bounds__(256) void k_softmax_probs(const float* __restrict__ aff, const float* __restrict__ neg, int K,
                                                       int64_t B, float* __restrict__ probs) {
    const int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x;   // one (site, head) pair per thread
    if (i >= B * 2 * K) return;
    const int64_t b = i / (2 * K);
    const int k = int(i - b * 2 * K);
    const float* o = (k < K ? aff + (int64_t(k) * B + b) * 2 : neg + (int64_t(k - K) * B + b) * 2);
    probs[i * 2 + 0] = softmax_p0(o);
    probs[i * 2 + 1] = softmax_p1(o);
}

// 2-way softmax of n rows [n][2] (the modules' own `apply_softmax`, clairs/model.py:255-259 / 461-465): the arithmetic of k_softmax_probs
__global__ void k_softmax_pairs(const float* __restrict__ logits, int64_t n, float* __restrict__ out) {
    const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float o[2] = {logits[i * 2], logits[i * 2 + 1]};
    out[i * 2 + 0] = softmax_p0(o);
    out[i * 2 + 1] = softmax_p1(o);
}

// number of sites whose QUAL waits for the host half (decision[.][1] bit 2), added to *count
__global__ void k_qual_pending(const int32_t* __restrict__ decision, int64_t B, int32_t* __restrict__ count) {
    const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    const bool flagged = i < B && (decision[i * 4 + 1] & 4) != 0;
    const unsigned long long m = __ballot(flagged);
    if ((threadIdx.x & 63) == 0 && m) atomicAdd(count, int32_t(__popcll(m)));
}

}  // namespace

extern "C" int cto_softmax_pairs(const float* logits, int64_t n, float* out, void* stream) {
    CTO_REQUIRE(n >= 0 && (n == 0 || (logits && out)), CTO_EINVAL, "cto_softmax_pairs: bad argument");
    if (n == 0) return CTO_OK;
    hipLaunchKernelGGL(k_softmax_pairs, dim3(unsigned(cto::cdiv(n, 256))), dim3(256), 0, static_cast<hipStream_t>(stream), logits, n, out);
    CTO_HIP(hipGetLastError());
    return CTO_OK;
}

extern "C" int cto_qual_pending(const int32_t* decision, int64_t B, int32_t* count, void* stream) {
    CTO_REQUIRE(decision && count && B >= 0, CTO_EINVAL, "cto_qual_pending: bad argument");
    CTO_HIP(hipMemsetAsync(count, 0, sizeof(int32_t), static_cast<hipStream_t>(stream)));
    if (B == 0) return CTO_OK;
    hipLaunchKernelGGL(k_qual_pending, dim3(unsigned(cto::cdiv(B, 256))), dim3(256), 0, static_cast<hipStream_t>(stream), decision, B, count);
    CTO_HIP(hipGetLastError());
    return CTO_OK;
}

extern "C" int cto_softmax_probs(const float* aff_logits, const float* neg_logits, int K, int64_t B, float* probs, void* stream) {
    CTO_REQUIRE(aff_logits && neg_logits && probs, CTO_EINVAL, "cto_softmax_probs: null argument");
    CTO_REQUIRE(K == 4 || K == 6, CTO_EINVAL, "K must be 4 or 6");
    if (B == 0) return CTO_OK;
    hipLaunchKernelGGL(k_softmax_probs, dim3(unsigned(cto::cdiv(B * 2 * K, 256))), dim3(256), 0,
                       static_cast<hipStream_t>(stream), aff_logits, neg_logits, K, B, probs);
    CTO_HIP(hipGetLastError());
    return CTO_OK;
}

extern "C" int cto_posterior(const float* aff_logits, const float* neg_logits, int K, int64_t B, const double* lik,
                             const double* edges, float* probs, double* post, int32_t* decision, double* qual,
                             void* stream) {
    CTO_REQUIRE(aff_logits && neg_logits && lik && edges && post && decision && qual, CTO_EINVAL,
                "cto_posterior: null argument");
    CTO_REQUIRE(K == 4 || K == 6, CTO_EINVAL, "K must be 4 or 6");
    if (B == 0) return CTO_OK;
    hipLaunchKernelGGL(k_posterior<false>, dim3(unsigned(cto::cdiv(B, 256))), dim3(256), 0, static_cast<hipStream_t>(stream),
                       aff_logits, neg_logits, static_cast<const double*>(nullptr), K, B, lik, edges, probs, post, decision, qual);
    CTO_HIP(hipGetLastError());
    return CTO_OK;
}

extern "C" int cto_posterior_from_probs(const double* p1, int K, int64_t B, const double* lik, const double* edges,
                                        double* post, int32_t* decision, double* qual, void* stream) {
    CTO_REQUIRE(p1 && lik && edges && post && decision && qual, CTO_EINVAL, "cto_posterior_from_probs: null argument");
    CTO_REQUIRE(K == 4 || K == 6, CTO_EINVAL, "K must be 4 or 6");
    if (B == 0) return CTO_OK;
    hipLaunchKernelGGL(k_posterior<true>, dim3(unsigned(cto::cdiv(B, 256))), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<const float*>(nullptr), static_cast<const float*>(nullptr), p1, K, B, lik, edges,
                       static_cast<float*>(nullptr), post, decision, qual);
    CTO_HIP(hipGetLastError());
    return CTO_OK;
}

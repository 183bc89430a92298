// MFMA / activation primitives shared by the network kernels (gfx950 / CDNA4).
//
// All dense contractions run on the exact-fp32 matrix cores (v_mfma_f32_16x16x4_f32: fp32 in, fp32
// accumulate, bit-equal to an fmaf chain), because the parity bar is 1e-4 on probabilities and bf16/fp16
// inputs miss it.  Wave = 64 lanes; one MFMA computes a 16x16 tile over k = 4:
//     lane l supplies A[i = l & 15][k = l >> 4] and B[k = l >> 4][j = l & 15],
//     lane l receives  D[row = 4 * (l >> 4) + r][col = l & 15] in register r = 0..3.
// Every weight matrix is consumed in PyTorch's native Linear layout W[n][k] (k contiguous), so
// C[m][n] = sum_k A[m][k] * W[n][k] needs no transposition: lane l reads W[n0 + (l & 15)][k0 + (l >> 4)].
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace cto {

using f32x4 = __attribute__((ext_vector_type(4))) float;

// Row padding (floats) of the packed GRU weight matrices [3H][KP + H + pad]: see gru_kernel.h
#ifndef CTO_GRU_WPAD
#define CTO_GRU_WPAD 32
#endif
constexpr int GRU_WPAD = CTO_GRU_WPAD;

// Workgroup barrier for LDS-only hand-offs.  `__syncthreads()` is a workgroup-scope release/acquire fence: it drains EVERY
// outstanding memory operation (s_waitcnt vmcnt(0) lgkmcnt(0)) before s_barrier, so weight fragments that were requested to
// "fly under the barrier" are waited for right there - one exposed L2 round trip per barrier.  Waves of these kernels only talk
// through LDS, for which completing the LDS operations (lgkmcnt) is sufficient; register-destined global loads stay in flight.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// 16-byte weight load with the global address space spelled out.  Pointers that reach a loop through arrays / lambdas can lose
// it, and the compiler then emits flat_load, which counts on vmcnt AND lgkmcnt: every s_waitcnt for an LDS read then also waits
// for the weight prefetches in flight (seen in one k_cvt_block instantiation: lgkmcnt(0) before every chunk's MFMAs).
__device__ __forceinline__ float4 ldg4(const float* p) {
    typedef const __attribute__((address_space(1))) f32x4* gptr;
    const f32x4 v = *reinterpret_cast<gptr>(reinterpret_cast<uintptr_t>(p));
    return make_float4(v[0], v[1], v[2], v[3]);
}

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// Sum over the 16 lanes of a DPP row, result in every lane: four v_add_f32 with a DPP operand (quad swaps, half-row mirror, row
// mirror).  `__shfl_xor(v, o, 16)` compiles to ds_bpermute_b32 - a trip through the LDS crossbar and an s_waitcnt per step.
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float row16_sum(float v) {
    v += dpp_mov<0xB1>(v);     // quad_perm [1,0,3,2]
    v += dpp_mov<0x4E>(v);     // quad_perm [2,3,0,1]
    v += dpp_mov<0x141>(v);    // row_half_mirror: the other quad of the half row
    v += dpp_mov<0x140>(v);    // row_mirror: the other half row
    return v;
}

// ---- activations (clairs/model.py: nn.SELU, nn.GELU() exact-erf form) ----
__device__ __forceinline__ float selu_f(float x) {
    const float scale = 1.0507009873554804934193349852946f;
    const float alpha = 1.6732632423543772848170429916717f;
    return x > 0.f ? scale * x : scale * alpha * expm1f(x);
}
// branch-free SELU for MFMA epilogues: exp(x) - 1 by the hardware exponential; the cancellation near 0 costs at most one
// ulp of 1.0 (6e-8 absolute), far inside the 1e-4 parity bar, and there is no libm expm1f call (branches, ~50 VALU) per element
__device__ __forceinline__ float selu_fast(float x) {
#ifdef CTO_PRECISE_MATH     // tools/ab builds only: libm everywhere, to price what the fast forms cost in accuracy (range sweep)
    return selu_f(x);
#endif
    const float scale = 1.0507009873554804934193349852946f;
    const float alpha = 1.6732632423543772848170429916717f;
    const float neg = scale * alpha * (__expf(fminf(x, 0.f)) - 1.0f);
    return x > 0.f ? scale * x : neg;
}
// exact-erf GELU (nn.GELU() default).  erf by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7, i.e. fp32 round-off class):
// libm's erff costs ~45 VALU per call and the FFN epilogues evaluate it 160 times per lane per transformer block.
__device__ __forceinline__ float erf_as(float x) {
#ifdef CTO_PRECISE_MATH
    return erff(x);
#endif
    const float ax = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    const float r = 1.0f - p * t * __expf(-ax * ax);
    return copysignf(r, x);
}
__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.0f + erf_as(x * 0.70710678118654752440f)); }
// 1 / x where the kernels accept the hardware reciprocal (1 ulp)
__device__ __forceinline__ float rcp_fast(float x) {
#ifdef CTO_PRECISE_MATH
    return 1.0f / x;
#else
    return __builtin_amdgcn_rcpf(x);
#endif
}
__device__ __forceinline__ float sigmoid_f(float x) { return 1.0f / (1.0f + expf(-x)); }
// exp(x) for x <= 0 (softmax numerators) on the hardware exp2: the product x * log2(e) is carried in two floats (FMA residual +
// the constant's low part), so the result is within ~2 ulp like libm's expf, at 7 VALU instead of ~25 (no range / denormal
// branches: below 2^-126 the hardware flushes to 0, which is what a softmax wants)
__device__ __forceinline__ float exp_le0(float x) {
#ifdef CTO_PRECISE_MATH
    return expf(x);
#endif
    const float L_HI = 1.44269502162933349609375f, L_LO = 1.925963033500011e-8f;
    const float t = x * L_HI;
    const float e = fmaf(x, L_LO, fmaf(x, L_HI, -t));          // x * log2(e) - t
    const float r = __builtin_amdgcn_exp2f(t);
    return fmaf(r, e * 0.693147180559945309417f, r);            // 2^(t + e) = 2^t (1 + e ln 2 + ...)
}


}  // namespace cto

// Split-operand MFMA primitives (EXPERIMENT, side channel: gru_split_kernel.h, the split GEMMs of cvt_gemm.h).
// a = hi + lo with hi, lo 16-bit floats; a·b ≈ hi·hi + hi·lo + lo·hi on v_mfma_f32_16x16x32_{f16,bf16}, fp32 accumulation.
//     f16  : hi = RTZ(a) (11 bits; a - hi is then exact), lo = RNE(a - hi) (11 bits): 22 significant bits.  Needs |a| < 65504;
//            small operands go subnormal (absolute error below 2^-25) and the f16 MFMA does not flush subnormal inputs (probed).
//     bf16 : hi, lo = RNE: 16-17 significant bits, no range condition.
// Fragment of a 16x16x32 product: lane l = (kg << 4) | j holds row / column j, k = 8 kg .. 8 kg + 7 (16 bytes).
#pragma once
#include "mfma_common.h"

namespace cto {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));

// D[row 4 kg + r][col j] += sum_k A[row j][k] B[col j][k]   (a: the lane's A fragment, b: its B fragment)
template <bool F16>
__device__ __forceinline__ f32x4 mfma_split(const uint4& a, const uint4& b, f32x4 c) {
    if constexpr (F16)
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}

// (a, b) -> packed 16-bit pairs hi, lo with a ≈ hi.x + lo.x, b ≈ hi.y + lo.y
template <bool F16>
__device__ __forceinline__ void split_pair(float a, float b, unsigned& hi, unsigned& lo) {
    if constexpr (F16) {
        const auto h = __builtin_amdgcn_cvt_pkrtz(a, b);          // truncation: a - hi is exact and lo takes it up
        hi = __builtin_bit_cast(unsigned, h);
        const f16x2_t l = {(_Float16)(a - float(h[0])), (_Float16)(b - float(h[1]))};
        lo = __builtin_bit_cast(unsigned, l);
    } else {
        const bf16x2_t h = {(__bf16)a, (__bf16)b};
        hi = __builtin_bit_cast(unsigned, h);
        const float ra = a - __uint_as_float(hi << 16), rb = b - __uint_as_float(hi & 0xffff0000u);
        const bf16x2_t l = {(__bf16)ra, (__bf16)rb};
        lo = __builtin_bit_cast(unsigned, l);
    }
}

// An activation tile row that feeds a split GEMM keeps its fp32 row pitch and holds [hi: K 16-bit values][lo: K 16-bit values]:
// four consecutive channels / one channel of `row` (the row's first float), K = the row's channel count.
template <bool F16>
__device__ __forceinline__ void put_split4(float* row, int K, int c, float x, float y, float z, float w) {
    uint2 hi, lo;
    split_pair<F16>(x, y, hi.x, lo.x);
    split_pair<F16>(z, w, hi.y, lo.y);
    unsigned short* r = reinterpret_cast<unsigned short*>(row);
    *reinterpret_cast<uint2*>(r + c) = hi;
    *reinterpret_cast<uint2*>(r + K + c) = lo;
}
template <bool F16>
__device__ __forceinline__ void put_split1(float* row, int K, int c, float x) {
    unsigned hi, lo;
    split_pair<F16>(x, 0.f, hi, lo);
    unsigned short* r = reinterpret_cast<unsigned short*>(row);
    r[c] = static_cast<unsigned short>(hi);
    r[K + c] = static_cast<unsigned short>(lo);
}

// 16-byte load of a 16-bit weight fragment with the global address space spelled out (see ldg4)
__device__ __forceinline__ uint4 ldg16(const unsigned short* p) {
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    typedef const __attribute__((address_space(1))) u32x4* gptr;
    const u32x4 v = *reinterpret_cast<gptr>(reinterpret_cast<uintptr_t>(p));
    return make_uint4(v[0], v[1], v[2], v[3]);
}

}  // namespace cto

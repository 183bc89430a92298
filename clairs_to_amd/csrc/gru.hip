// BiGRU recurrent kernels (clairs/model.py:412-417, 442-448) - instantiations and launchers.
// Kept in a translation unit of their own so that edits to the CvT kernels cannot perturb their code generation.
#include <stdlib.h>
#include "common.h"
#include "gru_kernel.h"
#include "gru_split_kernel.h"

using namespace cto;

namespace {

#ifndef CTO_GRU_NW1
#define CTO_GRU_NW1 4      // waves per workgroup of layer 1.  8 (two per SIMD, half the columns each) measured slower on MI355X:
                           // layer 1 + tail 0.301 -> 0.328 ms at 4096 sites, 1.159 -> 1.260 ms at 16384 (profiles/round4_gru_l1_eight_waves.txt)
#endif

// W / fc1w: row-major (the plain schedule, CTO_GRU_ROT=0); Wf / fc1f: the same weights in fragment order (the rotated schedule)
template <int KIN, int KP, int H, int MS, bool FUSE>
int launch_gru_range(hipStream_t s, const float* x, const float* W, const float* Wf, const float* bias, float* out, const float* fc1w,
                     const float* fc1f, float* fc1_part, int64_t B, int64_t begin, int64_t end, const XRawArgs* raw = nullptr) {
    if (end <= begin) return CTO_OK;
    const size_t smem = size_t(2) * MS * 16 * ((H + 4) + (KP + 4)) * sizeof(float);   // h tiles + x tiles
    constexpr int NW = FUSE ? 4 : CTO_GRU_NW1;
    // rotated schedule (gate arithmetic under the next step's x-part MFMAs); CTO_GRU_ROT=0 selects the plain one
    static const bool rot = [] { const char* e = getenv("CTO_GRU_ROT"); return !(e && e[0] == '0'); }();
    static bool attr_set = false;
    if (!attr_set) {
        CTO_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gru_layer<KIN, KP, H, MS, 1, FUSE>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, int(smem)));
        CTO_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gru_layer_rot<KIN, KP, H, MS, FUSE, NW>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, int(smem)));
        attr_set = true;
    }
    const unsigned grid = unsigned(cdiv(end - begin, MS * 16)) * 2;
    if constexpr (!FUSE && KIN == 34) {
        if (raw) {       // layer 1 on the int16 tensor (x = its address): the rotated schedule only - the caller expands for the plain one
            static bool raw_attr_set = false;
            if (!raw_attr_set) {
                CTO_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gru_layer_rot<KIN, KP, H, MS, FUSE, NW, true>),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, int(smem)));
                raw_attr_set = true;
            }
            hipLaunchKernelGGL((k_gru_layer_rot<KIN, KP, H, MS, FUSE, NW, true>), dim3(grid), dim3(64 * NW), smem, s, x, Wf, bias, out, fc1f, fc1_part,
                               int(B), int(begin), int(end), *raw);
            CTO_HIP(hipGetLastError());
            return CTO_OK;
        }
    }
    if (rot)
        hipLaunchKernelGGL((k_gru_layer_rot<KIN, KP, H, MS, FUSE, NW>), dim3(grid), dim3(64 * NW), smem, s, x, Wf, bias, out, fc1f, fc1_part,
                           int(B), int(begin), int(end), XRawArgs{nullptr, 0, 0});
    else
        hipLaunchKernelGGL((k_gru_layer<KIN, KP, H, MS, 1, FUSE>), dim3(grid), dim3(256), smem, s, x, W, bias, out, fc1w, fc1_part,
                           int(B), int(begin), int(end));
    CTO_HIP(hipGetLastError());
    return CTO_OK;
}

// Tile height per launch.  A workgroup owns (MS*16 sites, one direction) for all 33 steps, so a launch is a whole number of
// "rounds" of one workgroup per CU.  32-site tiles use every weight fragment for twice as many MFMAs and are used for every
// full round (a multiple of 16 * CUs sites); what is left over runs as 32-site tiles if it still fills most of a round, else
// as 16-site tiles, whose workgroups finish in ~0.83x the time (the weight stream per workgroup is the same, the MFMA work is
// half) and which spread a small batch over twice as many CUs: measured 1.45 -> 1.20 ms for B <= 2048, -3 % for B = 10 000.
template <int KIN, int KP, int H, bool FUSE>
int launch_gru(hipStream_t s, const float* x, const float* W, const float* Wf, const float* bias, float* out, const float* fc1w,
               const float* fc1f, float* fc1_part, int64_t B, const XRawArgs* raw = nullptr) {
    static const int cus = [] {
        int dev = 0, n = 256;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
        return n > 0 ? n : 256;
    }();
    const int64_t round32 = int64_t(16) * cus;                   // sites of one round of 32-site tiles (2 directions)
    const int64_t full = (B / round32) * round32;
    const int64_t rest = B - full;
    int rc = launch_gru_range<KIN, KP, H, 2, FUSE>(s, x, W, Wf, bias, out, fc1w, fc1f, fc1_part, B, 0, full, raw);
    if (rc != CTO_OK || rest == 0) return rc;
    if (rest * 4 > round32 * 3) return launch_gru_range<KIN, KP, H, 2, FUSE>(s, x, W, Wf, bias, out, fc1w, fc1f, fc1_part, B, full, B, raw);
    return launch_gru_range<KIN, KP, H, 1, FUSE>(s, x, W, Wf, bias, out, fc1w, fc1f, fc1_part, B, full, B, raw);
}

}  // namespace

// true when launch_gru_layer1_raw runs natively (the rotated schedule); otherwise the caller expands the tensor first
bool gru_layer1_takes_raw() {
    static const bool rot = [] { const char* e = getenv("CTO_GRU_ROT"); return !(e && e[0] == '0'); }();
    return rot;
}
int launch_gru_layer1_raw(hipStream_t s, const int16_t* x_raw, const int32_t* site_info, int which, int min_rescale_cov, const float* Wf,
                          const float* bias, float* out, int64_t B) {
    const XRawArgs raw{site_info, which, min_rescale_cov};
    return launch_gru<34, 48, 128, false>(s, reinterpret_cast<const float*>(x_raw), nullptr, Wf, bias, out, nullptr, nullptr, nullptr, B, &raw);
}
int launch_gru_layer1(hipStream_t s, const float* x, const float* W, const float* Wf, const float* bias, float* out, int64_t B) {
    return launch_gru<34, 48, 128, false>(s, x, W, Wf, bias, out, nullptr, nullptr, nullptr, B);
}

// layer 2 with the head's fc1 folded in: writes one partial [B][128] slab per direction into fc1_part
int launch_gru_layer2_fc1(hipStream_t s, const float* x, const float* W, const float* Wf, const float* bias, const float* fc1w,
                          const float* fc1f, float* fc1_part, int64_t B) {
    return launch_gru<256, 256, 192, true>(s, x, W, Wf, bias, nullptr, fc1w, fc1f, fc1_part, B);
}

// the recurrent layers on split 16-bit operands (experiment behind CTO_GRU_SPLIT=f16|bf16; gru_split_kernel.h): same tiling rule as above
template <int KIN, int KP, int H, int MS, bool F16, bool FUSE>
static int launch_split_range(hipStream_t s, const float* x, const void* Wp, const float* bias, const void* Fp, float* fc1_part,
                              float* out, int64_t B, int64_t begin, int64_t end, const GruSplitScale& sc) {
    if (end <= begin) return CTO_OK;
    const size_t smem = size_t(4) * MS * 16 * ((H + 8) + (KP + 8)) * sizeof(unsigned short) + size_t(4) * H * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        CTO_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gru_split<KIN, KP, H, MS, F16, FUSE>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, int(smem)));
        attr_set = true;
    }
    const unsigned grid = unsigned(cdiv(end - begin, MS * 16)) * 2;
    hipLaunchKernelGGL((k_gru_split<KIN, KP, H, MS, F16, FUSE>), dim3(grid), dim3(256), smem, s, x, static_cast<const uint4*>(Wp), bias,
                       static_cast<const uint4*>(Fp), fc1_part, out, int(B), int(begin), int(end), sc);
    CTO_HIP(hipGetLastError());
    return CTO_OK;
}

template <int KIN, int KP, int H, bool F16, bool FUSE>
static int launch_split(hipStream_t s, const float* x, const void* Wp, const float* bias, const void* Fp, float* fc1_part, float* out,
                        int64_t B, const GruSplitScale& sc) {
    static const int cus = [] {
        int dev = 0, n = 256;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
        return n > 0 ? n : 256;
    }();
    const int64_t round32 = int64_t(16) * cus;
    const int64_t full = (B / round32) * round32;
    const int64_t rest = B - full;
    int rc = launch_split_range<KIN, KP, H, 2, F16, FUSE>(s, x, Wp, bias, Fp, fc1_part, out, B, 0, full, sc);
    if (rc != CTO_OK || rest == 0) return rc;
    if (rest * 4 > round32 * 3) return launch_split_range<KIN, KP, H, 2, F16, FUSE>(s, x, Wp, bias, Fp, fc1_part, out, B, full, B, sc);
    return launch_split_range<KIN, KP, H, 1, F16, FUSE>(s, x, Wp, bias, Fp, fc1_part, out, B, full, B, sc);
}

// scale5 = {sx, sh, s_total, inv_s, inv_f} (GruSplitScale; chosen by pack_gru_split)
int launch_gru_layer2_fc1_split(hipStream_t s, const float* x, const void* Wp, const float* bias, const void* Fp, float* fc1_part,
                                int64_t B, bool f16, const float* scale5) {
    const GruSplitScale sc{scale5[0], scale5[1], scale5[2], scale5[3], scale5[4]};
    return f16 ? launch_split<256, 256, 192, true, true>(s, x, Wp, bias, Fp, fc1_part, nullptr, B, sc)
               : launch_split<256, 256, 192, false, true>(s, x, Wp, bias, Fp, fc1_part, nullptr, B, sc);
}

int launch_gru_layer1_split(hipStream_t s, const float* x, const void* Wp, const float* bias, float* out, int64_t B, bool f16,
                            const float* scale5) {
    const GruSplitScale sc{scale5[0], scale5[1], scale5[2], scale5[3], scale5[4]};
    return f16 ? launch_split<34, 64, 128, true, false>(s, x, Wp, bias, nullptr, nullptr, out, B, sc)
               : launch_split<34, 64, 128, false, false>(s, x, Wp, bias, nullptr, nullptr, out, B, sc);
}

#ifdef CTO_GRU_CLOCKS
extern "C" int cto_debug_gru_clocks(long long* out16) {
    CTO_HIP(hipDeviceSynchronize());
    CTO_HIP(hipMemcpyFromSymbol(out16, HIP_SYMBOL(cto::g_gru_clk), 16 * sizeof(long long)));
    return CTO_OK;
}
#endif

// BiGRU recurrent kernels (clairs/model.py:412-417, 442-448) - instantiations and launchers.
// Kept in a translation unit of their own so that edits to the CvT kernels cannot perturb their code generation.
#include <stdlib.h>
#include "common.h"
#include "gru_kernel.h"

using namespace cto;

namespace {

template <int KIN, int KP, int H, int MS, int MH, bool FUSE>
int launch_gru(hipStream_t s, const float* x, const float* W, const float* bias, float* out, const float* fc1w,
               float* fc1_part, int64_t B) {
    const size_t smem = size_t(2) * MH * MS * 16 * ((H + 4) + (KP + 4)) * sizeof(float);   // h tiles + x tiles
    static bool attr_set = false;
    if (!attr_set) {
        CTO_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gru_layer<KIN, KP, H, MS, MH, FUSE>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, int(smem)));
        attr_set = true;
    }
    const unsigned grid = unsigned(cdiv(B, MH * MS * 16)) * 2;
    // rotated schedule (gate arithmetic under the next step's x-part MFMAs); CTO_GRU_ROT=0 selects the plain one
    static const bool rot = [] { const char* e = getenv("CTO_GRU_ROT"); return !(e && e[0] == '0'); }();
    if (rot && MH == 1) {
        static bool attr2 = false;
        if (!attr2) {
            CTO_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gru_layer_rot<KIN, KP, H, MS, FUSE>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, int(smem)));
            attr2 = true;
        }
        hipLaunchKernelGGL((k_gru_layer_rot<KIN, KP, H, MS, FUSE>), dim3(grid), dim3(256), smem, s, x, W, bias, out, fc1w,
                           fc1_part, int(B));
        CTO_HIP(hipGetLastError());
        return CTO_OK;
    }
    hipLaunchKernelGGL((k_gru_layer<KIN, KP, H, MS, MH, FUSE>), dim3(grid), dim3(256 * MH), smem, s, x, W, bias, out, fc1w, fc1_part,
                       int(B));
    CTO_HIP(hipGetLastError());
    return CTO_OK;
}

}  // namespace

int launch_gru_layer1(hipStream_t s, const float* x, const float* W, const float* bias, float* out, int64_t B) {
    return launch_gru<34, 48, 128, 2, 1, false>(s, x, W, bias, out, nullptr, nullptr, B);
}

// layer 2 with the head's fc1 folded in: writes one partial [B][128] slab per direction into fc1_part
int launch_gru_layer2_fc1(hipStream_t s, const float* x, const float* W, const float* bias, const float* fc1w, float* fc1_part,
                          int64_t B) {
    return launch_gru<256, 256, 192, 2, 1, true>(s, x, W, bias, nullptr, fc1w, fc1_part, B);
}

#ifdef CTO_GRU_CLOCKS
extern "C" int cto_debug_gru_clocks(long long* out8) {
    CTO_HIP(hipDeviceSynchronize());
    CTO_HIP(hipMemcpyFromSymbol(out8, HIP_SYMBOL(cto::g_gru_clk), 8 * sizeof(long long)));
    return CTO_OK;
}
#endif

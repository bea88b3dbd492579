// 256 x 256 x 64 tile, 8 waves, rotated k-step: product instances (see conv_common.h for the family map).
#include "conv_kernel256i.h"

int conv_launch_wave8(const ConvArgs& a, long long grid256, int gn_mode, hipStream_t s) {
    constexpr int MAXDEV = 64;
    static std::once_flag once[MAXDEV];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAXDEV) return UAV_EINVAL;
    std::call_once(once[dev], [] {
        const void* fns[] = {(const void*)conv_gemm256i_kernel<6>, (const void*)conv_gemm256i_kernel<6, 1>,
                             (const void*)conv_gemm256i_kernel<6, 2>, (const void*)conv_gemm256i_kernel<6, 3>};
        for (const void* f : fns) (void)hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * LSTAGE + LEPI_BYTES);
    });
    const size_t lds = 2 * LSTAGE + LEPI_BYTES;
    if (gn_mode == 0) hipLaunchKernelGGL(conv_gemm256i_kernel<6>, dim3((unsigned)grid256), dim3(512), lds, s, a);
    else if (gn_mode == 1) hipLaunchKernelGGL((conv_gemm256i_kernel<6, 1>), dim3((unsigned)grid256), dim3(512), lds, s, a);
    else if (gn_mode == 2) hipLaunchKernelGGL((conv_gemm256i_kernel<6, 2>), dim3((unsigned)grid256), dim3(512), lds, s, a);
    else hipLaunchKernelGGL((conv_gemm256i_kernel<6, 3>), dim3((unsigned)grid256), dim3(512), lds, s, a);
    return uav_launch_status();
}

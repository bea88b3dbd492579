// 256 x 256 x 64 tile, FOUR waves (one per SIMD): the product kernel of every big-tile launch (see conv_common.h for the family map).
#include "conv_kernel256w.h"

int conv_launch_wave4(const ConvArgs& a, long long grid256, int gn_mode, bool hilo, hipStream_t s) {
    constexpr int MAXDEV = 64;
    static std::once_flag once[MAXDEV];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAXDEV) return UAV_EINVAL;
    std::call_once(once[dev], [] {
        const void* fns[] = {(const void*)conv_gemm256w_kernel<0>, (const void*)conv_gemm256w_kernel<1>,
                             (const void*)conv_gemm256w_kernel<2>, (const void*)conv_gemm256w_kernel<3>,
                             (const void*)conv_gemm256w_kernel<0, 0, true>};
        for (const void* f : fns) (void)hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * LSTAGE + LEPI_BYTES);
    });
    const size_t lds = 2 * LSTAGE + LEPI_BYTES;
    if (hilo) hipLaunchKernelGGL((conv_gemm256w_kernel<0, 0, true>), dim3((unsigned)grid256), dim3(256), lds, s, a);
    else if (gn_mode == 0) hipLaunchKernelGGL(conv_gemm256w_kernel<0>, dim3((unsigned)grid256), dim3(256), lds, s, a);
    else if (gn_mode == 1) hipLaunchKernelGGL(conv_gemm256w_kernel<1>, dim3((unsigned)grid256), dim3(256), lds, s, a);
    else if (gn_mode == 2) hipLaunchKernelGGL(conv_gemm256w_kernel<2>, dim3((unsigned)grid256), dim3(256), lds, s, a);
    else hipLaunchKernelGGL(conv_gemm256w_kernel<3>, dim3((unsigned)grid256), dim3(256), lds, s, a);
    return uav_launch_status();
}

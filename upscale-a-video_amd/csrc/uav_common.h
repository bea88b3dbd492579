// Shared device/host helpers for the gfx950 kernels of libuav_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <mutex>
#include "../../include/uav_hip.h"

typedef _Float16 half_t;
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float    float2_t __attribute__((ext_vector_type(2)));
typedef float    float4_t __attribute__((ext_vector_type(4)));
typedef float    float16_t __attribute__((ext_vector_type(16)));
typedef uint32_t uint4_t __attribute__((ext_vector_type(4)));
typedef uint32_t uint2_t __attribute__((ext_vector_type(2)));

#define UAV_DEVINL __device__ __forceinline__

// Launch-error helper: returns the hipError_t (positive) of the last launch, 0 if none.
static inline int uav_launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}

// Per-device one-time raise of a kernel's dynamic-LDS limit (the attribute is per device; a function-local `static bool`
// would cover only the device that happened to be current at the first call and races between host threads).
struct UavDynLds { std::once_flag once[64]; };
static inline int uav_set_dyn_lds(UavDynLds& st, const void* fn, int bytes) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return UAV_EINVAL;
    std::call_once(st.once[dev], [fn, bytes] { (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes); });
    return 0;
}

UAV_DEVINL float uav_silu(float x) { return x / (1.0f + __expf(-x)); }
// erf by Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7, far below the fp16 rounding of every consumer), branch-free:
// the device-library erff is a two-branch routine that costs ~4x as many VALU instructions once a wave diverges, and
// the GEGLU epilogue evaluates it 64 times per lane per tile.
UAV_DEVINL float uav_erf(float x) {
    const float ax = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
    float poly = fmaf(1.061405429f, t, -1.453152027f);
    poly = fmaf(poly, t, 1.421413741f);
    poly = fmaf(poly, t, -0.284496736f);
    poly = fmaf(poly, t, 0.254829592f);
    const float e = __builtin_amdgcn_exp2f(-1.4426950408889634f * ax * ax);
    const float r = fmaf(-poly * t, e, 1.0f);
    return __builtin_copysignf(r, x);
}
UAV_DEVINL float uav_gelu_erf(float x) { return 0.5f * x * (1.0f + uav_erf(x * 0.70710678118654752f)); }

UAV_DEVINL float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
UAV_DEVINL float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// Bijective XCD-aware remap of a linear workgroup id (cdna guide T1): consecutive remapped
// ids land on the same XCD (hardware places block b on XCD b % 8), so neighbouring tiles
// share that XCD's L2.
UAV_DEVINL uint32_t xcd_remap(uint32_t bid, uint32_t nwg) {
    const uint32_t nx = 8;
    uint32_t q = nwg / nx, r = nwg % nx;
    uint32_t xcd = bid % nx, idx = bid / nx;
    uint32_t base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

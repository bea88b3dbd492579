// Shared device/host helpers for the gfx950 kernels of libuav_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
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

UAV_DEVINL float uav_silu(float x) { return x / (1.0f + __expf(-x)); }
UAV_DEVINL float uav_gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }

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

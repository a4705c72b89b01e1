// Shared helpers for the gfx950 kernels of libmhe_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdarg>
#include "../../include/mhe.h"

namespace mhe {

typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef short v8s __attribute__((ext_vector_type(8)));
typedef unsigned short u16;

void set_error(const char *fmt, ...);

static inline int check_launch(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return MHE_ERR_LAUNCH;
    }
    return MHE_OK;
}

#define MHE_REQUIRE(cond, ...)                 \
    do {                                       \
        if (!(cond)) {                         \
            mhe::set_error(__VA_ARGS__);       \
            return MHE_ERR_ARG;                \
        }                                      \
    } while (0)

// wave-local ordering of LDS traffic: all 64 lanes run in lockstep and one
// wave's DS operations complete in order, so only the compiler needs fencing.
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__device__ __forceinline__ float bf16_to_f32(u16 h) {
    return __uint_as_float(((unsigned)h) << 16);
}
// round-to-nearest-even; NaN stays NaN (plain cast lowers to v_cvt_pk_bf16_f32)
__device__ __forceinline__ u16 f32_to_bf16(float f) {
    __bf16 b = (__bf16)f;
    return __builtin_bit_cast(u16, b);
}

// 4 consecutive elements of f32 or bf16 storage <-> 4 floats (8/16-byte accesses)
template <typename T> __device__ __forceinline__ void load4(const T *p, float *v);
template <> __device__ __forceinline__ void load4<float>(const float *p, float *v) {
    const float4 t = *reinterpret_cast<const float4 *>(p); v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
}
template <> __device__ __forceinline__ void load4<u16>(const u16 *p, float *v) {
    const uint2 t = *reinterpret_cast<const uint2 *>(p);
    v[0] = __uint_as_float(t.x << 16); v[1] = __uint_as_float(t.x & 0xffff0000u);
    v[2] = __uint_as_float(t.y << 16); v[3] = __uint_as_float(t.y & 0xffff0000u);
}
template <typename T> __device__ __forceinline__ void store4(T *p, const float *v);
template <> __device__ __forceinline__ void store4<float>(float *p, const float *v) {
    *reinterpret_cast<float4 *>(p) = make_float4(v[0], v[1], v[2], v[3]);
}
template <> __device__ __forceinline__ void store4<u16>(u16 *p, const float *v) {
    uint2 o;
    o.x = (unsigned)f32_to_bf16(v[0]) | ((unsigned)f32_to_bf16(v[1]) << 16);
    o.y = (unsigned)f32_to_bf16(v[2]) | ((unsigned)f32_to_bf16(v[3]) << 16);
    *reinterpret_cast<uint2 *>(p) = o;
}

}  // namespace mhe

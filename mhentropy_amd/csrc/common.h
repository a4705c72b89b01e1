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

// ---- order-independent statistic accumulators ----------------------------------------------------------------------------------
// Every per-channel sum that many workgroups contribute to (BatchNorm batch statistics, Gram matrices, BatchNorm-reverse sums, bias
// gradients) is accumulated in 64-bit FIXED POINT with integer atomics: integer addition is associative, so the total does not depend
// on the order in which workgroups arrive - two runs of one launch, eager or replayed from a HIP graph, give the same bits (the f32
// atomics of rounds 1-3 did not: one flipped bf16 rounding after 53 train-mode BatchNorm layers moved the C2 loss by 3e-3 ... 2e-2).
//   two-word form (BatchNorm sums):  v = W0 * 2^-16 + W1 * 2^-56, W0 = rint(v 2^16), W1 = rint((v 2^16 - W0) 2^40): an f32 partial whose
//       lowest bit is >= 2^-56 is represented EXACTLY (|v| >= 2^-32 for a full 24-bit mantissa);
//   one-word form (Gram matrices of normalised activations): quantum 2^-20.
// RANGE (round 5): a partial sum is accepted up to |v| < 2^38 (two-word form; 2^34 in the one-word form) - a 1,024-pixel sum
// of y^2 at an RMS |y| of 1.6e4, where the f32 atomics these replace would already have lost every contribution below 2^14.  (Round 4
// drew the line at 2^32: un-normalised 0..255 images or a run about to diverge could reach that.)  A shard total at or beyond 2^55
// quanta takes the finalize kernels' double-precision path (wave_totals); |shard total| must stay below 2^61 quanta = 2^45 in value.
// A partial that is NaN / Inf / beyond the range plants a sticky marker (atomicMax to 2^62, above every legitimate total, so it survives
// later adds): the finalize kernels turn it into NaN - an explicit, detectable error, as the f32 sum would have become Inf / NaN.
namespace fx {
typedef long long acc_t;
constexpr int NSH = 64;                                   // shards per unit: workgroup b adds into shard b % NSH
constexpr long long MARK = 1ll << 62, LIMIT = 1ll << 61;
__device__ __forceinline__ void add_word(acc_t *p, float x) {
    if (x != 0.f) atomicAdd(reinterpret_cast<unsigned long long *>(p), (unsigned long long)(long long)x);
}
// one-word form, quantum 2^-20
__device__ __forceinline__ void add1(acc_t *p, float v) {
    const float s = v * 0x1p20f;
    if (fabsf(s) < 0x1p54f) add_word(p, rintf(s));
    else atomicMax(p, MARK);
}
__device__ __forceinline__ bool marked(acc_t w) { return w >= LIMIT || w <= -LIMIT; }
__host__ __device__ __forceinline__ double value1(acc_t w) { return (double)w * 0x1p-20; }
// two-word form: plane 0 = [NSH][2][C] words of quantum 2^-16, plane 1 (at + NSH * 2 * C) the remainders of quantum 2^-56
__device__ __forceinline__ size_t plane(int C) { return (size_t)NSH * 2 * C; }
__device__ __forceinline__ void add2(acc_t *p, size_t lo_off, float v) {
    const float s = v * 0x1p16f;
    if (fabsf(s) < 0x1p54f) {
        const float h = rintf(s);
        add_word(p, h);
        add_word(p + lo_off, rintf((s - h) * 0x1p40f));
    } else atomicMax(p, MARK);
}
// unit = the accumulators of one BatchNorm unit, [2 planes][NSH][2 statistics][C]
__device__ __forceinline__ void add(acc_t *unit, int shard, int stat, int C, int c, float v) {
    add2(unit + ((size_t)shard * 2 + stat) * C + c, plane(C), v);
}
__host__ __device__ __forceinline__ double value2(acc_t hi, acc_t lo) { return (double)hi * 0x1p-16 + (double)lo * 0x1p-56; }
// lane = shard: the unit's totals of both statistics of channel c over the 64 shards, exact (integer wave reductions), NaN if any shard
// carries the marker.  All four words are loaded before anything is reduced (one memory round trip per channel, as the f32 form had);
// clear: the words are zeroed on the way (self-cleaning arena)
__device__ __forceinline__ void wave_totals(acc_t *unit, int C, int c, int lane, bool clear, double &t0, double &t1) {
    acc_t *p0 = unit + ((size_t)lane * 2) * C + c, *p1 = p0 + C;
    const size_t pl = plane(C);
    acc_t hi0 = p0[0], hi1 = p1[0], lo0 = p0[pl], lo1 = p1[pl];
    if (clear) { p0[0] = 0; p1[0] = 0; p0[pl] = 0; p1[pl] = 0; }
    const bool bad0 = __any(marked(hi0)), bad1 = __any(marked(hi1));
    if (__any(hi0 >= (1ll << 55) || hi0 <= -(1ll << 55) || hi1 >= (1ll << 55) || hi1 <= -(1ll << 55))) {
        // (the integer sum of 64 such words could wrap: fixed-order sum of doubles)
        double d0 = value2(hi0, lo0), d1 = value2(hi1, lo1);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { d0 += __shfl_xor(d0, o, 64); d1 += __shfl_xor(d1, o, 64); }
        t0 = bad0 ? __builtin_nan("") : d0; t1 = bad1 ? __builtin_nan("") : d1;
        return;
    }
    // Four words per lane, 64 lanes -> four totals with 7 cross-lane exchanges of a 64-bit word instead of the 24 of four butterflies (the
    // exchanges - ds_bpermute pairs - were most of this kernel's time above its launch floor): at xor 32 a lane hands over the statistic its
    // half does not keep, at xor 16 the word its quarter does not keep; four butterfly steps finish one word per lane.  Integer sums:
    // the same totals whatever the order.
    const bool up = (lane & 32) != 0, lw = (lane & 16) != 0;
    acc_t kh = up ? hi1 : hi0, kl = up ? lo1 : lo0;
    kh += __shfl_xor(up ? hi0 : hi1, 32, 64);
    kl += __shfl_xor(up ? lo0 : lo1, 32, 64);
    acc_t k = lw ? kl : kh;
    k += __shfl_xor(lw ? kh : kl, 16, 64);
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) k += __shfl_xor(k, o, 64);
    // lane 0: hi of statistic 0, lane 16: its lo; lanes 32 / 48: statistic 1
    const acc_t H0 = __shfl(k, 0, 64), L0 = __shfl(k, 16, 64), H1 = __shfl(k, 32, 64), L1 = __shfl(k, 48, 64);
    t0 = bad0 ? __builtin_nan("") : value2(H0, L0);
    t1 = bad1 ? __builtin_nan("") : value2(H1, L1);
}
__device__ __forceinline__ double wave_total(acc_t *unit, int stat, int C, int c, int lane, bool clear) {
    acc_t *p = unit + ((size_t)lane * 2 + stat) * C + c;
    acc_t hi = p[0], lo = p[plane(C)];
    if (clear) { p[0] = 0; p[plane(C)] = 0; }
    const bool bad = __any(marked(hi));
    if (__any(hi >= (1ll << 55) || hi <= -(1ll << 55))) {
        double d = value2(hi, lo);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) d += __shfl_xor(d, o, 64);
        return bad ? __builtin_nan("") : d;
    }
    // (as above: the upper half keeps the low word, the lower half the high word - 6 exchanges instead of 12)
    const bool up = (lane & 32) != 0;
    acc_t k = up ? lo : hi;
    k += __shfl_xor(up ? hi : lo, 32, 64);
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) k += __shfl_xor(k, o, 64);
    const acc_t Hs = __shfl(k, 0, 64), Ls = __shfl(k, 32, 64);
    return bad ? __builtin_nan("") : value2(Hs, Ls);
}
}  // namespace fx

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

// Reverse kernels of the ResNet trunk's non-convolution stages (train-mode BatchNorm + ReLU, max / average
// pooling, stride-2 scatter) and the optimizer tail of the train step (gradient norm, clip + Adam).
// Reference: torchvision ResNet differentiated by autograd in hand/CrossModalHand.py:455-470
// (`total_loss.backward(); clip_grad_norm_(encoderRGB.parameters(), 1.); optimizer.step()` with
// torch.optim.Adam defaults, :201).  Activations NHWC of storage type T (f32 or bf16), statistics f32/f64.
//
// All of these are HBM-bound elementwise / reduction passes; algorithmic bytes are the tensors they name.
#include "common.h"
#include <cstdlib>

namespace mhe { namespace tb {
constexpr int NSH = fx::NSH;  // statistic shards, as the forward's (conv.hip)

// Per-channel sums of g' = g [a > 0] and g' * xhat, xhat = (y - mean) invstd, over a slab of pixels per block.
// Threads: (C/4 channel quads, capped at 256) x row lanes; LDS reduce over row lanes; one sharded atomic per value.
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const T *__restrict__ g, const T *__restrict__ a,
                                                            const T *__restrict__ y, const float *__restrict__ mean_invstd,
                                                            mhe_stat_t *__restrict__ stats, long P, int C, long rows_per_block) {
    __shared__ float red[256 * 8];
    const int Q = C / 4;
    const int qpb = Q < 256 ? Q : 256;          // quads handled at once
    const int rl_n = 256 / qpb;                  // row lanes
    const int tid = threadIdx.x;
    const int ql = tid % qpb, rl = tid / qpb;
    const long r0 = (long)blockIdx.x * rows_per_block;
    const long r1 = r0 + rows_per_block < P ? r0 + rows_per_block : P;
    const int shard = (int)(blockIdx.x % NSH);
    for (int q0 = 0; q0 < Q; q0 += qpb) {
        const int q = q0 + ql;
        float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
        if (rl < rl_n && q < Q) {
            const int c = q * 4;
            const float4 mu = *reinterpret_cast<const float4 *>(mean_invstd + c);
            const float4 is = *reinterpret_cast<const float4 *>(mean_invstd + C + c);
            const float m[4] = {mu.x, mu.y, mu.z, mu.w}, iv[4] = {is.x, is.y, is.z, is.w};
            // four rows' loads in flight per thread (one row at a time ran at 1.5 TB/s: 89 us for the 134 MB of layer4's last block)
            long r = r0 + rl;
            for (; r + 3 * rl_n < r1; r += 4 * rl_n) {
                float gv[4][4], yv[4][4], av[4][4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    load4<T>(g + (r + u * rl_n) * C + c, gv[u]);
                    load4<T>(y + (r + u * rl_n) * C + c, yv[u]);
                    if (a) load4<T>(a + (r + u * rl_n) * C + c, av[u]);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float gg = (!a || av[u][k] > 0.f) ? gv[u][k] : 0.f;
                        s1[k] += gg;
                        s2[k] = fmaf(gg, (yv[u][k] - m[k]) * iv[k], s2[k]);
                    }
            }
            for (; r < r1; r += rl_n) {
                float gv[4], yv[4], av[4];
                load4<T>(g + r * C + c, gv);
                load4<T>(y + r * C + c, yv);
                if (a) load4<T>(a + r * C + c, av);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float gg = (!a || av[k] > 0.f) ? gv[k] : 0.f;
                    s1[k] += gg;
                    s2[k] = fmaf(gg, (yv[k] - m[k]) * iv[k], s2[k]);
                }
            }
        }
        if (rl_n > 1) {
            __syncthreads();
#pragma unroll
            for (int k = 0; k < 4; ++k) { red[tid * 8 + k] = s1[k]; red[tid * 8 + 4 + k] = s2[k]; }
            __syncthreads();
            if (rl == 0) {
                for (int o = 1; o < rl_n; ++o)
#pragma unroll
                    for (int k = 0; k < 4; ++k) { s1[k] += red[(o * qpb + ql) * 8 + k]; s2[k] += red[(o * qpb + ql) * 8 + 4 + k]; }
            }
        }
        if (rl == 0 && q < Q) {
#pragma unroll
            for (int k = 0; k < 4; ++k) { fx::add(stats, shard, 0, C, q * 4 + k, s1[k]); fx::add(stats, shard, 1, C, q * 4 + k, s2[k]); }
        }
    }
}

// dbeta = sum g', dgamma = sum g' xhat; coefficients of gy = k2 g' + k1 y + k0 with
//   gy = gamma invstd (g' - dbeta/M - xhat dgamma/M)       (F.batch_norm backward, training mode)
// (one wavefront per channel, lane = statistic shard: see bn_finalize_kernel in conv.hip)
__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(const mhe_stat_t *__restrict__ stats, const float *__restrict__ gamma,
                                                              const float *__restrict__ mean_invstd, float *__restrict__ dgamma,
                                                              float *__restrict__ dbeta, float *__restrict__ coef, int C, double count) {
    static_assert(NSH == 64, "one lane per statistic shard");
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (c >= C) return;
    double s1, s2;
    fx::wave_totals(const_cast<mhe_stat_t *>(stats), C, c, lane, false, s1, s2);
    if (lane) return;
    const float db = (float)s1, dg = (float)s2;
    dbeta[c] = db;
    dgamma[c] = dg;
    const float mean = mean_invstd[c], invstd = mean_invstd[C + c];
    const float k2 = gamma[c] * invstd;
    const float k1 = -k2 * invstd * dg / (float)count;
    coef[c] = k2;
    coef[C + c] = k1;
    coef[2 * C + c] = -k2 * db / (float)count - k1 * mean;
}

template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const T *__restrict__ g, const T *__restrict__ a,
                                                           const T *__restrict__ y, const float *__restrict__ coef,
                                                           T *__restrict__ gy, T *__restrict__ g_masked, size_t n4, int C) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const size_t e = i * 4;
        const int c = (int)(e % C);
        float gv[4], yv[4], av[4], o[4];
        load4<T>(g + e, gv);
        load4<T>(y + e, yv);
        if (a) {
            load4<T>(a + e, av);
#pragma unroll
            for (int k = 0; k < 4; ++k) gv[k] = av[k] > 0.f ? gv[k] : 0.f;
        }
        const float4 k2 = *reinterpret_cast<const float4 *>(coef + c), k1 = *reinterpret_cast<const float4 *>(coef + C + c),
                     k0 = *reinterpret_cast<const float4 *>(coef + 2 * C + c);
        o[0] = fmaf(k2.x, gv[0], fmaf(k1.x, yv[0], k0.x)); o[1] = fmaf(k2.y, gv[1], fmaf(k1.y, yv[1], k0.y));
        o[2] = fmaf(k2.z, gv[2], fmaf(k1.z, yv[2], k0.z)); o[3] = fmaf(k2.w, gv[3], fmaf(k1.w, yv[3], k0.w));
        store4<T>(gy + e, o);
        if (g_masked) store4<T>(g_masked + e, gv);
    }
}

// mean / invstd of the batch statistics the forward accumulated (same shards, same f64 combine as bn_finalize)
__global__ __launch_bounds__(256) void bn_mean_invstd_kernel(const mhe_stat_t *__restrict__ stats, float *__restrict__ mean_invstd, int C, double count, float eps) {
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (c >= C) return;
    double s1, s2;
    fx::wave_totals(const_cast<mhe_stat_t *>(stats), C, c, lane, false, s1, s2);
    if (lane) return;
    const double dmean = s1 / (double)count;
    const double dvar0 = s2 / (double)count - dmean * dmean, dvar = dvar0 != dvar0 ? dvar0 : fmax(dvar0, 0.0);      // (NaN stays NaN)
    mean_invstd[c] = (float)dmean;
    mean_invstd[C + c] = 1.f / sqrtf((float)dvar + eps);
}

// 3x3 stride-2 pad-1 max pool that also records which of the 9 taps won (first maximum in scan order, as
// torch's max_pool2d does) so that the reverse pass is a gather
template <typename T>
__global__ void maxpool_idx_kernel(const T *__restrict__ x, T *__restrict__ y, unsigned char *__restrict__ idx, int B, int H,
                                   int W, int C, int Ho, int Wo) {
    const size_t n4 = (size_t)B * Ho * Wo * C / 4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const size_t e = i * 4;
        const int c = (int)(e % C);
        size_t t = e / C;
        const int wo = (int)(t % Wo); t /= Wo;
        const int ho = (int)(t % Ho);
        const int b = (int)(t / Ho);
        float m[4] = {-3.0e38f, -3.0e38f, -3.0e38f, -3.0e38f};
        unsigned am[4] = {0, 0, 0, 0};
#pragma unroll
        for (int dh = 0; dh < 3; ++dh)
#pragma unroll
            for (int dw = 0; dw < 3; ++dw) {
                const int hi = 2 * ho - 1 + dh, wi = 2 * wo - 1 + dw;
                if ((unsigned)hi >= (unsigned)H || (unsigned)wi >= (unsigned)W) continue;
                float v[4];
                load4<T>(x + (((size_t)b * H + hi) * W + wi) * C + c, v);
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (v[k] > m[k]) { m[k] = v[k]; am[k] = dh * 3 + dw; }
            }
        store4<T>(y + e, m);
        *reinterpret_cast<unsigned *>(idx + e) = am[0] | (am[1] << 8) | (am[2] << 16) | (am[3] << 24);
    }
}

// gx[b,h,w,c] = sum over the (at most 4) windows containing (h,w) whose recorded winner is (h,w)
template <typename T>
__global__ void maxpool_bwd_kernel(const T *__restrict__ gy, const unsigned char *__restrict__ idx, T *__restrict__ gx, int B,
                                   int H, int W, int C, int Ho, int Wo) {
    const size_t n4 = (size_t)B * H * W * C / 4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const size_t e = i * 4;
        const int c = (int)(e % C);
        size_t t = e / C;
        const int w = (int)(t % W); t /= W;
        const int h = (int)(t % H);
        const int b = (int)(t / H);
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        for (int ho = h / 2; ho <= (h + 1) / 2; ++ho) {
            if (ho >= Ho) continue;
            const int dh = h - (2 * ho - 1);
            for (int wo = w / 2; wo <= (w + 1) / 2; ++wo) {
                if (wo >= Wo) continue;
                const unsigned tap = dh * 3 + (w - (2 * wo - 1));
                const size_t o = (((size_t)b * Ho + ho) * Wo + wo) * C + c;
                const unsigned am = *reinterpret_cast<const unsigned *>(idx + o);
                float gv[4];
                load4<T>(gy + o, gv);
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (((am >> (8 * k)) & 0xffu) == tap) acc[k] += gv[k];
            }
        }
        store4<T>(gx + e, acc);
    }
}

// ---- the stem's pool with its BatchNorm folded in (train step).  Forward: y = maxpool(relu(x * scale + shift)) + the winning tap, straight
// from the raw stem output (the normalised copy is never written); reverse: the pool's scatter, the ReLU gate recomputed from the raw
// output, and the BatchNorm-reverse sums of the gated gradient in ONE pass (separately: scatter 0.4 ms + a reduction over three
// full-resolution tensors 0.33 ms + an apply pass over four).  16-byte lanes: E = 8 (bf16) or 4 (f32) channels per thread.
template <typename T> struct Lane;
template <> struct Lane<float> {
    static constexpr int E = 4;
    static __device__ __forceinline__ void get(const float *p, float *v) { const float4 r = *reinterpret_cast<const float4 *>(p); v[0] = r.x; v[1] = r.y; v[2] = r.z; v[3] = r.w; }
    static __device__ __forceinline__ void put(float *p, const float *v) { *reinterpret_cast<float4 *>(p) = make_float4(v[0], v[1], v[2], v[3]); }
    static __device__ __forceinline__ void unpack(uint4 r, float *v) { v[0] = __uint_as_float(r.x); v[1] = __uint_as_float(r.y); v[2] = __uint_as_float(r.z); v[3] = __uint_as_float(r.w); }
    static __device__ __forceinline__ float stored(float v) { return v; }
};
template <> struct Lane<u16> {
    static constexpr int E = 8;
    static __device__ __forceinline__ void unpack(uint4 r, float *v) {
        const unsigned in[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) { v[2 * i] = __uint_as_float(in[i] << 16); v[2 * i + 1] = __uint_as_float(in[i] & 0xffff0000u); }
    }
    static __device__ __forceinline__ void get(const u16 *p, float *v) { unpack(*reinterpret_cast<const uint4 *>(p), v); }
    static __device__ __forceinline__ void put(u16 *p, const float *v) {
        unsigned o[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = (unsigned)f32_to_bf16(v[2 * i]) | ((unsigned)f32_to_bf16(v[2 * i + 1]) << 16);
        *reinterpret_cast<uint4 *>(p) = make_uint4(o[0], o[1], o[2], o[3]);
    }
    static __device__ __forceinline__ float stored(float v) { return __uint_as_float((unsigned)f32_to_bf16(v) << 16); }     // the value a bf16 tensor would hold
};

// bn_bwd_apply_kernel on 16-byte lanes (bf16: eight elements per thread; the four-element form moved 8 bytes per lane - the 8-byte
// accesses run at 0.54-0.70 of the 16-byte rate on this chip, MI355X_MICROARCH.md - and took a 64-bit modulo per element group): the channel
// chunk of a thread does not change along its walk (the grid's stride is a multiple of 2048 elements >= C, C a power of two or a divisor of
// the stride: checked by the launcher), so the three coefficient rows are read once.  Same arithmetic, same results.
template <typename T, int U>
__global__ __launch_bounds__(256) void bn_bwd_apply_wide_kernel(const T *__restrict__ g, const T *__restrict__ a, const T *__restrict__ y,
                                                                const float *__restrict__ coef, T *__restrict__ gy, T *__restrict__ g_masked, size_t nl, int C) {
    constexpr int E = Lane<T>::E;
    const size_t i0 = (size_t)blockIdx.x * 256 + threadIdx.x;
    const int c = (int)((i0 * E) % (size_t)C);
    float k2[E], k1[E], k0[E];
#pragma unroll
    for (int k = 0; k < E; ++k) { k2[k] = coef[c + k]; k1[k] = coef[C + c + k]; k0[k] = coef[2 * C + c + k]; }
    // U 16-byte pieces per thread and tensor in flight.  U = 4 for the 33 / 17 MB tensors of layer3 / layer4: with one piece per thread a launch
    // was four dependent load -> store rounds of workgroups per CU (100 MB in 39 us = 2.6 TB/s; 21-26 us with four).  U = 1 for the large
    // tensors (eight waves per SIMD carry the latency there: 5.6-6.3 TB/s, 5-10 % slower with U = 4 at half the occupancy)
    const size_t S = (size_t)gridDim.x * 256;
    for (size_t i = i0; i < nl; i += U * S) {
        uint4 rg[U], ry[U], ra[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t j = i + u * S < nl ? i + u * S : i;                               // (past the end: a harmless re-read, never stored)
            rg[u] = reinterpret_cast<const uint4 *>(g)[j];
            ry[u] = reinterpret_cast<const uint4 *>(y)[j];
            if (a) ra[u] = reinterpret_cast<const uint4 *>(a)[j];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (i + u * S >= nl) break;
            float gv[E], yv[E], o[E];
            Lane<T>::unpack(rg[u], gv);
            Lane<T>::unpack(ry[u], yv);
            if (a) {
                float av[E];
                Lane<T>::unpack(ra[u], av);
#pragma unroll
                for (int k = 0; k < E; ++k) gv[k] = av[k] > 0.f ? gv[k] : 0.f;
            }
#pragma unroll
            for (int k = 0; k < E; ++k) o[k] = fmaf(k2[k], gv[k], fmaf(k1[k], yv[k], k0[k]));
            const size_t e = (i + u * S) * E;
            Lane<T>::put(gy + e, o);
            if (g_masked) Lane<T>::put(g_masked + e, gv);
        }
    }
}

// y = maxpool3x3s2(relu(x * scale + shift)) and the winning tap; the comparison runs on the values as a stored activation would hold them
// (bf16-rounded for T = u16), first maximum in scan order - the same winners as maxpool_idx_kernel on the materialised activation
template <typename T>
__global__ __launch_bounds__(256) void maxpool_idx_affine_kernel(const T *__restrict__ x, const float *__restrict__ scale, const float *__restrict__ shift,
                                                                 T *__restrict__ y, unsigned char *__restrict__ idx, int B, int H, int W, int C, int Ho, int Wo,
                                                                 T *__restrict__ xwin = nullptr) {
    constexpr int E = Lane<T>::E;
    const int cpp = C / E;                                   // chunks per pixel
    const unsigned n = (unsigned)B * Ho * Wo * cpp;
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int c = (int)(i % cpp) * E;
        unsigned t = i / cpp;
        const int wo = (int)(t % Wo); t /= Wo;
        const int ho = (int)(t % Ho);
        const int b = (int)(t / Ho);
        float v[9][E];
        bool ok[9];
#pragma unroll
        for (int dh = 0; dh < 3; ++dh)
#pragma unroll
            for (int dw = 0; dw < 3; ++dw) {                 // nine loads in flight from clamped addresses; out-of-range taps dropped afterwards
                const int hi = 2 * ho - 1 + dh, wi = 2 * wo - 1 + dw;
                ok[dh * 3 + dw] = (unsigned)hi < (unsigned)H && (unsigned)wi < (unsigned)W;
                const int hc = hi < 0 ? 0 : (hi >= H ? H - 1 : hi), wc = wi < 0 ? 0 : (wi >= W ? W - 1 : wi);
                Lane<T>::get(x + (((size_t)b * H + hc) * W + wc) * C + c, v[dh * 3 + dw]);
            }
        float sc[E], sf[E], m[E], xw[E];
        unsigned am[E];
#pragma unroll
        for (int k = 0; k < E; ++k) { sc[k] = scale[c + k]; sf[k] = shift[c + k]; m[k] = -3.0e38f; am[k] = 0; xw[k] = 0.f; }
#pragma unroll
        for (int j = 0; j < 9; ++j)
#pragma unroll
            for (int k = 0; k < E; ++k) {
                const float a = Lane<T>::stored(fmaxf(fmaf(v[j][k], sc[k], sf[k]), 0.f));
                const bool win = ok[j] && a > m[k];
                m[k] = win ? a : m[k];
                am[k] = win ? (unsigned)j : am[k];
                xw[k] = win ? v[j][k] : xw[k];               // the RAW input at the winner (what the reverse pass normalises again)
            }
        const size_t e = (size_t)i * E;
        Lane<T>::put(y + e, m);
        if (xwin) Lane<T>::put(xwin + e, xw);
        unsigned w0 = 0, w1 = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) w0 |= am[k] << (8 * k);
        if constexpr (E == 8) {
#pragma unroll
            for (int k = 0; k < 4; ++k) w1 |= am[4 + k] << (8 * k);
            *reinterpret_cast<uint2 *>(idx + e) = make_uint2(w0, w1);
        } else *reinterpret_cast<unsigned *>(idx + e) = w0;
    }
}

// The BatchNorm-reverse sums of the stem WITHOUT a walk over its full-resolution output (round 4).  The gradient that reaches the stem's
// output through the max pool is non-zero only at the pool's winners, and the forward pass can keep what the sums need of them: per POOLED
// element its gradient g, the pooled value a = relu(bn(y_win)) (the gate: a > 0) and the raw winner y_win (maxpool_idx_affine_kernel's
// xwin):  sum g' = sum_pooled g [a > 0],  sum g' xhat = sum_pooled g [a > 0] (y_win - mean) invstd.  A quarter of the elements of
// maxpool_bwd_bn_kernel's first walk and no window gathers.  One difference to that walk, in the last bf16 bit: where two windows picked
// the same pixel, the walk rounds the SUM of their gradients to the storage type before adding it up (as the apply walk stores it), here
// each term enters exactly.
template <typename T>
__global__ __launch_bounds__(256) void pooled_bn_sums_kernel(const T *__restrict__ g, const T *__restrict__ a, const T *__restrict__ ywin,
                                                             const float *__restrict__ mean_invstd, mhe_stat_t *__restrict__ stats, size_t nl, int C) {
    constexpr int E = Lane<T>::E;
    __shared__ float red[256 * 2 * E];
    const int tid = threadIdx.x, cpp = C / E;
    const size_t i0 = (size_t)blockIdx.x * 256 + tid, S = (size_t)gridDim.x * 256;
    const int c = (int)((i0 * E) % (size_t)C);
    float mu[E], iv[E], s1[E], s2[E];
#pragma unroll
    for (int k = 0; k < E; ++k) { mu[k] = mean_invstd[c + k]; iv[k] = mean_invstd[C + c + k]; s1[k] = s2[k] = 0.f; }
    for (size_t i = i0; i < nl; i += 2 * S) {
        uint4 rg[2], ra[2], ry[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const size_t j = i + u * S < nl ? i + u * S : i;
            rg[u] = reinterpret_cast<const uint4 *>(g)[j]; ra[u] = reinterpret_cast<const uint4 *>(a)[j]; ry[u] = reinterpret_cast<const uint4 *>(ywin)[j];
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (i + u * S >= nl) break;
            float gv[E], av[E], yv[E];
            Lane<T>::unpack(rg[u], gv); Lane<T>::unpack(ra[u], av); Lane<T>::unpack(ry[u], yv);
#pragma unroll
            for (int k = 0; k < E; ++k) {
                const float t = av[k] > 0.f ? gv[k] : 0.f;
                s1[k] += t;
                s2[k] = fmaf(t, (yv[k] - mu[k]) * iv[k], s2[k]);
            }
        }
    }
    // threads tid, tid + cpp, ... of a block share a channel chunk (the launcher keeps the grid's stride a multiple of cpp)
#pragma unroll
    for (int k = 0; k < E; ++k) { red[tid * 2 * E + k] = s1[k]; red[tid * 2 * E + E + k] = s2[k]; }
    __syncthreads();
    if (tid < cpp) {
        const int shard = (int)(blockIdx.x % NSH);
        for (int k = 0; k < E; ++k) {
            float x1 = 0.f, x2 = 0.f;
            for (int o = tid; o < 256; o += cpp) { x1 += red[o * 2 * E + k]; x2 += red[o * 2 * E + E + k]; }
            fx::add(stats, shard, 0, C, c + k, x1);
            fx::add(stats, shard, 1, C, c + k, x2);
        }
    }
}

// gx = (scatter of gy to the recorded winners) [relu(y * scale + shift) > 0], written as T, and the BatchNorm-reverse sums of the stored gx:
// stats[shard][0][c] += sum gx, [1][c] += sum gx * (y - mean) * invstd.  blockDim * gridDim is a multiple of the chunks per pixel, so a thread
// keeps one channel chunk for its whole walk: sums in registers, folded through LDS at the end.
// Two further forms of the same walk, so that the scattered gradient never exists in memory (it is as large as the stem's output):
// stats only (gx = NULL), and - coef = the BatchNorm reverse's k2 | k1 | k0 - gx = k2 (scattered, gated, rounded gradient) + k1 y + k0, i.e.
// bn_bwd_apply_kernel's result in its arithmetic, without the sums.
template <typename T>
__global__ __launch_bounds__(256) void maxpool_bwd_bn_kernel(const T *__restrict__ gy, const unsigned char *__restrict__ idx, const T *__restrict__ y,
                                                             const float *__restrict__ scale, const float *__restrict__ shift,
                                                             const float *__restrict__ mean_invstd, mhe_stat_t *__restrict__ stats, T *__restrict__ gx,
                                                             int B, int H, int W, int C, int Ho, int Wo, const float *__restrict__ coef = nullptr) {
    constexpr int E = Lane<T>::E;
    __shared__ float red[256 * 2 * E];
    const int cpp = C / E;
    const unsigned n = (unsigned)B * H * W * cpp;
    const int tid = threadIdx.x;
    const int c = (int)((blockIdx.x * blockDim.x + tid) % cpp) * E;
    float sc[E], sf[E], mu[E], iv[E], s1[E], s2[E];
#pragma unroll
    for (int k = 0; k < E; ++k) { sc[k] = scale[c + k]; sf[k] = shift[c + k]; mu[k] = mean_invstd[c + k]; iv[k] = mean_invstd[C + c + k]; s1[k] = s2[k] = 0.f; }
    float k2[E], k1[E], k0[E];
#pragma unroll
    for (int k = 0; k < E; ++k) { k2[k] = coef ? coef[c + k] : 0.f; k1[k] = coef ? coef[C + c + k] : 0.f; k0[k] = coef ? coef[2 * C + c + k] : 0.f; }
    // (power-of-two maps - the stem's 128 x 128 - take shifts instead of three integer divisions per element group; the second window row
    // exists for odd h only, which is uniform over a wave: 8 threads share a pixel and a wave's 8 / 16 pixels lie in one row)
    const bool p2 = !(W & (W - 1)) && !(H & (H - 1)) && !(cpp & (cpp - 1));
    const int cs = __ffs(cpp) - 1, wsh = __ffs(W) - 1, hsh = __ffs(H) - 1;
    for (unsigned i = blockIdx.x * blockDim.x + tid; i < n; i += gridDim.x * blockDim.x) {
        int w, h, b;
        if (p2) { const unsigned t = i >> cs; w = (int)(t & (unsigned)(W - 1)); h = (int)((t >> wsh) & (unsigned)(H - 1)); b = (int)(t >> (wsh + hsh)); }
        else { unsigned t = i / cpp; w = (int)(t % W); t /= W; h = (int)(t % H); b = (int)(t / H); }
        // windows (ho, wo) containing (h, w): ho in {h / 2, (h + 1) / 2}, the second one only for odd h (and inside the pooled grid)
        const size_t e = (size_t)i * E;
        float yv[E], acc[E];
        Lane<T>::get(y + e, yv);
#pragma unroll
        for (int k = 0; k < E; ++k) acc[k] = 0.f;
        // the windows of one row, in the order (u, 0), (u, 1): loads first, then the adds (the order the sums have always had)
        auto window_row = [&](int u) __attribute__((always_inline)) {
            float g[2][E];
            unsigned a0[2], a1[2], tap[2];
            bool ok[2];
#pragma unroll
            for (int v = 0; v < 2; ++v) {
                const int ho = (h + u) / 2, wo = (w + v) / 2;
                ok[v] = (v == 0 || (w & 1)) && ho < Ho && wo < Wo;
                tap[v] = (unsigned)((h - (2 * ho - 1)) * 3 + (w - (2 * wo - 1)));
                const size_t o = (((size_t)b * Ho + (ho < Ho ? ho : Ho - 1)) * Wo + (wo < Wo ? wo : Wo - 1)) * C + c;
                Lane<T>::get(gy + o, g[v]);
                if constexpr (E == 8) { const uint2 r = *reinterpret_cast<const uint2 *>(idx + o); a0[v] = r.x; a1[v] = r.y; }
                else { a0[v] = *reinterpret_cast<const unsigned *>(idx + o); a1[v] = 0; }
            }
#pragma unroll
            for (int v = 0; v < 2; ++v)
#pragma unroll
                for (int k = 0; k < E; ++k) {
                    const unsigned am = ((k < 4 ? a0[v] : a1[v]) >> (8 * (k & 3))) & 0xffu;
                    acc[k] += (ok[v] && am == tap[v]) ? g[v][k] : 0.f;
                }
        };
        window_row(0);
        if (h & 1) window_row(1);
#pragma unroll
        for (int k = 0; k < E; ++k) {
            const bool on = Lane<T>::stored(fmaxf(fmaf(yv[k], sc[k], sf[k]), 0.f)) > 0.f;
            acc[k] = on ? Lane<T>::stored(acc[k]) : 0.f;
            s1[k] += acc[k];
            s2[k] = fmaf(acc[k], (yv[k] - mu[k]) * iv[k], s2[k]);
            if (coef) acc[k] = fmaf(k2[k], acc[k], fmaf(k1[k], yv[k], k0[k]));
        }
        if (gx) Lane<T>::put(gx + e, acc);
    }
    if (!stats) return;
    // threads tid, tid + cpp, ... of a block share a channel chunk
#pragma unroll
    for (int k = 0; k < E; ++k) { red[tid * 2 * E + k] = s1[k]; red[tid * 2 * E + E + k] = s2[k]; }
    __syncthreads();
    if (tid < cpp) {
        const int shard = (int)(blockIdx.x % NSH);
        for (int k = 0; k < E; ++k) {
            float a = 0.f, b2 = 0.f;
            for (int o = tid; o < 256; o += cpp) { a += red[o * 2 * E + k]; b2 += red[o * 2 * E + E + k]; }
            fx::add(stats, shard, 0, C, c + k, a);
            fx::add(stats, shard, 1, C, c + k, b2);
        }
    }
}

// gx[b,p,c] = g[b,c] / HW  (global average pool)
template <typename T>
__global__ void avgpool_bwd_kernel(const float *__restrict__ g, const T *__restrict__ mask, T *__restrict__ gx, int HW, int C, size_t n4) {
    const float inv = 1.f / (float)HW;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const size_t e = i * 4;
        const int c = (int)(e % C);
        const size_t b = e / ((size_t)HW * C);
        const float4 v = *reinterpret_cast<const float4 *>(g + b * C + c);
        float o[4] = {v.x * inv, v.y * inv, v.z * inv, v.w * inv};
        if (mask) {
            float mv[4];
            load4<T>(mask + e, mv);
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k] = mv[k] > 0.f ? o[k] : 0.f;
        }
        store4<T>(gx + e, o);
    }
}

// out[b, 2ho, 2wo, c] = g[b, ho, wo, c] (+ base), everything else 0 (+ base): the data gradient of a stride-2
// sampling, either as the zero-dilated operand of a 3x3 data-gradient convolution or added to `base`
template <typename T>
__global__ void upsample2_kernel(const T *__restrict__ g, const T *__restrict__ base, T *__restrict__ out, int B, int H, int W,
                                 int C, int Ho, int Wo) {
    const size_t n4 = (size_t)B * H * W * C / 4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const size_t e = i * 4;
        const int c = (int)(e % C);
        size_t t = e / C;
        const int w = (int)(t % W); t /= W;
        const int h = (int)(t % H);
        const int b = (int)(t / H);
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (base) load4<T>(base + e, v);
        if (!(h & 1) && !(w & 1) && (h >> 1) < Ho && (w >> 1) < Wo) {
            float gv[4];
            load4<T>(g + (((size_t)b * Ho + (h >> 1)) * Wo + (w >> 1)) * C + c, gv);
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] += gv[k];
        }
        store4<T>(out + e, v);
    }
}

// sum of squares of the flat gradient buffer -> out[0], in a FIXED summation order (per-block partials, then one block
// folds them): data-parallel replicas that hold the same all-reduced gradient must compute the same clip coefficient bit
// for bit, or they drift apart (atomics would make the order, hence the rounding, vary from rank to rank)
constexpr int SQ_BLOCKS = 1024;
__global__ __launch_bounds__(256) void sqnorm_partial_kernel(const float *__restrict__ g, size_t n, float *__restrict__ partial) {
    __shared__ float red[4];
    // 16-byte loads, two in flight per thread (4-byte lanes read the 184 MB gradient of config C2 at 2.4 TB/s); a fixed order all the same
    float acc = 0.f;
    const size_t n4 = n / 4, S = (size_t)SQ_BLOCKS * 256;
    const float4 *g4 = reinterpret_cast<const float4 *>(g);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += 2 * S) {
        const float4 a = g4[i], b = i + S < n4 ? g4[i + S] : make_float4(0.f, 0.f, 0.f, 0.f);
        acc = fmaf(a.x, a.x, acc); acc = fmaf(a.y, a.y, acc); acc = fmaf(a.z, a.z, acc); acc = fmaf(a.w, a.w, acc);
        acc = fmaf(b.x, b.x, acc); acc = fmaf(b.y, b.y, acc); acc = fmaf(b.z, b.z, acc); acc = fmaf(b.w, b.w, acc);
    }
    if (blockIdx.x == 0 && threadIdx.x < (unsigned)(n - 4 * n4)) { const float t = g[4 * n4 + threadIdx.x]; acc = fmaf(t, t, acc); }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ __launch_bounds__(256) void sqnorm_final_kernel(const float *__restrict__ partial, float *__restrict__ out) {
    __shared__ float red[4];
    float acc = 0.f;
    for (int i = threadIdx.x; i < SQ_BLOCKS; i += 256) acc += partial[i];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = (red[0] + red[1]) + (red[2] + red[3]);
}

// clip_grad_norm_(max_norm) folded into torch.optim.Adam's update (defaults: no weight decay, no amsgrad):
//   coef = min(1, max_norm / (|g| + 1e-6));  g *= coef * grad_scale;  m = b1 m + (1-b1) g;  v = b2 v + (1-b2) g^2;
//   p -= lr / (1 - b1^t) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
// t is read from device memory (state[0], advanced by the caller's tick kernel) so a captured graph stays valid.
__global__ __launch_bounds__(256) void adam_kernel(float *__restrict__ p, const float *__restrict__ g, float *__restrict__ m,
                                                   float *__restrict__ v, size_t n, const float *__restrict__ sqnorm,
                                                   const int *__restrict__ step, float lr, float b1, float b2, float eps,
                                                   float max_norm, float grad_scale) {
    const float t = (float)step[0];
    const float bc1 = 1.f - powf(b1, t), bc2s = sqrtf(1.f - powf(b2, t));
    float coef = grad_scale;
    if (max_norm > 0.f) {
        const float nrm = sqrtf(sqnorm[0]) * grad_scale;
        const float cc = max_norm / (nrm + 1e-6f);
        coef *= cc < 1.f ? cc : 1.f;
    }
    const float step_size = lr / bc1;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float gg = g[i] * coef;
        const float mm = b1 * m[i] + (1.f - b1) * gg;
        const float vv = b2 * v[i] + (1.f - b2) * gg * gg;
        m[i] = mm; v[i] = vv;
        p[i] -= step_size * mm / (sqrtf(vv) / bc2s + eps);
    }
}

__global__ void tick_kernel(int *step, float *sqnorm) { step[0] += 1; sqnorm[0] = 0.f; }
}}  // namespace mhe::tb

using namespace mhe;
static inline unsigned ewg(size_t n) { size_t b = (n + 255) / 256; return (unsigned)(b < 1 ? 1 : (b > 8192 ? 8192 : b)); }
#define DISPATCH_T(dtype, KERNEL, grid, ...)                                                                           \
    do {                                                                                                                \
        if ((dtype) == MHE_F32) hipLaunchKernelGGL(KERNEL<float>, grid, dim3(256), 0, (hipStream_t)stream, __VA_ARGS__); \
        else hipLaunchKernelGGL(KERNEL<u16>, grid, dim3(256), 0, (hipStream_t)stream, __VA_ARGS__);                      \
    } while (0)

extern "C" int mhe_bn_mean_invstd(const mhe_stat_t *stats, float *mean_invstd, int C, double count, float eps, void *stream) {
    MHE_REQUIRE(stats && mean_invstd && C > 0 && count > 0.f, "mhe_bn_mean_invstd: bad arguments");
    hipLaunchKernelGGL(tb::bn_mean_invstd_kernel, dim3((C + 3) / 4), dim3(256), 0, (hipStream_t)stream, stats, mean_invstd, C, count, eps);
    return check_launch("bn_mean_invstd_kernel");
}

extern "C" int mhe_bn_bwd_reduce_nhwc(const void *g, const void *a, const void *y, const float *mean_invstd, mhe_stat_t *stats,
                                      long P, int C, int dtype, void *stream) {
    MHE_REQUIRE(g && y && mean_invstd && stats && P > 0 && C > 0 && C % 4 == 0, "mhe_bn_bwd_reduce_nhwc: bad arguments");
    long blocks = P / 64;
    if (blocks < 1) blocks = 1;
    if (blocks > 2048) blocks = 2048;
    const long rpb = (P + blocks - 1) / blocks;
    blocks = (P + rpb - 1) / rpb;
    if (dtype == MHE_F32)
        hipLaunchKernelGGL(tb::bn_bwd_reduce_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const float *)g,
                           (const float *)a, (const float *)y, mean_invstd, stats, P, C, rpb);
    else
        hipLaunchKernelGGL(tb::bn_bwd_reduce_kernel<u16>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const u16 *)g,
                           (const u16 *)a, (const u16 *)y, mean_invstd, stats, P, C, rpb);
    return check_launch("bn_bwd_reduce_kernel");
}

extern "C" int mhe_bn_bwd_finalize(const mhe_stat_t *stats, const float *gamma, const float *mean_invstd, float *dgamma,
                                   float *dbeta, float *coef, int C, double count, void *stream) {
    MHE_REQUIRE(stats && gamma && mean_invstd && dgamma && dbeta && coef && C > 0 && count > 0.f, "mhe_bn_bwd_finalize: bad arguments");
    hipLaunchKernelGGL(tb::bn_bwd_finalize_kernel, dim3((C + 3) / 4), dim3(256), 0, (hipStream_t)stream, stats, gamma,
                       mean_invstd, dgamma, dbeta, coef, C, count);
    return check_launch("bn_bwd_finalize_kernel");
}

extern "C" int mhe_bn_bwd_apply_nhwc(const void *g, const void *a, const void *y, const float *coef, void *gy, void *g_masked,
                                     long P, int C, int dtype, void *stream) {
    MHE_REQUIRE(g && y && coef && gy && P > 0 && C > 0 && C % 4 == 0, "mhe_bn_bwd_apply_nhwc: bad arguments");
    const size_t n4 = (size_t)P * C / 4;
    // 16-byte lanes where a thread's channel chunk is the same at every step of its walk (stride = blocks * 256 lanes, a multiple of C / E)
    const int E = dtype == MHE_F32 ? 4 : 8;
    const size_t nl = (size_t)P * C / E;
    static const int wide_env = getenv("MHE_BN_BWD_APPLY_WIDE") ? atoi(getenv("MHE_BN_BWD_APPLY_WIDE")) : 1;
    const bool deep = nl * 16 <= ((size_t)80 << 20);                  // up to 80 MB per tensor: four 16-byte pieces per thread and tensor in flight
    const unsigned wblocks = ewg(deep ? (nl + 3) / 4 : nl);
    if (wide_env && C % E == 0 && ((size_t)wblocks * 256) % (size_t)(C / E) == 0 &&
        ((uintptr_t)g | (uintptr_t)a | (uintptr_t)y | (uintptr_t)gy | (uintptr_t)g_masked) % 16 == 0) {
#define MHE_APPLY_WIDE(T, U) hipLaunchKernelGGL((tb::bn_bwd_apply_wide_kernel<T, U>), dim3(wblocks), dim3(256), 0, (hipStream_t)stream, (const T *)g, (const T *)a, \
                                                (const T *)y, coef, (T *)gy, (T *)g_masked, nl, C)
        if (dtype == MHE_F32) { if (deep) MHE_APPLY_WIDE(float, 4); else MHE_APPLY_WIDE(float, 1); }
        else { if (deep) MHE_APPLY_WIDE(u16, 4); else MHE_APPLY_WIDE(u16, 1); }
#undef MHE_APPLY_WIDE
        return check_launch("bn_bwd_apply_wide_kernel");
    }
    if (dtype == MHE_F32)
        hipLaunchKernelGGL(tb::bn_bwd_apply_kernel<float>, dim3(ewg(n4)), dim3(256), 0, (hipStream_t)stream, (const float *)g,
                           (const float *)a, (const float *)y, coef, (float *)gy, (float *)g_masked, n4, C);
    else
        hipLaunchKernelGGL(tb::bn_bwd_apply_kernel<u16>, dim3(ewg(n4)), dim3(256), 0, (hipStream_t)stream, (const u16 *)g,
                           (const u16 *)a, (const u16 *)y, coef, (u16 *)gy, (u16 *)g_masked, n4, C);
    return check_launch("bn_bwd_apply_kernel");
}

extern "C" int mhe_maxpool3x3s2_idx_nhwc(const void *x, void *y, unsigned char *idx, int B, int H, int W, int C, int dtype, void *stream) {
    MHE_REQUIRE(x && y && idx && B > 0 && H > 0 && W > 0 && C % 4 == 0, "mhe_maxpool3x3s2_idx_nhwc: bad arguments");
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    const size_t n4 = (size_t)B * Ho * Wo * C / 4;
    if (dtype == MHE_F32)
        hipLaunchKernelGGL(tb::maxpool_idx_kernel<float>, dim3(ewg(n4)), dim3(256), 0, (hipStream_t)stream, (const float *)x, (float *)y, idx, B, H, W, C, Ho, Wo);
    else
        hipLaunchKernelGGL(tb::maxpool_idx_kernel<u16>, dim3(ewg(n4)), dim3(256), 0, (hipStream_t)stream, (const u16 *)x, (u16 *)y, idx, B, H, W, C, Ho, Wo);
    return check_launch("maxpool_idx_kernel");
}

extern "C" int mhe_maxpool3x3s2_bwd_nhwc(const void *gy, const unsigned char *idx, void *gx, int B, int H, int W, int C, int dtype, void *stream) {
    MHE_REQUIRE(gy && idx && gx && B > 0 && H > 0 && W > 0 && C % 4 == 0, "mhe_maxpool3x3s2_bwd_nhwc: bad arguments");
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    const size_t n4 = (size_t)B * H * W * C / 4;
    if (dtype == MHE_F32)
        hipLaunchKernelGGL(tb::maxpool_bwd_kernel<float>, dim3(ewg(n4)), dim3(256), 0, (hipStream_t)stream, (const float *)gy, idx, (float *)gx, B, H, W, C, Ho, Wo);
    else
        hipLaunchKernelGGL(tb::maxpool_bwd_kernel<u16>, dim3(ewg(n4)), dim3(256), 0, (hipStream_t)stream, (const u16 *)gy, idx, (u16 *)gx, B, H, W, C, Ho, Wo);
    return check_launch("maxpool_bwd_kernel");
}

extern "C" int mhe_maxpool3x3s2_idx_affine_nhwc(const void *x, const float *scale, const float *shift, void *y, unsigned char *idx, int B, int H, int W,
                                                int C, int dtype, void *stream) {
    MHE_REQUIRE(x && scale && shift && y && idx && B > 0 && H > 0 && W > 0, "mhe_maxpool3x3s2_idx_affine_nhwc: bad arguments");
    MHE_REQUIRE(dtype == MHE_F32 || dtype == MHE_BF16, "mhe_maxpool3x3s2_idx_affine_nhwc: dtype=%d", dtype);
    const int E = dtype == MHE_F32 ? 4 : 8;
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    MHE_REQUIRE(C > 0 && C % E == 0 && (size_t)B * Ho * Wo * (C / E) < (1ull << 31), "mhe_maxpool3x3s2_idx_affine_nhwc: C=%d must be a multiple of %d (and < 2^31 lanes)", C, E);
    const size_t n = (size_t)B * Ho * Wo * (C / E);
    if (dtype == MHE_F32)
        hipLaunchKernelGGL(tb::maxpool_idx_affine_kernel<float>, dim3(ewg(n)), dim3(256), 0, (hipStream_t)stream, (const float *)x, scale, shift, (float *)y, idx, B, H, W, C, Ho, Wo);
    else
        hipLaunchKernelGGL(tb::maxpool_idx_affine_kernel<u16>, dim3(ewg(n)), dim3(256), 0, (hipStream_t)stream, (const u16 *)x, scale, shift, (u16 *)y, idx, B, H, W, C, Ho, Wo);
    return check_launch("maxpool_idx_affine_kernel");
}

extern "C" int mhe_maxpool3x3s2_idx_affine_win_nhwc(const void *x, const float *scale, const float *shift, void *y, unsigned char *idx, void *xwin, int B,
                                                    int H, int W, int C, int dtype, void *stream) {
    MHE_REQUIRE(x && scale && shift && y && idx && xwin && B > 0 && H > 0 && W > 0, "mhe_maxpool3x3s2_idx_affine_win_nhwc: bad arguments");
    MHE_REQUIRE(dtype == MHE_F32 || dtype == MHE_BF16, "mhe_maxpool3x3s2_idx_affine_win_nhwc: dtype=%d", dtype);
    const int E = dtype == MHE_F32 ? 4 : 8;
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    MHE_REQUIRE(C > 0 && C % E == 0 && (size_t)B * Ho * Wo * (C / E) < (1ull << 31), "mhe_maxpool3x3s2_idx_affine_win_nhwc: C=%d must be a multiple of %d (and < 2^31 lanes)", C, E);
    const size_t n = (size_t)B * Ho * Wo * (C / E);
    if (dtype == MHE_F32)
        hipLaunchKernelGGL(tb::maxpool_idx_affine_kernel<float>, dim3(ewg(n)), dim3(256), 0, (hipStream_t)stream, (const float *)x, scale, shift, (float *)y, idx, B, H, W, C, Ho, Wo, (float *)xwin);
    else
        hipLaunchKernelGGL(tb::maxpool_idx_affine_kernel<u16>, dim3(ewg(n)), dim3(256), 0, (hipStream_t)stream, (const u16 *)x, scale, shift, (u16 *)y, idx, B, H, W, C, Ho, Wo, (u16 *)xwin);
    return check_launch("maxpool_idx_affine_kernel");
}

extern "C" int mhe_pooled_bn_sums_nhwc(const void *g, const void *pooled, const void *xwin, const float *mean_invstd, mhe_stat_t *stats, long P, int C,
                                       int dtype, void *stream) {
    MHE_REQUIRE(g && pooled && xwin && mean_invstd && stats && P > 0, "mhe_pooled_bn_sums_nhwc: bad arguments");
    MHE_REQUIRE(dtype == MHE_F32 || dtype == MHE_BF16, "mhe_pooled_bn_sums_nhwc: dtype=%d", dtype);
    const int E = dtype == MHE_F32 ? 4 : 8;
    MHE_REQUIRE(C > 0 && C % E == 0 && 256 % (C / E) == 0, "mhe_pooled_bn_sums_nhwc: C=%d: C / %d must divide 256 (a thread keeps one channel chunk)", C, E);
    MHE_REQUIRE(((uintptr_t)g | (uintptr_t)pooled | (uintptr_t)xwin) % 16 == 0, "mhe_pooled_bn_sums_nhwc: 16-byte aligned tensors");
    const size_t nl = (size_t)P * C / E;
    // few, long-lived workgroups: every workgroup ends with an LDS fold and 4 x C / 8 fixed-point adds by eight of its threads (8,192
    // one-or-two-iteration workgroups took 143 us for 0.4 GB)
    unsigned blocks = ewg((nl + 1) / 2);
    if (blocks > 2048) blocks = 2048;
    if (dtype == MHE_F32)
        hipLaunchKernelGGL(tb::pooled_bn_sums_kernel<float>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const float *)g, (const float *)pooled, (const float *)xwin,
                           mean_invstd, stats, nl, C);
    else
        hipLaunchKernelGGL(tb::pooled_bn_sums_kernel<u16>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const u16 *)g, (const u16 *)pooled, (const u16 *)xwin,
                           mean_invstd, stats, nl, C);
    return check_launch("pooled_bn_sums_kernel");
}

extern "C" int mhe_maxpool3x3s2_bwd_bn_nhwc(const void *gy, const unsigned char *idx, const void *y, const float *scale, const float *shift,
                                            const float *mean_invstd, mhe_stat_t *stats, void *gx, int B, int H, int W, int C, int dtype, void *stream) {
    MHE_REQUIRE(gy && idx && y && scale && shift && mean_invstd && (stats || gx) && B > 0 && H > 0 && W > 0, "mhe_maxpool3x3s2_bwd_bn_nhwc: bad arguments");
    MHE_REQUIRE(dtype == MHE_F32 || dtype == MHE_BF16, "mhe_maxpool3x3s2_bwd_bn_nhwc: dtype=%d", dtype);
    const int E = dtype == MHE_F32 ? 4 : 8;
    MHE_REQUIRE(C > 0 && C % E == 0 && 256 % (C / E) == 0 && (size_t)B * H * W * (C / E) < (1ull << 31),
                "mhe_maxpool3x3s2_bwd_bn_nhwc: C=%d: C / %d must divide 256 (a thread keeps one channel chunk)", C, E);
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    const size_t n = (size_t)B * H * W * (C / E);
    if (dtype == MHE_F32)
        hipLaunchKernelGGL(tb::maxpool_bwd_bn_kernel<float>, dim3(ewg(n)), dim3(256), 0, (hipStream_t)stream, (const float *)gy, idx, (const float *)y, scale, shift,
                           mean_invstd, stats, (float *)gx, B, H, W, C, Ho, Wo, (const float *)nullptr);
    else
        hipLaunchKernelGGL(tb::maxpool_bwd_bn_kernel<u16>, dim3(ewg(n)), dim3(256), 0, (hipStream_t)stream, (const u16 *)gy, idx, (const u16 *)y, scale, shift,
                           mean_invstd, stats, (u16 *)gx, B, H, W, C, Ho, Wo, (const float *)nullptr);
    return check_launch("maxpool_bwd_bn_kernel");
}

extern "C" int mhe_maxpool3x3s2_bwd_bn_apply_nhwc(const void *gy, const unsigned char *idx, const void *y, const float *scale, const float *shift,
                                                  const float *mean_invstd, const float *coef, void *gy_out, int B, int H, int W, int C, int dtype,
                                                  void *stream) {
    MHE_REQUIRE(gy && idx && y && scale && shift && mean_invstd && coef && gy_out && B > 0 && H > 0 && W > 0, "mhe_maxpool3x3s2_bwd_bn_apply_nhwc: bad arguments");
    MHE_REQUIRE(dtype == MHE_F32 || dtype == MHE_BF16, "mhe_maxpool3x3s2_bwd_bn_apply_nhwc: dtype=%d", dtype);
    const int E = dtype == MHE_F32 ? 4 : 8;
    MHE_REQUIRE(C > 0 && C % E == 0 && 256 % (C / E) == 0 && (size_t)B * H * W * (C / E) < (1ull << 31),
                "mhe_maxpool3x3s2_bwd_bn_apply_nhwc: C=%d: C / %d must divide 256 (a thread keeps one channel chunk)", C, E);
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    const size_t n = (size_t)B * H * W * (C / E);
    if (dtype == MHE_F32)
        hipLaunchKernelGGL(tb::maxpool_bwd_bn_kernel<float>, dim3(ewg(n)), dim3(256), 0, (hipStream_t)stream, (const float *)gy, idx, (const float *)y, scale, shift,
                           mean_invstd, (mhe_stat_t *)nullptr, (float *)gy_out, B, H, W, C, Ho, Wo, coef);
    else
        hipLaunchKernelGGL(tb::maxpool_bwd_bn_kernel<u16>, dim3(ewg(n)), dim3(256), 0, (hipStream_t)stream, (const u16 *)gy, idx, (const u16 *)y, scale, shift,
                           mean_invstd, (mhe_stat_t *)nullptr, (u16 *)gy_out, B, H, W, C, Ho, Wo, coef);
    return check_launch("maxpool_bwd_bn_kernel");
}

extern "C" int mhe_avgpool_bwd_nhwc(const float *g, const void *mask, void *gx, int B, int HW, int C, int dtype, void *stream) {
    MHE_REQUIRE(g && gx && B > 0 && HW > 0 && C > 0 && C % 4 == 0, "mhe_avgpool_bwd_nhwc: bad arguments");
    const size_t n4 = (size_t)B * HW * C / 4;
    if (dtype == MHE_F32)
        hipLaunchKernelGGL(tb::avgpool_bwd_kernel<float>, dim3(ewg(n4)), dim3(256), 0, (hipStream_t)stream, g, (const float *)mask, (float *)gx, HW, C, n4);
    else
        hipLaunchKernelGGL(tb::avgpool_bwd_kernel<u16>, dim3(ewg(n4)), dim3(256), 0, (hipStream_t)stream, g, (const u16 *)mask, (u16 *)gx, HW, C, n4);
    return check_launch("avgpool_bwd_kernel");
}

extern "C" int mhe_upsample2_nhwc(const void *g, const void *base, void *out, int B, int H, int W, int C, int dtype, void *stream) {
    MHE_REQUIRE(g && out && B > 0 && H > 0 && W > 0 && C % 4 == 0, "mhe_upsample2_nhwc: bad arguments");
    const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
    const size_t n4 = (size_t)B * H * W * C / 4;
    if (dtype == MHE_F32)
        hipLaunchKernelGGL(tb::upsample2_kernel<float>, dim3(ewg(n4)), dim3(256), 0, (hipStream_t)stream, (const float *)g, (const float *)base, (float *)out, B, H, W, C, Ho, Wo);
    else
        hipLaunchKernelGGL(tb::upsample2_kernel<u16>, dim3(ewg(n4)), dim3(256), 0, (hipStream_t)stream, (const u16 *)g, (const u16 *)base, (u16 *)out, B, H, W, C, Ho, Wo);
    return check_launch("upsample2_kernel");
}

extern "C" size_t mhe_sqnorm_workspace_floats(void) { return tb::SQ_BLOCKS; }

extern "C" int mhe_sqnorm_f32(const float *g, size_t n, float *workspace, float *out, void *stream) {
    MHE_REQUIRE(g && workspace && out && n > 0 && (uintptr_t)g % 16 == 0, "mhe_sqnorm_f32: bad arguments (g 16-byte aligned)");
    hipLaunchKernelGGL(tb::sqnorm_partial_kernel, dim3(tb::SQ_BLOCKS), dim3(256), 0, (hipStream_t)stream, g, n, workspace);
    hipLaunchKernelGGL(tb::sqnorm_final_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, workspace, out);
    return check_launch("sqnorm_kernel");
}

extern "C" int mhe_train_tick(int *step, float *sqnorm, void *stream) {
    MHE_REQUIRE(step && sqnorm, "mhe_train_tick: null pointer");
    hipLaunchKernelGGL(tb::tick_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, step, sqnorm);
    return check_launch("tick_kernel");
}

extern "C" int mhe_adam_step_f32(float *p, const float *g, float *m, float *v, size_t n, const float *sqnorm, const int *step,
                                 float lr, float beta1, float beta2, float eps, float max_norm, float grad_scale, void *stream) {
    MHE_REQUIRE(p && g && m && v && step && n > 0, "mhe_adam_step_f32: bad arguments");
    MHE_REQUIRE(max_norm <= 0.f || sqnorm, "mhe_adam_step_f32: clipping needs the squared gradient norm");
    size_t blocks = (n + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(tb::adam_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, sqnorm, step, lr,
                       beta1, beta2, eps, max_norm, grad_scale);
    return check_launch("adam_kernel");
}

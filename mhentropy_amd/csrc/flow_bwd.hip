// Elementwise stages of the RealNVP reverse pass (reference hand/flows.py:97-122,210-217 differentiated by
// autograd in hand/CrossModalHand.py:455-470).  The train step walks the couplings last -> first; for each it
// re-evaluates the two coupling nets layer by layer on the coupling's OUTPUT (the masked half is unchanged by
// the coupling, so the nets see exactly their forward input), inverts the affine update to recover the
// coupling's input, and back-propagates.  The dense products run on the implicit-GEMM kernel
// (mhe_conv2d_nhwc as Y = X W^T) and mhe_conv_wgrad_nhwc; this file holds what sits between them.
// All tensors f32; the 45-wide flow variable is zero-padded to 64 columns where it is a GEMM operand.
#include "common.h"
#include "../../include/mhe.h"

namespace mhe { namespace flowbwd {
constexpr int XP = 64;        // padded width of the flow variable as a GEMM operand

__global__ __launch_bounds__(256) void mask_pad_kernel(const float *__restrict__ x, const float *__restrict__ mask,
                                                       float *__restrict__ xp, u16 *__restrict__ xp_b, long R, int dim) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= R * XP) return;
    const long r = i / XP;
    const int d = (int)(i % XP);
    const float v = d < dim ? x[r * dim + d] * mask[d] : 0.f;
    if (xp) xp[i] = v;
    if (xp_b) xp_b[i] = f32_to_bf16(v);
}

// P[r][c] = leaky_relu(P[r][c] + cond[r % B][c], 0.01)      (hand/flows.py:108-117)
__global__ __launch_bounds__(256) void cond_lrelu_kernel(float *__restrict__ P, const float *__restrict__ cond,
                                                         long cond_stride, long R, int B, int H) {
    const long n4 = R * H / 4;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const long r = i / (H / 4);
        const int c = (int)(i % (H / 4)) * 4;
        v4f v = *reinterpret_cast<v4f *>(P + i * 4);
        const v4f b = *reinterpret_cast<const v4f *>(cond + (r % B) * cond_stride + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float t = v[e] + b[e];
            v[e] = t > 0.f ? t : 0.01f * t;
        }
        *reinterpret_cast<v4f *>(P + i * 4) = v;
    }
}

// G[r][c] *= (H[r][c] > 0 ? 1 : slope)   (slope 0.01: leaky_relu of the coupling nets; 0: the det head's ReLU)
__global__ __launch_bounds__(256) void lrelu_bwd_kernel(float *__restrict__ G, const float *__restrict__ Hact, long n4, float slope) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        v4f g = *reinterpret_cast<v4f *>(G + i * 4);
        const v4f h = *reinterpret_cast<const v4f *>(Hact + i * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) g[e] = h[e] > 0.f ? g[e] : slope * g[e];
        *reinterpret_cast<v4f *>(G + i * 4) = g;
    }
}

__global__ __launch_bounds__(256) void add_kernel(const float *__restrict__ a, const float *__restrict__ b, float *__restrict__ out, long n) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) out[i] = a[i] + b[i];
}


// Mixed-precision forms for the bf16 performance mode: the three hidden x hidden products of each net (forward l1, its
// data gradient and its weight gradient: 85 % of the reverse pass's FLOP) run on bf16 MFMA, so their operands
// are kept as bf16 copies while everything that feeds an exp/tanh or a per-image reduction stays f32.
template <typename T> __device__ __forceinline__ v4f ld4(const T *p);
template <> __device__ __forceinline__ v4f ld4<float>(const float *p) { return *reinterpret_cast<const v4f *>(p); }
template <> __device__ __forceinline__ v4f ld4<u16>(const u16 *p) {
    const uint2 r = *reinterpret_cast<const uint2 *>(p);
    v4f o;
    o[0] = __uint_as_float(r.x << 16); o[1] = __uint_as_float(r.x & 0xffff0000u);
    o[2] = __uint_as_float(r.y << 16); o[3] = __uint_as_float(r.y & 0xffff0000u);
    return o;
}
__device__ __forceinline__ void st4b(u16 *p, const v4f &v) {
    uint2 o;
    o.x = (unsigned)f32_to_bf16(v[0]) | ((unsigned)f32_to_bf16(v[1]) << 16);
    o.y = (unsigned)f32_to_bf16(v[2]) | ((unsigned)f32_to_bf16(v[3]) << 16);
    *reinterpret_cast<uint2 *>(p) = o;
}

// out = leaky_relu(pre + cond[r % B]) written as f32 and / or bf16
template <typename TI>
__global__ __launch_bounds__(256) void cond_lrelu_mixed_kernel(const TI *__restrict__ pre, const float *__restrict__ cond,
                                                               long cond_stride, float *__restrict__ out_f, u16 *__restrict__ out_b,
                                                               long R, int B, int H) {
    const long n4 = R * H / 4;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const long r = i / (H / 4);
        const int c = (int)(i % (H / 4)) * 4;
        v4f v = ld4<TI>(pre + i * 4);
        const v4f b = *reinterpret_cast<const v4f *>(cond + (r % B) * cond_stride + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float t = v[e] + b[e];
            v[e] = t > 0.f ? t : 0.01f * t;
        }
        if (out_f) *reinterpret_cast<v4f *>(out_f + i * 4) = v;
        if (out_b) st4b(out_b + i * 4, v);
    }
}

// out = g * (h > 0 ? 1 : slope) written as f32 and / or bf16
template <typename TG, typename TH>
__global__ __launch_bounds__(256) void lrelu_bwd_mixed_kernel(const TG *__restrict__ g, const TH *__restrict__ h,
                                                              float *__restrict__ out_f, u16 *__restrict__ out_b, long n4, float slope) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        v4f gv = ld4<TG>(g + i * 4);
        const v4f hv = ld4<TH>(h + i * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) gv[e] = hv[e] > 0.f ? gv[e] : slope * gv[e];
        if (out_f) *reinterpret_cast<v4f *>(out_f + i * 4) = gv;
        if (out_b) st4b(out_b + i * 4, gv);
    }
}

// out[n*B+b][c] = g * (h > 0 ? 1 : slope)  AND  sum_out[b][c] = sum_n out[n*B+b][c]  (the gradient of the per-image conditioning
// term): one thread per (image, column pair) walks the N hypotheses - the separate sum-over-hypotheses pass re-read the whole
// f32 tensor (48 launches of ~17 us per train step at the bench size)
template <typename TG, typename TH>
__global__ __launch_bounds__(256) void lrelu_bwd_sum_kernel(const TG *__restrict__ g, const TH *__restrict__ h, float *__restrict__ out_f,
                                                            u16 *__restrict__ out_b, float *__restrict__ sum_out, long sum_stride,
                                                            float *__restrict__ sum_t, int N, int B, int H, float slope) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int hc = H / 2;
    if (t >= B * hc) return;
    const int b = t / hc, c = (t % hc) * 2;
    float a0 = 0.f, a1 = 0.f;
#pragma unroll 8
    for (int n = 0; n < N; ++n) {
        const size_t e = ((size_t)n * B + b) * H + c;
        float g0, g1, h0, h1;
        if constexpr (sizeof(TG) == 4) { const float2 v = *reinterpret_cast<const float2 *>(g + e); g0 = v.x; g1 = v.y; }
        else { const unsigned v = *reinterpret_cast<const unsigned *>(g + e); g0 = __uint_as_float(v << 16); g1 = __uint_as_float(v & 0xffff0000u); }
        if constexpr (sizeof(TH) == 4) { const float2 v = *reinterpret_cast<const float2 *>(h + e); h0 = v.x; h1 = v.y; }
        else { const unsigned v = *reinterpret_cast<const unsigned *>(h + e); h0 = __uint_as_float(v << 16); h1 = __uint_as_float(v & 0xffff0000u); }
        g0 = h0 > 0.f ? g0 : slope * g0;
        g1 = h1 > 0.f ? g1 : slope * g1;
        if (out_f) *reinterpret_cast<float2 *>(out_f + e) = make_float2(g0, g1);
        if (out_b) *reinterpret_cast<unsigned *>(out_b + e) = (unsigned)f32_to_bf16(g0) | ((unsigned)f32_to_bf16(g1) << 16);
        a0 += g0; a1 += g1;
    }
    *reinterpret_cast<float2 *>(sum_out + (size_t)b * sum_stride + c) = make_float2(a0, a1);
    if (sum_t) { sum_t[(size_t)c * B + b] = a0; sum_t[(size_t)(c + 1) * B + b] = a1; }      // the same, [column][image]
}

// the same at H = 512 with bf16 in and out (the train step's case): one workgroup per image, a wave reads whole 1 KiB rows with 16-byte
// lanes and takes every 8th hypothesis, partial sums meet in LDS - 2,048 waves with 16-byte accesses instead of 1,024 with 4-byte ones
__global__ __launch_bounds__(512) void lrelu_bwd_sum512_kernel(const u16 *__restrict__ g, const u16 *__restrict__ h, u16 *__restrict__ out_b,
                                                               float *__restrict__ sum_out, long sum_stride, float *__restrict__ sum_t,
                                                               int N, int B, float slope) {
    __shared__ float part[8][512];
    const int b = blockIdx.x, w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float a[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) a[k] = 0.f;
#pragma unroll 4
    for (int n = w; n < N; n += 8) {
        const size_t e = ((size_t)n * B + b) * 512 + lane * 8;
        const uint4 gv = *reinterpret_cast<const uint4 *>(g + e), hv = *reinterpret_cast<const uint4 *>(h + e);
        const unsigned gw[4] = {gv.x, gv.y, gv.z, gv.w}, hw[4] = {hv.x, hv.y, hv.z, hv.w};
        unsigned ow[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float g0 = __uint_as_float(gw[k] << 16), g1 = __uint_as_float(gw[k] & 0xffff0000u);
            const float h0 = __uint_as_float(hw[k] << 16), h1 = __uint_as_float(hw[k] & 0xffff0000u);
            g0 = h0 > 0.f ? g0 : slope * g0;
            g1 = h1 > 0.f ? g1 : slope * g1;
            ow[k] = (unsigned)f32_to_bf16(g0) | ((unsigned)f32_to_bf16(g1) << 16);
            a[2 * k] += g0; a[2 * k + 1] += g1;
        }
        *reinterpret_cast<uint4 *>(out_b + e) = make_uint4(ow[0], ow[1], ow[2], ow[3]);
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) part[w][lane * 8 + k] = a[k];
    __syncthreads();
    const int c = threadIdx.x;
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) s += part[k][c];
    sum_out[(size_t)b * sum_stride + c] = s;
    if (sum_t) sum_t[(size_t)c * B + b] = s;
}

// One coupling, reverse: from its output x_out and the nets' raw outputs Os, Ot (bias included, 64-wide)
//   s = tanh(Os)(1-m), t = Ot(1-m), x_in = m x_out + (1-m)(x_out - t) e^{-s}          (hand/flows.py:213-216)
// and the adjoints of x_out (g_out) and of log q (a_q per row; log q = logN(z0) - sum s):
//   GOs = (1-m)(g_out x_in e^s - a_q)(1 - tanh^2),  GOt = (1-m) g_out,  g_part = g_out (m + (1-m) e^s)
// a workgroup takes 64 rows: thread (rg = tid / 64, d = tid % 64) walks rows rg, rg + 4, ...; the column sums of GOs / GOt (the l2 bias
// gradients of the two nets, formerly two separate column-sum launches per coupling) are folded through LDS and added atomically
__global__ __launch_bounds__(256) void couple_bwd_kernel(
    const float *__restrict__ x_out, const float *__restrict__ Os, const float *__restrict__ Ot,
    const float *__restrict__ mask, const float *__restrict__ g_out, const float *__restrict__ g_logp, float q_weight,
    float *__restrict__ x_in, float *__restrict__ GOs, float *__restrict__ GOt, float *__restrict__ g_part,
    long R, int B, int dim, u16 *__restrict__ GOs_b, u16 *__restrict__ GOt_b, float *__restrict__ db_s, float *__restrict__ db_t) {
    __shared__ float red[2][4][XP];
    const int d = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const long r0 = (long)blockIdx.x * 64;
    const float m = d < dim ? mask[d] : 1.f;
    float cs = 0.f, ct = 0.f;
    for (int k = 0; k < 16; ++k) {
        const long r = r0 + rg + 4 * k;
        if (r >= R) break;
        const long i = r * XP + d;
        float gos = 0.f, got = 0.f;
        if (d < dim) {
            const float xo = x_out[r * dim + d], go = g_out[r * dim + d];
            float xi = xo, gp = go;
            if (m == 0.f) {
                const float s = tanhf(Os[i]), t = Ot[i];
                const float es = expf(s);
                xi = (xo - t) / es;
                const float a_q = g_logp ? g_logp[r % B] * q_weight : 0.f;
                gos = (go * xi * es - a_q) * (1.f - s * s);
                got = go;
                gp = go * es;
            }
            x_in[r * dim + d] = xi;
            g_part[r * dim + d] = gp;
        }
        GOs[i] = gos;
        GOt[i] = got;
        if (GOs_b) { GOs_b[i] = f32_to_bf16(gos); GOt_b[i] = f32_to_bf16(got); }
        cs += gos; ct += got;
    }
    if (db_s) {
        red[0][rg][d] = cs; red[1][rg][d] = ct;
        __syncthreads();
        if (threadIdx.x < 128) {
            const int w = threadIdx.x >> 6;
            const float v = red[w][0][d] + red[w][1][d] + red[w][2][d] + red[w][3][d];
            atomicAdd((w ? db_t : db_s) + d, v);
        }
    }
}

// g_in = g_part + m (GXs + GXt): the nets' input is m * x
__global__ __launch_bounds__(256) void couple_accum_kernel(const float *__restrict__ g_part, const float *__restrict__ GXs,
                                                           const float *__restrict__ GXt, const float *__restrict__ mask,
                                                           float *__restrict__ g_in, long R, int dim) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= R * dim) return;
    const long r = i / dim;
    const int d = (int)(i % dim);
    g_in[i] = g_part[i] + mask[d] * (GXs[r * XP + d] + GXt[r * XP + d]);
}
}}  // namespace mhe::flowbwd

using namespace mhe;
static inline unsigned ew_grid(long n) { long b = (n + 255) / 256; return (unsigned)(b < 1 ? 1 : (b > 16384 ? 16384 : b)); }

extern "C" int mhe_flow_mask_pad_mixed(const float *x, const float *mask, float *xp, void *xp_bf16, long R, int dim, void *stream) {
    MHE_REQUIRE(x && mask && (xp || xp_bf16) && R > 0 && dim > 0 && dim <= flowbwd::XP, "mhe_flow_mask_pad: bad arguments");
    hipLaunchKernelGGL(flowbwd::mask_pad_kernel, dim3((unsigned)((R * flowbwd::XP + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, mask, xp,
                       (u16 *)xp_bf16, R, dim);
    return check_launch("mask_pad_kernel");
}

extern "C" int mhe_flow_mask_pad_f32(const float *x, const float *mask, float *xp, long R, int dim, void *stream) {
    return mhe_flow_mask_pad_mixed(x, mask, xp, nullptr, R, dim, stream);
}

extern "C" int mhe_flow_cond_lrelu_f32(float *P, const float *cond, long cond_stride, long R, int B, int H, void *stream) {
    MHE_REQUIRE(P && cond && R > 0 && B > 0 && H > 0 && H % 4 == 0 && cond_stride % 4 == 0, "mhe_flow_cond_lrelu_f32: bad arguments");
    hipLaunchKernelGGL(flowbwd::cond_lrelu_kernel, dim3(ew_grid(R * H / 4)), dim3(256), 0, (hipStream_t)stream, P, cond, cond_stride, R, B, H);
    return check_launch("cond_lrelu_kernel");
}

extern "C" int mhe_add_f32(const float *a, const float *b, float *out, long n, void *stream) {
    MHE_REQUIRE(a && b && out && n > 0, "mhe_add_f32: bad arguments");
    hipLaunchKernelGGL(flowbwd::add_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, a, b, out, n);
    return check_launch("add_kernel");
}

extern "C" int mhe_flow_lrelu_bwd_f32(float *G, const float *Hact, long n, float slope, void *stream) {
    MHE_REQUIRE(G && Hact && n > 0 && n % 4 == 0, "mhe_flow_lrelu_bwd_f32: bad arguments");
    hipLaunchKernelGGL(flowbwd::lrelu_bwd_kernel, dim3(ew_grid(n / 4)), dim3(256), 0, (hipStream_t)stream, G, Hact, n / 4, slope);
    return check_launch("lrelu_bwd_kernel");
}

extern "C" int mhe_flow_couple_bwd_mixed(const float *x_out, const float *Os, const float *Ot, const float *mask,
                                         const float *g_out, const float *g_log_p, float q_weight, float *x_in, float *GOs,
                                         float *GOt, float *g_part, void *GOs_bf16, void *GOt_bf16, float *db_s, float *db_t, long R, int B,
                                         int dim, void *stream) {
    MHE_REQUIRE(x_out && Os && Ot && mask && g_out && x_in && GOs && GOt && g_part, "mhe_flow_couple_bwd: null pointer");
    MHE_REQUIRE((GOs_bf16 == nullptr) == (GOt_bf16 == nullptr), "mhe_flow_couple_bwd: the bf16 copies come together");
    MHE_REQUIRE((db_s == nullptr) == (db_t == nullptr), "mhe_flow_couple_bwd: the two bias-gradient accumulators come together");
    MHE_REQUIRE(R > 0 && B > 0 && R % B == 0 && dim > 0 && dim <= flowbwd::XP, "mhe_flow_couple_bwd: bad sizes");
    hipLaunchKernelGGL(flowbwd::couple_bwd_kernel, dim3((unsigned)((R + 63) / 64)), dim3(256), 0, (hipStream_t)stream,
                       x_out, Os, Ot, mask, g_out, g_log_p, q_weight, x_in, GOs, GOt, g_part, R, B, dim, (u16 *)GOs_bf16, (u16 *)GOt_bf16, db_s, db_t);
    return check_launch("couple_bwd_kernel");
}

extern "C" int mhe_flow_couple_bwd_f32(const float *x_out, const float *Os, const float *Ot, const float *mask,
                                       const float *g_out, const float *g_log_p, float q_weight, float *x_in, float *GOs,
                                       float *GOt, float *g_part, long R, int B, int dim, void *stream) {
    return mhe_flow_couple_bwd_mixed(x_out, Os, Ot, mask, g_out, g_log_p, q_weight, x_in, GOs, GOt, g_part, nullptr, nullptr, nullptr, nullptr, R, B, dim,
                                     stream);
}

extern "C" int mhe_flow_lrelu_bwd_sum(const void *g, int g_dtype, const void *h, int h_dtype, float *out_f32, void *out_bf16,
                                      float *sum_out, long sum_stride, float *sum_out_t, int N, int B, int H, float slope, void *stream) {
    MHE_REQUIRE(g && h && sum_out && N > 0 && B > 0 && H > 0 && H % 2 == 0 && sum_stride >= H && sum_stride % 2 == 0,
                "mhe_flow_lrelu_bwd_sum: bad arguments");
    const dim3 grid((unsigned)(((long)B * (H / 2) + 255) / 256)), block(256);
    hipStream_t s = (hipStream_t)stream;
    u16 *ob = (u16 *)out_bf16;
    if (g_dtype == MHE_F32 && h_dtype == MHE_F32)
        hipLaunchKernelGGL((flowbwd::lrelu_bwd_sum_kernel<float, float>), grid, block, 0, s, (const float *)g, (const float *)h, out_f32, ob, sum_out, sum_stride, sum_out_t, N, B, H, slope);
    else if (g_dtype == MHE_F32)
        hipLaunchKernelGGL((flowbwd::lrelu_bwd_sum_kernel<float, u16>), grid, block, 0, s, (const float *)g, (const u16 *)h, out_f32, ob, sum_out, sum_stride, sum_out_t, N, B, H, slope);
    else if (h_dtype == MHE_F32)
        hipLaunchKernelGGL((flowbwd::lrelu_bwd_sum_kernel<u16, float>), grid, block, 0, s, (const u16 *)g, (const float *)h, out_f32, ob, sum_out, sum_stride, sum_out_t, N, B, H, slope);
    else if (H == 512 && ob && !out_f32)
        hipLaunchKernelGGL(flowbwd::lrelu_bwd_sum512_kernel, dim3(B), dim3(512), 0, s, (const u16 *)g, (const u16 *)h, ob, sum_out, sum_stride, sum_out_t, N, B, slope);
    else
        hipLaunchKernelGGL((flowbwd::lrelu_bwd_sum_kernel<u16, u16>), grid, block, 0, s, (const u16 *)g, (const u16 *)h, out_f32, ob, sum_out, sum_stride, sum_out_t, N, B, H, slope);
    return check_launch("lrelu_bwd_sum_kernel");
}

extern "C" int mhe_flow_couple_accum_f32(const float *g_part, const float *GXs, const float *GXt, const float *mask,
                                         float *g_in, long R, int dim, void *stream) {
    MHE_REQUIRE(g_part && GXs && GXt && mask && g_in && R > 0 && dim > 0 && dim <= flowbwd::XP, "mhe_flow_couple_accum_f32: bad arguments");
    hipLaunchKernelGGL(flowbwd::couple_accum_kernel, dim3((unsigned)((R * dim + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       g_part, GXs, GXt, mask, g_in, R, dim);
    return check_launch("couple_accum_kernel");
}

extern "C" int mhe_flow_cond_lrelu_mixed(const void *pre, int pre_dtype, const float *cond, long cond_stride, float *out_f32,
                                         void *out_bf16, long R, int B, int H, void *stream) {
    MHE_REQUIRE(pre && cond && (out_f32 || out_bf16) && R > 0 && B > 0 && H > 0 && H % 4 == 0 && cond_stride % 4 == 0,
                "mhe_flow_cond_lrelu_mixed: bad arguments");
    if (pre_dtype == MHE_F32)
        hipLaunchKernelGGL(flowbwd::cond_lrelu_mixed_kernel<float>, dim3(ew_grid(R * H / 4)), dim3(256), 0, (hipStream_t)stream,
                           (const float *)pre, cond, cond_stride, out_f32, (u16 *)out_bf16, R, B, H);
    else
        hipLaunchKernelGGL(flowbwd::cond_lrelu_mixed_kernel<u16>, dim3(ew_grid(R * H / 4)), dim3(256), 0, (hipStream_t)stream,
                           (const u16 *)pre, cond, cond_stride, out_f32, (u16 *)out_bf16, R, B, H);
    return check_launch("cond_lrelu_mixed_kernel");
}

extern "C" int mhe_flow_lrelu_bwd_mixed(const void *g, int g_dtype, const void *h, int h_dtype, float *out_f32, void *out_bf16,
                                        long n, float slope, void *stream) {
    MHE_REQUIRE(g && h && (out_f32 || out_bf16) && n > 0 && n % 4 == 0, "mhe_flow_lrelu_bwd_mixed: bad arguments");
    const dim3 grid(ew_grid(n / 4)), block(256);
    hipStream_t s = (hipStream_t)stream;
    u16 *ob = (u16 *)out_bf16;
    if (g_dtype == MHE_F32 && h_dtype == MHE_F32)
        hipLaunchKernelGGL((flowbwd::lrelu_bwd_mixed_kernel<float, float>), grid, block, 0, s, (const float *)g, (const float *)h, out_f32, ob, n / 4, slope);
    else if (g_dtype == MHE_F32)
        hipLaunchKernelGGL((flowbwd::lrelu_bwd_mixed_kernel<float, u16>), grid, block, 0, s, (const float *)g, (const u16 *)h, out_f32, ob, n / 4, slope);
    else if (h_dtype == MHE_F32)
        hipLaunchKernelGGL((flowbwd::lrelu_bwd_mixed_kernel<u16, float>), grid, block, 0, s, (const u16 *)g, (const float *)h, out_f32, ob, n / 4, slope);
    else
        hipLaunchKernelGGL((flowbwd::lrelu_bwd_mixed_kernel<u16, u16>), grid, block, 0, s, (const u16 *)g, (const u16 *)h, out_f32, ob, n / 4, slope);
    return check_launch("lrelu_bwd_mixed_kernel");
}

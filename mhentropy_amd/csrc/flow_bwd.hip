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
                                                       float *__restrict__ xp, long R, int dim) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= R * XP) return;
    const long r = i / XP;
    const int d = (int)(i % XP);
    xp[i] = d < dim ? x[r * dim + d] * mask[d] : 0.f;
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

// One coupling, reverse: from its output x_out and the nets' raw outputs Os, Ot (bias included, 64-wide)
//   s = tanh(Os)(1-m), t = Ot(1-m), x_in = m x_out + (1-m)(x_out - t) e^{-s}          (hand/flows.py:213-216)
// and the adjoints of x_out (g_out) and of log q (a_q per row; log q = logN(z0) - sum s):
//   GOs = (1-m)(g_out x_in e^s - a_q)(1 - tanh^2),  GOt = (1-m) g_out,  g_part = g_out (m + (1-m) e^s)
__global__ __launch_bounds__(256) void couple_bwd_kernel(
    const float *__restrict__ x_out, const float *__restrict__ Os, const float *__restrict__ Ot,
    const float *__restrict__ mask, const float *__restrict__ g_out, const float *__restrict__ g_logp, float q_weight,
    float *__restrict__ x_in, float *__restrict__ GOs, float *__restrict__ GOt, float *__restrict__ g_part,
    long R, int B, int dim) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= R * XP) return;
    const long r = i / XP;
    const int d = (int)(i % XP);
    float gos = 0.f, got = 0.f;
    if (d < dim) {
        const float m = mask[d], xo = x_out[r * dim + d], go = g_out[r * dim + d];
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

extern "C" int mhe_flow_mask_pad_f32(const float *x, const float *mask, float *xp, long R, int dim, void *stream) {
    MHE_REQUIRE(x && mask && xp && R > 0 && dim > 0 && dim <= flowbwd::XP, "mhe_flow_mask_pad_f32: bad arguments");
    hipLaunchKernelGGL(flowbwd::mask_pad_kernel, dim3((unsigned)((R * flowbwd::XP + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, mask, xp, R, dim);
    return check_launch("mask_pad_kernel");
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

extern "C" int mhe_flow_couple_bwd_f32(const float *x_out, const float *Os, const float *Ot, const float *mask,
                                       const float *g_out, const float *g_log_p, float q_weight, float *x_in, float *GOs,
                                       float *GOt, float *g_part, long R, int B, int dim, void *stream) {
    MHE_REQUIRE(x_out && Os && Ot && mask && g_out && x_in && GOs && GOt && g_part, "mhe_flow_couple_bwd_f32: null pointer");
    MHE_REQUIRE(R > 0 && B > 0 && R % B == 0 && dim > 0 && dim <= flowbwd::XP, "mhe_flow_couple_bwd_f32: bad sizes");
    hipLaunchKernelGGL(flowbwd::couple_bwd_kernel, dim3((unsigned)((R * flowbwd::XP + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       x_out, Os, Ot, mask, g_out, g_log_p, q_weight, x_in, GOs, GOt, g_part, R, B, dim);
    return check_launch("couple_bwd_kernel");
}

extern "C" int mhe_flow_couple_accum_f32(const float *g_part, const float *GXs, const float *GXt, const float *mask,
                                         float *g_in, long R, int dim, void *stream) {
    MHE_REQUIRE(g_part && GXs && GXt && mask && g_in && R > 0 && dim > 0 && dim <= flowbwd::XP, "mhe_flow_couple_accum_f32: bad arguments");
    hipLaunchKernelGGL(flowbwd::couple_accum_kernel, dim3((unsigned)((R * dim + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       g_part, GXs, GXt, mask, g_in, R, dim);
    return check_launch("couple_accum_kernel");
}

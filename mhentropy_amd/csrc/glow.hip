// Elementwise stages of the conditional Glow (ActNorm -> LU linear -> affine coupling with a context-conditioned
// residual MLP), the flow the reference builds for q_z_giv_i_model == 'glow' (hand/network.py:342-344; call sites
// :693-694, :736-742).  The class itself (ConditionalGlow of the unpinned git dependency nkolot/nflows,
// hand/environment.yml:284) is absent from the reference tree: these kernels follow the published nflows
// algorithm restated in oracle/glow_ref.py - PARITY UNPINNED.  The dense products run on mhe_linear_f32; the
// context-only terms (initial-layer context columns, GLU gates) are evaluated once per image and indexed here by
// image = (row / row_div) % n_img  (row_div = 1: sample-major rows n*B+b; row_div = N: nflows' batch-major b*N+n).
#include "common.h"
#include "../../include/mhe.h"

namespace mhe { namespace glow {

// H[r][c] += img[(image of r)][c]
__global__ __launch_bounds__(256) void add_image_rows_kernel(float *__restrict__ H, const float *__restrict__ img, long img_stride,
                                                             long R, int C, int row_div, int n_img) {
    const long n4 = R * C / 4;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const long r = i / (C / 4);
        const int c = (int)(i % (C / 4)) * 4;
        v4f v = *reinterpret_cast<v4f *>(H + i * 4);
        const v4f b = *reinterpret_cast<const v4f *>(img + ((r / row_div) % n_img) * img_stride + c);
        v += b;
        *reinterpret_cast<v4f *>(H + i * 4) = v;
    }
}

template <typename TO>
__global__ __launch_bounds__(256) void relu_copy_kernel(const float *__restrict__ in, TO *__restrict__ out, long n4) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        float v[4];
        load4<float>(in + i * 4, v);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
        store4<TO>(out + i * 4, v);
    }
}

// residual block tail: H += T * sigmoid(gate[(image of r)])      (F.glu(cat(T, gate)) = T * sigmoid(gate))
template <typename TT>
__global__ __launch_bounds__(256) void glu_residual_kernel(float *__restrict__ H, const TT *__restrict__ T,
                                                           const float *__restrict__ gate, long gate_stride, long R, int C,
                                                           int row_div, int n_img) {
    const long n4 = R * C / 4;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const long r = i / (C / 4);
        const int c = (int)(i % (C / 4)) * 4;
        v4f h = *reinterpret_cast<v4f *>(H + i * 4);
        float t[4];
        load4<TT>(T + i * 4, t);
        const v4f g = *reinterpret_cast<const v4f *>(gate + ((r / row_div) % n_img) * gate_stride + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) h[e] += t[e] / (1.f + expf(-g[e]));
        *reinterpret_cast<v4f *>(H + i * 4) = h;
    }
}

// Affine coupling on a [R,64]-padded variable u with the net's raw output prm [R,64] = [shift(T) | unconstrained scale(T)]:
//   scale = sigmoid(us + 2) + 1e-3;  forward: y_t = u_t * scale + shift, logdet += sum log scale
//                                    inverse: y_t = (u_t - shift) / scale, logdet -= sum log scale
// transform feature j (0..T-1) is column first + 2*j (the alternating mask); identity columns are copied.
// (ld / ldp: row pitches of the padded variable and of the parameter rows, multiples of 64 - 64 for the 45-D hand flow, 192 for
// a 144-D body pose)
__global__ __launch_bounds__(256) void coupling_kernel(const float *__restrict__ u, const float *__restrict__ prm, float *__restrict__ y,
                                                       float *__restrict__ logdet, long R, int dim, int first, int T, int inverse, int ld,
                                                       int ldp) {
    const int lane = threadIdx.x & 63;
    const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= R) return;
    float ls = 0.f;
    for (int c = lane; c < ld; c += 64) {
        float v = c < dim ? u[r * ld + c] : 0.f;
        const int j = (c - first) >> 1;
        if (c < dim && c >= first && ((c - first) & 1) == 0 && j < T) {
            const float shift = prm[r * ldp + j], us = prm[r * ldp + T + j];
            const float scale = 1.f / (1.f + expf(-(us + 2.f))) + 1e-3f;
            ls += logf(scale);
            v = inverse ? (v - shift) / scale : v * scale + shift;
        }
        y[r * ld + c] = v;
    }
    ls = wave_sum(ls);
    if (lane == 0) logdet[r] += inverse ? -ls : ls;
}

// Reverse of the INVERSE coupling (the sampling direction the train step differentiates):
//   y_t = (v_t - shift) / scale, scale = sigmoid(us + 2) + 1e-3, and log q gains + sum_t log scale
// given g_y = dL/dy and a_q = dL/dlog q (= g_log_p[r % B] * q_weight): g_v (identity columns pass through),
// g_prm [R,64] = [d/d shift (T) | d/d us (T) | 0].
__global__ __launch_bounds__(256) void coupling_inv_bwd_kernel(const float *__restrict__ v, const float *__restrict__ prm,
                                                               const float *__restrict__ g_y, const float *__restrict__ g_logp,
                                                               float q_weight, float *__restrict__ g_v, float *__restrict__ g_prm, long R,
                                                               int B, int dim, int first, int T) {
    const int lane = threadIdx.x & 63;
    const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= R) return;
    const float gy = lane < dim ? g_y[r * 64 + lane] : 0.f;
    float gv = gy;
    if (lane >= 2 * T) g_prm[r * 64 + lane] = 0.f;
    const int j = (lane - first) >> 1;
    if (lane < dim && lane >= first && ((lane - first) & 1) == 0 && j < T) {
        const float shift = prm[r * 64 + j], us = prm[r * 64 + T + j];
        const float sig = 1.f / (1.f + expf(-(us + 2.f))), scale = sig + 1e-3f;
        const float a_q = g_logp ? g_logp[r % B] * q_weight : 0.f;
        gv = gy / scale;
        g_prm[r * 64 + j] = -gv;
        g_prm[r * 64 + T + j] = (-gv * (v[r * 64 + lane] - shift) / scale + a_q / scale) * sig * (1.f - sig);
    }
    g_v[r * 64 + lane] = lane < dim ? gv : 0.f;
}

// residual block tail reverse: H_out = H_in + T3 * sigmoid(gate[image]):  g_t3 = g_h * s,  g_gate_rows = g_h * T3 * s (1 - s)
template <typename TT>       // storage type of t3 and g_t3 (operands of the bf16 products in performance mode); g_gate stays f32
__global__ __launch_bounds__(256) void glu_bwd_kernel(const float *__restrict__ g_h, const TT *__restrict__ t3,
                                                      const float *__restrict__ gate, long gate_stride, TT *__restrict__ g_t3,
                                                      float *__restrict__ g_gate, long R, int C, int row_div, int n_img) {
    const long n4 = R * C / 4;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const long r = i / (C / 4);
        const int c = (int)(i % (C / 4)) * 4;
        const v4f gh = *reinterpret_cast<const v4f *>(g_h + i * 4);
        float t[4], a[4];
        load4<TT>(t3 + i * 4, t);
        const v4f g = *reinterpret_cast<const v4f *>(gate + ((r / row_div) % n_img) * gate_stride + c);
        v4f b;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float s = 1.f / (1.f + expf(-g[e]));
            a[e] = gh[e] * s;
            b[e] = gh[e] * t[e] * s * (1.f - s);
        }
        store4<TT>(g_t3 + i * 4, a);
        *reinterpret_cast<v4f *>(g_gate + i * 4) = b;
    }
}

// acc += g * [h > 0]   (reverse of t = relu(h) feeding a layer, accumulated onto the residual path's gradient)
template <typename TG, typename TH = float>
__global__ __launch_bounds__(256) void relu_bwd_add_kernel(float *__restrict__ acc, const TG *__restrict__ g, const TH *__restrict__ h, long n4) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        v4f a = *reinterpret_cast<v4f *>(acc + i * 4);
        float gg[4], hh[4];
        load4<TG>(g + i * 4, gg);
        load4<TH>(h + i * 4, hh);
#pragma unroll
        for (int e = 0; e < 4; ++e) a[e] += hh[e] > 0.f ? gg[e] : 0.f;
        *reinterpret_cast<v4f *>(acc + i * 4) = a;
    }
}

// x [R,dim] <-> xp [R,64] zero padded; and the base density: out[r] = -|z|^2/2 - dim/2 log(2 pi) + sign * logdet[r] + const
__global__ __launch_bounds__(256) void pad64_kernel(const float *__restrict__ x, float *__restrict__ xp, long R, int dim, int ld) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= R * ld) return;
    const int d = (int)(i % ld);
    xp[i] = d < dim ? x[(i / ld) * dim + d] : 0.f;
}

__global__ __launch_bounds__(256) void finish_kernel(const float *__restrict__ zp, const float *__restrict__ vp, const float *__restrict__ logdet,
                                                     float *__restrict__ v_out, float *__restrict__ logp, long R, int dim, float sign,
                                                     float logdet_const, int ld, const float *__restrict__ const_parts, int n_parts) {
    const int lane = threadIdx.x & 63;
    const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= R) return;
    // the per-layer constants sum(log_scale) + sum(log diag U) as mhe_glow_affine_f64 left them on the device: + in the density direction,
    // - in the sampling direction, i.e. sign * sum inside the bracket below
    for (int i = 0; i < n_parts; ++i) logdet_const += sign * const_parts[i];
    float sq = 0.f;
    for (int c = lane; c < dim; c += 64) {
        const float z = zp[r * ld + c];
        sq = fmaf(z, z, sq);
        if (v_out) v_out[r * dim + c] = vp[r * ld + c];
    }
    sq = wave_sum(sq);
    if (lane == 0) logp[r] = -0.5f * sq - 0.5f * dim * 1.8378770664093453f + sign * (logdet[r] + logdet_const);
}
// ---- reverse stages of the one-launch kernel's tape (round 5): a workgroup owns ONE IMAGE = its N sample-major rows r = n B + b, so the sums
// over an image's hypotheses (gate gradients, per-image bias-gradient rows) fall out of the same pass that writes the gradient - the
// glu_bwd + sum_over_hypotheses + colsum (and dropout + relu-mask + colsum) chains were 6 launches and 5 passes over [R, 512] per block.
//   glu_bwd_sum:     g_t3 = g_h s (bf16),  gct[b] = s (1 - s) sum_n g_h t3,  bsum[b] = s sum_n g_h          (s = sigmoid(gate[b]))
//   mask_scale_sum:  g <- g scale [t2 > 0] in place (t2 = dropout(relu(.)): zero where dropped or inactive),  bsum[b] = sum_n g
// 512 threads: column quad c = t % (C / 4), row lane t / (C / 4); four rows per lane in flight; lanes folded through LDS in lane order.
template <bool GLU>
__global__ __launch_bounds__(512) void image_rows_bwd_kernel(const float *__restrict__ g_h, u16 *__restrict__ g_io, const u16 *__restrict__ act,
                                                             const float *__restrict__ gate, long gate_stride, float scale, float *__restrict__ gct,
                                                             long gct_stride, float *__restrict__ bsum, long bsum_stride, int N, int B, int C) {
    __shared__ float part[2][512][4];
    const int b = blockIdx.x, quads = C / 4, nl = 512 / quads, c = (threadIdx.x % quads) * 4, l0 = threadIdx.x / quads;
    float s[4] = {1.f, 1.f, 1.f, 1.f};
    if constexpr (GLU) {
        const v4f gt = *reinterpret_cast<const v4f *>(gate + (size_t)b * gate_stride + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) s[e] = 1.f / (1.f + expf(-gt[e]));
    }
    float ag[4] = {0.f, 0.f, 0.f, 0.f}, ab[4] = {0.f, 0.f, 0.f, 0.f};
    for (int n0 = l0; n0 < N; n0 += 4 * nl) {
        float gh[4][4], tv[4][4];
        bool on[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int n = n0 + u * nl;
            on[u] = n < N;
            const size_t o = ((size_t)(on[u] ? n : 0) * B + b) * C + c;
            if constexpr (GLU) load4<float>(g_h + o, gh[u]); else load4<u16>(g_io + o, gh[u]);
            load4<u16>(act + o, tv[u]);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (!on[u]) continue;
            const size_t o = ((size_t)(n0 + u * nl) * B + b) * C + c;
            float out[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if constexpr (GLU) { out[e] = gh[u][e] * s[e]; ag[e] = fmaf(gh[u][e], tv[u][e], ag[e]); ab[e] += gh[u][e]; }
                else { out[e] = tv[u][e] > 0.f ? gh[u][e] * scale : 0.f; }
            }
            store4<u16>(g_io + o, out);
            if constexpr (!GLU) {               // the bias gradient sums the values AS STORED (bf16): what the weight gradient multiplies
#pragma unroll
                for (int e = 0; e < 4; ++e) ab[e] += bf16_to_f32(f32_to_bf16(out[e]));
            }
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) { part[0][threadIdx.x][e] = ag[e]; part[1][threadIdx.x][e] = ab[e]; }
    __syncthreads();
    if (l0) return;
#pragma unroll 1
    for (int l = 1; l < nl; ++l)
#pragma unroll
        for (int e = 0; e < 4; ++e) { ag[e] += part[0][l * quads + threadIdx.x][e]; ab[e] += part[1][l * quads + threadIdx.x][e]; }
    if constexpr (GLU) {
        v4f og, ob;
#pragma unroll
        for (int e = 0; e < 4; ++e) { og[e] = s[e] * (1.f - s[e]) * ag[e]; ob[e] = s[e] * ab[e]; }
        *reinterpret_cast<v4f *>(gct + (size_t)b * gct_stride + c) = og;
        *reinterpret_cast<v4f *>(bsum + (size_t)b * bsum_stride + c) = ob;
    } else {
        *reinterpret_cast<v4f *>(bsum + (size_t)b * bsum_stride + c) = v4f{ab[0], ab[1], ab[2], ab[3]};
    }
}
}}  // namespace mhe::glow

using namespace mhe;
static inline unsigned gg(long n) { long b = (n + 255) / 256; return (unsigned)(b < 1 ? 1 : (b > 16384 ? 16384 : b)); }

extern "C" int mhe_glow_add_image_rows_f32(float *H, const float *img, long img_stride, long R, int C, int row_div, int n_img, void *stream) {
    MHE_REQUIRE(H && img && R > 0 && C > 0 && C % 4 == 0 && img_stride % 4 == 0 && row_div > 0 && n_img > 0, "mhe_glow_add_image_rows_f32: bad arguments");
    hipLaunchKernelGGL(glow::add_image_rows_kernel, dim3(gg(R * C / 4)), dim3(256), 0, (hipStream_t)stream, H, img, img_stride, R, C, row_div, n_img);
    return check_launch("add_image_rows_kernel");
}

extern "C" int mhe_relu_copy_f32(const float *in, void *out, long n, int out_dtype, void *stream) {
    MHE_REQUIRE(in && out && n > 0 && n % 4 == 0 && (out_dtype == MHE_F32 || out_dtype == MHE_BF16), "mhe_relu_copy_f32: bad arguments");
    if (out_dtype == MHE_F32)
        hipLaunchKernelGGL(glow::relu_copy_kernel<float>, dim3(gg(n / 4)), dim3(256), 0, (hipStream_t)stream, in, (float *)out, n / 4);
    else
        hipLaunchKernelGGL(glow::relu_copy_kernel<u16>, dim3(gg(n / 4)), dim3(256), 0, (hipStream_t)stream, in, (u16 *)out, n / 4);
    return check_launch("relu_copy_kernel");
}

extern "C" int mhe_glow_glu_residual_f32(float *H, const void *T, int t_dtype, const float *gate, long gate_stride, long R, int C,
                                         int row_div, int n_img, void *stream) {
    MHE_REQUIRE(H && T && gate && R > 0 && C > 0 && C % 4 == 0 && gate_stride % 4 == 0 && row_div > 0 && n_img > 0 &&
                    (t_dtype == MHE_F32 || t_dtype == MHE_BF16), "mhe_glow_glu_residual_f32: bad arguments");
    if (t_dtype == MHE_F32)
        hipLaunchKernelGGL(glow::glu_residual_kernel<float>, dim3(gg(R * C / 4)), dim3(256), 0, (hipStream_t)stream, H, (const float *)T, gate,
                           gate_stride, R, C, row_div, n_img);
    else
        hipLaunchKernelGGL(glow::glu_residual_kernel<u16>, dim3(gg(R * C / 4)), dim3(256), 0, (hipStream_t)stream, H, (const u16 *)T, gate,
                           gate_stride, R, C, row_div, n_img);
    return check_launch("glu_residual_kernel");
}

static inline int pad_cols(int n) { return (n + 63) / 64 * 64; }

extern "C" int mhe_glow_coupling_f32(const float *u, const float *params, float *y, float *logdet, long R, int dim, int first,
                                     int n_transform, int inverse, void *stream) {
    MHE_REQUIRE(u && params && y && logdet && R > 0 && dim > 0 && dim <= 256 && (first == 0 || first == 1) && n_transform > 0 &&
                    first + 2 * (n_transform - 1) < dim,
                "mhe_glow_coupling_f32: bad arguments");
    hipLaunchKernelGGL(glow::coupling_kernel, dim3((unsigned)((R + 3) / 4)), dim3(256), 0, (hipStream_t)stream, u, params, y, logdet, R, dim,
                       first, n_transform, inverse, pad_cols(dim), pad_cols(2 * n_transform));
    return check_launch("coupling_kernel");
}

extern "C" int mhe_pad64_f32(const float *x, float *xp, long R, int dim, void *stream) {
    MHE_REQUIRE(x && xp && R > 0 && dim > 0 && dim <= 256, "mhe_pad64_f32: bad arguments");
    const int ld = pad_cols(dim);
    hipLaunchKernelGGL(glow::pad64_kernel, dim3((unsigned)((R * ld + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, xp, R, dim, ld);
    return check_launch("pad64_kernel");
}

extern "C" int mhe_glow_finish_f32(const float *z_padded, const float *v_padded, const float *logdet, float *v_out, float *log_prob,
                                   long R, int dim, float sign, float logdet_const, void *stream) {
    MHE_REQUIRE(z_padded && logdet && log_prob && (!v_out || v_padded) && R > 0 && dim > 0 && dim <= 256, "mhe_glow_finish_f32: bad arguments");
    hipLaunchKernelGGL(glow::finish_kernel, dim3((unsigned)((R + 3) / 4)), dim3(256), 0, (hipStream_t)stream, z_padded, v_padded, logdet, v_out,
                       log_prob, R, dim, sign, logdet_const, pad_cols(dim), (const float *)nullptr, 0);
    return check_launch("finish_kernel");
}

// ... with the log-determinant constant read from the device: sum of const_parts[0 .. n_parts) (mhe_glow_affine_f64) - a captured HIP graph
// then follows the parameters from step to step (a host scalar would be baked into the graph)
extern "C" int mhe_glow_finish_dev_f32(const float *z_padded, const float *v_padded, const float *logdet, float *v_out, float *log_prob,
                                       long R, int dim, float sign, const float *const_parts, int n_parts, void *stream) {
    MHE_REQUIRE(z_padded && logdet && log_prob && (!v_out || v_padded) && R > 0 && dim > 0 && dim <= 256 && const_parts && n_parts > 0 && n_parts <= 64,
                "mhe_glow_finish_dev_f32: bad arguments");
    hipLaunchKernelGGL(glow::finish_kernel, dim3((unsigned)((R + 3) / 4)), dim3(256), 0, (hipStream_t)stream, z_padded, v_padded, logdet, v_out,
                       log_prob, R, dim, sign, 0.f, pad_cols(dim), const_parts, n_parts);
    return check_launch("finish_kernel");
}

extern "C" int mhe_glow_coupling_inv_bwd_f32(const float *v, const float *params, const float *g_y, const float *g_log_p, float q_weight,
                                             float *g_v, float *g_params, long R, int B, int dim, int first, int n_transform, void *stream) {
    MHE_REQUIRE(v && params && g_y && g_v && g_params && R > 0 && B > 0 && dim > 0 && dim <= 64 && (first == 0 || first == 1) &&
                    n_transform > 0 && first + 2 * (n_transform - 1) < dim && 2 * n_transform <= 64,
                "mhe_glow_coupling_inv_bwd_f32: bad arguments");
    hipLaunchKernelGGL(glow::coupling_inv_bwd_kernel, dim3((unsigned)((R + 3) / 4)), dim3(256), 0, (hipStream_t)stream, v, params, g_y, g_log_p,
                       q_weight, g_v, g_params, R, B, dim, first, n_transform);
    return check_launch("coupling_inv_bwd_kernel");
}

extern "C" int mhe_glow_glu_bwd_f32(const float *g_h, const void *t3, const float *gate, long gate_stride, void *g_t3, float *g_gate_rows,
                                    long R, int C, int row_div, int n_img, int t_dtype, void *stream) {
    MHE_REQUIRE(g_h && t3 && gate && g_t3 && g_gate_rows && R > 0 && C > 0 && C % 4 == 0 && gate_stride % 4 == 0 && row_div > 0 && n_img > 0 &&
                    (t_dtype == MHE_F32 || t_dtype == MHE_BF16), "mhe_glow_glu_bwd_f32: bad arguments");
    if (t_dtype == MHE_F32)
        hipLaunchKernelGGL(glow::glu_bwd_kernel<float>, dim3(gg(R * C / 4)), dim3(256), 0, (hipStream_t)stream, g_h, (const float *)t3, gate,
                           gate_stride, (float *)g_t3, g_gate_rows, R, C, row_div, n_img);
    else
        hipLaunchKernelGGL(glow::glu_bwd_kernel<u16>, dim3(gg(R * C / 4)), dim3(256), 0, (hipStream_t)stream, g_h, (const u16 *)t3, gate,
                           gate_stride, (u16 *)g_t3, g_gate_rows, R, C, row_div, n_img);
    return check_launch("glu_bwd_kernel");
}

extern "C" int mhe_relu_bwd_add_f32(float *acc, const void *g, const float *h, long n, int g_dtype, void *stream) {
    MHE_REQUIRE(acc && g && h && n > 0 && n % 4 == 0 && (g_dtype == MHE_F32 || g_dtype == MHE_BF16), "mhe_relu_bwd_add_f32: bad arguments");
    if (g_dtype == MHE_F32)
        hipLaunchKernelGGL(glow::relu_bwd_add_kernel<float>, dim3(gg(n / 4)), dim3(256), 0, (hipStream_t)stream, acc, (const float *)g, h, n / 4);
    else
        hipLaunchKernelGGL(glow::relu_bwd_add_kernel<u16>, dim3(gg(n / 4)), dim3(256), 0, (hipStream_t)stream, acc, (const u16 *)g, h, n / 4);
    return check_launch("relu_bwd_add_kernel");
}

// ... with the ReLU's argument (or its output: same sign) kept in bf16 - the fused Glow kernel's tape (csrc/glow_fwd.hip) holds relu(h) as bf16
extern "C" int mhe_relu_bwd_add_mixed(float *acc, const void *g, const void *h, long n, int g_dtype, int h_dtype, void *stream) {
    MHE_REQUIRE(acc && g && h && n > 0 && n % 4 == 0 && (g_dtype == MHE_F32 || g_dtype == MHE_BF16) && (h_dtype == MHE_F32 || h_dtype == MHE_BF16),
                "mhe_relu_bwd_add_mixed: bad arguments");
    if (h_dtype == MHE_F32) return mhe_relu_bwd_add_f32(acc, g, (const float *)h, n, g_dtype, stream);
    if (g_dtype == MHE_F32)
        hipLaunchKernelGGL((glow::relu_bwd_add_kernel<float, u16>), dim3(gg(n / 4)), dim3(256), 0, (hipStream_t)stream, acc, (const float *)g, (const u16 *)h, n / 4);
    else
        hipLaunchKernelGGL((glow::relu_bwd_add_kernel<u16, u16>), dim3(gg(n / 4)), dim3(256), 0, (hipStream_t)stream, acc, (const u16 *)g, (const u16 *)h, n / 4);
    return check_launch("relu_bwd_add_kernel");
}

// the two per-image reverse stages of the one-launch kernel's tape (sample-major rows r = n B + b; C in {64 .. 2048}, 512 % (C / 4) == 0)
extern "C" int mhe_glow_glu_bwd_sum(const float *g_h, const void *t3, const float *gate, long gate_stride, void *g_t3, float *gct, long gct_stride,
                                    float *bsum, long bsum_stride, int N, int B, int C, void *stream) {
    MHE_REQUIRE(g_h && t3 && gate && g_t3 && gct && bsum && N > 0 && B > 0 && C >= 4 && C % 4 == 0 && C / 4 <= 512 && 512 % (C / 4) == 0 &&
                    gate_stride % 4 == 0 && gct_stride % 4 == 0 && bsum_stride % 4 == 0, "mhe_glow_glu_bwd_sum: bad arguments");
    hipLaunchKernelGGL(glow::image_rows_bwd_kernel<true>, dim3(B), dim3(512), 0, (hipStream_t)stream, g_h, (u16 *)g_t3, (const u16 *)t3, gate, gate_stride,
                       1.f, gct, gct_stride, bsum, bsum_stride, N, B, C);
    return check_launch("image_rows_bwd_kernel<glu>");
}

extern "C" int mhe_glow_mask_scale_sum(void *g, const void *t2, float scale, float *bsum, long bsum_stride, int N, int B, int C, void *stream) {
    MHE_REQUIRE(g && t2 && bsum && N > 0 && B > 0 && C >= 4 && C % 4 == 0 && C / 4 <= 512 && 512 % (C / 4) == 0 && bsum_stride % 4 == 0,
                "mhe_glow_mask_scale_sum: bad arguments");
    hipLaunchKernelGGL(glow::image_rows_bwd_kernel<false>, dim3(B), dim3(512), 0, (hipStream_t)stream, (const float *)nullptr, (u16 *)g, (const u16 *)t2,
                       (const float *)nullptr, 0L, scale, (float *)nullptr, 0L, bsum, bsum_stride, N, B, C);
    return check_launch("image_rows_bwd_kernel<mask>");
}

// Reverse of a bottleneck's conv3 (1x1) + train-mode BatchNorm WITHOUT the convolution's raw output y3 and without the BatchNorm reverse's
// gy3 = k2 g + k1 y3 + k0 ever being materialised (train step, layer1 / layer2 of ResNet-50: y3 is the widest tensor of the block and the
// forward pass no longer writes it - conv_fuse.hip).  Reference: hand/network.py (torchvision Bottleneck: conv3 -> bn3), differentiated by
// autograd in hand/CrossModalHand.py:455-470.
//
// With y3 = A W^T (A = conv3's input [M][Cb], W [C][Cb]), g the gated gradient at bn3's output and D = g^T A [C][Cb] (ONE weight-gradient
// launch on g itself), everything the reverse needs is linear algebra on small matrices:
//   sum_p g y3 [c]   = sum_k W[c][k] D[c][k]                      -> dgamma, and with sum_p g (taken by the producer of g): k2, k1, k0
//   dW = gy3^T A     = k2 D + k1 (W G) + k0 m^T                   G = A^T A [Cb][Cb], m = 1^T A [Cb]: the forward's Gram statistics
//   gy3 W            = g (k2 W) + A (W^T diag(k1) W) + k0^T W     -> one data-gradient launch on g with weights k2 W, a Cb x Cb product
//                                                                    on A as its residual, a per-channel constant as its bias
// Against the form that reads y3 (written by the forward or evaluated again) twice, writes gy3 and reads it back: 2.2 GB less HBM traffic
// per layer1 block at C2.  The sums over a million pixels are combined in f64 (k1 (W G) + k0 m^T is k1 M Cov(y3, A): a difference of
// large numbers when a channel's mean is large against its spread).
#include "common.h"
#include "../../include/mhe.h"

namespace mhe { namespace fold {

// one wave per output channel c of conv3
template <int CB>
__global__ __launch_bounds__(256) void channel_kernel(float *__restrict__ D, const u16 *__restrict__ w, const double *__restrict__ tot,
                                                      const mhe_stat_t *__restrict__ stats, const float *__restrict__ gamma,
                                                      const float *__restrict__ mean_invstd, float *__restrict__ dgamma, float *__restrict__ dbeta,
                                                      float *__restrict__ dW, u16 *__restrict__ w_dg, float *__restrict__ coef, int C, int ldg,
                                                      double count) {
    constexpr int NC = CB / 64;
    __shared__ double wl[4][CB];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63, c = blockIdx.x * 4 + wv;
    if (c >= C) return;                                          // (whole waves leave together: the LDS exchange below is wave-local)
    const double s1 = fx::wave_total(const_cast<mhe_stat_t *>(stats), 0, C, c, lane, false);       // lane = statistic shard
    double wk[NC], dk[NC], sw = 0.0;
#pragma unroll
    for (int k = 0; k < NC; ++k) {
        const size_t e = (size_t)c * CB + lane + 64 * k;
        wk[k] = (double)bf16_to_f32(w[e]);
        dk[k] = (double)D[e];
        D[e] = 0.f;                                              // the accumulator is clean for the next step
        wl[wv][lane + 64 * k] = wk[k];
        sw = fma(wk[k], dk[k], sw);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sw += __shfl_xor(sw, o, 64);
    wave_sync();
    const double mean = (double)mean_invstd[c], invstd = (double)mean_invstd[C + c];
    const double s2 = invstd * (sw - mean * s1);                 // sum g xhat
    const double k2 = (double)gamma[c] * invstd, k1 = -k2 * invstd * s2 / count, k0 = -k2 * s1 / count - k1 * mean;
    // (W G)[c][k], k = lane + 64 j
    double t[NC];
#pragma unroll
    for (int k = 0; k < NC; ++k) t[k] = 0.0;
    for (int i = 0; i < CB; ++i) {
        const double wi = wl[wv][i];
#pragma unroll
        for (int k = 0; k < NC; ++k) t[k] = fma(wi, tot[(size_t)i * CB + lane + 64 * k], t[k]);
    }
#pragma unroll
    for (int k = 0; k < NC; ++k) {
        const int kk = lane + 64 * k;
        dW[(size_t)c * CB + kk] += (float)(k2 * dk[k] + k1 * t[k] + k0 * tot[(size_t)CB * CB + kk]);
        w_dg[(size_t)kk * ldg + c] = f32_to_bf16((float)(k2 * wk[k]));
    }
    if (lane == 0) {
        dbeta[c] = (float)s1;
        dgamma[c] = (float)s2;
        coef[c] = (float)k1;
        coef[C + c] = (float)k0;
    }
}

// S = W^T diag(k1) W [Cb][Cb] (row j per workgroup) and c0 = k0^T W [Cb] (one more workgroup).  1,024 threads = 1,024 / CB channel groups x CB
// columns: a thread walks its group's channels (rows of W: coalesced), the groups are folded through LDS.  (One wave per row walking all C
// channels with a strided scalar per step took 174 us at Cb = 128: eight times the rest of the fold.)
template <int CB>
__global__ __launch_bounds__(1024) void gram_side_kernel(const u16 *__restrict__ w, const float *__restrict__ coef, u16 *__restrict__ S,
                                                         float *__restrict__ c0, int C, int ldS) {
    constexpr int NG = 1024 / CB;
    __shared__ float red[NG][CB];
    const int j = blockIdx.x, k = threadIdx.x % CB, grp = threadIdx.x / CB;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    const bool row = j < CB;
    int c = grp;
    for (; c + 3 * NG < C; c += 4 * NG) {                    // four channels in flight per thread (the loads are what the loop waits for)
        float f[4], x[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int cc = c + u * NG;
            f[u] = row ? coef[cc] * bf16_to_f32(w[(size_t)cc * CB + j]) : coef[C + cc];
            x[u] = bf16_to_f32(w[(size_t)cc * CB + k]);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) acc[u] = fmaf(f[u], x[u], acc[u]);
    }
    for (; c < C; c += NG) {
        const float f0 = row ? coef[c] * bf16_to_f32(w[(size_t)c * CB + j]) : coef[C + c];
        acc[0] = fmaf(f0, bf16_to_f32(w[(size_t)c * CB + k]), acc[0]);
    }
    red[grp][k] = (acc[0] + acc[1]) + (acc[2] + acc[3]);
    __syncthreads();
    if (grp == 0) {
        float v = 0.f;
#pragma unroll
        for (int g = 0; g < NG; ++g) v += red[g][k];
        if (row) S[(size_t)j * ldS + k] = f32_to_bf16(v);
        else c0[k] = v;
    }
}

}}  // namespace mhe::fold

using namespace mhe;

extern "C" int mhe_conv3_bn_fold(float *D, const void *w_bf16, const double *gram_totals, const mhe_stat_t *rev_stats, const float *gamma,
                                 const float *mean_invstd, double count, float *dgamma, float *dbeta, float *dW, void *w_dg_bf16, int ld_dg,
                                 void *S_bf16, int ld_S, float *c0, float *coef_ws, int C, int Cb, void *stream) {
    MHE_REQUIRE(D && w_bf16 && gram_totals && rev_stats && gamma && mean_invstd && dgamma && dbeta && dW && w_dg_bf16 && S_bf16 && c0 && coef_ws,
                "mhe_conv3_bn_fold: null pointer");
    MHE_REQUIRE((Cb == 64 || Cb == 128) && C > 0 && ld_dg >= C && ld_S >= Cb && count > 1.f, "mhe_conv3_bn_fold: Cb=%d (64 | 128), C=%d, ld_dg=%d, ld_S=%d", Cb, C, ld_dg, ld_S);
    hipStream_t s = (hipStream_t)stream;
    if (Cb == 64) {
        hipLaunchKernelGGL(fold::channel_kernel<64>, dim3((C + 3) / 4), dim3(256), 0, s, D, (const u16 *)w_bf16, gram_totals, rev_stats, gamma,
                           mean_invstd, dgamma, dbeta, dW, (u16 *)w_dg_bf16, coef_ws, C, ld_dg, (double)count);
        if (int rc = check_launch("fold::channel_kernel")) return rc;
        hipLaunchKernelGGL(fold::gram_side_kernel<64>, dim3(Cb + 1), dim3(1024), 0, s, (const u16 *)w_bf16, coef_ws, (u16 *)S_bf16, c0, C, ld_S);
    } else {
        hipLaunchKernelGGL(fold::channel_kernel<128>, dim3((C + 3) / 4), dim3(256), 0, s, D, (const u16 *)w_bf16, gram_totals, rev_stats, gamma,
                           mean_invstd, dgamma, dbeta, dW, (u16 *)w_dg_bf16, coef_ws, C, ld_dg, (double)count);
        if (int rc = check_launch("fold::channel_kernel")) return rc;
        hipLaunchKernelGGL(fold::gram_side_kernel<128>, dim3(Cb + 1), dim3(1024), 0, s, (const u16 *)w_bf16, coef_ws, (u16 *)S_bf16, c0, C, ld_S);
    }
    return check_launch("fold::gram_side_kernel");
}

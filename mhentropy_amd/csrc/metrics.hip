// Per-image evaluation metrics of MHEntLoss (hand/criteria.py:91-168, helper
// hand/utils.py:21-30): per-joint 3D / 2D error over the N hypotheses, masked means
// renormalised by the number of valid images, best/worst of N, unbiased per-joint
// spread over N.  One wavefront per image, one lane per joint.
#include "common.h"

namespace mhe { namespace metrics {

constexpr int K = 21;
constexpr int ROOT = 12;      // criteria.py:112

// out[14][B]: for sup in (3d, 2d): sample, sample_std, vis, vis_std, vis_mean, invis, invis_std
template <int D>
__device__ __forceinline__ void one_sup(const float *__restrict__ coord, const float *gt_b, float cscale,
                                        float escale, int N, int B, int b, int lane, const float *w, const float *nvis,
                                        const float *nvalid, float *__restrict__ out) {
    const int k = lane < K ? lane : K - 1;
    float g[D];
#pragma unroll
    for (int d = 0; d < D; ++d) g[d] = gt_b[k * D + d];
    // pass 1: per-hypothesis masked mean error (best / worst of N), per-joint mean error and coordinate mean
    float best[3] = {3.0e38f, 3.0e38f, 3.0e38f};
    float worst_vis = -3.0e38f;
    float esum = 0.f, cm[D];
#pragma unroll
    for (int d = 0; d < D; ++d) cm[d] = 0.f;
    for (int n = 0; n < N; ++n) {
        const float *c = coord + ((size_t)n * B + b) * K * D + k * D;
        float e = 0.f;
#pragma unroll
        for (int d = 0; d < D; ++d) { const float v = c[d]; const float df = v - g[d]; e = fmaf(df, df, e); cm[d] += v * cscale; }
        e = sqrtf(e) * escale;
        esum += e;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            float m = wave_sum(lane < K ? e * w[a] : 0.f) / (nvis[a] + 1e-16f);
            m = nvalid[a] > 0.f ? m * (float)B / (nvalid[a] + 1e-16f) : m * 0.f;     // criteria.py:125-131
            best[a] = fminf(best[a], m);
            if (a == 1) worst_vis = fmaxf(worst_vis, m);
        }
    }
    // pass 2: unbiased std over N of every coordinate, volume -> length (criteria.py:155-162)
    float sp = 0.f;
    if (N > 1) {
        float var[D];
#pragma unroll
        for (int d = 0; d < D; ++d) { cm[d] /= (float)N; var[d] = 0.f; }
        for (int n = 0; n < N; ++n) {
            const float *c = coord + ((size_t)n * B + b) * K * D + k * D;
#pragma unroll
            for (int d = 0; d < D; ++d) { const float df = c[d] * cscale - cm[d]; var[d] = fmaf(df, df, var[d]); }
        }
        sp = 1.f;
#pragma unroll
        for (int d = 0; d < D; ++d) sp *= sqrtf(var[d] / (float)(N - 1));
    }
    sp = (D == 3 ? powf(sp, 1.f / 3.f) : sqrtf(sp)) * sqrtf((float)D);
    const float emean = esum / (float)N;
    float stdv[3], meanv = 0.f;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        float m = wave_sum(lane < K ? sp * w[a] : 0.f) / (nvis[a] + 1e-16f);
        stdv[a] = nvalid[a] > 0.f ? m * (float)B / (nvalid[a] + 1e-16f) : m * 0.f;
    }
    {
        float m = wave_sum(lane < K ? emean * w[1] : 0.f) / (nvis[1] + 1e-16f);
        meanv = nvalid[1] > 0.f ? m * (float)B / (nvalid[1] + 1e-16f) : m * 0.f;
    }
    if (lane == 0) {
        out[0 * B + b] = best[0];  out[1 * B + b] = stdv[0];
        out[2 * B + b] = (D == 2) ? worst_vis : best[1];       // 2D visible: worst hypothesis (criteria.py:148-152)
        out[3 * B + b] = stdv[1];  out[4 * B + b] = meanv;
        out[5 * B + b] = best[2];  out[6 * B + b] = stdv[2];
    }
}

__global__ __launch_bounds__(256) void metrics_kernel(const float *__restrict__ xyz, const float *__restrict__ uv,
                                                      const float *__restrict__ pose3d, const float *__restrict__ scale,
                                                      const float *__restrict__ crop_uv, const float *__restrict__ vis,
                                                      float *__restrict__ out, int N, int B) {
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (b >= B) return;
    // number of images that have at least one counted joint, per attribute (criteria.py:128-130)
    float cnt_vis = 0.f, cnt_inv = 0.f;
    for (int i = lane; i < B; i += 64) {
        int nv = 0, ni = 0;
        for (int k = 0; k < K; ++k) {
            if (k == ROOT) continue;
            if (vis[(size_t)i * K + k] == 1.f) ++nv; else ++ni;
        }
        cnt_vis += nv > 0 ? 1.f : 0.f;
        cnt_inv += ni > 0 ? 1.f : 0.f;
    }
    const float nvalid[3] = {(float)B, wave_sum(cnt_vis), wave_sum(cnt_inv)};
    const int k = lane < K ? lane : K - 1;
    const float v = vis[(size_t)b * K + k];
    float w[3];
    w[0] = 1.f;
    w[1] = (k != ROOT && v == 1.f) ? 1.f : 0.f;
    w[2] = (k != ROOT && v != 1.f) ? 1.f : 0.f;
    const float nvis[3] = {(float)K, wave_sum(lane < K ? w[1] : 0.f), wave_sum(lane < K ? w[2] : 0.f)};
    const float sc = scale[b];
    one_sup<3>(xyz, pose3d + (size_t)b * K * 3, sc, sc, N, B, b, lane, w, nvis, nvalid, out);
    // 2D ground truth in pixels: (crop_uv + 1) / 2 * 256   (criteria.py:96)
    __shared__ float gt2[4][K * 2];
    float *g2 = gt2[threadIdx.x >> 6];
    if (lane < K * 2) g2[lane] = (crop_uv[(size_t)b * K * 2 + lane] + 1.f) / 2.f * 256.f;
    wave_sync();
    one_sup<2>(uv, g2, 1.f, 1.f, N, B, b, lane, w, nvis, nvalid, out + 7 * B);
}

}}  // namespace mhe::metrics

using namespace mhe;

extern "C" int mhe_metrics_f32(const float *xyz, const float *uv, const float *pose3d, const float *scale,
                               const float *crop_uv, const float *vis, float *out, int N, int B, void *stream) {
    MHE_REQUIRE(xyz && uv && pose3d && scale && crop_uv && vis && out && N > 0 && B > 0, "mhe_metrics_f32: bad arguments");
    hipLaunchKernelGGL(metrics::metrics_kernel, dim3((B + 3) / 4), dim3(256), 0, (hipStream_t)stream, xyz, uv, pose3d,
                       scale, crop_uv, vis, out, N, B);
    return check_launch("metrics_kernel");
}

// ---------------------------------------------------------------------------
// Top-Q hypothesis selection of MHEnt.sample (hand/network.py:866-871): per image keep the Q hypotheses
// of highest log q, in descending order (torch.topk semantics), and gather their flow samples.
// One wavefront per image; rank by counting (N is a few hundred at most).
namespace mhe { namespace metrics {
__global__ __launch_bounds__(256) void topk_gather_kernel(const float *__restrict__ score, const float *__restrict__ rows,
                                                          int *__restrict__ idx_out, float *__restrict__ rows_out,
                                                          int N, int B, int Q, int D) {
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (b >= B) return;
    for (int n = lane; n < N; n += 64) {
        const float v = score[(size_t)n * B + b];
        int rank = 0;
        for (int m = 0; m < N; ++m) {
            const float u = score[(size_t)m * B + b];
            rank += (u > v || (u == v && m < n)) ? 1 : 0;
        }
        if (rank < Q) idx_out[(size_t)rank * B + b] = n;
    }
    wave_sync();
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_waitcnt(0);              // this wave's index stores are visible to its own later loads
    for (int r = 0; r < Q; ++r) {
        const int n = idx_out[(size_t)r * B + b];
        for (int d = lane; d < D; d += 64) rows_out[((size_t)r * B + b) * D + d] = rows[((size_t)n * B + b) * D + d];
    }
}
}}  // namespace mhe::metrics

extern "C" int mhe_topk_gather_f32(const float *score, const float *rows, int *idx_out, float *rows_out, int N, int B,
                                   int Q, int D, void *stream) {
    using namespace mhe;
    MHE_REQUIRE(score && rows && idx_out && rows_out, "mhe_topk_gather_f32: null pointer");
    MHE_REQUIRE(N > 0 && B > 0 && Q > 0 && Q <= N && D > 0, "mhe_topk_gather_f32: need 0 < Q <= N (N=%d Q=%d)", N, Q);
    hipLaunchKernelGGL(metrics::topk_gather_kernel, dim3((B + 3) / 4), dim3(256), 0, (hipStream_t)stream, score, rows,
                       idx_out, rows_out, N, B, Q, D);
    return check_launch("topk_gather_kernel");
}

// Per-image evaluation metrics of MHEntLoss (hand/criteria.py:91-168, helper
// hand/utils.py:21-30): per-joint 3D / 2D error over the N hypotheses, masked means
// renormalised by the number of valid images, best/worst of N, unbiased per-joint
// spread over N.  One workgroup per image.
#include "common.h"

namespace mhe { namespace metrics {

constexpr int K = 21;
constexpr int ROOT = 12;      // criteria.py:112

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// Round 5: one WORKGROUP per image, the image's hypotheses staged through LDS in chunks of CH rows.  The first version (one wavefront per
// image, a serial walk over the N hypotheses with three wave reductions each) was pure latency: 542 us for 21 MB at B = 256, N = 200 - a
// seventh of the reference iteration's metrics pass.  Here: (load) the chunk's rows by coalesced loads, (a) one thread per HYPOTHESIS: the 21
// joint errors, the three masked means -> best / worst of N, (b) one thread per COORDINATE column: sums over the chunk (every wave takes a
// quarter of the rows), second pass (the rows are still in LDS when N <= CH) for the unbiased spread.  Sums run in a different order than the
// serial walk did: last-bit differences.
constexpr int CH = 256;                 // hypotheses per chunk: 256 x 63 floats = 63 KiB
constexpr int LDS_FLOATS = CH * K * 3 + CH * K + 4 * 64 * 2 + 4 * 64 + 64 * 4;

// out[14][B]: for sup in (3d, 2d): sample, sample_std, vis, vis_std, vis_mean, invis, invis_std
template <int D>
__device__ __forceinline__ void one_sup(float *lds, const float *__restrict__ coord, const float *gt_b, float cscale, float escale, int N, int B,
                                        int b, const float *w, const float *nvis, const float *nvalid, float *__restrict__ out) {
    constexpr int KD = K * D;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    float *buf = lds;                               // [CH][KD] coordinates
    float *eb = lds + CH * K * 3;                   // [CH][K]  joint errors
    float *part = eb + CH * K;                      // [4 waves][64] column sums, [4][64] error sums
    float *red = part + 4 * 64 * 2;                 // [4 waves][64]: best[3], worst per wave
    float *fin = red + 4 * 64;                      // [64]: column means; [64..]: scratch
    float best[3] = {3.0e38f, 3.0e38f, 3.0e38f}, worst_vis = -3.0e38f;
    float csum = 0.f, esum = 0.f;                   // lane < KD: column `lane`; lane < K: joint `lane` (this wave's rows)
    for (int n0 = 0; n0 < N; n0 += CH) {
        const int cn = N - n0 < CH ? N - n0 : CH;
        __syncthreads();
        for (int i = tid; i < cn * KD; i += 256) {
            const int n = i / KD, j = i - n * KD;
            buf[n * KD + j] = coord[((size_t)(n0 + n) * B + b) * KD + j];
        }
        __syncthreads();
        if (tid < cn) {                             // (a) hypothesis tid: row stride KD (odd for D = 3; 42 for D = 2: two-way conflicts, small)
            const float *c = buf + tid * KD;
            float m[3] = {0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < K; ++k) {
                float e = 0.f;
#pragma unroll
                for (int d = 0; d < D; ++d) { const float df = c[k * D + d] - gt_b[k * D + d]; e = fmaf(df, df, e); }
                e = sqrtf(e) * escale;
                eb[tid * K + k] = e;
#pragma unroll
                for (int a = 0; a < 3; ++a) m[a] = fmaf(e, w[a * K + k], m[a]);
            }
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                float v = m[a] / (nvis[a] + 1e-16f);
                v = nvalid[a] > 0.f ? v * (float)B / (nvalid[a] + 1e-16f) : v * 0.f;     // criteria.py:125-131
                best[a] = fminf(best[a], v);
                if (a == 1) worst_vis = fmaxf(worst_vis, v);
            }
        }
        __syncthreads();
        if (lane < KD) for (int n = wave; n < cn; n += 4) csum += buf[n * KD + lane] * cscale;          // (b)
        if (lane < K) for (int n = wave; n < cn; n += 4) esum += eb[n * K + lane];
    }
    // best / worst over the workgroup's threads
#pragma unroll
    for (int a = 0; a < 3; ++a) best[a] = -wave_max(-best[a]);
    worst_vis = wave_max(worst_vis);
    if (lane == 0) { red[wave * 4 + 0] = best[0]; red[wave * 4 + 1] = best[1]; red[wave * 4 + 2] = best[2]; red[wave * 4 + 3] = worst_vis; }
    part[wave * 64 + lane] = csum;
    part[4 * 64 + wave * 64 + lane] = esum;
    __syncthreads();
    if (tid < 64) fin[tid] = (part[tid] + part[64 + tid] + part[128 + tid] + part[192 + tid]) / (float)N;              // column means
    const float emean = lane < K ? (part[256 + lane] + part[320 + lane] + part[384 + lane] + part[448 + lane]) / (float)N : 0.f;
    __syncthreads();
    // pass 2: unbiased std over N of every coordinate, volume -> length (criteria.py:155-162)
    float var = 0.f;
    if (N > 1) {
        const float cm = fin[lane];
        for (int n0 = 0; n0 < N; n0 += CH) {
            const int cn = N - n0 < CH ? N - n0 : CH;
            if (N > CH) {                           // more than one chunk: the rows have to come in again
                __syncthreads();
                for (int i = tid; i < cn * KD; i += 256) {
                    const int n = i / KD, j = i - n * KD;
                    buf[n * KD + j] = coord[((size_t)(n0 + n) * B + b) * KD + j];
                }
                __syncthreads();
            }
            if (lane < KD) for (int n = wave; n < cn; n += 4) { const float df = buf[n * KD + lane] * cscale - cm; var = fmaf(df, df, var); }
        }
    }
    __syncthreads();
    part[wave * 64 + lane] = var;
    __syncthreads();
    if (wave == 0) {
        float sp = 0.f;
        if (N > 1) {
            sp = 1.f;
            const int k = lane < K ? lane : K - 1;
#pragma unroll
            for (int d = 0; d < D; ++d) {
                const int j = k * D + d;
                sp *= sqrtf((part[j] + part[64 + j] + part[128 + j] + part[192 + j]) / (float)(N - 1));
            }
        }
        sp = (D == 3 ? powf(sp, 1.f / 3.f) : sqrtf(sp)) * sqrtf((float)D);
        float stdv[3], meanv;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float m = wave_sum(lane < K ? sp * w[a * K + lane] : 0.f) / (nvis[a] + 1e-16f);
            stdv[a] = nvalid[a] > 0.f ? m * (float)B / (nvalid[a] + 1e-16f) : m * 0.f;
        }
        {
            const float m = wave_sum(lane < K ? emean * w[K + lane] : 0.f) / (nvis[1] + 1e-16f);
            meanv = nvalid[1] > 0.f ? m * (float)B / (nvalid[1] + 1e-16f) : m * 0.f;
        }
        if (lane == 0) {
            float bst[3], wst = -3.0e38f;
#pragma unroll
            for (int a = 0; a < 3; ++a) bst[a] = fminf(fminf(red[a], red[4 + a]), fminf(red[8 + a], red[12 + a]));
            wst = fmaxf(fmaxf(red[3], red[7]), fmaxf(red[11], red[15]));
            out[0 * B + b] = bst[0];  out[1 * B + b] = stdv[0];
            out[2 * B + b] = (D == 2) ? wst : bst[1];          // 2D visible: worst hypothesis (criteria.py:148-152)
            out[3 * B + b] = stdv[1];  out[4 * B + b] = meanv;
            out[5 * B + b] = bst[2];  out[6 * B + b] = stdv[2];
        }
    }
    __syncthreads();
}

__global__ __launch_bounds__(256) void metrics_kernel(const float *__restrict__ xyz, const float *__restrict__ uv,
                                                      const float *__restrict__ pose3d, const float *__restrict__ scale,
                                                      const float *__restrict__ crop_uv, const float *__restrict__ vis,
                                                      float *__restrict__ out, int N, int B) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    __shared__ float wk[3 * K], gt[K * 3], hdr[8];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    if (tid < 64) {
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
        cnt_vis = wave_sum(cnt_vis); cnt_inv = wave_sum(cnt_inv);
        const int k = lane < K ? lane : K - 1;
        const float v = vis[(size_t)b * K + k];
        const float w1 = (k != ROOT && v == 1.f) ? 1.f : 0.f, w2 = (k != ROOT && v != 1.f) ? 1.f : 0.f;
        const float n1 = wave_sum(lane < K ? w1 : 0.f), n2 = wave_sum(lane < K ? w2 : 0.f);
        if (lane < K) { wk[lane] = 1.f; wk[K + lane] = w1; wk[2 * K + lane] = w2; }
        if (lane == 0) { hdr[0] = (float)K; hdr[1] = n1; hdr[2] = n2; hdr[3] = (float)B; hdr[4] = cnt_vis; hdr[5] = cnt_inv; }
        if (lane < K * 3) gt[lane] = pose3d[(size_t)b * K * 3 + lane];
    }
    __syncthreads();
    const float nvis[3] = {hdr[0], hdr[1], hdr[2]}, nvalid[3] = {hdr[3], hdr[4], hdr[5]};
    const float sc = scale[b];
    one_sup<3>(lds, xyz, gt, sc, sc, N, B, b, wk, nvis, nvalid, out);
    // 2D ground truth in pixels: (crop_uv + 1) / 2 * 256   (criteria.py:96)
    if (tid < K * 2) gt[tid] = (crop_uv[(size_t)b * K * 2 + tid] + 1.f) / 2.f * 256.f;
    __syncthreads();
    one_sup<2>(lds, uv, gt, 1.f, 1.f, N, B, b, wk, nvis, nvalid, out + 7 * B);
}

}}  // namespace mhe::metrics

using namespace mhe;

extern "C" int mhe_metrics_f32(const float *xyz, const float *uv, const float *pose3d, const float *scale,
                               const float *crop_uv, const float *vis, float *out, int N, int B, void *stream) {
    MHE_REQUIRE(xyz && uv && pose3d && scale && crop_uv && vis && out && N > 0 && B > 0, "mhe_metrics_f32: bad arguments");
    static const bool set = [] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(metrics::metrics_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  metrics::LDS_FLOATS * (int)sizeof(float));
        return true;
    }();
    (void)set;
    hipLaunchKernelGGL(metrics::metrics_kernel, dim3(B), dim3(256), metrics::LDS_FLOATS * sizeof(float), (hipStream_t)stream, xyz, uv, pose3d,
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

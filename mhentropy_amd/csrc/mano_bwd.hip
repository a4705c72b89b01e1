// Reverse mode of the MANO loss pass: d(sum_r g_r * log_p_r) / d(th45, det) for every hypothesis row,
// one wavefront per hypothesis.  The wave first re-runs joint_pass (forward values stay in its LDS scratch),
// then walks the chain backwards keeping all adjoints in a second per-wave LDS scratch:
//   Laplace likelihood + soft priors  (hand/network.py:155-165,233-258,612-667)
//   orthographic projection            (hand/network.py:497-514)
//   root/bone normalisation            (hand/utils.py:46-66)
//   reorder / centre / mm              (hand/manopth/manolayer.py:260-273, hand/ManoLayer.py:54-56)
//   fingertip skinning + blend shapes  (manolayer.py:181-188,236-251)
//   kinematic chain                    (manolayer.py:193-234)
//   Rodrigues through the quaternion   (rodrigues_layer.py:15-54)
//   PCA pose coefficients              (manolayer.py:131-143)
// It stands where autograd differentiates those lines in the reference's train step
// (hand/CrossModalHand.py:455-470 `loss.backward()`).
#include "mano_joint_pass.h"

namespace mhe { namespace mano {

// adjoint scratch per wave (floats)
constexpr int A_ROT = 0;       // [16][9]   local rotations
constexpr int A_JR = 144;      // [16][3]   rest joints
constexpr int A_G = 192;       // [16][12]  global rotation (9) + joint position (3)
constexpr int A_GR = 384;      // [16][12]  skinning transform
constexpr int A_PRE = 576;     // [21][3]   chain joints + skinned tips (metres, pre-reorder)
constexpr int A_TIPV = 640;    // [5][3]    posed tip vertices (rest frame)
constexpr int A_POSE = 656;    // [48]      axis-angles
constexpr int A_SCRATCH = 704;

__global__ __launch_bounds__(256) void mano_joints_bwd_kernel(
    const float *__restrict__ th45_g, const float *__restrict__ det_g, const float *__restrict__ crop_uv,
    const float *__restrict__ vis, const float *__restrict__ tables, const float *__restrict__ g_logp,
    float *__restrict__ g_th45_o, float *__restrict__ g_det_o, int R, int B, float lap_b, float th45_alpha, float row_w) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *tb = smem;
    for (int i = threadIdx.x; i < JOINT_FLOATS / 4; i += 256)
        reinterpret_cast<float4 *>(tb)[i] = reinterpret_cast<const float4 *>(tables)[i];
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float *sc = smem + JOINT_FLOATS + wave * SCRATCH;
    float *ad = smem + JOINT_FLOATS + 4 * SCRATCH + wave * A_SCRATCH;

    for (int r = blockIdx.x * 4 + wave; r < R; r += gridDim.x * 4) {
        const int b = r % B;
        const float th45 = lane < 45 ? th45_g[(size_t)r * 45 + lane] : 0.f;
        const float det = lane < 16 ? det_g[b * 16 + lane] : 0.f;
        const RowOut o = joint_pass(tb, sc, lane, th45, det);
        const float g = g_logp[b] * row_w;                 // d loss / d log_p of this row
        const int c3 = lane % 3, k21 = lane < 63 ? lane / 3 : 20;

        // ---- Laplace on the projected joints and the projection itself
        const float s_cam = expf(bcast(det, 13));
        const float t_cam = (lane & 1) ? bcast(det, 15) : bcast(det, 14);
        const int lu = lane < 42 ? lane : 41;
        const float uv = s_cam * __shfl(o.xyz, 3 * (lu >> 1) + (lu & 1), 64) + t_cam;
        float a_uv = 0.f;
        if (lane < 42) {
            const float d = uv - crop_uv[b * 42 + lane];
            if (vis[b * 21 + (lane >> 1)] == 1.f && fabsf(d) > 1e-4f) a_uv = (d > 0.f ? -g : g) / lap_b;
        }
        const float a_logs = wave_sum(a_uv * (uv - t_cam));
        const float a_t0 = wave_sum((lane & 1) ? 0.f : a_uv);
        const float a_t1 = wave_sum((lane & 1) ? a_uv : 0.f);
        const float a_uv_j = __shfl(a_uv, 2 * k21 + (c3 < 2 ? c3 : 0), 64);
        const float a_xyz = (lane < 63 && c3 < 2) ? s_cam * a_uv_j : 0.f;

        // ---- xyz = (J - J_root) / |J_norm - J_root|
        const float L = o.bone;
        const float S0 = wave_sum(c3 == 0 ? a_xyz : 0.f), S1 = wave_sum(c3 == 1 ? a_xyz : 0.f), S2 = wave_sum(c3 == 2 ? a_xyz : 0.f);
        const float Sc = c3 == 0 ? S0 : (c3 == 1 ? S1 : S2);
        const float a_L = -wave_sum(lane < 63 ? a_xyz * o.xyz : 0.f) / L;
        const float dc = sc[S_J21 + 3 * kNormIdx + c3] - sc[S_J21 + 3 * kRootIdx + c3];
        float aJ = a_xyz / L;
        if (k21 == kNormIdx) aJ += a_L * dc / L;
        if (k21 == kRootIdx) aJ -= Sc / L + a_L * dc / L;
        // ---- undo the reorder and the metre -> mm scale; the centring joint receives sum_k aJ = 0
        if (lane < 63) ad[A_PRE + 3 * kJointReorder[kFreihand2Rhd[k21]] + c3] = 1000.f * aJ;
        wave_sync();

        // ---- fingertip skinning  v' = sum_j w_j (Gr_j [v;1])
        if (lane < 15) {
            const int tip = lane / 3, d = lane % 3;
            float ap = 0.f;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                float T = 0.f;
#pragma unroll
                for (int j = 0; j < 16; ++j) T = fmaf(tb[TIP_W + tip * 16 + j], sc[S_GR + 12 * j + 3 * c + d], T);
                ap = fmaf(ad[A_PRE + 48 + 3 * tip + c], T, ap);
            }
            ad[A_TIPV + lane] = ap;
        }
        for (int e = lane; e < 192; e += 64) {
            const int j = e / 12, q = e % 12;
            float acc = 0.f;
#pragma unroll
            for (int tip = 0; tip < 5; ++tip) {
                const float av = ad[A_PRE + 48 + 3 * tip + (q < 9 ? q / 3 : q - 9)];
                acc = fmaf(tb[TIP_W + tip * 16 + j] * av, q < 9 ? sc[S_TIPV + 3 * tip + q % 3] : 1.f, acc);
            }
            ad[A_GR + e] = acc;
            ad[A_G + e] = q < 9 ? 0.f : ad[A_PRE + 3 * j + (q - 9)];      // chain joints are the transforms' translations
        }
        wave_sync();
        // ---- blend shapes of the tips: pose-corrective -> local rotations, shape -> beta
        for (int k = lane; k < 144; k += 64) {
            float a = 0.f;
            if (k >= 9) {
#pragma unroll
                for (int tc = 0; tc < 15; ++tc) a = fmaf(tb[TIP_PD + tc * 135 + (k - 9)], ad[A_TIPV + tc], a);
            }
            ad[A_ROT + k] = a;
        }
        float a_beta = 0.f;
        if (lane < 10) {
#pragma unroll
            for (int tc = 0; tc < 15; ++tc) a_beta = fmaf(tb[TIP_SD + tc * 10 + lane], ad[A_TIPV + tc], a_beta);
        }
        // ---- Gr_j = [Rg_j | t_j - Rg_j jr_j]
        if (lane < 16) {
            const int j = lane;
            const float *Rg = sc + S_G + 12 * j, *jr = sc + S_JR + 3 * j;
            float atr[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) atr[c] = ad[A_GR + 12 * j + 9 + c];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                ad[A_G + 12 * j + 9 + c] += atr[c];
#pragma unroll
                for (int d = 0; d < 3; ++d) ad[A_G + 12 * j + 3 * c + d] += ad[A_GR + 12 * j + 3 * c + d] - atr[c] * jr[d];
                ad[A_JR + 3 * j + c] = -(Rg[c] * atr[0] + Rg[3 + c] * atr[1] + Rg[6 + c] * atr[2]);
            }
        }
        wave_sync();
        // ---- kinematic chain, tips of the fingers back to the wrist; one finger per lane
        float rootv[15];
#pragma unroll
        for (int e = 0; e < 15; ++e) rootv[e] = 0.f;
        if (lane < 5) {
#pragma unroll
            for (int lvl = 2; lvl >= 0; --lvl) {
                const int j = 1 + 3 * lane + lvl, parent = lvl ? j - 1 : 0;
                float PR[9], Rj[9], aRg[9], at[3], rel[3], aPR[9], arel[3];
#pragma unroll
                for (int e = 0; e < 9; ++e) { PR[e] = sc[S_G + 12 * parent + e]; Rj[e] = sc[S_ROT + 9 * j + e]; aRg[e] = ad[A_G + 12 * j + e]; }
#pragma unroll
                for (int c = 0; c < 3; ++c) { at[c] = ad[A_G + 12 * j + 9 + c]; rel[c] = sc[S_JR + 3 * j + c] - sc[S_JR + 3 * parent + c]; }
#pragma unroll
                for (int a = 0; a < 3; ++a)
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        // Rg_j = PR Rj ;  t_j = PR rel + t_parent
                        ad[A_ROT + 9 * j + 3 * a + k] += PR[a] * aRg[k] + PR[3 + a] * aRg[3 + k] + PR[6 + a] * aRg[6 + k];
                        aPR[3 * a + k] = aRg[3 * a] * Rj[3 * k] + aRg[3 * a + 1] * Rj[3 * k + 1] + aRg[3 * a + 2] * Rj[3 * k + 2] + at[a] * rel[k];
                    }
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    arel[k] = PR[k] * at[0] + PR[3 + k] * at[1] + PR[6 + k] * at[2];
                    ad[A_JR + 3 * j + k] += arel[k];
                }
                if (lvl) {
#pragma unroll
                    for (int e = 0; e < 9; ++e) ad[A_G + 12 * parent + e] += aPR[e];
#pragma unroll
                    for (int c = 0; c < 3; ++c) { ad[A_G + 12 * parent + 9 + c] += at[c]; ad[A_JR + 3 * parent + c] -= arel[c]; }
                } else {
#pragma unroll
                    for (int e = 0; e < 9; ++e) rootv[e] = aPR[e];
#pragma unroll
                    for (int c = 0; c < 3; ++c) { rootv[9 + c] = at[c]; rootv[12 + c] = -arel[c]; }
                }
            }
        }
#pragma unroll
        for (int e = 0; e < 15; ++e) rootv[e] = wave_sum(rootv[e]);
        if (lane == 0) {
            // wrist: Rg_0 = R_0, t_0 = jr_0
#pragma unroll
            for (int e = 0; e < 9; ++e) ad[A_ROT + e] += ad[A_G + e] + rootv[e];
#pragma unroll
            for (int c = 0; c < 3; ++c) ad[A_JR + c] += rootv[12 + c] + ad[A_G + 9 + c] + rootv[9 + c];
        }
        wave_sync();
        // ---- rest joints are affine in beta
        if (lane < 10) {
#pragma unroll 8
            for (int i = 0; i < 48; ++i) a_beta = fmaf(tb[JSD + i * 10 + lane], ad[A_JR + i], a_beta);
        }
        // ---- Rodrigues (axis-angle -> unit quaternion -> matrix), 16 joints on 16 lanes
        if (lane < 16) {
            const float ax = sc[S_POSE + 3 * lane], ay = sc[S_POSE + 3 * lane + 1], az = sc[S_POSE + 3 * lane + 2];
            const float px = ax + 1e-8f, py = ay + 1e-8f, pz = az + 1e-8f;
            const float angle = sqrtf(px * px + py * py + pz * pz);
            const float nx = ax / angle, ny = ay / angle, nz = az / angle;
            const float half = angle * 0.5f;
            const float cs = cosf(half), sn = sinf(half);
            const float w0 = cs, x0 = sn * nx, y0 = sn * ny, z0 = sn * nz;
            const float qn = sqrtf(w0 * w0 + x0 * x0 + y0 * y0 + z0 * z0);
            const float w = w0 / qn, x = x0 / qn, y = y0 / qn, z = z0 / qn;
            const float *A = ad + A_ROT + 9 * lane;
            const float A0 = A[0], A1 = A[1], A2 = A[2], A3 = A[3], A4 = A[4], A5 = A[5], A6 = A[6], A7 = A[7], A8 = A[8];
            const float aw = 2.f * (w * (A0 + A4 + A8) + (-z * A1 + y * A2 + z * A3 - x * A5 - y * A6 + x * A7));
            const float axq = 2.f * (x * (A0 - A4 - A8) + (y * A1 + z * A2 + y * A3 - w * A5 + z * A6 + w * A7));
            const float ayq = 2.f * (y * (-A0 + A4 - A8) + (x * A1 + w * A2 + x * A3 + z * A5 - w * A6 + z * A7));
            const float azq = 2.f * (z * (-A0 - A4 + A8) + (-w * A1 + x * A2 + w * A3 + y * A5 + x * A6 + y * A7));
            const float qa = w * aw + x * axq + y * ayq + z * azq;
            const float bw = (aw - w * qa) / qn, bx = (axq - x * qa) / qn, by = (ayq - y * qa) / qn, bz = (azq - z * qa) / qn;
            const float a_sn = bx * nx + by * ny + bz * nz;
            const float anx = sn * bx, any_ = sn * by, anz = sn * bz;
            const float a_half = -sn * bw + cs * a_sn;
            const float a_angle = 0.5f * a_half - (anx * ax + any_ * ay + anz * az) / (angle * angle);
            ad[A_POSE + 3 * lane] = anx / angle + a_angle * px / angle;
            ad[A_POSE + 3 * lane + 1] = any_ / angle + a_angle * py / angle;
            ad[A_POSE + 3 * lane + 2] = anz / angle + a_angle * pz / angle;
        }
        wave_sync();
        // ---- PCA coefficients and the soft priors
        {
            const int lc = lane < 45 ? lane : 44;
            float a45 = 0.f;
#pragma unroll 9
            for (int l = 0; l < 45; ++l) a45 = fmaf(tb[COMPS + lc * 45 + l], ad[A_POSE + 3 + l], a45);
            const float v45 = fmaxf(fabsf(th45) / 2.f - 1.f, 0.f);
            a45 -= g * th45_alpha * v45 * (th45 > 0.f ? 1.f : -1.f);        // d/dx -alpha relu(|x|/2-1)^2
            if (lane < 45) g_th45_o[(size_t)r * 45 + lane] = a45;
        }
        {
            const float d0 = bcast(det, 0), d1 = bcast(det, 1), d2 = bcast(det, 2);
            const float r3 = sqrtf(d0 * d0 + d1 * d1 + d2 * d2);
            const float v3 = fmaxf(r3 / 3.14159265358979323846f - 1.f, 0.f);
            const float ab = __shfl(a_beta, lane >= 3 && lane < 13 ? lane - 3 : 0, 64);
            float out = 0.f;
            if (lane < 3) out = ad[A_POSE + lane] - (v3 > 0.f ? g * 10.f * v3 / 3.14159265358979323846f * det / r3 : 0.f);
            else if (lane < 13) {
                const float vb = fmaxf(fabsf(det) / 0.03f - 1.f, 0.f);
                out = ab - g * 100.f * vb / 0.03f * (det > 0.f ? 1.f : -1.f);
            } else if (lane == 13) out = a_logs;
            else if (lane == 14) out = a_t0;
            else if (lane == 15) out = a_t1;
            if (lane < 16) g_det_o[(size_t)r * 16 + lane] = out;
        }
        wave_sync();
    }
}

// out[b][c] = sum_n in[(n*B + b)][c]: per-image totals of per-hypothesis rows (det head and conditioning gradients)
__global__ __launch_bounds__(256) void sum_over_hypotheses_kernel(const float *__restrict__ in, float *__restrict__ out,
                                                                  int N, int B, int C, int accumulate, long out_stride) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)B * C) return;
    float acc = 0.f;
    for (int n = 0; n < N; ++n) acc += in[(size_t)n * B * C + i];
    float *o = out + (i / C) * out_stride + i % C;
    *o = accumulate ? *o + acc : acc;
}
}}  // namespace mhe::mano

using namespace mhe;

extern "C" int mhe_mano_joints_bwd_f32(const float *th45, const float *det, const float *crop_uv, const float *vis,
                                       const float *tables, const float *g_log_p, float *g_th45, float *g_det_rows,
                                       int R, int B, float laplace_b, float th45_alpha, float row_weight, void *stream) {
    MHE_REQUIRE(th45 && det && crop_uv && vis && tables && g_log_p && g_th45 && g_det_rows, "mhe_mano_joints_bwd_f32: null pointer");
    MHE_REQUIRE(R > 0 && B > 0 && R % B == 0, "mhe_mano_joints_bwd_f32: R=%d must be a positive multiple of B=%d", R, B);
    MHE_REQUIRE(laplace_b > 0.f, "mhe_mano_joints_bwd_f32: laplace_b must be > 0");
    const int blocks = (R + 3) / 4 < 2048 ? (R + 3) / 4 : 2048;
    const size_t lds = (mano::JOINT_FLOATS + 4 * (mano::SCRATCH + mano::A_SCRATCH)) * sizeof(float);
    hipLaunchKernelGGL(mano::mano_joints_bwd_kernel, dim3(blocks), dim3(256), lds, (hipStream_t)stream, th45, det, crop_uv,
                       vis, tables, g_log_p, g_th45, g_det_rows, R, B, laplace_b, th45_alpha, row_weight);
    return check_launch("mano_joints_bwd_kernel");
}

extern "C" int mhe_sum_over_hypotheses_f32(const float *rows, float *out, int N, int B, int C, int accumulate, long out_stride,
                                           void *stream) {
    MHE_REQUIRE(rows && out && N > 0 && B > 0 && C > 0, "mhe_sum_over_hypotheses_f32: bad arguments");
    if (out_stride <= 0) out_stride = C;
    MHE_REQUIRE(out_stride >= C, "mhe_sum_over_hypotheses_f32: out_stride=%ld < C=%d", out_stride, C);
    const size_t n = (size_t)B * C;
    hipLaunchKernelGGL(mano::sum_over_hypotheses_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       rows, out, N, B, C, accumulate, out_stride);
    return check_launch("sum_over_hypotheses_kernel");
}

// MANO forward kinematics, fingertip / full linear-blend skinning, projection and
// the per-hypothesis log-likelihood + priors, one 64-lane wavefront per hypothesis.
//
// Reference arithmetic being replaced (all fp32):
//   hand/manopth/manolayer.py:131-273, rodrigues_layer.py:15-54, tensutils.py:6-22,
//   hand/ManoLayer.py:45-60,150-165, hand/utils.py:46-66,
//   hand/network.py:155-165,233-258,455-558,612-667,703-717,787-788.
//
// HBM view (loss pass): reads 180 B (th45) per row + per-image det/crop_uv/vis,
// writes <= 0.6 KB per row; the 19.5 KB of joint-section tables live in LDS.
#include "mano_joint_pass.h"

namespace mhe { namespace mano {

__global__ __launch_bounds__(256) void mano_joints_kernel(
    const float *__restrict__ th45_g, const float *__restrict__ det_g, const float *__restrict__ crop_uv,
    const float *__restrict__ vis, const float *__restrict__ tables,
    float *__restrict__ z_o, float *__restrict__ xyz_o, float *__restrict__ uv_o, float *__restrict__ terms_o,
    float *__restrict__ logp_o, float *__restrict__ norms_o, float *__restrict__ jmm_o,
    int R, int B, float lap_b, float th45_alpha, int inv_norm, float image_size) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *tb = smem;
    for (int i = threadIdx.x; i < JOINT_FLOATS / 4; i += 256)
        reinterpret_cast<float4 *>(tb)[i] = reinterpret_cast<const float4 *>(tables)[i];
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float *sc = smem + JOINT_FLOATS + wave * SCRATCH;
    const float log2b = logf(2.f * lap_b);

    for (int r = blockIdx.x * 4 + wave; r < R; r += gridDim.x * 4) {
        const int b = r % B;
        const float th45 = lane < 45 ? th45_g[(size_t)r * 45 + lane] : 0.f;
        const float det = lane < 16 ? det_g[b * 16 + lane] : 0.f;
        const RowOut o = joint_pass(tb, sc, lane, th45, det);

        if (xyz_o && lane < 63) xyz_o[(size_t)r * 63 + lane] = o.xyz;
        if (jmm_o && lane < 63) jmm_o[(size_t)r * 63 + lane] = sc[S_J21 + lane];
        // -- orthographic projection (hand/network.py:497-514, ManoLayer.py:150-165)
        const float s_cam = expf(bcast(det, 13));
        const float t_cam = (lane & 1) ? bcast(det, 15) : bcast(det, 14);
        const int lu = lane < 42 ? lane : 41;
        float uv = s_cam * __shfl(o.xyz, 3 * (lu >> 1) + (lu & 1), 64) + t_cam;
        if (inv_norm) uv = (uv + 1.f) / 2.f * image_size;
        if (uv_o && lane < 42) uv_o[(size_t)r * 42 + lane] = uv;

        if (z_o) {      // z = [th3 th45 bt logs t]  (hand/network.py:703-717)
            const float a = __shfl(th45, lane >= 3 ? lane - 3 : 0, 64);
            const float d = __shfl(det, lane >= 45 ? lane - 45 : 0, 64);
            if (lane < 61) z_o[(size_t)r * 61 + lane] = lane < 3 ? det : (lane < 48 ? a : d);
        }
        // |theta|, |beta|  (hand/network.py:787-788)
        const float th_sq = wave_sum((lane < 45 ? th45 * th45 : 0.f) + (lane < 3 ? det * det : 0.f));
        const float bt_sq = wave_sum((lane >= 3 && lane < 13) ? det * det : 0.f);
        if (norms_o && lane == 0) { norms_o[(size_t)r * 2] = sqrtf(th_sq); norms_o[(size_t)r * 2 + 1] = sqrtf(bt_sq); }

        if (terms_o || logp_o) {
            // visibility-masked Laplace over the 42 projected coordinates (hand/network.py:255-257)
            float lt = 0.f;
            if (lane < 42) {
                const float y = crop_uv[b * 42 + lane];
                const float w = vis[b * 21 + (lane >> 1)];
                const float e = -(fmaxf(fabsf(y - uv) - 1e-4f, 0.f) + 1e-4f) / lap_b - log2b;
                lt = (w == 1.f) ? e : 0.f;
            }
            const float lp_uv = wave_sum(lt);
            // soft box on th45 (+-2) and beta (+-0.03), soft ball on th3 (pi)  (hand/network.py:155-163,429-435)
            float v45 = lane < 45 ? fmaxf(fabsf(th45) / 2.f - 1.f, 0.f) : 0.f;
            const float lp_45 = -wave_sum(th45_alpha * v45 * v45);
            float vbt = (lane >= 3 && lane < 13) ? fmaxf(fabsf(det) / 0.03f - 1.f, 0.f) : 0.f;
            const float lp_bt = -wave_sum(50.f * vbt * vbt);
            const float r3 = sqrtf(bcast(det, 0) * bcast(det, 0) + bcast(det, 1) * bcast(det, 1) + bcast(det, 2) * bcast(det, 2));
            const float v3 = fmaxf(r3 / 3.14159265358979323846f - 1.f, 0.f);
            const float lp_3 = -5.f * v3 * v3;
            if (lane == 0) {
                if (terms_o) {
                    float4 t4 = make_float4(lp_uv, lp_3, lp_45, lp_bt);
                    reinterpret_cast<float4 *>(terms_o)[r] = t4;
                }
                if (logp_o) logp_o[r] = ((lp_uv + lp_3) + lp_45) + lp_bt;
            }
        }
    }
}

// ---------------------------------------------------------------------------
// Full mesh in two launches.
//  (1) mano_pose_kernel: one wavefront per hypothesis runs joint_pass and leaves what skinning needs in a
//      352-float workspace row: pose map (135), beta (10), the 16 skinning transforms (192), centre/root/bone (7).
//  (2) mano_skin_kernel: one thread per vertex, HB hypotheses per workgroup.  Every per-hypothesis quantity
//      is wave-uniform, so it is read with SCALAR loads and enters the FMAs as an SGPR operand; the blend-shape
//      tables (vertex-fastest, coalesced) are fetched once per workgroup and reused from registers across
//      the HB hypotheses (1.5 MB / HB of L2 traffic per hypothesis).  No LDS at all in this kernel: the first
//      version kept the per-hypothesis values in LDS and was LDS-issue bound (1536 broadcast reads per vertex).
constexpr int WS_PM = 0, WS_BT = 135, WS_GR = 145, WS_NRM = 337, WS_STRIDE = 352;

__global__ __launch_bounds__(256) void mano_pose_kernel(const float *__restrict__ z_g, const float *__restrict__ tables,
                                                        float *__restrict__ ws, int R) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *tb = smem;
    for (int i = threadIdx.x; i < JOINT_FLOATS / 4; i += 256)
        reinterpret_cast<float4 *>(tb)[i] = reinterpret_cast<const float4 *>(tables)[i];
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float *sc = smem + JOINT_FLOATS + wave * SCRATCH;
    for (int r = blockIdx.x * 4 + wave; r < R; r += gridDim.x * 4) {
        const float *zr = z_g + (size_t)r * 61;
        const float th45 = lane < 45 ? zr[3 + lane] : 0.f;
        // det lane layout [th3 bt logs t] from z = [th3 th45 bt logs t]
        const float det = lane < 3 ? zr[lane] : (lane < 16 ? zr[45 + lane] : 0.f);
        const RowOut o = joint_pass(tb, sc, lane, th45, det);
        float *w = ws + (size_t)r * WS_STRIDE;
        for (int k = lane; k < 135; k += 64) {
            const int e = k % 9;
            w[WS_PM + k] = sc[S_ROT + 9 + k] - ((e == 0 || e == 4 || e == 8) ? 1.f : 0.f);
        }
        if (lane < 10) w[WS_BT + lane] = zr[48 + lane];
        for (int k = lane; k < 192; k += 64) w[WS_GR + k] = sc[S_GR + k];
        if (lane < 3) { w[WS_NRM + lane] = o.center_c; w[WS_NRM + 3 + lane] = o.root_c; }
        if (lane == 0) w[WS_NRM + 6] = o.bone;
        wave_sync();
    }
}

template <int HB>
__global__ __launch_bounds__(256) void mano_skin_kernel(const float *__restrict__ ws, const float *__restrict__ tables,
                                                        float *__restrict__ verts_o, int R, int mm_mode) {
    const int v = blockIdx.x * 256 + threadIdx.x;            // vertex (VP = 832 padded: 4 vertex groups, last partial)
    const int r0 = blockIdx.y * HB;
    const int vc = v < VP ? v : VP - 1;
    const float *Vt = tables + V_T, *Vsd = tables + V_SD, *Vpd = tables + V_PD, *Vw = tables + V_W;
    // wave-uniform row pointers (rows past R are clamped; their results are not stored)
    const float *wrow[HB];
#pragma unroll
    for (int h = 0; h < HB; ++h) wrow[h] = ws + (size_t)(r0 + h < R ? r0 + h : R - 1) * WS_STRIDE;
    float acc[HB][3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float t = Vt[c * VP + vc];
#pragma unroll
        for (int h = 0; h < HB; ++h) acc[h][c] = t;
    }
    // shape blend (manolayer.py:181-183) then pose-corrective blend (:187-188)
#pragma unroll 2
    for (int k = 0; k < 10; ++k) {
        const float s0 = Vsd[(k * 3 + 0) * VP + vc], s1 = Vsd[(k * 3 + 1) * VP + vc], s2 = Vsd[(k * 3 + 2) * VP + vc];
#pragma unroll
        for (int h = 0; h < HB; ++h) {
            const float bk = wrow[h][WS_BT + k];
            acc[h][0] = fmaf(s0, bk, acc[h][0]); acc[h][1] = fmaf(s1, bk, acc[h][1]); acc[h][2] = fmaf(s2, bk, acc[h][2]);
        }
    }
#pragma unroll 5
    for (int k = 0; k < 135; ++k) {
        const float p0 = Vpd[(k * 3 + 0) * VP + vc], p1 = Vpd[(k * 3 + 1) * VP + vc], p2 = Vpd[(k * 3 + 2) * VP + vc];
#pragma unroll
        for (int h = 0; h < HB; ++h) {
            const float m = wrow[h][WS_PM + k];
            acc[h][0] = fmaf(p0, m, acc[h][0]); acc[h][1] = fmaf(p1, m, acc[h][1]); acc[h][2] = fmaf(p2, m, acc[h][2]);
        }
    }
    float w[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) w[j] = Vw[j * VP + vc];
#pragma unroll
    for (int h = 0; h < HB; ++h) {
        // T = sum_j w_j Gr_j ; v' = T [v;1]   (manolayer.py:236-246)
        float T[12];
#pragma unroll
        for (int e = 0; e < 12; ++e) T[e] = 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j)
#pragma unroll
            for (int e = 0; e < 12; ++e) T[e] = fmaf(wrow[h][WS_GR + j * 12 + e], w[j], T[e]);
        const int r = r0 + h;
        if (v < NV && r < R) {
            const float bone = wrow[h][WS_NRM + 6];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float vp = T[3 * c] * acc[h][0] + T[3 * c + 1] * acc[h][1] + T[3 * c + 2] * acc[h][2] + T[9 + c];
                const float mesh = (vp - wrow[h][WS_NRM + c]) * 1000.f;            // manolayer.py:262-273
                verts_o[((size_t)r * NV + v) * 3 + c] = mm_mode ? mesh : (mesh - wrow[h][WS_NRM + 3 + c]) / bone;   // network.py:480
            }
        }
    }
}

// ManoLayer.xyz_from_vertice (hand/ManoLayer.py:141-148,108-139): 16 joints regressed
// from the mesh + 5 tip vertices, FreiHand order, then the RHD reorder (:54-56).
// One workgroup per hypothesis; wave w handles joints w, w+4, ...
__device__ constexpr int kWrapJointMap[16] = {0, 5, 6, 7, 9, 10, 11, 17, 18, 19, 13, 14, 15, 1, 2, 3};   // mano id -> FreiHand id
__device__ constexpr int kWrapTipVert[5] = {744, 320, 443, 555, 672};                                    // FreiHand ids 4,8,12,16,20
__global__ __launch_bounds__(256) void regress_joints_kernel(const float *__restrict__ verts, const float *__restrict__ tables,
                                                             float *__restrict__ joints, int R) {
    __shared__ float kp[21 * 3];
    const int r = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const float *v = verts + (size_t)r * NV * 3;
    for (int j = wave; j < 16; j += 4) {
        float a0 = 0.f, a1 = 0.f, a2 = 0.f;
        for (int i = lane; i < NV; i += 64) {
            const float w = tables[V_JR + j * VP + i];
            a0 = fmaf(v[3 * i], w, a0); a1 = fmaf(v[3 * i + 1], w, a1); a2 = fmaf(v[3 * i + 2], w, a2);
        }
        a0 = wave_sum(a0); a1 = wave_sum(a1); a2 = wave_sum(a2);
        if (lane == 0) { float *o = kp + 3 * kWrapJointMap[j]; o[0] = a0; o[1] = a1; o[2] = a2; }
    }
    if (threadIdx.x < 15) kp[3 * (4 + 4 * (threadIdx.x / 3)) + threadIdx.x % 3] = v[3 * kWrapTipVert[threadIdx.x / 3] + threadIdx.x % 3];
    __syncthreads();
    if (threadIdx.x < 63) joints[(size_t)r * 63 + threadIdx.x] = kp[3 * kFreihand2Rhd[threadIdx.x / 3] + threadIdx.x % 3];
}

__global__ void elbo_reduce_kernel(const float *__restrict__ lp_rows, const float *__restrict__ lq_rows,
                                   float *__restrict__ q_log_p, float *__restrict__ h_o, float *__restrict__ log_p,
                                   int N, int B) {
    // one wavefront per image: lanes stride over the N hypotheses, LDS-free wave reduce
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (wave >= B) return;
    float a = 0.f, q = 0.f;
    for (int n = lane; n < N; n += 64) {
        a += lp_rows[(size_t)n * B + wave];
        q += lq_rows ? -lq_rows[(size_t)n * B + wave] : 0.f;
    }
    a = wave_sum(a) / (float)N;
    q = wave_sum(q) / (float)N;
    if (lane == 0) {
        if (q_log_p) q_log_p[wave] = a;
        if (h_o) h_o[wave] = q;
        if (log_p) log_p[wave] = q + a;
    }
}

}}  // namespace mhe::mano

using namespace mhe;

extern "C" size_t mhe_mano_table_floats(void) { return mano::TOTAL_FLOATS; }

extern "C" int mhe_mano_joints_f32(const float *th45, const float *det, const float *crop_uv, const float *vis,
                                   const float *tables, float *z, float *xyz, float *uv, float *terms, float *log_p,
                                   float *norms, float *joints_mm, int R, int B, float laplace_b, float th45_alpha,
                                   int inv_norm, float image_size, void *stream) {
    MHE_REQUIRE(th45 && det && tables, "mhe_mano_joints_f32: null input");
    MHE_REQUIRE(R > 0 && B > 0 && R % B == 0, "mhe_mano_joints_f32: R=%d must be a positive multiple of B=%d", R, B);
    MHE_REQUIRE(!(terms || log_p) || (crop_uv && vis), "mhe_mano_joints_f32: likelihood outputs need crop_uv and vis");
    MHE_REQUIRE(laplace_b > 0.f, "mhe_mano_joints_f32: laplace_b must be > 0");
    const int blocks = (R + 3) / 4 < 2048 ? (R + 3) / 4 : 2048;
    const size_t lds = (mano::JOINT_FLOATS + 4 * mano::SCRATCH) * sizeof(float);
    hipLaunchKernelGGL(mano::mano_joints_kernel, dim3(blocks), dim3(256), lds, (hipStream_t)stream, th45, det, crop_uv,
                       vis, tables, z, xyz, uv, terms, log_p, norms, joints_mm, R, B, laplace_b, th45_alpha, inv_norm,
                       image_size);
    return check_launch("mano_joints_kernel");
}

extern "C" size_t mhe_mano_verts_workspace_floats(int R) { return R > 0 ? (size_t)R * mano::WS_STRIDE : 0; }

extern "C" int mhe_mano_verts_f32(const float *z, const float *tables, float *verts, float *workspace, int R, int mm_mode,
                                  void *stream) {
    MHE_REQUIRE(z && tables && verts && workspace, "mhe_mano_verts_f32: null pointer");
    MHE_REQUIRE(R > 0, "mhe_mano_verts_f32: R=%d", R);
    constexpr int HB = 16;
    const size_t lds = (mano::JOINT_FLOATS + 4 * mano::SCRATCH) * sizeof(float);
    const int pb = (R + 3) / 4 < 2048 ? (R + 3) / 4 : 2048;
    hipLaunchKernelGGL(mano::mano_pose_kernel, dim3(pb), dim3(256), lds, (hipStream_t)stream, z, tables, workspace, R);
    if (int rc = check_launch("mano_pose_kernel")) return rc;
    hipLaunchKernelGGL(mano::mano_skin_kernel<HB>, dim3((mano::VP + 255) / 256, (R + HB - 1) / HB), dim3(256), 0,
                       (hipStream_t)stream, workspace, tables, verts, R, mm_mode);
    return check_launch("mano_skin_kernel");
}

extern "C" int mhe_mano_regress_joints_f32(const float *verts, const float *tables, float *joints, int R, void *stream) {
    MHE_REQUIRE(verts && tables && joints && R > 0, "mhe_mano_regress_joints_f32: bad arguments");
    hipLaunchKernelGGL(mano::regress_joints_kernel, dim3(R), dim3(256), 0, (hipStream_t)stream, verts, tables, joints, R);
    return check_launch("regress_joints_kernel");
}

extern "C" int mhe_elbo_reduce_f32(const float *log_p_rows, const float *log_q_rows, float *q_log_p, float *h,
                                   float *log_p, int N, int B, void *stream) {
    MHE_REQUIRE(log_p_rows && N > 0 && B > 0, "mhe_elbo_reduce_f32: bad arguments");
    hipLaunchKernelGGL(mano::elbo_reduce_kernel, dim3((B + 3) / 4), dim3(256), 0, (hipStream_t)stream, log_p_rows,
                       log_q_rows, q_log_p, h, log_p, N, B);
    return check_launch("elbo_reduce_kernel");
}

// Body-model path of a ProHMR-style multi-hypothesis head (reference README.md:26-42):
//   6D rotation representation -> R   (hand/manopth/rot6d.py:4-24 Gram-Schmidt form, :26-51 symmetric 'robust' form; used by
//                                      hand/manopth/manolayer.py:150-156) and its reverse,
//   linear-blend skinning for a model of runtime size (J <= 32 joints, V vertices, nb shape / 9(J-1) pose-blend coefficients,
//   any kinematic tree with parents[j] < j): the arithmetic of hand/manopth/manolayer.py:181-246 at other sizes
//   (SMPL: 24 joints, 6,890 vertices, 207 pose-blend coefficients).
// Two launches like the hand mesh (mano.hip): a pose kernel (one wavefront per hypothesis: rest joints, kinematic chain, skinning
// transforms, pose map -> one workspace row) and a skinning kernel (one thread per vertex, HB hypotheses per workgroup, every
// per-hypothesis value a wave-uniform scalar operand, blend-shape tables read vertex-fastest and reused across the HB hypotheses).
// Algorithmic bytes per hypothesis: 12 V written (+ workspace row); 3 V (nb + 9(J-1) + 4 J + 4) FMA.
#include "common.h"

namespace mhe { namespace body {

__device__ __forceinline__ void cross3(const float *u, const float *v, float *o) {
    o[0] = u[1] * v[2] - u[2] * v[1]; o[1] = u[2] * v[0] - u[0] * v[2]; o[2] = u[0] * v[1] - u[1] * v[0];
}
__device__ __forceinline__ float norm3c(const float *v) { return fmaxf(sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]), 1e-8f); }   // rot6d.py:54-60

__global__ void rot6d_kernel(const float *__restrict__ p6, float *__restrict__ Rm, long n, int robust) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float a[3], b[3], x[3], y[3], z[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) { a[c] = p6[i * 6 + c]; b[c] = p6[i * 6 + 3 + c]; }
    if (!robust) {                       // x = n(a), z = n(x x b), y = z x x
        const float na = norm3c(a);
#pragma unroll
        for (int c = 0; c < 3; ++c) x[c] = a[c] / na;
        float t[3];
        cross3(x, b, t);
        const float nt = norm3c(t);
#pragma unroll
        for (int c = 0; c < 3; ++c) z[c] = t[c] / nt;
        cross3(z, x, y);
    } else {                             // symmetric: orthogonalise the normalised pair around its bisector
        const float na = norm3c(a), nb = norm3c(b);
        float m[3], o[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) { x[c] = a[c] / na; y[c] = b[c] / nb; m[c] = x[c] + y[c]; o[c] = x[c] - y[c]; }
        const float nm = norm3c(m), no = norm3c(o);
#pragma unroll
        for (int c = 0; c < 3; ++c) { m[c] /= nm; o[c] /= no; x[c] = m[c] + o[c]; y[c] = m[c] - o[c]; }
        const float nx = norm3c(x), ny = norm3c(y);
#pragma unroll
        for (int c = 0; c < 3; ++c) { x[c] /= nx; y[c] /= ny; }
        float t[3];
        cross3(x, y, t);
        const float nt = norm3c(t);
#pragma unroll
        for (int c = 0; c < 3; ++c) z[c] = t[c] / nt;
    }
#pragma unroll
    for (int r = 0; r < 3; ++r) { Rm[i * 9 + 3 * r] = x[r]; Rm[i * 9 + 3 * r + 1] = y[r]; Rm[i * 9 + 3 * r + 2] = z[r]; }   // columns x, y, z
}

// reverse of the Gram-Schmidt form: g6 = (dR/dp6)^T gR
__global__ void rot6d_bwd_kernel(const float *__restrict__ p6, const float *__restrict__ gR, float *__restrict__ g6, long n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float a[3], b[3], x[3], z[3], t[3], gx[3], gy[3], gz[3], tmp[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        a[c] = p6[i * 6 + c]; b[c] = p6[i * 6 + 3 + c];
        gx[c] = gR[i * 9 + 3 * c]; gy[c] = gR[i * 9 + 3 * c + 1]; gz[c] = gR[i * 9 + 3 * c + 2];
    }
    const float ra = sqrtf(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]), na = fmaxf(ra, 1e-8f);
#pragma unroll
    for (int c = 0; c < 3; ++c) x[c] = a[c] / na;
    cross3(x, b, t);
    const float rt = sqrtf(t[0] * t[0] + t[1] * t[1] + t[2] * t[2]), nt = fmaxf(rt, 1e-8f);
#pragma unroll
    for (int c = 0; c < 3; ++c) z[c] = t[c] / nt;
    // y = z x x :  gz += x x gy,  gx += gy x z
    cross3(x, gy, tmp);
#pragma unroll
    for (int c = 0; c < 3; ++c) gz[c] += tmp[c];
    cross3(gy, z, tmp);
#pragma unroll
    for (int c = 0; c < 3; ++c) gx[c] += tmp[c];
    // z = t / max(|t|, eps)
    float gt[3];
    const float dz = rt > 1e-8f ? z[0] * gz[0] + z[1] * gz[1] + z[2] * gz[2] : 0.f;
#pragma unroll
    for (int c = 0; c < 3; ++c) gt[c] = (gz[c] - z[c] * dz) / nt;
    // t = x x b :  gx += b x gt,  gb = gt x x
    float gb[3];
    cross3(b, gt, tmp);
#pragma unroll
    for (int c = 0; c < 3; ++c) gx[c] += tmp[c];
    cross3(gt, x, gb);
    const float dx = ra > 1e-8f ? x[0] * gx[0] + x[1] * gx[1] + x[2] * gx[2] : 0.f;
#pragma unroll
    for (int c = 0; c < 3; ++c) { g6[i * 6 + c] = (gx[c] - x[c] * dx) / na; g6[i * 6 + 3 + c] = gb[c]; }
}

// workspace row: pose map [9(J-1)] | betas [nb] | skinning transforms [J][12] (3x3 row-major, then translation) | posed joints [J][3]
__host__ __device__ inline int ws_pm(int, int) { return 0; }
__host__ __device__ inline int ws_bt(int J, int) { return 9 * (J - 1); }
__host__ __device__ inline int ws_a(int J, int nb) { return 9 * (J - 1) + nb; }
__host__ __device__ inline int ws_jp(int J, int nb) { return 9 * (J - 1) + nb + 12 * J; }
__host__ __device__ inline int ws_stride(int J, int nb) { return (9 * (J - 1) + nb + 15 * J + 15) / 16 * 16; }
constexpr int MAXJ = 32;

__global__ __launch_bounds__(256) void lbs_pose_kernel(const float *__restrict__ rot, const float *__restrict__ betas,
                                                       const float *__restrict__ jt, const float *__restrict__ jsd,
                                                       const int *__restrict__ parents, float *__restrict__ ws,
                                                       float *__restrict__ joints_o, int R, int J, int nb) {
    __shared__ float sG[4][MAXJ * 12], sJ[4][MAXJ * 3];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float *G = sG[wave], *Jr = sJ[wave];
    const int stride = ws_stride(J, nb);
    for (int r = blockIdx.x * 4 + wave; r < R; r += gridDim.x * 4) {
        const float *Rr = rot + (size_t)r * J * 9, *bt = betas + (size_t)r * nb;
        float *w = ws + (size_t)r * stride;
        // rest joints: J_template + J_shapedirs beta  (manolayer.py:181-184, joint regression folded into the tables)
        for (int e = lane; e < 3 * J; e += 64) {
            float v = jt[e];
            for (int k = 0; k < nb; ++k) v = fmaf(jsd[e * nb + k], bt[k], v);
            Jr[e] = v;
        }
        for (int e = lane; e < 9 * (J - 1); e += 64) {
            const int d = e % 9;
            w[ws_pm(J, nb) + e] = Rr[9 + e] - ((d == 0 || d == 4 || d == 8) ? 1.f : 0.f);                 // manolayer.py:187
        }
        if (lane < nb) w[ws_bt(J, nb) + lane] = bt[lane];
        wave_sync();
        // kinematic chain (manolayer.py:193-229): G_j = G_parent [R_j | j_j - j_parent]; lane e < 12 owns element (row e/4... ) of G_j
        if (lane < 12) {
            const int rr = lane / 4, cc = lane % 4;
            G[lane] = cc < 3 ? Rr[3 * rr + cc] : Jr[rr];
        }
        wave_sync();
        for (int j = 1; j < J; ++j) {
            const int p = parents[j];
            float v = 0.f;
            if (lane < 12) {
                const int rr = lane / 4, cc = lane % 4;
                const float *Gp = G + 12 * p;
                if (cc < 3) {
                    v = Gp[4 * rr] * Rr[9 * j + cc] + Gp[4 * rr + 1] * Rr[9 * j + 3 + cc] + Gp[4 * rr + 2] * Rr[9 * j + 6 + cc];
                } else {
                    const float t0 = Jr[3 * j] - Jr[3 * p], t1 = Jr[3 * j + 1] - Jr[3 * p + 1], t2 = Jr[3 * j + 2] - Jr[3 * p + 2];
                    v = Gp[4 * rr] * t0 + Gp[4 * rr + 1] * t1 + Gp[4 * rr + 2] * t2 + Gp[4 * rr + 3];
                }
            }
            wave_sync();
            if (lane < 12) G[12 * j + lane] = v;
            wave_sync();
        }
        // skinning transforms: rotation of G_j, translation G_j.t - G_j.R j_rest  (manolayer.py:231-234); posed joints = G_j.t
        for (int e = lane; e < 12 * J; e += 64) {
            const int j = e / 12, k = e % 12;
            const float *Gj = G + 12 * j;
            float v;
            if (k < 9) v = Gj[4 * (k / 3) + k % 3];
            else {
                const int rr = k - 9;
                v = Gj[4 * rr + 3] - (Gj[4 * rr] * Jr[3 * j] + Gj[4 * rr + 1] * Jr[3 * j + 1] + Gj[4 * rr + 2] * Jr[3 * j + 2]);
            }
            w[ws_a(J, nb) + e] = v;
        }
        for (int e = lane; e < 3 * J; e += 64) {
            const float v = G[12 * (e / 3) + 4 * (e % 3) + 3];
            w[ws_jp(J, nb) + e] = v;
            if (joints_o) joints_o[(size_t)r * 3 * J + e] = v;
        }
        wave_sync();
    }
}

template <int HB>
__global__ __launch_bounds__(256) void lbs_skin_kernel(const float *__restrict__ ws, const float *__restrict__ Vt, const float *__restrict__ Vsd,
                                                       const float *__restrict__ Vpd, const float *__restrict__ Vw, float *__restrict__ verts_o,
                                                       int R, int J, int nb, int NV, int VP, float scale) {
    const int v = blockIdx.x * 256 + threadIdx.x;
    const int r0 = blockIdx.y * HB;
    const int vc = v < NV ? v : NV - 1;
    const int stride = ws_stride(J, nb), NP = 9 * (J - 1), oa = ws_a(J, nb), ob = ws_bt(J, nb);
    const float *wrow[HB];
#pragma unroll
    for (int h = 0; h < HB; ++h) wrow[h] = ws + (size_t)(r0 + h < R ? r0 + h : R - 1) * stride;
    float acc[HB][3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float t = Vt[c * VP + vc];
#pragma unroll
        for (int h = 0; h < HB; ++h) acc[h][c] = t;
    }
    for (int k = 0; k < nb; ++k) {                 // shape blend (manolayer.py:181-183)
        const float s0 = Vsd[(k * 3 + 0) * VP + vc], s1 = Vsd[(k * 3 + 1) * VP + vc], s2 = Vsd[(k * 3 + 2) * VP + vc];
#pragma unroll
        for (int h = 0; h < HB; ++h) {
            const float bk = wrow[h][ob + k];
            acc[h][0] = fmaf(s0, bk, acc[h][0]); acc[h][1] = fmaf(s1, bk, acc[h][1]); acc[h][2] = fmaf(s2, bk, acc[h][2]);
        }
    }
#pragma unroll 3
    for (int k = 0; k < NP; ++k) {                 // pose-corrective blend (:187-188)
        const float p0 = Vpd[(k * 3 + 0) * VP + vc], p1 = Vpd[(k * 3 + 1) * VP + vc], p2 = Vpd[(k * 3 + 2) * VP + vc];
#pragma unroll
        for (int h = 0; h < HB; ++h) {
            const float m = wrow[h][k];
            acc[h][0] = fmaf(p0, m, acc[h][0]); acc[h][1] = fmaf(p1, m, acc[h][1]); acc[h][2] = fmaf(p2, m, acc[h][2]);
        }
    }
    float T[HB][12];
#pragma unroll
    for (int h = 0; h < HB; ++h)
#pragma unroll
        for (int e = 0; e < 12; ++e) T[h][e] = 0.f;
    for (int j = 0; j < J; ++j) {                  // T = sum_j w_j A_j  (:236)
        const float wj = Vw[j * VP + vc];
#pragma unroll
        for (int h = 0; h < HB; ++h)
#pragma unroll
            for (int e = 0; e < 12; ++e) T[h][e] = fmaf(wrow[h][oa + j * 12 + e], wj, T[h][e]);
    }
#pragma unroll
    for (int h = 0; h < HB; ++h) {
        const int r = r0 + h;
        if (v < NV && r < R) {
#pragma unroll
            for (int c = 0; c < 3; ++c)
                verts_o[((size_t)r * NV + v) * 3 + c] =
                    (T[h][3 * c] * acc[h][0] + T[h][3 * c + 1] * acc[h][1] + T[h][3 * c + 2] * acc[h][2] + T[h][9 + c]) * scale;   // :245-246
        }
    }
}

}}  // namespace mhe::body

using namespace mhe;

extern "C" int mhe_rot6d_to_rotmat_f32(const float *poses6, float *rotmats, long n, int robust, void *stream) {
    MHE_REQUIRE(poses6 && rotmats && n > 0, "mhe_rot6d_to_rotmat_f32: bad arguments");
    hipLaunchKernelGGL(body::rot6d_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, poses6, rotmats, n, robust);
    return check_launch("rot6d_kernel");
}

extern "C" int mhe_rot6d_to_rotmat_bwd_f32(const float *poses6, const float *g_rotmats, float *g_poses6, long n, void *stream) {
    MHE_REQUIRE(poses6 && g_rotmats && g_poses6 && n > 0, "mhe_rot6d_to_rotmat_bwd_f32: bad arguments");
    hipLaunchKernelGGL(body::rot6d_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, poses6, g_rotmats, g_poses6, n);
    return check_launch("rot6d_bwd_kernel");
}

extern "C" size_t mhe_lbs_workspace_floats(int R, int J, int nb) {
    return (R > 0 && J > 0 && J <= body::MAXJ && nb >= 0) ? (size_t)R * body::ws_stride(J, nb) : 0;
}

extern "C" int mhe_lbs_pose_f32(const float *rotmats, const float *betas, const float *j_template, const float *j_shapedirs,
                                const int *parents, float *workspace, float *joints, int R, int J, int nb, void *stream) {
    MHE_REQUIRE(rotmats && betas && j_template && j_shapedirs && parents && workspace, "mhe_lbs_pose_f32: null pointer");
    MHE_REQUIRE(R > 0 && J > 0 && J <= body::MAXJ && nb > 0 && nb <= 64, "mhe_lbs_pose_f32: R=%d J=%d nb=%d", R, J, nb);
    const int blocks = (R + 3) / 4 < 2048 ? (R + 3) / 4 : 2048;
    hipLaunchKernelGGL(body::lbs_pose_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, rotmats, betas, j_template, j_shapedirs, parents,
                       workspace, joints, R, J, nb);
    return check_launch("lbs_pose_kernel");
}

extern "C" int mhe_lbs_skin_f32(const float *workspace, const float *v_template, const float *v_shapedirs, const float *v_posedirs,
                                const float *v_weights, float *verts, int R, int J, int nb, int NV, int VP, float scale, void *stream) {
    MHE_REQUIRE(workspace && v_template && v_shapedirs && v_posedirs && v_weights && verts, "mhe_lbs_skin_f32: null pointer");
    MHE_REQUIRE(R > 0 && J > 0 && J <= body::MAXJ && nb > 0 && NV > 0 && VP >= NV, "mhe_lbs_skin_f32: R=%d J=%d nb=%d NV=%d VP=%d", R, J, nb, NV, VP);
    constexpr int HB = 8;
    hipLaunchKernelGGL(body::lbs_skin_kernel<HB>, dim3((NV + 255) / 256, (R + HB - 1) / HB), dim3(256), 0, (hipStream_t)stream, workspace,
                       v_template, v_shapedirs, v_posedirs, v_weights, verts, R, J, nb, NV, VP, scale);
    return check_launch("lbs_skin_kernel");
}

// The ActNorm + LU re-parameterisation of the conditional Glow ON THE DEVICE (round 5).
//
// Per layer the flow applies x -> W (exp(log_scale) * x + shift) + bias with W = L U (unit lower L, upper U with a softplus(+eps) diagonal):
// nflows `transforms.ActNorm` + `transforms.LULinear`, restated in oracle/glow_ref.py (PARITY UNPINNED: the class is a third-party
// dependency absent from the reference tree; call sites hand/network.py:342-344,693-694,736-742).  The sampling direction needs
// A^-1 with A = W diag(exp(log_scale)), the density the constant sum(log_scale) + sum(log diag U), and the train step the gradients of the
// six small parameter tensors from dL/dA^-1, dL/dc^-1 and the constant.  Rounds 2-4 did this algebra in float64 numpy on the HOST - a
// device->host copy, a few 45 x 45 products and a copy back per step - which drained the launch queue every step and kept the Glow branch
// out of HIP graphs.  Here: one workgroup per layer, float64 in LDS, no host round trip; the inverse comes from two triangular
// substitutions (A^-1 = diag(1/s) U^-1 L^-1) instead of a general inverse.
//   mhe_glow_affine_f64        parameters -> A, c, A^-1, (A^-1)^T, c^-1 = -A^-1 c (f32, zero-padded to 64) + the per-layer constant + a
//                              float64 workspace (L, U, W, A^-1, s, shift, diag, udiag, c) kept for the reverse kernel
//   mhe_glow_reparam_bwd_f64   dA^-1 [64][64], dc^-1 [64], dL/dlog q per image -> gradients of log_scale, shift, lower, upper,
//                              unconstrained diagonal, bias:  G = dA^-1 - dc^-1 c^T;  dc = -A^-T dc^-1;  dA = -A^-T G A^-T;
//                              dW = dA diag(s) + dc shift^T;  dlog_scale = colsum(dA o W) s + S;  dshift = W^T dc;  dL = dW U^T (strict lower);
//                              dU = L^T dW (upper);  dudiag = (diag(dU) + S / diag) sigmoid(udiag);  dbias = dc;   S = sum_r dL/dlog q[r]
#include "common.h"
#include "../../include/mhe.h"

namespace mhe { namespace glowaff {

constexpr int MAXD = 64, NT = 256;

struct Ptrs { const float *p[6]; };           // log_scale, shift, lower_entries, upper_entries, unconstrained_upper_diag, bias
struct GPtrs { float *p[6]; };

__device__ __forceinline__ int low_idx(int i, int j) { return i * (i - 1) / 2 + j; }                               // i > j   (np.tril_indices(D, -1))
__device__ __forceinline__ int up_idx(int i, int j, int D) { return i * (D - 1) - i * (i - 1) / 2 + (j - i - 1); }   // i < j   (np.triu_indices(D, 1))
__host__ __device__ __forceinline__ size_t ws_doubles(int D) { return (size_t)4 * D * D + 5 * D; }

// one workgroup per layer
__global__ __launch_bounds__(NT) void affine_kernel(const Ptrs *__restrict__ params, int D, float eps, float *__restrict__ A, float *__restrict__ c_out,
                                                    float *__restrict__ Ainv, float *__restrict__ AinvT, float *__restrict__ cinv,
                                                    float *__restrict__ const_parts, double *__restrict__ ws) {
    __shared__ double Lm[MAXD * MAXD], U[MAXD * MAXD], X[MAXD * MAXD], Y[MAXD * MAXD];       // 128 KiB
    __shared__ double sc[MAXD], sh[MAXD], dg[MAXD], cc[MAXD], red[NT];
    const int l = blockIdx.x, tid = threadIdx.x;
    const Ptrs P = params[l];
    double *w = ws + (size_t)l * ws_doubles(D);
    double *wL = w, *wU = w + D * D, *wW = w + 2 * D * D, *wAi = w + 3 * D * D, *wv = w + 4 * D * D;
    for (int i = tid; i < D * D; i += NT) {
        const int r = i / D, k = i % D;
        Lm[i] = r == k ? 1.0 : r > k ? (double)P.p[2][low_idx(r, k)] : 0.0;
        U[i] = r < k ? (double)P.p[3][up_idx(r, k, D)] : 0.0;
    }
    double part = 0.0;
    if (tid < D) {
        const double ud = (double)P.p[4][tid], ls = (double)P.p[0][tid];
        const double d = (ud > 30.0 ? ud : log1p(exp(ud))) + (double)eps;                  // softplus + eps (nflows LULinear)
        dg[tid] = d; sc[tid] = exp(ls); sh[tid] = (double)P.p[1][tid];
        wv[tid] = sc[tid]; wv[D + tid] = sh[tid]; wv[2 * D + tid] = d; wv[3 * D + tid] = ud;
        part = ls + log(d);
    }
    red[tid] = part;
    __syncthreads();
    if (tid < D) U[tid * D + tid] = dg[tid];
    if (tid == 0) {
        double s = 0.0;
        for (int i = 0; i < D; ++i) s += red[i];
        const_parts[l] = (float)s;
    }
    __syncthreads();
    // W = L U -> X;  A = W diag(s), c = W shift + bias
    for (int i = tid; i < D * D; i += NT) {
        const int r = i / D, k = i % D;
        double a = 0.0;
        const int kmax = r < k ? r : k;                       // L[r][m] = 0 for m > r, U[m][k] = 0 for m > k
        for (int m = 0; m <= kmax; ++m) a = fma(Lm[r * D + m], U[m * D + k], a);
        X[i] = a;
        wW[i] = a; wL[i] = Lm[i]; wU[i] = U[i];
    }
    __syncthreads();
    if (tid < D) {
        double a = (double)P.p[5][tid];
        for (int k = 0; k < D; ++k) a = fma(X[tid * D + k], sh[k], a);
        cc[tid] = a; wv[4 * D + tid] = a;
    }
    for (int i = tid; i < 64 * 64; i += NT) {
        const int r = i >> 6, k = i & 63;
        A[(size_t)l * 4096 + i] = (r < D && k < D) ? (float)(X[r * D + k] * sc[k]) : 0.f;
    }
    __syncthreads();
    // L^-1 -> Y and U^-1 -> X, column by column (a thread owns a column; its own earlier writes are what it reads back)
    if (tid < D) {
        const int j = tid;
        for (int i = 0; i < D; ++i) Y[i * D + j] = i == j ? 1.0 : 0.0;
        for (int i = j + 1; i < D; ++i) {
            double a = 0.0;
            for (int k = j; k < i; ++k) a = fma(Lm[i * D + k], Y[k * D + j], a);
            Y[i * D + j] = -a;
        }
    } else if (tid >= 64 && tid < 64 + D) {
        const int j = tid - 64;
        for (int i = 0; i < D; ++i) X[i * D + j] = 0.0;
        X[j * D + j] = 1.0 / U[j * D + j];
        for (int i = j - 1; i >= 0; --i) {
            double a = 0.0;
            for (int k = i + 1; k <= j; ++k) a = fma(U[i * D + k], X[k * D + j], a);
            X[i * D + j] = -a / U[i * D + i];
        }
    }
    __syncthreads();
    // A^-1 = diag(1/s) U^-1 L^-1 -> Lm (no longer needed: it is in the workspace)
    for (int i = tid; i < D * D; i += NT) {
        const int r = i / D, k = i % D;
        double a = 0.0;
        const int m0 = r > k ? r : k;                         // U^-1[r][m] = 0 for m < r, L^-1[m][k] = 0 for m < k
        for (int m = m0; m < D; ++m) a = fma(X[r * D + m], Y[m * D + k], a);
        a /= sc[r];
        Lm[i] = a; wAi[i] = a;
    }
    __syncthreads();
    for (int i = tid; i < 64 * 64; i += NT) {
        const int r = i >> 6, k = i & 63;
        const bool in = r < D && k < D;
        Ainv[(size_t)l * 4096 + i] = in ? (float)Lm[r * D + k] : 0.f;
        AinvT[(size_t)l * 4096 + i] = in ? (float)Lm[k * D + r] : 0.f;
    }
    if (tid < 64) {
        double a = 0.0;
        if (tid < D) for (int k = 0; k < D; ++k) a = fma(Lm[tid * D + k], cc[k], a);
        cinv[(size_t)l * 64 + tid] = tid < D ? (float)(-a) : 0.f;
        c_out[(size_t)l * 64 + tid] = tid < D ? (float)cc[tid] : 0.f;
    }
}

__global__ __launch_bounds__(NT) void reparam_bwd_kernel(const float *const *__restrict__ g_ainv, const float *const *__restrict__ g_cinv,
                                                         const float *__restrict__ g_logp, int n_logp, float q_sign, int D,
                                                         const double *__restrict__ ws, const GPtrs *__restrict__ grads) {
    __shared__ double Ai[MAXD * MAXD], G[MAXD * MAXD], T[MAXD * MAXD], dA[MAXD * MAXD];      // 128 KiB
    __shared__ double gc[MAXD], dc[MAXD], red[NT];
    const int l = blockIdx.x, tid = threadIdx.x;
    const double *w = ws + (size_t)l * ws_doubles(D);
    const double *wL = w, *wU = w + D * D, *wW = w + 2 * D * D, *wAi = w + 3 * D * D, *wv = w + 4 * D * D;
    const double *sc = wv, *sh = wv + D, *dg = wv + 2 * D, *ud = wv + 3 * D, *cc = wv + 4 * D;
    const float *Gin = g_ainv[l], *gcin = g_cinv[l];
    const GPtrs O = grads[l];
    // S = sum_r dL/dlog q[r] = q_sign * sum_b g_logp[b]   (q_sign = -1: the entropy term of log_p = h + q_log_p, each image's K rows share g_logp[b] / K)
    double s = 0.0;
    if (g_logp) for (int i = tid; i < n_logp; i += NT) s += (double)g_logp[i];
    red[tid] = s;
    if (tid < D) gc[tid] = (double)gcin[tid];
    __syncthreads();
    for (int o = NT / 2; o > 0; o >>= 1) { if (tid < o) red[tid] += red[tid + o]; __syncthreads(); }
    const double S = (double)q_sign * red[0];
    for (int i = tid; i < D * D; i += NT) {
        const int r = i / D, k = i % D;
        Ai[i] = wAi[i];
        G[i] = (double)Gin[r * 64 + k] - gc[r] * cc[k];        // c^-1 = -A^-1 c
    }
    __syncthreads();
    if (tid < D) {
        double a = 0.0;
        for (int k = 0; k < D; ++k) a = fma(Ai[k * D + tid], gc[k], a);
        dc[tid] = -a;
    }
    for (int i = tid; i < D * D; i += NT) {                     // T = A^-T G
        const int r = i / D, k = i % D;
        double a = 0.0;
        for (int m = 0; m < D; ++m) a = fma(Ai[m * D + r], G[m * D + k], a);
        T[i] = a;
    }
    __syncthreads();
    for (int i = tid; i < D * D; i += NT) {                     // dA = -T A^-T
        const int r = i / D, k = i % D;
        double a = 0.0;
        for (int m = 0; m < D; ++m) a = fma(T[r * D + m], Ai[k * D + m], a);
        dA[i] = -a;
    }
    __syncthreads();
    for (int i = tid; i < D * D; i += NT) {                     // dW -> G
        const int r = i / D, k = i % D;
        G[i] = dA[i] * sc[k] + dc[r] * sh[k];
    }
    if (tid < D) {
        double a = 0.0, b = 0.0;
        for (int m = 0; m < D; ++m) { a = fma(dA[m * D + tid], wW[m * D + tid], a); b = fma(wW[m * D + tid], dc[m], b); }
        O.p[0][tid] = (float)(a * sc[tid] + S);                 // log_scale
        O.p[1][tid] = (float)b;                                 // shift
        O.p[5][tid] = (float)dc[tid];                           // bias
    }
    __syncthreads();
    for (int i = tid; i < D * D; i += NT) {
        const int r = i / D, k = i % D;
        if (r > k) {                                            // dL = dW U^T, strictly lower entries
            double a = 0.0;
            for (int m = 0; m < D; ++m) a = fma(G[r * D + m], wU[k * D + m], a);
            O.p[2][low_idx(r, k)] = (float)a;
        } else {                                                // dU = L^T dW, upper entries and the diagonal
            double a = 0.0;
            for (int m = 0; m < D; ++m) a = fma(wL[m * D + r], G[m * D + k], a);
            if (r < k) O.p[3][up_idx(r, k, D)] = (float)a;
            else O.p[4][r] = (float)((a + S / dg[r]) / (1.0 + exp(-ud[r])));
        }
    }
}

}}  // namespace mhe::glowaff

using namespace mhe;

extern "C" size_t mhe_glow_affine_workspace_doubles(int layers, int features) {
    return layers > 0 && features > 0 && features <= glowaff::MAXD ? (size_t)layers * glowaff::ws_doubles(features) : 0;
}

extern "C" int mhe_glow_affine_f64(const void *param_ptrs, int layers, int features, float eps, float *A, float *c, float *Ainv, float *AinvT,
                                   float *cinv, float *const_parts, double *workspace, void *stream) {
    MHE_REQUIRE(param_ptrs && A && c && Ainv && AinvT && cinv && const_parts && workspace, "mhe_glow_affine_f64: null pointer");
    MHE_REQUIRE(layers > 0 && features > 1 && features <= glowaff::MAXD, "mhe_glow_affine_f64: features=%d (2..%d), layers=%d", features, glowaff::MAXD, layers);
    hipLaunchKernelGGL(glowaff::affine_kernel, dim3(layers), dim3(glowaff::NT), 0, (hipStream_t)stream, (const glowaff::Ptrs *)param_ptrs, features, eps,
                       A, c, Ainv, AinvT, cinv, const_parts, workspace);
    return check_launch("glowaff::affine_kernel");
}

extern "C" int mhe_glow_reparam_bwd_f64(const void *g_ainv_ptrs, const void *g_cinv_ptrs, const float *g_log_p, int n_log_p, float q_sign, int layers,
                                        int features, const double *workspace, const void *grad_ptrs, void *stream) {
    MHE_REQUIRE(g_ainv_ptrs && g_cinv_ptrs && workspace && grad_ptrs, "mhe_glow_reparam_bwd_f64: null pointer");
    MHE_REQUIRE(layers > 0 && features > 1 && features <= glowaff::MAXD && (n_log_p >= 0), "mhe_glow_reparam_bwd_f64: bad arguments");
    hipLaunchKernelGGL(glowaff::reparam_bwd_kernel, dim3(layers), dim3(glowaff::NT), 0, (hipStream_t)stream, (const float *const *)g_ainv_ptrs,
                       (const float *const *)g_cinv_ptrs, g_log_p, g_log_p ? n_log_p : 0, q_sign, features, workspace, (const glowaff::GPtrs *)grad_ptrs);
    return check_launch("glowaff::reparam_bwd_kernel");
}

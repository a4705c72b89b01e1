// Train-mode BatchNorm statistics of a 1x1 convolution's output WITHOUT evaluating the convolution (forward-only path of ResNet-50's
// layer1 / layer2 conv3 -> bn3, reference hand/network.py:54-61,110 = torchvision Bottleneck.bn3 in training mode).
//
// For y[p][c] = sum_i w[c][i] a[p][i] (a = relu(bn2(y2)) rounded to the storage type, the operand conv3 multiplies):
//     sum_p y[p][c]    = w[c] . m,          m[i]    = sum_p a[p][i]
//     sum_p y[p][c]^2  = w[c]^T G w[c],     G[i][j] = sum_p a[p][i] a[p][j]            (the input's Gram matrix)
// so bn3's batch mean / variance over the 256 / 512 output channels follow from the 64 / 128-channel input's first and second moments:
// 8x / 4x fewer multiply-adds than the product itself, no per-output-element work at all, and the only HBM traffic is one read of the
// bottleneck-wide operand (134 / 67 MB at config C2).  The statistics-only launch of the streaming kernel it replaces evaluated the whole
// product, rounded and summed every output element: 86 / 95 us, VALU-bound; this is a weight-gradient-shaped kernel (K = pixels) at the
// HBM read rate.  The statistics are those of the f32 products (the stored-output form rounded them to bf16 first): closer to the
// reference's f32 arithmetic, different from the unfused path by the rounding noise of the sums (~1e-4 of the variance).
//   gram_kernel     : persistent workgroups, 128-pixel tiles register-staged into a double-buffered LDS image ([pixel][channel] rows, 64-byte
//                     segments XOR-swizzled as in wgrad.hip), G accumulated on v_mfma_f32_32x32x16_bf16 from transposing reads
//                     (ds_read_b64_tr_b16) over the workgroup's whole life; at the end the workgroup STORES its partial G and m (f32) as slab
//                     blockIdx of the workspace - no atomics (up to round 3: f32 atomics into 16 shards, order-dependent; round 4 first:
//                     one round of Cb^2 64-bit fixed-point atomics per workgroup, 10-20 us of a 35-50 us launch);
//   gram_combine    : slabs -> f64 totals, summed in slab order: the totals do not depend on the order the workgroups finish in, and
//                     nothing has to be zeroed or cleared (a launch overwrites its slabs and records how many it wrote);
//   gram_finalize   : one wave per output channel: w^T G w and w . m in f64 -> mean, variance -> scale / shift (+ running statistics).
#include "conv_shared.h"
#include <cstdlib>

namespace mhe { namespace conv {

namespace {
typedef short v4s __attribute__((ext_vector_type(4)));
typedef short v8s __attribute__((ext_vector_type(8)));
typedef __attribute__((ext_vector_type(8))) __bf16 bf8g;
typedef __attribute__((address_space(3))) v4s lds_v4s;
constexpr int GMAXWG = 512;               // most workgroups (= partial slabs) of a launch
constexpr int GPX = 128;                  // pixels per tile
__device__ __forceinline__ int gseg_swz(int pitch, int row) { return pitch >= 256 ? (row & 3) : ((row >> 1) & 1); }
}

template <int CB>
__global__ __launch_bounds__(256) void gram_kernel(const u16 *__restrict__ x, const float *__restrict__ in_scale, const float *__restrict__ in_shift,
                                                   float *__restrict__ gram, int M, int relu, u16 *__restrict__ a_out) {
    constexpr int PA = CB * 2, CPR = CB / 8, RPP = 256 / CPR, NJ = GPX / RPP;      // row pitch (bytes), 16-byte chunks per row, rows per pass, passes
    constexpr int TW = CB / 64;                                                    // 32 x 32 tiles per wave and dimension (waves 2 x 2)
    constexpr int GE = CB * CB + CB;                                               // floats per slab: G then m
    __shared__ __attribute__((aligned(1024))) char tile[2][GPX * PA];
    __shared__ float red[256 * 8];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int cch = tid % CPR, r0 = tid / CPR;
    const int ntiles = M / GPX;
    float sc[8], sh[8], cs[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { sc[i] = in_scale ? in_scale[cch * 8 + i] : 1.f; sh[i] = in_shift ? in_shift[cch * 8 + i] : 0.f; cs[i] = 0.f; }
    // two tiles in flight in registers (one workgroup per CU: every workgroup's atomics are a full round of Cb^2 integer adds, so fewer,
    // longer-lived workgroups - with the second tile in flight doing what the second workgroup per CU did for the load latency)
    struct Regs { uint4 v[NJ]; };
    Regs ra, rb;
    auto load_tile = [&](int L, Regs &r) __attribute__((always_inline)) {
        const int Lc = L < ntiles ? L : ntiles - 1;                                // (past the end: a harmless re-read, never stored)
#pragma unroll
        for (int j = 0; j < NJ; ++j) r.v[j] = *reinterpret_cast<const uint4 *>(x + ((size_t)Lc * GPX + r0 + RPP * j) * CB + cch * 8);
    };
    auto store_tile = [&](int buf, int L, const Regs &r) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int row = r0 + RPP * j;
            float f[8];
            Chunk<u16>::unpack(r.v[j], f);
            if (in_scale) {
#pragma unroll
                for (int i = 0; i < 8; ++i) { f[i] = fmaf(f[i], sc[i], sh[i]); if (relu) f[i] = fmaxf(f[i], 0.f); }
            }
            const uint4 o = Chunk<u16>::pack(f);
            Chunk<u16>::unpack(o, f);                                              // the values the matrix cores multiply
#pragma unroll
            for (int i = 0; i < 8; ++i) cs[i] += f[i];
            const int pc = (((cch >> 2) ^ gseg_swz(PA, row)) << 2) | (cch & 3);
            *reinterpret_cast<uint4 *>(tile[buf] + row * PA + pc * 16) = o;
            // (train step: the normalised operand itself, which the reverse pass multiplies again - one pass over the raw tensor instead of two)
            if (a_out) *reinterpret_cast<uint4 *>(a_out + ((size_t)L * GPX + row) * CB + cch * 8) = o;
        }
    };
    // transposing-read geometry (wgrad.hip): 16-lane group g reads pixels 8 (g >> 1) + q (+ 4), channels 16 (g & 1) + 4 pq of a 32-channel block
    const int g = lane >> 4, q = (lane >> 2) & 3, pq = lane & 3;
    const int fr = 8 * (g >> 1) + q;
    int a_off[TW], b_off[TW];
#pragma unroll
    for (int i = 0; i < TW; ++i) {
        const int sa = wm * TW + i, sb = wn * TW + i;
        a_off[i] = fr * PA + ((sa ^ gseg_swz(PA, fr)) * 64) + (16 * (g & 1) + 4 * pq) * 2;
        b_off[i] = fr * PA + ((sb ^ gseg_swz(PA, fr)) * 64) + (16 * (g & 1) + 4 * pq) * 2;
    }
    v16f acc[TW][TW];
#pragma unroll
    for (int i = 0; i < TW; ++i)
#pragma unroll
        for (int j = 0; j < TW; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int G = (int)gridDim.x;
    auto products = [&](int buf) __attribute__((always_inline)) {
        const char *base = tile[buf];
#pragma unroll
        for (int kk = 0; kk < GPX; kk += 16) {
            v8s fa[TW], fb[TW];
#pragma unroll
            for (int i = 0; i < TW; ++i) {
                const v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s *)(base + a_off[i] + kk * PA));
                const v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s *)(base + a_off[i] + (kk + 4) * PA));
                fa[i] = v8s{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                const v4s lo2 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s *)(base + b_off[i] + kk * PA));
                const v4s hi2 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s *)(base + b_off[i] + (kk + 4) * PA));
                fb[i] = v8s{lo2[0], lo2[1], lo2[2], lo2[3], hi2[0], hi2[1], hi2[2], hi2[3]};
            }
#pragma unroll
            for (int i = 0; i < TW; ++i)
#pragma unroll
                for (int j = 0; j < TW; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8g, fa[i]), __builtin_bit_cast(bf8g, fb[j]), acc[i][j], 0, 0, 0);
        }
    };
    load_tile(blockIdx.x, ra);
    load_tile(blockIdx.x + G, rb);
    for (int L = blockIdx.x; L < ntiles; L += 2 * G) {
        store_tile(0, L, ra);
        load_tile(L + 2 * G, ra);                                // in flight during two tiles' products
        __syncthreads();                                         // (also: every wave is done reading the buffer written two tiles ago)
        products(0);
        if (L + G < ntiles) {                                    // (uniform over the workgroup)
            store_tile(1, L + G, rb);
            load_tile(L + 3 * G, rb);
            __syncthreads();
            products(1);
        }
    }
    if (blockIdx.x == 0 && tid == 0) reinterpret_cast<int *>(gram)[0] = (int)gridDim.x;      // slabs written by this launch (gram_combine)
    float *dst = gram + 2 + (size_t)blockIdx.x * GE;
#pragma unroll
    for (int i = 0; i < TW; ++i)
#pragma unroll
        for (int j = 0; j < TW; ++j) {
            const int n = (wn * TW + j) * 32 + (lane & 31);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = (wm * TW + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                dst[m * CB + n] = acc[i][j][r];
            }
        }
    // column sums: RPP threads share a channel chunk
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 8; ++i) red[tid * 8 + i] = cs[i];
    __syncthreads();
    if (tid < CB) {
        const int c8 = tid >> 3, e = tid & 7;
        float a = 0.f;
        for (int k = 0; k < RPP; ++k) a += red[(c8 + CPR * k) * 8 + e];
        dst[CB * CB + tid] = a;
    }
}

// totals in f64: entry e = sum over the launch's slabs in slab order (four slab lanes per entry, folded in lane order: a fixed order)
__global__ __launch_bounds__(256) void gram_combine_kernel(const float *__restrict__ gram, double *__restrict__ tot, int ge) {
    __shared__ double part[256];
    const int el = threadIdx.x & 63, zl = threadIdx.x >> 6, e = blockIdx.x * 64 + el;
    const int n = reinterpret_cast<const int *>(gram)[0];
    double a = 0.0;
    if (e < ge) {
        const float *src = gram + 2 + e;
#pragma unroll 8
        for (int b = zl; b < n; b += 4) a += (double)src[(size_t)b * ge];
    }
    part[threadIdx.x] = a;
    __syncthreads();
    if (zl == 0 && e < ge) tot[e] = ((part[el] + part[64 + el]) + part[128 + el]) + part[192 + el];
}

// one wave per output channel c: q = w^T G w, s = w . m in f64 -> the BatchNorm affine (torch semantics, as bn_finalize_kernel)
template <int CB>
__global__ __launch_bounds__(256) void gram_finalize_kernel(const double *__restrict__ tot, const u16 *__restrict__ w, const float *__restrict__ gamma,
                                                            const float *__restrict__ beta, float *__restrict__ rmean, float *__restrict__ rvar,
                                                            float *__restrict__ scale, float *__restrict__ shift, float *__restrict__ mean_invstd, int C,
                                                            double count, float momentum, float eps, long long *__restrict__ nbt) {
    constexpr int NC = CB / 64;
    __shared__ double wl[4][CB];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63, c = blockIdx.x * 4 + wv;
    if (nbt && blockIdx.x == 0 && threadIdx.x == 0) *nbt += 1;
    if (c >= C) return;                                          // (whole waves leave together: the LDS exchange below is wave-local)
#pragma unroll
    for (int k = 0; k < NC; ++k) wl[wv][lane + 64 * k] = (double)bf16_to_f32(w[(size_t)c * CB + lane + 64 * k]);
    wave_sync();
    double t[NC];
#pragma unroll
    for (int k = 0; k < NC; ++k) t[k] = 0.0;
    for (int i = 0; i < CB; ++i) {
        const double wi = wl[wv][i];
#pragma unroll
        for (int k = 0; k < NC; ++k) t[k] = fma(wi, tot[(size_t)i * CB + lane + 64 * k], t[k]);
    }
    double qv = 0.0, sv = 0.0;
#pragma unroll
    for (int k = 0; k < NC; ++k) { qv = fma(t[k], wl[wv][lane + 64 * k], qv); sv = fma(wl[wv][lane + 64 * k], tot[(size_t)CB * CB + lane + 64 * k], sv); }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { qv += __shfl_xor(qv, o, 64); sv += __shfl_xor(sv, o, 64); }
    if (lane) return;
    const double dmean = sv / count, dvar0 = qv / count - dmean * dmean, dvar = dvar0 != dvar0 ? dvar0 : fmax(dvar0, 0.0);      // (NaN stays NaN)
    const float mean = (float)dmean, var = (float)dvar;
    const float scv = gamma[c] / sqrtf(var + eps);
    scale[c] = scv;
    shift[c] = beta[c] - mean * scv;
    if (mean_invstd) { mean_invstd[c] = mean; mean_invstd[C + c] = 1.f / sqrtf(var + eps); }
    if (rmean) rmean[c] = (1.f - momentum) * rmean[c] + momentum * mean;
    if (rvar) rvar[c] = (1.f - momentum) * rvar[c] + momentum * var * (float)(count / (count - 1.0));
}

}}  // namespace mhe::conv

using namespace mhe;

// 8-byte words of the partial-slab workspace: a count word, then GMAXWG slabs of Cb^2 + Cb floats
extern "C" size_t mhe_gram_stats_words(int Cb) { return (Cb == 64 || Cb == 128) ? 1 + ((size_t)conv::GMAXWG * ((size_t)Cb * Cb + Cb) + 1) / 2 : 0; }
extern "C" size_t mhe_gram_stats_workspace_bytes(int Cb) { return (Cb == 64 || Cb == 128) ? ((size_t)Cb * Cb + Cb) * sizeof(double) : 0; }

extern "C" int mhe_conv1x1_gram_nhwc(const void *x, const float *in_scale, const float *in_shift, int relu_in, mhe_stat_t *gram, long pixels, int Cb,
                                     void *stream) {
    return mhe_conv1x1_gram_store_nhwc(x, in_scale, in_shift, relu_in, gram, nullptr, pixels, Cb, stream);
}

extern "C" int mhe_conv1x1_gram_store_nhwc(const void *x, const float *in_scale, const float *in_shift, int relu_in, mhe_stat_t *gram, void *a_out,
                                           long pixels, int Cb, void *stream) {
    MHE_REQUIRE(x && gram && pixels > 0 && pixels % conv::GPX == 0 && pixels < (1l << 31) && (Cb == 64 || Cb == 128),
                "mhe_conv1x1_gram_nhwc: bf16 rows of 64 / 128 channels, pixel count a multiple of %d (pixels=%ld Cb=%d)", conv::GPX, pixels, Cb);
    MHE_REQUIRE((in_scale == nullptr) == (in_shift == nullptr), "mhe_conv1x1_gram_nhwc: in_scale/in_shift must come together");
    const int ntiles = (int)(pixels / conv::GPX);
    // workgroups = partial slabs (a workgroup ends with Cb^2 + Cb plain stores whatever it has summed)
    static const int wgs64 = getenv("MHE_GRAM_WGS_64") ? atoi(getenv("MHE_GRAM_WGS_64")) : 512;
    static const int wgs128 = getenv("MHE_GRAM_WGS_128") ? atoi(getenv("MHE_GRAM_WGS_128")) : 256;
    int want = Cb == 64 ? wgs64 : wgs128;
    if (want < 1) want = 1;
    if (want > conv::GMAXWG) want = conv::GMAXWG;
    const dim3 grid((unsigned)(ntiles < want ? ntiles : want));
    if (Cb == 64) hipLaunchKernelGGL(conv::gram_kernel<64>, grid, dim3(256), 0, (hipStream_t)stream, (const u16 *)x, in_scale, in_shift, (float *)gram, (int)pixels, relu_in, (u16 *)a_out);
    else hipLaunchKernelGGL(conv::gram_kernel<128>, grid, dim3(256), 0, (hipStream_t)stream, (const u16 *)x, in_scale, in_shift, (float *)gram, (int)pixels, relu_in, (u16 *)a_out);
    return check_launch("gram_kernel");
}

extern "C" int mhe_gram_bn_finalize(mhe_stat_t *gram, void *workspace, const void *w, const float *gamma, const float *beta, float *running_mean,
                                    float *running_var, float *scale, float *shift, float *mean_invstd, int C, int Cb, double count,
                                    float momentum, float eps, long long *num_batches_tracked, void *stream) {
    MHE_REQUIRE(gram && workspace && w && gamma && beta && scale && shift && C > 0 && (Cb == 64 || Cb == 128) && count > 1.f,
                "mhe_gram_bn_finalize: bad arguments");
    const int ge = Cb * Cb + Cb;
    hipLaunchKernelGGL(conv::gram_combine_kernel, dim3((ge + 63) / 64), dim3(256), 0, (hipStream_t)stream, (const float *)gram, (double *)workspace, ge);
    if (int rc = check_launch("gram_combine_kernel")) return rc;
    if (Cb == 64)
        hipLaunchKernelGGL(conv::gram_finalize_kernel<64>, dim3((C + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const double *)workspace, (const u16 *)w, gamma,
                           beta, running_mean, running_var, scale, shift, mean_invstd, C, (double)count, momentum, eps, num_batches_tracked);
    else
        hipLaunchKernelGGL(conv::gram_finalize_kernel<128>, dim3((C + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const double *)workspace, (const u16 *)w, gamma,
                           beta, running_mean, running_var, scale, shift, mean_invstd, C, (double)count, momentum, eps, num_batches_tracked);
    return check_launch("gram_finalize_kernel");
}

// NHWC convolution / dense layer as an implicit GEMM on MFMA (f32 or bf16 storage,
// f32 accumulate), with the BatchNorm affine + ReLU of the PRODUCER folded into the
// operand load, the BatchNorm affine / residual / ReLU of THIS layer folded into the
// epilogue, and the train-mode batch statistics accumulated from the accumulators.
//
// Replaces the torchvision ResNet trunk the reference instantiates
// (hand/network.py:54-61,110) and, with KH=KW=1, the Linear layers
// (hand/network.py:87,121,380-383; hand/flows.py:108-109).
//
// GEMM view: y^T[n][m] = sum_k w[n][k] * xcol[m][k],  m = output pixel, n = output
// channel, k = (kh, kw, cin) with cin fastest (NHWC keeps a k-chunk contiguous).
// Both operands are staged as rows of 128 bytes of K (32 f32 / 64 bf16) in LDS,
// 16-byte chunks XOR-swizzled by row so the ds_read_b128 fragment reads are
// conflict-free.  A 16x16 MFMA tile reads one 16-byte chunk per lane per operand:
//   f32 : 4 x v_mfma_f32_16x16x4_f32   (lane group q <-> k = 4q+i)
//   bf16: 1 x v_mfma_f32_16x16x32_bf16 (lane group q <-> k = 8q+j)
// The accumulator holds y^T (row = channel 4q+r, column = pixel lane&15), so each
// lane owns 4 consecutive channels of one pixel: the epilogue's per-channel
// scale/shift are one float4 load and the store is one 16 B (f32) / 8 B (bf16) write.
#include "conv_shared.h"
#include <cstdlib>

namespace mhe { namespace conv {

// MODE 0: plain operand load; 1: producer BatchNorm(+ReLU) applied to the operand; 2: residual-tail (dual input);
// 3: plain, the K range continued on a second tensor (1x1 only: p.xcat, p.Cin2 - the data-gradient launch of csrc/conv_fold.hip)
// Tile BM x BN computed by WM x WN wavefronts (64 * WM * WN threads); each wave owns (BM/WM) x (BN/WN).
// Shipped shapes: 128x64 and 128x128 on 2x2 waves (2 workgroups per CU), 256x256 on 2x4 waves (one per CU,
// half the L2->LDS bytes per MAC of 128x128 - the 128-tiles measure L2-fill-bound at ~11 TB/s).
template <typename T, int BM, int BN, int WM, int WN, bool FAST, int MODE, bool DG = false>
__global__ __launch_bounds__(64 * WM * WN, WM * WN == 4 ? 2 : 1) void conv_kernel(const Params p) {
    constexpr int NTH = 64 * WM * WN;
    constexpr int CE = El<T>::CE, BKE = 8 * CE;
    constexpr int RSTEP = NTH / 8;                  // rows covered by one pass of the thread block
    constexpr int NJ_A = BM / RSTEP, NJ_B = BN / RSTEP;
    constexpr int MTW = BM / WM / 16, NTW = BN / WN / 16;     // 16x16 tiles per wave
    __shared__ uint4 lds[2][(BM + BN) * 8];
    constexpr bool AFFM = MODE == 1 || MODE == 2;
    __shared__ __attribute__((aligned(16))) float aff[2][AFFM ? MAXC : 4];     // producer BatchNorm scale / shift

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int q = lane >> 4, l15 = lane & 15;
    const int wr = wave / WN, wc = wave % WN;
    int mtile, ntile;
    tile_of_block(mtile, ntile);
    const int m0 = mtile * BM, n0 = ntile * BN;
    const int s = tid & 7, rbase = tid >> 3;
    const T *xg = reinterpret_cast<const T *>(p.x);
    const T *wg = reinterpret_cast<const T *>(p.w);

    // ---- per-thread activation rows (NJ_A rows, same 16-byte slot)
    int hi0[NJ_A], wi0[NJ_A];
    size_t xb[NJ_A];
    bool mv[NJ_A];
#pragma unroll
    for (int j = 0; j < NJ_A; ++j) {
        const int m = m0 + rbase + RSTEP * j;
        mv[j] = m < p.M;
        const int mm = mv[j] ? m : 0;
        const int wo = mm % p.Wo, t = mm / p.Wo, ho = t % p.Ho, b = t / p.Ho;
        hi0[j] = ho * p.stride - p.pad;
        wi0[j] = wo * p.stride - p.pad;
        xb[j] = (size_t)b * p.H * p.W;
    }
    if constexpr (AFFM) {
        for (int i = tid; i < p.Cin; i += NTH) { aff[0][i] = p.in_scale[i]; aff[1][i] = p.in_shift[i]; }
        __syncthreads();
    }
    const int ntaps = p.KH * p.KW;
    const int nk = p.Kpad / BKE;

    // Loads are only ISSUED in load_stage; the producer's BatchNorm (+ residual tail) is applied in
    // store_stage, one K-stage of MFMAs later, so the transform never waits on a load it has just issued.
    constexpr bool dual = MODE == 2;
    uint4 ra[NJ_A], ra2[dual ? NJ_A : 1], rb[NJ_B];
    size_t aoff[dual ? NJ_A : 1];
    int c_ld = 0;
    unsigned okbits = 0;
    const T *x2g = reinterpret_cast<const T *>(p.x2);
    auto load_stage = [&](int ks) {
        const int kc = ks * BKE + s * CE;
        int tap, c;
        if constexpr (FAST) { tap = (ks * BKE) / p.Cin; c = kc - tap * p.Cin; }
        else                { tap = kc / p.Cin; c = kc - tap * p.Cin; }
        const T *src = xg;
        int pitch = p.Cin;
        if constexpr (MODE == 3) {                        // (uniform per K stage)
            if (ks * BKE >= p.Cin) { src = reinterpret_cast<const T *>(p.xcat); pitch = p.Cin2; c = kc - p.Cin; }
            tap = 0;
        }
        const int kh = tap / p.KW, kw = tap - kh * p.KW;
        const bool tv = tap < ntaps;
        c_ld = c;
        okbits = 0;
#pragma unroll
        for (int j = 0; j < NJ_A; ++j) {
            const int hi = hi0[j] + kh, wi = wi0[j] + kw;
            const bool ok = tv && mv[j] && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
            uint4 v = make_uint4(0, 0, 0, 0);
            if constexpr (dual) ra2[j] = v;
            if (ok) {
                const size_t off = (xb[j] + (size_t)hi * p.W + wi) * pitch + c;
                v = *reinterpret_cast<const uint4 *>(src + off);
                if constexpr (dual) { ra2[j] = *reinterpret_cast<const uint4 *>(x2g + off); aoff[j] = off; }
                if constexpr (AFFM) okbits |= 1u << j;
            }
            ra[j] = v;
        }
#pragma unroll
        for (int j = 0; j < NJ_B; ++j) {
            const int n = n0 + rbase + RSTEP * j;
            rb[j] = n < p.Cout ? *reinterpret_cast<const uint4 *>(wg + (size_t)n * p.Kpad + kc) : make_uint4(0, 0, 0, 0);
        }
    };
    auto store_stage = [&](int buf) {
#pragma unroll
        for (int j = 0; j < NJ_A; ++j) {
            uint4 v = ra[j];
            if constexpr (AFFM) {
                if ((okbits >> j) & 1u) {          // padding stays zero
                    v = in_transform<T>(v, aff[0], aff[1], c_ld, p.relu_in, dual, ra2[dual ? j : 0], p.x2_scale, p.x2_shift);
                    if constexpr (dual) {
                        if (p.a_out && ntile == 0) *reinterpret_cast<uint4 *>(reinterpret_cast<T *>(p.a_out) + aoff[j]) = v;
                    }
                }
            }
            lds[buf][swz(rbase + RSTEP * j, s)] = v;
        }
#pragma unroll
        for (int j = 0; j < NJ_B; ++j) lds[buf][BM * 8 + swz(rbase + RSTEP * j, s)] = rb[j];
    };

    v4f acc[NTW][MTW];
#pragma unroll
    for (int a = 0; a < NTW; ++a)
#pragma unroll
        for (int b = 0; b < MTW; ++b) acc[a][b] = v4f{0.f, 0.f, 0.f, 0.f};

    load_stage(0);
    store_stage(0);
    __syncthreads();
    int cur = 0;
    for (int ks = 0; ks < nk; ++ks) {
        const bool more = ks + 1 < nk;
        if (more) load_stage(ks + 1);
        const uint4 *LA = lds[cur], *LB = lds[cur] + BM * 8;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            uint4 fa[MTW], fb[NTW];
#pragma unroll
            for (int mt = 0; mt < MTW; ++mt) fa[mt] = LA[swz(wr * (BM / WM) + mt * 16 + l15, kk * 4 + q)];
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) fb[nt] = LB[swz(wc * (BN / WN) + nt * 16 + l15, kk * 4 + q)];
            if constexpr (sizeof(T) == 4) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
                        for (int mt = 0; mt < MTW; ++mt) {
                            const float wv = __uint_as_float(reinterpret_cast<const unsigned *>(&fb[nt])[i]);
                            const float xv = __uint_as_float(reinterpret_cast<const unsigned *>(&fa[mt])[i]);
                            acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv, xv, acc[nt][mt], 0, 0, 0);
                        }
            } else {
#pragma unroll
                for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
                    for (int mt = 0; mt < MTW; ++mt)
                        acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                            __builtin_bit_cast(__attribute__((ext_vector_type(8))) __bf16, fb[nt]),
                            __builtin_bit_cast(__attribute__((ext_vector_type(8))) __bf16, fa[mt]), acc[nt][mt], 0, 0, 0);
            }
        }
        if (more) store_stage(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }

    epilogue<T, BM, BN, WM, WN, DG>(p, acc, lds, mtile % NSH, n0, [&](int row) { return out_pixel(p, m0 + row); });
}

// ---------------------------------------------------------------------------
// Stem: 7x7 stride-2 pad-3 convolution 3 -> 64 read straight from the NCHW f32 image (no layout-change
// pass, no channel padding, every input element fetched from HBM once per 8x16 output tile).
// K is ordered (kh, kw, c) with every kh row padded from 21 to 24: k = 24*kh + 3*kw + c.  With the input
// patch stored channel-interleaved in LDS ([row][3*col + c]) the 24 k-values of one (pixel, kh) are 24
// CONSECUTIVE patch elements, so MFMA operand fragments are read directly from the patch (4-byte aligned
// 16-byte pieces) - no im2col buffer is ever built.  k-slots 21..23 of a row read the next pixels' data
// against zero weights.  Algorithmic HBM bytes: 12 B/pixel in, 128 (bf16) | 256 (f32) B/pixel out.
template <typename T>
__global__ __launch_bounds__(256) void stem_kernel(const float *__restrict__ x, const Params p, int tiles_x, int tiles_y) {
    constexpr int CE = El<T>::CE;                       // k per 16-byte chunk
    constexpr int KP = 192, NCH = KP / CE, CPK = 24 / CE;     // chunks in all, chunks per kh row
    constexpr int TH = 8, TW = 16, PH = 2 * TH + 5, PW = 2 * TW + 5;
    constexpr int PSI = 3 * PW + 1;                     // 112 elements per interleaved patch row
    constexpr int PROWS = PH + 3;                       // rows 21..23 are zero: k-slots of the K padding (kh = 7) land there
    constexpr int BM = 128, BN = 64, WM = 2, WN = 2, MTW = 4, NTW = 2;
    constexpr int NST = NCH / 8;
    __shared__ uint4 lds[NST * BN * 8 > BM * BN * (int)sizeof(T) / 16 ? NST * BN * 8 : BM * BN * (int)sizeof(T) / 16];
    __shared__ __attribute__((aligned(16))) T patch[PROWS * PSI];

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int q = lane >> 4, l15 = lane & 15, wr = wave / WN, wc = wave % WN;
    int t = blockIdx.x;
    const int tx = t % tiles_x; t /= tiles_x;
    const int ty = t % tiles_y;
    const int b = t / tiles_y;
    const int oy0 = ty * TH, ox0 = tx * TW;
    const T *wg = reinterpret_cast<const T *>(p.w);

    // every global load is issued (clamped address, no branch) before the first LDS store
    const int iy0 = 2 * oy0 - 3, ix0 = 2 * ox0 - 3;
    {
        constexpr int NP = (3 * PH * PW + 255) / 256, NWL = BN * NCH / 256;
        float pv[NP];
        uint4 wv[NWL];
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            const int i = tid + 256 * j, ic = i < 3 * PH * PW ? i : 0;
            const int c = ic / (PH * PW), r = (ic / PW) % PH, col = ic % PW;
            const int iy = iy0 + r, ix = ix0 + col;
            const bool ok = (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            const float v = x[(((size_t)b * 3 + c) * p.H + (ok ? iy : 0)) * p.W + (ok ? ix : 0)];
            pv[j] = ok ? v : 0.f;
        }
#pragma unroll
        for (int j = 0; j < NWL; ++j) {
            const int i = tid + 256 * j, n = i / NCH, ch = i % NCH;
            wv[j] = *reinterpret_cast<const uint4 *>(wg + (size_t)n * KP + ch * CE);
        }
        for (int i = tid; i < (PROWS - PH) * PSI + PH; i += 256) {      // zero rows + the pad column of real rows
            const int idx = i < (PROWS - PH) * PSI ? PH * PSI + i : (i - (PROWS - PH) * PSI) * PSI + 3 * PW;
            if constexpr (sizeof(T) == 4) patch[idx] = 0.f; else patch[idx] = 0;
        }
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            const int i = tid + 256 * j;
            if (i < 3 * PH * PW) {
                const int c = i / (PH * PW), r = (i / PW) % PH, col = i % PW;
                if constexpr (sizeof(T) == 4) patch[r * PSI + 3 * col + c] = pv[j]; else patch[r * PSI + 3 * col + c] = f32_to_bf16(pv[j]);
            }
        }
#pragma unroll
        for (int j = 0; j < NWL; ++j) {
            const int i = tid + 256 * j, n = i / NCH, ch = i % NCH;
            lds[(ch >> 3) * BN * 8 + swz(n, ch & 7)] = wv[j];
        }
    }
    __syncthreads();

    v4f acc[NTW][MTW];
#pragma unroll
    for (int a = 0; a < NTW; ++a)
#pragma unroll
        for (int c = 0; c < MTW; ++c) acc[a][c] = v4f{0.f, 0.f, 0.f, 0.f};
    const unsigned *pw = reinterpret_cast<const unsigned *>(patch);
#pragma unroll
    for (int ks = 0; ks < NCH / 4; ++ks) {              // 4 chunks (one per lane group q) per step
        const int chunk = 4 * ks + q, kh = chunk / CPK, jj = chunk - kh * CPK;
        uint4 fa[MTW], fb[NTW];
#pragma unroll
        for (int mt = 0; mt < MTW; ++mt) {
            // tile row 16*(4wr+mt) + l15  <->  pixel (dy = 4wr+mt, dx = l15)
            const int e = (2 * (4 * wr + mt) + kh) * PSI + 6 * l15 + CE * jj;       // first element; byte offset is 4-aligned
            const unsigned *src = pw + e * (int)sizeof(T) / 4;
            fa[mt] = make_uint4(src[0], src[1], src[2], src[3]);
        }
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) fb[nt] = lds[(chunk >> 3) * BN * 8 + swz(wc * 32 + nt * 16 + l15, chunk & 7)];
        if constexpr (sizeof(T) == 4) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
                    for (int mt = 0; mt < MTW; ++mt) {
                        const float wv = __uint_as_float(reinterpret_cast<const unsigned *>(&fb[nt])[i]);
                        const float xv = __uint_as_float(reinterpret_cast<const unsigned *>(&fa[mt])[i]);
                        acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv, xv, acc[nt][mt], 0, 0, 0);
                    }
        } else {
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
                for (int mt = 0; mt < MTW; ++mt)
                    acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                        __builtin_bit_cast(__attribute__((ext_vector_type(8))) __bf16, fb[nt]),
                        __builtin_bit_cast(__attribute__((ext_vector_type(8))) __bf16, fa[mt]), acc[nt][mt], 0, 0, 0);
        }
    }
    __syncthreads();
    epilogue<T, BM, BN, WM, WN, false>(p, acc, lds, (int)(blockIdx.x % NSH), 0, [&](int row) {
        const int oy = oy0 + (row >> 4), ox = ox0 + (row & 15);
        return (oy < p.Ho && ox < p.Wo) ? ((long)b * p.Ho + oy) * p.Wo + ox : -1l;
    });
}

// ---------------------------------------------------------------------------
// one wavefront per channel: lane = statistic shard (NSH == 64); the shard words are summed as INTEGERS (exact, order-free) and
// decoded to f64 once (the serial walk over the 64 shards made this ~6 us latency-bound kernel, launched once per BatchNorm, 3 % of
// the forward step).  `stats` is read and - clear != 0 - zeroed through the SAME pointer (loads first, then the zero stores).
__global__ __launch_bounds__(256) void bn_finalize_kernel(mhe_stat_t *stats, const float *__restrict__ gamma,
                                                          const float *__restrict__ beta, float *__restrict__ rmean, float *__restrict__ rvar,
                                                          float *__restrict__ scale, float *__restrict__ shift, float *__restrict__ mean_invstd, int C,
                                                          double count, float momentum, float eps, int clear,
                                                          long long *__restrict__ num_batches_tracked) {
    static_assert(NSH == 64, "one lane per statistic shard");
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (num_batches_tracked && blockIdx.x == 0 && threadIdx.x == 0) *num_batches_tracked += 1;      // nn.BatchNorm2d's counter (int64)
    if (c >= C) return;
    // self-cleaning accumulators: the arena is zero again when the next forward starts (no memset launch per step)
    double s1, s2;
    fx::wave_totals(stats, C, c, lane, clear != 0, s1, s2);
    if (lane) return;
    const double dmean = s1 / count;
    // biased, as F.batch_norm normalises with; a NaN total (the accumulators' out-of-range / non-finite marker) must stay NaN: fmax(NaN, 0) is 0,
    // which made an overflowed sum of squares a silent variance of ZERO up to round 4
    const double dvar0 = s2 / count - dmean * dmean, dvar = dvar0 != dvar0 ? dvar0 : fmax(dvar0, 0.0);
    const float mean = (float)dmean, var = (float)dvar;
    const float sc = gamma[c] / sqrtf(var + eps);
    scale[c] = sc;
    shift[c] = beta[c] - mean * sc;
    if (mean_invstd) { mean_invstd[c] = mean; mean_invstd[C + c] = 1.f / sqrtf(var + eps); }     // kept for the reverse pass
    if (rmean) rmean[c] = (1.f - momentum) * rmean[c] + momentum * mean;
    if (rvar) rvar[c] = (1.f - momentum) * rvar[c] + momentum * var * (float)(count / (count - 1.0));
}

// y = relu?(x*scale+shift + (res*rscale+rshift | res))   4 channels per thread
template <typename T>
__global__ void bn_act_kernel(const T *__restrict__ x, const float *__restrict__ scale, const float *__restrict__ shift,
                              const T *__restrict__ res, const float *__restrict__ rscale, const float *__restrict__ rshift,
                              T *__restrict__ y, size_t n4, int C, int relu) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const size_t e = i * 4;
        const int c = (int)(e % C);
        float v[4], r[4];
        load4<T>(x + e, v);
        const float4 s = *reinterpret_cast<const float4 *>(scale + c), t = *reinterpret_cast<const float4 *>(shift + c);
        v[0] = fmaf(v[0], s.x, t.x); v[1] = fmaf(v[1], s.y, t.y); v[2] = fmaf(v[2], s.z, t.z); v[3] = fmaf(v[3], s.w, t.w);
        if (res) {
            load4<T>(res + e, r);
            if (rscale) {
                const float4 rs = *reinterpret_cast<const float4 *>(rscale + c), rt = *reinterpret_cast<const float4 *>(rshift + c);
                r[0] = fmaf(r[0], rs.x, rt.x); r[1] = fmaf(r[1], rs.y, rt.y); r[2] = fmaf(r[2], rs.z, rt.z); r[3] = fmaf(r[3], rs.w, rt.w);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] += r[k];
        }
        if (relu) {
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = fmaxf(v[k], 0.f);
        }
        store4<T>(y + e, v);
    }
}

// 3x3 stride-2 pad-1 max pool over relu(x*scale+shift); 4 channels per thread
template <typename T>
__global__ void maxpool_kernel(const T *__restrict__ x, const float *__restrict__ scale, const float *__restrict__ shift,
                               T *__restrict__ y, int B, int H, int W, int C, int Ho, int Wo) {
    const size_t n4 = (size_t)B * Ho * Wo * C / 4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const size_t e = i * 4;
        const int c = (int)(e % C);
        size_t t = e / C;
        const int wo = (int)(t % Wo); t /= Wo;
        const int ho = (int)(t % Ho);
        const int b = (int)(t / Ho);
        float4 s = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
        if (scale) { s = *reinterpret_cast<const float4 *>(scale + c); sh = *reinterpret_cast<const float4 *>(shift + c); }
        float m[4] = {-3.0e38f, -3.0e38f, -3.0e38f, -3.0e38f};
#pragma unroll
        for (int dh = 0; dh < 3; ++dh)
#pragma unroll
            for (int dw = 0; dw < 3; ++dw) {
                const int hi = 2 * ho - 1 + dh, wi = 2 * wo - 1 + dw;
                if ((unsigned)hi >= (unsigned)H || (unsigned)wi >= (unsigned)W) continue;
                float v[4];
                load4<T>(x + (((size_t)b * H + hi) * W + wi) * C + c, v);
                v[0] = fmaf(v[0], s.x, sh.x); v[1] = fmaf(v[1], s.y, sh.y); v[2] = fmaf(v[2], s.z, sh.z); v[3] = fmaf(v[3], s.w, sh.w);
#pragma unroll
                for (int k = 0; k < 4; ++k) m[k] = fmaxf(m[k], v[k]);
            }
        if (scale) {
#pragma unroll
            for (int k = 0; k < 4; ++k) m[k] = fmaxf(m[k], 0.f);       // relu commutes with max
        }
        store4<T>(y + e, m);
    }
}

// the same for bf16 with 8 channels (16 bytes) per thread: the nine window loads are issued unconditionally from clamped addresses
// (out-of-range taps are dropped by a select afterwards), so they are in flight together instead of one guarded 8-byte load at a time
__global__ __launch_bounds__(256) void maxpool8_kernel(const u16 *__restrict__ x, const float *__restrict__ scale, const float *__restrict__ shift,
                                                       u16 *__restrict__ y, int B, int H, int W, int C, int Ho, int Wo) {
    const size_t n8 = (size_t)B * Ho * Wo * C / 8;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (size_t)gridDim.x * blockDim.x) {
        const size_t e = i * 8;
        const int c = (int)(e % C);
        size_t t = e / C;
        const int wo = (int)(t % Wo); t /= Wo;
        const int ho = (int)(t % Ho);
        const int b = (int)(t / Ho);
        uint4 raw[9];
        bool ok[9];
#pragma unroll
        for (int dh = 0; dh < 3; ++dh)
#pragma unroll
            for (int dw = 0; dw < 3; ++dw) {
                const int hi = 2 * ho - 1 + dh, wi = 2 * wo - 1 + dw;
                ok[dh * 3 + dw] = (unsigned)hi < (unsigned)H && (unsigned)wi < (unsigned)W;
                const int hc = hi < 0 ? 0 : (hi >= H ? H - 1 : hi), wc = wi < 0 ? 0 : (wi >= W ? W - 1 : wi);
                raw[dh * 3 + dw] = *reinterpret_cast<const uint4 *>(x + (((size_t)b * H + hc) * W + wc) * C + c);
            }
        float sc[8], sf[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) { sc[k] = scale ? scale[c + k] : 1.f; sf[k] = scale ? shift[c + k] : 0.f; }
        float m[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) m[k] = -3.0e38f;
#pragma unroll
        for (int j = 0; j < 9; ++j) {
            float v[8];
            Chunk<u16>::unpack(raw[j], v);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float a = fmaf(v[k], sc[k], sf[k]);
                m[k] = ok[j] ? fmaxf(m[k], a) : m[k];
            }
        }
        if (scale) {
#pragma unroll
            for (int k = 0; k < 8; ++k) m[k] = fmaxf(m[k], 0.f);       // relu commutes with max
        }
        *reinterpret_cast<uint4 *>(y + e) = Chunk<u16>::pack(m);
    }
}

// global average pool: block = (image b, 64-channel group); 4 pixel groups x 64 channels
template <typename T>
__global__ __launch_bounds__(256) void avgpool_kernel(const T *__restrict__ x, float *__restrict__ y, int HW, int C) {
    __shared__ float part[4][64];
    const int b = blockIdx.y, c = blockIdx.x * 64 + (threadIdx.x & 63), g = threadIdx.x >> 6;
    float a = 0.f;
    if (c < C)
        for (int p = g; p < HW; p += 4) {
            if constexpr (sizeof(T) == 4) a += x[((size_t)b * HW + p) * C + c];
            else a += bf16_to_f32(x[((size_t)b * HW + p) * C + c]);
        }
    part[g][threadIdx.x & 63] = a;
    __syncthreads();
    if (g == 0 && c < C) y[(size_t)b * C + c] = (part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x]) / (float)HW;
}

// the last block's tail and the global average pool in one pass (round 4; the forward ended with a 42 us bn_act pass writing the
// 8 x 8 x 2048 block output that only the 24 us pool read): y[b][c] = mean_p round_T(relu(x * scale + shift + (res * rscale + rshift | res))).
// Rounded to the storage type before it is summed and summed in avgpool_kernel's order (four pixel groups p = g, g + 4, ...; then
// (s0 + s1 + s2 + s3) / HW): the pooled feature equals the two launches' bit for bit.  Thread = 4 channels of one pixel group.
template <typename T>
__global__ __launch_bounds__(256) void bn_act_avgpool_kernel(const T *__restrict__ x, const float *__restrict__ scale, const float *__restrict__ shift,
                                                             const T *__restrict__ res, const float *__restrict__ rscale, const float *__restrict__ rshift,
                                                             float *__restrict__ y, int HW, int C, int relu) {
    __shared__ float part[4][256];
    const int b = blockIdx.y, l = threadIdx.x & 63, g = threadIdx.x >> 6, c = blockIdx.x * 256 + l * 4;
    float a[4] = {0.f, 0.f, 0.f, 0.f};
    if (c < C) {
        const float4 s = *reinterpret_cast<const float4 *>(scale + c), t = *reinterpret_cast<const float4 *>(shift + c);
        float4 rs = make_float4(1.f, 1.f, 1.f, 1.f), rt = make_float4(0.f, 0.f, 0.f, 0.f);
        if (rscale) { rs = *reinterpret_cast<const float4 *>(rscale + c); rt = *reinterpret_cast<const float4 *>(rshift + c); }
        for (int p = g; p < HW; p += 4) {
            const size_t e = ((size_t)b * HW + p) * C + c;
            float v[4], r[4];
            load4<T>(x + e, v);
            v[0] = fmaf(v[0], s.x, t.x); v[1] = fmaf(v[1], s.y, t.y); v[2] = fmaf(v[2], s.z, t.z); v[3] = fmaf(v[3], s.w, t.w);
            if (res) {
                load4<T>(res + e, r);
                if (rscale) { r[0] = fmaf(r[0], rs.x, rt.x); r[1] = fmaf(r[1], rs.y, rt.y); r[2] = fmaf(r[2], rs.z, rt.z); r[3] = fmaf(r[3], rs.w, rt.w); }
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k] += r[k];
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (relu) v[k] = fmaxf(v[k], 0.f);
                if constexpr (sizeof(T) == 2) v[k] = bf16_to_f32(f32_to_bf16(v[k]));      // what bn_act_kernel would have stored
                a[k] += v[k];
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) part[g][l * 4 + k] = a[k];
    __syncthreads();
    const int cc = blockIdx.x * 256 + threadIdx.x;
    if (cc < C) y[(size_t)b * C + cc] = (part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x]) / (float)HW;
}

// NCHW f32 -> NHWC with the channel dimension zero-padded to Cp
template <typename T>
__global__ void nchw_to_nhwc_kernel(const float *__restrict__ x, T *__restrict__ y, int B, int C, int HW, int Cp) {
    const size_t n = (size_t)B * HW;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const size_t b = i / HW, p = i % HW;
        for (int c = 0; c < Cp; ++c) {
            const float v = c < C ? x[(b * C + c) * HW + p] : 0.f;
            if constexpr (sizeof(T) == 4) y[i * Cp + c] = v;
            else y[i * Cp + c] = f32_to_bf16(v);
        }
    }
}

// the same for 3 channels padded to 4 in bf16: one 8-byte store per pixel (the stem's weight gradient reads the image as pixel pairs)
__global__ void nchw3_to_nhwc4_bf16_kernel(const float *__restrict__ x, uint2 *__restrict__ y, int B, int HW) {
    const size_t n = (size_t)B * HW;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const size_t b = i / HW, p = i % HW;
        const float *src = x + b * 3 * HW + p;
        const float c0 = src[0], c1 = src[HW], c2 = src[2 * (size_t)HW];
        y[i] = make_uint2((unsigned)f32_to_bf16(c0) | ((unsigned)f32_to_bf16(c1) << 16), (unsigned)f32_to_bf16(c2));
    }
}

template <typename T, int BM, int BN, int WM, int WN, bool FAST>
static void launch_mode(const Params &p, hipStream_t s) {
    const dim3 grid((p.M + BM - 1) / BM, (p.Cout + BN - 1) / BN), block(64 * WM * WN);
    if (p.x2) {
        if constexpr (FAST) {
            // dual-input operand load with the data-gradient epilogue (BatchNorm-reverse apply on load, train.py): 128-row tiles only
            if constexpr (BM == 128) { if (p.mask) { hipLaunchKernelGGL((conv_kernel<T, BM, BN, WM, WN, true, 2, true>), grid, block, 0, s, p); return; } }
            hipLaunchKernelGGL((conv_kernel<T, BM, BN, WM, WN, true, 2>), grid, block, 0, s, p);
        }
    } else if (p.in_scale) {
        hipLaunchKernelGGL((conv_kernel<T, BM, BN, WM, WN, FAST, 1>), grid, block, 0, s, p);
    } else if (p.xcat) {
        if constexpr (FAST && BM == 128 && sizeof(T) == 2) {
            if (p.mask) hipLaunchKernelGGL((conv_kernel<T, BM, BN, WM, WN, true, 3, true>), grid, block, 0, s, p);
            else hipLaunchKernelGGL((conv_kernel<T, BM, BN, WM, WN, true, 3>), grid, block, 0, s, p);       // (ungated: a shortcut's data gradient)
        }
    } else if (p.mask) {
        hipLaunchKernelGGL((conv_kernel<T, BM, BN, WM, WN, FAST, 0, true>), grid, block, 0, s, p);
    } else {
        hipLaunchKernelGGL((conv_kernel<T, BM, BN, WM, WN, FAST, 0>), grid, block, 0, s, p);
    }
}

bool p8_supports(const Params &p);             // conv_p8.hip: the phase-pipelined 256x256 bf16 kernel (variant 7)
int launch_p8(const Params &p, hipStream_t s);
int launch_p8h(const Params &p, hipStream_t s);   // ... on a 256 x 128 tile (variant 13)
bool stream_supports(const Params &p);         // conv_stream.hip: the streaming 1x1 kernel for K = 64 / 128 (variant 8)
int launch_stream(const Params &p, hipStream_t s);
bool stream3_supports(const Params &p);        // conv_stream.hip: the row-streaming 3x3 kernel for 64 -> 64 channels (variant 9)
int launch_stream3(const Params &p, hipStream_t s);
bool wide_supports(const Params &p);           // conv_wide.hip: resident weight slab + transfer waves for 256 / 512 -> many channels (variant 11)
int launch_wide(const Params &p, hipStream_t s);
bool fuse_supports(const Params &p, int cb);   // conv_fuse.hip: bottleneck tail with conv3 re-evaluated + the next conv1 (variant 12)
int launch_fuse(const Params &p, int cb, hipStream_t s);
bool tail_supports(const Params &p);           // conv_tail.hip: residual tail + conv1 on a 128 x 256 tile with transfer waves (variant 10)
int launch_tail(const Params &p, hipStream_t s);

// tile choice: 0 = 128x64, 1 = 128x128, 2 = 256x256 (FAST only; needs >= ~3/4 of the CUs' worth of tiles)
static int choose_tile(const Params &p, bool fast, bool bf16) {
    static const int env_force = getenv("MHE_CONV_TILE") ? atoi(getenv("MHE_CONV_TILE")) : -1;    // tuning knob
    const int force = p.force >= 0 ? p.force : env_force;
    // data-gradient form with a per-channel constant (mhe_conv2d_masked_bias_nhwc): the kernels with the shared epilogue only
    if ((p.mask && p.out_shift) || p.xcat) return force == 0 || (force < 0 && p.Cout <= 64) ? 0 : 1;
    static const int env_stream = getenv("MHE_CONV_STREAM") ? atoi(getenv("MHE_CONV_STREAM")) : 1;
    // (not chosen for the data-gradient form at 256 input channels - layer3's conv1 reverse, 256 -> 1024: 178 us there against 157 on the
    // phase-pipelined kernel, 220 with the gate as bits; at 64 / 128 it leads by 10-25 %: tools/dg_l3.py)
    if (bf16 && (force == 8 || (force < 0 && env_stream && !(p.mask && p.Cin == 256))) && stream_supports(p)) return 8;
    if (bf16 && (force == 9 || (force < 0 && env_stream)) && stream3_supports(p)) return 9;
    static const int env_wide = getenv("MHE_CONV_WIDE") ? atoi(getenv("MHE_CONV_WIDE")) : 1;
    // (selected by itself at 256 input channels only: at 512 the 64-channel slabs make 32 workgroups re-read every activation tile through
    // one L2 - 62 us against the phase-pipelined kernel's 46 at layer4's conv3)
    if (bf16 && (force == 11 || (force < 0 && env_wide && p.Cin == 256)) && wide_supports(p)) return 11;
    static const int env_tail = getenv("MHE_CONV_TAIL") ? atoi(getenv("MHE_CONV_TAIL")) : 1;
    if (bf16 && (force == 10 || (force < 0 && env_tail)) && tail_supports(p)) return 10;
    if ((force & 15) == 7 && bf16 && p8_supports(p)) return 7;      // (higher bits: ablation builds of tuning runs)
    if (force == 13 && bf16 && p8_supports(p)) return 13;
    // dual-input load + data-gradient epilogue: instantiated on the 128-row tiles only - a forced 256-row tile (descriptor or
    // MHE_CONV_TILE) must not fall through to the plain dual-input kernel, which has no gate / BatchNorm-reverse sums
    if (p.x2 && p.mask) return force == 0 ? 0 : 1;
    if (force >= 0 && force <= 4 && (force < 2 || (fast && bf16))) return force;
    if (p.Cout <= 64) return 0;
    if (fast && bf16 && p.Cout >= 256) {      // (an f32 256x256 output tile would not fit the LDS staging buffers)
        const long tiles = (long)((p.M + 255) / 256) * ((p.Cout + 255) / 256);
        // plain operands go to the phase-pipelined kernel (conv_p8.hip: 0.85-1.0x the time of the register-staged 256x256
        // kernel on every trunk shape, tools/conv_variants.py); the producer-BatchNorm / residual-tail operand loads need
        // the register path
        // residual-tail loads run on the 128x128 tile, two workgroups per CU, whatever the tile count: 113 vs 140 us (layer3's conv1,
        // 256 tiles of 256x256 = a single wave of workgroups with nothing to overlap their epilogue with), 240 vs 254 us (layer3.0),
        // equal on layer4.0 (tools/conv_variants.py --tail)
        if (p.x2) return 1;
        if (tiles >= 192) return p8_supports(p) && force < 0 ? 7 : 2;
    }
    // 3x3 layers with too few pixels for 256-channel tiles (layer4's conv2 and their data gradients: 16k pixels x 512 channels = 128 tiles
    // of 256 x 256): the phase-pipelined kernel on a 256 x 128 tile, one per CU - 78 us against 93 on the 128 x 128 tile at C2.  Long
    // K loops only: at 128 channels (layer2's conv2: 1024 tiles of 18 K tiles, one workgroup per CU) the tile's prologue and epilogue
    // are not covered and the register-staged 128 x 128 tile with two workgroups per CU stays ahead (107 us against 117; tools/conv_variants.py)
    static const int env_p8h = getenv("MHE_CONV_P8H") ? atoi(getenv("MHE_CONV_P8H")) : 1;
    if (fast && bf16 && env_p8h && force < 0 && !p.x2 && p.KH * p.KW > 1 && p.KH * p.KW * p.Cin >= 2304 && p.Cout % 128 == 0 && p8_supports(p) &&
        (long)((p.M + 255) / 256) * (p.Cout / 128) >= 192)
        return 13;
    return 1;
}

template <typename T>
static int launch_conv(const Params &p, hipStream_t s) {
    constexpr int BKE = 8 * El<T>::CE;
    const bool fast = (p.Cin % BKE) == 0;
    if constexpr (sizeof(T) == 2) {
        const int t0 = choose_tile(p, fast, true);
        if (t0 == 8) return launch_stream(p, s);
        if (t0 == 9) return launch_stream3(p, s);
        if (t0 == 10) return launch_tail(p, s);
        if (t0 == 11) return launch_wide(p, s);
        if (t0 == 7) return launch_p8(p, s);
        if (t0 == 13) return launch_p8h(p, s);
    }
    if (p.x2 && !fast) { set_error("residual-tail prologue needs Cin %% %d == 0", BKE); return MHE_ERR_ARG; }
    switch (choose_tile(p, fast, sizeof(T) == 2)) {
        case 0: if (fast) launch_mode<T, 128, 64, 2, 2, true>(p, s); else launch_mode<T, 128, 64, 2, 2, false>(p, s); break;
        case 2: if constexpr (sizeof(T) == 2) launch_mode<T, 256, 256, 2, 4, true>(p, s); break;
        case 3: if constexpr (sizeof(T) == 2) launch_mode<T, 256, 128, 4, 2, true>(p, s); break;
        case 4: if constexpr (sizeof(T) == 2) launch_mode<T, 256, 64, 4, 2, true>(p, s); break;
        default: if (fast) launch_mode<T, 128, 128, 2, 2, true>(p, s); else launch_mode<T, 128, 128, 2, 2, false>(p, s); break;
    }
    return check_launch("conv_kernel");
}

}}  // namespace mhe::conv

using namespace mhe;

static inline int elem_chunk(int dtype) { return dtype == MHE_F32 ? 4 : 8; }

static int conv_entry(const mhe_conv_desc *d, const void *x, const void *w, void *y, const float *in_scale,
                      const float *in_shift, const float *out_scale, const float *out_shift, const void *residual,
                      mhe_stat_t *stats, const void *x2, const float *x2_scale, const float *x2_shift, void *a_out, void *stream,
                      const void *mask = nullptr, const struct BnRev *bn = nullptr, float *y32 = nullptr, const int *scatter = nullptr,
                      const void *xcat = nullptr, int cin2 = 0, const void *mask_bits = nullptr);
struct BnRev { const void *y[2]; const float *mi[2]; mhe_stat_t *stats[2]; };

extern "C" int mhe_conv2d_nhwc(const mhe_conv_desc *d, const void *x, const void *w, void *y, const float *in_scale,
                               const float *in_shift, const float *out_scale, const float *out_shift,
                               const void *residual, mhe_stat_t *stats, void *stream) {
    return conv_entry(d, x, w, y, in_scale, in_shift, out_scale, out_shift, residual, stats, nullptr, nullptr, nullptr,
                      nullptr, stream);
}

extern "C" int mhe_conv2d_f32out_nhwc(const mhe_conv_desc *d, const void *x, const void *w, float *y_f32, const float *out_shift, void *stream) {
    MHE_REQUIRE(y_f32, "mhe_conv2d_f32out_nhwc: null output");
    return conv_entry(d, x, w, nullptr, nullptr, nullptr, nullptr, out_shift, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, stream, nullptr,
                      nullptr, y_f32);
}

extern "C" int mhe_conv1x1_residual_in_nhwc(const mhe_conv_desc *d, const void *x, const void *x2, const void *w, void *y,
                                            const float *in_scale, const float *in_shift, const float *x2_scale,
                                            const float *x2_shift, void *a_out, mhe_stat_t *stats, void *stream) {
    MHE_REQUIRE(d && x2 && in_scale && in_shift, "mhe_conv1x1_residual_in_nhwc: x2, in_scale and in_shift are required");
    MHE_REQUIRE(d->KH == 1 && d->KW == 1 && d->stride == 1 && d->pad == 0, "mhe_conv1x1_residual_in_nhwc: 1x1 stride-1 only");
    MHE_REQUIRE((x2_scale == nullptr) == (x2_shift == nullptr), "mhe_conv1x1_residual_in_nhwc: x2_scale/x2_shift must come together");
    return conv_entry(d, x, w, y, in_scale, in_shift, nullptr, nullptr, nullptr, stats, x2, x2_scale, x2_shift, a_out, stream);
}

extern "C" int mhe_conv1x1_cat_bias_nhwc(const mhe_conv_desc *d, const void *x, const void *xcat, int cin2, const void *w, void *y, const void *residual,
                                         const float *bias, void *stream) {
    MHE_REQUIRE(d && xcat && d->dtype == MHE_BF16 && d->KH == 1 && d->KW == 1 && d->stride == 1 && d->pad == 0 && cin2 > 0 && cin2 % 64 == 0 && d->Cin % 64 == 0,
                "mhe_conv1x1_cat_bias_nhwc: a bf16 1x1 stride-1 launch on two operand tensors with Cin and cin2 multiples of 64");
    MHE_REQUIRE(d->tile == 0 || d->tile == 1 || d->tile == 2, "mhe_conv1x1_cat_bias_nhwc: 128-row register-staged tiles only (tile 0, 1 or 2)");
    return conv_entry(d, x, w, y, nullptr, nullptr, nullptr, bias, residual, nullptr, nullptr, nullptr, nullptr, nullptr, stream, nullptr, nullptr, nullptr,
                      nullptr, xcat, cin2);
}

extern "C" int mhe_conv1x1_residual_in_masked_nhwc(const mhe_conv_desc *d, const void *x, const void *x2, const void *w, void *y,
                                                   const float *in_scale, const float *in_shift, const float *x2_scale,
                                                   const float *x2_shift, void *a_out, const void *residual, const void *mask,
                                                   const void *bn_y0, const float *bn_mean_invstd0, mhe_stat_t *bn_stats0, void *stream) {
    MHE_REQUIRE(d && x2 && in_scale && in_shift && mask, "mhe_conv1x1_residual_in_masked_nhwc: x2, in_scale, in_shift and mask are required");
    MHE_REQUIRE(d->KH == 1 && d->KW == 1 && d->stride == 1 && d->pad == 0, "mhe_conv1x1_residual_in_masked_nhwc: 1x1 stride-1 only");
    MHE_REQUIRE((x2_scale == nullptr) == (x2_shift == nullptr), "mhe_conv1x1_residual_in_masked_nhwc: x2_scale/x2_shift must come together");
    MHE_REQUIRE(!bn_y0 || (bn_mean_invstd0 && bn_stats0), "mhe_conv1x1_residual_in_masked_nhwc: bn_y needs its mean_invstd and stats");
    MHE_REQUIRE(d->tile == 0 || d->tile == 1 || d->tile == 2 || d->tile == 11,
                "mhe_conv1x1_residual_in_masked_nhwc: 128-row tiles only (tile 0, 1, 2, or 11 = the transfer-wave kernel)");
    const BnRev bn = {{bn_y0, nullptr}, {bn_mean_invstd0, nullptr}, {bn_stats0, nullptr}};
    return conv_entry(d, x, w, y, in_scale, in_shift, nullptr, nullptr, residual, nullptr, x2, x2_scale, x2_shift, a_out, stream, mask, &bn);
}

extern "C" int mhe_conv2d_masked_nhwc(const mhe_conv_desc *d, const void *x, const void *w, void *y, const void *residual,
                                      const void *mask, const void *bn_y0, const float *bn_mean_invstd0, mhe_stat_t *bn_stats0,
                                      const void *bn_y1, const float *bn_mean_invstd1, mhe_stat_t *bn_stats1, void *stream) {
    MHE_REQUIRE(mask, "mhe_conv2d_masked_nhwc: mask is required");
    MHE_REQUIRE((!bn_y0 || (bn_mean_invstd0 && bn_stats0)) && (!bn_y1 || (bn_y0 && bn_mean_invstd1 && bn_stats1)),
                "mhe_conv2d_masked_nhwc: each bn_y needs its mean_invstd and stats (and bn_y1 needs bn_y0)");
    const BnRev bn = {{bn_y0, bn_y1}, {bn_mean_invstd0, bn_mean_invstd1}, {bn_stats0, bn_stats1}};
    return conv_entry(d, x, w, y, nullptr, nullptr, nullptr, nullptr, residual, nullptr, nullptr, nullptr, nullptr, nullptr, stream, mask, &bn);
}

extern "C" int mhe_conv2d_masked_bits_nhwc(const mhe_conv_desc *d, const void *x, const void *w, void *y, const void *residual,
                                           const void *mask, const void *mask_bits, const void *bn_y0, const float *bn_mean_invstd0, mhe_stat_t *bn_stats0,
                                           const void *bn_y1, const float *bn_mean_invstd1, mhe_stat_t *bn_stats1, void *stream) {
    MHE_REQUIRE(mask, "mhe_conv2d_masked_bits_nhwc: mask is required (kernels without the bit path read it)");
    MHE_REQUIRE((!bn_y0 || (bn_mean_invstd0 && bn_stats0)) && (!bn_y1 || (bn_y0 && bn_mean_invstd1 && bn_stats1)),
                "mhe_conv2d_masked_bits_nhwc: each bn_y needs its mean_invstd and stats (and bn_y1 needs bn_y0)");
    MHE_REQUIRE(!mask_bits || (d && d->Cout % 8 == 0), "mhe_conv2d_masked_bits_nhwc: Cout must be a multiple of 8");
    const BnRev bn = {{bn_y0, bn_y1}, {bn_mean_invstd0, bn_mean_invstd1}, {bn_stats0, bn_stats1}};
    return conv_entry(d, x, w, y, nullptr, nullptr, nullptr, nullptr, residual, nullptr, nullptr, nullptr, nullptr, nullptr, stream, mask, &bn, nullptr,
                      nullptr, nullptr, 0, mask_bits);
}

extern "C" int mhe_conv2d_masked_bias_nhwc(const mhe_conv_desc *d, const void *x, const void *xcat, int cin2, const void *w, void *y,
                                           const void *residual, const void *mask, const float *bias, const void *bn_y0,
                                           const float *bn_mean_invstd0, mhe_stat_t *bn_stats0, void *stream) {
    MHE_REQUIRE(mask && bias, "mhe_conv2d_masked_bias_nhwc: mask and bias are required");
    MHE_REQUIRE(!xcat || (d && d->dtype == MHE_BF16 && d->KH == 1 && d->KW == 1 && d->stride == 1 && d->pad == 0 && cin2 > 0 && cin2 % 64 == 0 && d->Cin % 64 == 0),
                "mhe_conv2d_masked_bias_nhwc: a concatenated operand needs a bf16 1x1 stride-1 launch with Cin and cin2 multiples of 64");
    MHE_REQUIRE(!bn_y0 || (bn_mean_invstd0 && bn_stats0), "mhe_conv2d_masked_bias_nhwc: bn_y needs its mean_invstd and stats");
    MHE_REQUIRE(d && (d->tile == 0 || d->tile == 1 || d->tile == 2), "mhe_conv2d_masked_bias_nhwc: 128-row register-staged tiles only (tile 0, 1 or 2)");
    const BnRev bn = {{bn_y0, nullptr}, {bn_mean_invstd0, nullptr}, {bn_stats0, nullptr}};
    return conv_entry(d, x, w, y, nullptr, nullptr, nullptr, bias, residual, nullptr, nullptr, nullptr, nullptr, nullptr, stream, mask, &bn, nullptr,
                      nullptr, xcat, xcat ? cin2 : 0);
}

extern "C" int mhe_conv3x3s2_dgrad_nhwc(int B, int Ho, int Wo, int Cout, int Cin, int dtype, const void *gy, const void *const *w4,
                                        void *dx, const void *residual, const void *mask, const void *bn_y0,
                                        const float *bn_mean_invstd0, mhe_stat_t *bn_stats0, const void *bn_y1,
                                        const float *bn_mean_invstd1, mhe_stat_t *bn_stats1, int tile, void *stream) {
    MHE_REQUIRE(gy && w4 && dx && w4[0] && w4[1] && w4[2] && w4[3], "mhe_conv3x3s2_dgrad_nhwc: null pointer");
    MHE_REQUIRE((!bn_y0 || (bn_mean_invstd0 && bn_stats0 && mask)) && (!bn_y1 || (bn_y0 && bn_mean_invstd1 && bn_stats1)),
                "mhe_conv3x3s2_dgrad_nhwc: each bn_y needs its mean_invstd and stats (and the gate)");
    const BnRev bn = {{bn_y0, bn_y1}, {bn_mean_invstd0, bn_mean_invstd1}, {bn_stats0, bn_stats1}};
    for (int py = 0; py < 2; ++py)
        for (int px = 0; px < 2; ++px) {
            // output rows 2i + py: py = 0 sees forward tap kh = 1 at gy row i; py = 1 sees kh = 2 at row i and kh = 0 at row i + 1
            mhe_conv_desc d = {B, Ho, Wo, Cout, Cin, 1 + py, 1 + px, 1, 0, dtype, 0, 0, tile};
            const int sc[2] = {py, px};
            const int rc = conv_entry(&d, gy, w4[2 * py + px], dx, nullptr, nullptr, nullptr, nullptr, residual, nullptr, nullptr, nullptr,
                                      nullptr, nullptr, stream, mask, mask ? &bn : nullptr, nullptr, sc);
            if (rc) return rc;
        }
    return MHE_OK;
}

static int conv_entry(const mhe_conv_desc *d, const void *x, const void *w, void *y, const float *in_scale,
                      const float *in_shift, const float *out_scale, const float *out_shift, const void *residual,
                      mhe_stat_t *stats, const void *x2, const float *x2_scale, const float *x2_shift, void *a_out, void *stream,
                      const void *mask, const BnRev *bn, float *y32, const int *scatter, const void *xcat, int cin2, const void *mask_bits) {
    MHE_REQUIRE(d && x && w && (y || y32), "mhe_conv2d_nhwc: null pointer");
    MHE_REQUIRE(d->dtype == MHE_F32 || d->dtype == MHE_BF16, "mhe_conv2d_nhwc: dtype=%d", d->dtype);
    const int ce = elem_chunk(d->dtype), bke = 8 * ce;
    MHE_REQUIRE(d->B > 0 && d->H > 0 && d->W > 0 && d->KH > 0 && d->KW > 0 && d->stride > 0 && d->pad >= 0,
                "mhe_conv2d_nhwc: bad geometry");
    MHE_REQUIRE(d->Cin % ce == 0, "mhe_conv2d_nhwc: Cin=%d must be a multiple of %d (pad channels)", d->Cin, ce);
    MHE_REQUIRE(d->Cout % ce == 0, "mhe_conv2d_nhwc: Cout=%d must be a multiple of %d", d->Cout, ce);
    MHE_REQUIRE((in_scale == nullptr) == (in_shift == nullptr), "mhe_conv2d_nhwc: in_scale/in_shift must come together");
    MHE_REQUIRE(!in_scale || d->Cin <= conv::MAXC, "mhe_conv2d_nhwc: fused input affine supports Cin <= %d", conv::MAXC);
    conv::Params p{};
    p.x = x; p.w = w; p.y = y; p.in_scale = in_scale; p.in_shift = in_shift; p.out_scale = out_scale;
    p.out_shift = out_shift; p.residual = residual; p.stats = stats; p.mask = mask;
    for (int u = 0; u < 2; ++u) { p.bn_y[u] = bn ? bn->y[u] : nullptr; p.bn_mi[u] = bn ? bn->mi[u] : nullptr; p.bn_stats[u] = bn ? bn->stats[u] : nullptr; }
    p.x2 = x2; p.x2_scale = x2_scale; p.x2_shift = x2_shift; p.a_out = a_out;
    p.B = d->B; p.H = d->H; p.W = d->W; p.Cin = d->Cin; p.Cout = d->Cout; p.KH = d->KH; p.KW = d->KW;
    p.stride = d->stride; p.pad = d->pad;
    p.Ho = (d->H + 2 * d->pad - d->KH) / d->stride + 1;
    p.Wo = (d->W + 2 * d->pad - d->KW) / d->stride + 1;
    p.os2 = 0; p.os_py = p.os_px = 0;
    p.res_s2 = d->res_half ? 1 : 0;
    MHE_REQUIRE(!p.res_s2 || (residual && !scatter && !y32), "mhe_conv2d_nhwc: res_half needs a residual (and no output scatter)");
    if (scatter) {     // parity class of a stride-2 data gradient: one output per input position (reads past the edge give zeros)
        p.Ho = d->H; p.Wo = d->W; p.os2 = 1; p.os_py = scatter[0]; p.os_px = scatter[1];
    }
    MHE_REQUIRE(p.Ho > 0 && p.Wo > 0, "mhe_conv2d_nhwc: empty output");
    const long long M = (long long)p.B * p.Ho * p.Wo;
    MHE_REQUIRE(M < (1ll << 31), "mhe_conv2d_nhwc: too many output pixels");
    p.M = (int)M;
    p.xcat = xcat; p.Cin2 = cin2; p.mask_bits = (const unsigned char *)mask_bits;
    const int ktot = d->KH * d->KW * d->Cin + cin2;
    p.Kpad = (ktot + bke - 1) / bke * bke;       // weight rows are zero-padded to this length by the packer
    p.relu_in = d->relu_in; p.relu_out = d->relu_out;
    p.force = d->tile - 1;
    p.y32 = nullptr;
    if (y32) {        // f32 result of a bf16 product: register-staged kernels only (their epilogue stores straight from the accumulators)
        MHE_REQUIRE(d->dtype == MHE_BF16 && !in_scale && !out_scale && !residual && !stats && !x2 && !mask && !d->relu_out && d->Cout % 4 == 0,
                    "mhe_conv2d_f32out_nhwc: bf16 operands, optional out_shift only");
        p.y32 = y32;
        if (p.force < 0 || p.force > 4) p.force = p.Cout <= 64 ? 0 : 1;
    }
    if (d->dtype == MHE_F32) return conv::launch_conv<float>(p, (hipStream_t)stream);
    return conv::launch_conv<u16>(p, (hipStream_t)stream);
}

// ---- the statistics-only pass and the fused bottleneck tail that re-evaluates conv3 (conv_fuse.hip)
extern "C" int mhe_conv1x1_stats_nhwc(const mhe_conv_desc *d, const void *x, const void *w, const float *in_scale, const float *in_shift,
                                      mhe_stat_t *stats, void *stream) {
    MHE_REQUIRE(d && x && w && stats, "mhe_conv1x1_stats_nhwc: null pointer");
    MHE_REQUIRE(d->dtype == MHE_BF16 && d->KH == 1 && d->KW == 1 && d->stride == 1 && d->pad == 0 && (d->Cin == 64 || d->Cin == 128 || d->Cin == 256) && d->Cout % 256 == 0,
                "mhe_conv1x1_stats_nhwc: bf16 1x1 stride-1 with 64 / 128 / 256 input channels and a multiple of 256 output channels");
    MHE_REQUIRE((in_scale == nullptr) == (in_shift == nullptr), "mhe_conv1x1_stats_nhwc: in_scale/in_shift must come together");
    conv::Params p{};
    p.x = x; p.w = w; p.y = nullptr; p.in_scale = in_scale; p.in_shift = in_shift; p.stats = stats;
    p.B = d->B; p.H = d->H; p.W = d->W; p.Cin = d->Cin; p.Cout = d->Cout; p.KH = 1; p.KW = 1; p.stride = 1; p.pad = 0; p.Ho = d->H; p.Wo = d->W;
    const long long M = (long long)d->B * d->H * d->W;
    MHE_REQUIRE(M > 0 && M < (1ll << 31), "mhe_conv1x1_stats_nhwc: bad pixel count");
    p.M = (int)M; p.Kpad = d->Cin; p.relu_in = d->relu_in; p.force = 8; p.stats_only = 1;
    if (d->Cin == 256) {       // the resident-slab kernel with its stores dropped (conv_wide.hip): the sums of the products as they would be stored
        p.force = 11;
        MHE_REQUIRE(conv::wide_supports(p), "mhe_conv1x1_stats_nhwc: geometry not taken by the resident-slab kernel (M=%lld Cout=%d)", M, d->Cout);
        return conv::launch_wide(p, (hipStream_t)stream);
    }
    MHE_REQUIRE(conv::stream_supports(p), "mhe_conv1x1_stats_nhwc: geometry not taken by the streaming kernel");
    return conv::launch_stream(p, (hipStream_t)stream);
}

extern "C" int mhe_bottleneck_tail_supported(const mhe_conv_desc *d, int Cb) {
    if (!d || d->dtype != MHE_BF16) return 0;
    static const float dummy = 0.f;
    conv::Params p{};
    p.x = p.x2 = p.w = p.w3 = p.y = p.a_out = (void *)&dummy; p.in_scale = p.in_shift = p.mid_scale = p.mid_shift = &dummy;
    p.Cin = d->Cin; p.Cout = d->Cout; p.KH = d->KH; p.KW = d->KW; p.stride = d->stride; p.pad = d->pad; p.Kpad = d->Cin;
    const long long M = (long long)d->B * d->H * d->W;
    if (M <= 0 || M >= (1ll << 31)) return 0;
    p.M = (int)M;
    return conv::fuse_supports(p, Cb) ? 1 : 0;
}

extern "C" int mhe_bottleneck_tail_nhwc(const mhe_conv_desc *d, int Cb, const void *y2, const float *bn2_scale, const float *bn2_shift,
                                        const void *w3, const float *bn3_scale, const float *bn3_shift, const void *identity,
                                        const float *id_scale, const float *id_shift, const void *w1, void *a_out, void *y1, mhe_stat_t *stats,
                                        void *stream) {
    return mhe_bottleneck_tail_bits_nhwc(d, Cb, y2, bn2_scale, bn2_shift, w3, bn3_scale, bn3_shift, identity, id_scale, id_shift, w1, a_out, nullptr, y1, stats, stream);
}

extern "C" int mhe_bottleneck_tail_bits_nhwc(const mhe_conv_desc *d, int Cb, const void *y2, const float *bn2_scale, const float *bn2_shift,
                                             const void *w3, const float *bn3_scale, const float *bn3_shift, const void *identity,
                                             const float *id_scale, const float *id_shift, const void *w1, void *a_out, void *a_bits, void *y1,
                                             mhe_stat_t *stats, void *stream) {
    MHE_REQUIRE(d && y2 && bn2_scale && bn2_shift && w3 && bn3_scale && bn3_shift && identity && w1 && a_out && y1, "mhe_bottleneck_tail_nhwc: null pointer");
    MHE_REQUIRE(d->dtype == MHE_BF16, "mhe_bottleneck_tail_nhwc: bf16 storage only");
    MHE_REQUIRE((id_scale == nullptr) == (id_shift == nullptr), "mhe_bottleneck_tail_nhwc: id_scale/id_shift must come together");
    conv::Params p{};
    p.x = y2; p.in_scale = bn2_scale; p.in_shift = bn2_shift; p.w3 = w3; p.mid_scale = bn3_scale; p.mid_shift = bn3_shift;
    p.x2 = identity; p.x2_scale = id_scale; p.x2_shift = id_shift; p.w = w1; p.a_out = a_out; p.y = y1; p.stats = stats;
    p.a_bits = (unsigned char *)a_bits;
    p.B = d->B; p.H = d->H; p.W = d->W; p.Cin = d->Cin; p.Cout = d->Cout; p.KH = d->KH; p.KW = d->KW; p.stride = d->stride; p.pad = d->pad;
    p.Ho = d->H; p.Wo = d->W; p.Kpad = d->Cin; p.relu_in = 1; p.force = -1;
    const long long M = (long long)d->B * d->H * d->W;
    MHE_REQUIRE(M > 0 && M < (1ll << 31), "mhe_bottleneck_tail_nhwc: bad pixel count");
    p.M = (int)M;
    MHE_REQUIRE(conv::fuse_supports(p, Cb), "mhe_bottleneck_tail_nhwc: needs Cin = 4 Cb, Cb 64 / 128, Cout 64 / 128, pixels %% 128 == 0 (Cin=%d Cb=%d Cout=%d M=%lld)",
                d->Cin, Cb, d->Cout, M);
    return conv::launch_fuse(p, Cb, (hipStream_t)stream);
}

extern "C" int mhe_conv_stat_shards(void) { return conv::NSH; }
extern "C" size_t mhe_stat_words(int C) { return C > 0 ? (size_t)2 * conv::NSH * 2 * (size_t)C : 0; }

// which kernel variant the launcher picks for a geometry with plain operands (see mhe_conv_desc.tile); with a producer-BatchNorm /
// residual-tail operand load variant 7 becomes 2
extern "C" int mhe_conv_tile(const mhe_conv_desc *d) {
    if (!d) return -1;
    conv::Params p{};
    p.B = d->B; p.H = d->H; p.W = d->W; p.Cin = d->Cin; p.Cout = d->Cout; p.KH = d->KH; p.KW = d->KW; p.stride = d->stride; p.pad = d->pad;
    const int Ho = (d->H + 2 * d->pad - d->KH) / d->stride + 1, Wo = (d->W + 2 * d->pad - d->KW) / d->stride + 1;
    p.M = d->B * Ho * Wo;
    p.force = d->tile - 1;
    const int bke = d->dtype == MHE_F32 ? 32 : 64;
    p.Kpad = (d->KH * d->KW * d->Cin + bke - 1) / bke * bke;
    return conv::choose_tile(p, d->Cin % bke == 0, d->dtype == MHE_BF16);
}

// the same for an operand-load form: 1 = producer BatchNorm on load (mhe_conv2d_nhwc with in_scale), 2 = residual-block tail
// (mhe_conv1x1_residual_in_nhwc); 0 = plain (mhe_conv_tile)
extern "C" int mhe_conv_tile_mode(const mhe_conv_desc *d, int mode) {
    if (!d || mode < 0 || mode > 3) return -1;
    static const float dummy = 0.f;
    conv::Params p{};
    p.B = d->B; p.H = d->H; p.W = d->W; p.Cin = d->Cin; p.Cout = d->Cout; p.KH = d->KH; p.KW = d->KW; p.stride = d->stride; p.pad = d->pad;
    const int Ho = (d->H + 2 * d->pad - d->KH) / d->stride + 1, Wo = (d->W + 2 * d->pad - d->KW) / d->stride + 1;
    p.M = d->B * Ho * Wo;
    p.force = d->tile - 1;
    const int bke = d->dtype == MHE_F32 ? 32 : 64;
    p.Kpad = (d->KH * d->KW * d->Cin + bke - 1) / bke * bke;
    if (mode == 1 || mode == 2) p.in_scale = &dummy;
    if (mode == 2) p.x2 = &dummy;
    if (mode == 3) p.residual = &dummy;            // forward form with a residual operand (the streaming kernels do not take it)
    return conv::choose_tile(p, d->Cin % bke == 0, d->dtype == MHE_BF16);
}

extern "C" int mhe_stem_conv7x7s2(const float *x_nchw, const void *w, void *y, mhe_stat_t *stats, int B, int H, int W, int dtype,
                                  void *stream) {
    MHE_REQUIRE(x_nchw && w && y, "mhe_stem_conv7x7s2: null pointer");
    MHE_REQUIRE(B > 0 && H > 0 && W > 0, "mhe_stem_conv7x7s2: bad geometry");
    MHE_REQUIRE(dtype == MHE_F32 || dtype == MHE_BF16, "mhe_stem_conv7x7s2: dtype=%d", dtype);
    conv::Params p{};
    p.force = -1;
    p.w = w; p.y = y; p.stats = stats;
    p.B = B; p.H = H; p.W = W; p.Cin = 3; p.Cout = 64; p.KH = p.KW = 7; p.stride = 2; p.pad = 3;
    p.Ho = (H + 6 - 7) / 2 + 1; p.Wo = (W + 6 - 7) / 2 + 1;
    p.M = B * p.Ho * p.Wo;
    const int tx = (p.Wo + 15) / 16, ty = (p.Ho + 7) / 8;
    const dim3 grid((unsigned)(tx * ty * B));
    if (dtype == MHE_F32) hipLaunchKernelGGL(conv::stem_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, x_nchw, p, tx, ty);
    else hipLaunchKernelGGL(conv::stem_kernel<u16>, grid, dim3(256), 0, (hipStream_t)stream, x_nchw, p, tx, ty);
    return check_launch("stem_kernel");
}

extern "C" int mhe_linear_skinny_f32(const float *X, const float *W, const float *bias, float *Y, int M, int N, int K,
                                     int act, void *stream);

extern "C" int mhe_linear_f32(const float *X, const float *W, const float *bias, float *Y, int M, int N, int K, int act,
                              void *stream) {
    MHE_REQUIRE(X && W && Y && M > 0 && N > 0 && K > 0, "mhe_linear_f32: bad arguments");
    MHE_REQUIRE(K % 32 == 0, "mhe_linear_f32: K=%d must be a multiple of 32", K);
    MHE_REQUIRE(N % 4 == 0, "mhe_linear_f32: N=%d must be a multiple of 4", N);
    if (M <= 256 && K % 64 == 0) return mhe_linear_skinny_f32(X, W, bias, Y, M, N, K, act, stream);
    mhe_conv_desc d = {M, 1, 1, K, N, 1, 1, 1, 0, MHE_F32, 0, act == MHE_ACT_RELU};
    return mhe_conv2d_nhwc(&d, X, W, Y, nullptr, nullptr, nullptr, bias, nullptr, nullptr, stream);
}

extern "C" int mhe_bn_finalize(const mhe_stat_t *stats, const float *gamma, const float *beta, float *running_mean,
                               float *running_var, float *scale, float *shift, float *mean_invstd, int C, double count,
                               float momentum, float eps, void *stream) {
    MHE_REQUIRE(stats && gamma && beta && scale && shift && C > 0 && count > 1.f, "mhe_bn_finalize: bad arguments");
    hipLaunchKernelGGL(conv::bn_finalize_kernel, dim3((C + 3) / 4), dim3(256), 0, (hipStream_t)stream, const_cast<mhe_stat_t *>(stats), gamma,
                       beta, running_mean, running_var, scale, shift, mean_invstd, C, count, momentum, eps, 0, (long long *)nullptr);
    return check_launch("bn_finalize_kernel");
}

extern "C" int mhe_bn_finalize_step(mhe_stat_t *stats, const float *gamma, const float *beta, float *running_mean,
                                    float *running_var, float *scale, float *shift, float *mean_invstd, int C, double count,
                                    float momentum, float eps, int clear_stats, long long *num_batches_tracked, void *stream) {
    MHE_REQUIRE(stats && gamma && beta && scale && shift && C > 0 && count > 1.f, "mhe_bn_finalize_step: bad arguments");
    hipLaunchKernelGGL(conv::bn_finalize_kernel, dim3((C + 3) / 4), dim3(256), 0, (hipStream_t)stream, stats, gamma,
                       beta, running_mean, running_var, scale, shift, mean_invstd, C, count, momentum, eps, clear_stats,
                       num_batches_tracked);
    return check_launch("bn_finalize_kernel");
}

static inline int ew_blocks(size_t n) { size_t b = (n + 255) / 256; return (int)(b < 4096 ? (b ? b : 1) : 4096); }

extern "C" int mhe_bn_act_nhwc(const void *x, const float *scale, const float *shift, const void *res,
                               const float *res_scale, const float *res_shift, void *y, long P, int C, int relu,
                               int dtype, void *stream) {
    MHE_REQUIRE(x && scale && shift && y && P > 0 && C > 0 && C % 4 == 0, "mhe_bn_act_nhwc: bad arguments");
    const size_t n4 = (size_t)P * C / 4;
    // (16-byte-lane forms with the scale / shift rows read once per thread were built in round 4 and measured SLOWER: one piece per thread
    // 25.57 -> 25.77 ms in the train step over its 18 launches, twice; four pieces per thread in flight on the small tensors 27 us against
    // 15 us for 67 MB - while the same two changes took bn_bwd_apply_kernel from 39 to 21-26 us per 100 MB: profiles/EXPERIMENTS.md)
    if (dtype == MHE_F32)
        hipLaunchKernelGGL(conv::bn_act_kernel<float>, dim3(ew_blocks(n4)), dim3(256), 0, (hipStream_t)stream,
                           (const float *)x, scale, shift, (const float *)res, res_scale, res_shift, (float *)y, n4, C, relu);
    else
        hipLaunchKernelGGL(conv::bn_act_kernel<u16>, dim3(ew_blocks(n4)), dim3(256), 0, (hipStream_t)stream,
                           (const u16 *)x, scale, shift, (const u16 *)res, res_scale, res_shift, (u16 *)y, n4, C, relu);
    return check_launch("bn_act_kernel");
}

extern "C" int mhe_maxpool3x3s2_nhwc(const void *x, const float *scale, const float *shift, void *y, int B, int H, int W,
                                     int C, int dtype, void *stream) {
    MHE_REQUIRE(x && y && B > 0 && H > 0 && W > 0 && C % 4 == 0, "mhe_maxpool3x3s2_nhwc: bad arguments");
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    const size_t n4 = (size_t)B * Ho * Wo * C / 4;
    if (dtype == MHE_F32)
        hipLaunchKernelGGL(conv::maxpool_kernel<float>, dim3(ew_blocks(n4)), dim3(256), 0, (hipStream_t)stream,
                           (const float *)x, scale, shift, (float *)y, B, H, W, C, Ho, Wo);
    else if (C % 8 == 0)
        hipLaunchKernelGGL(conv::maxpool8_kernel, dim3(ew_blocks(n4 / 2)), dim3(256), 0, (hipStream_t)stream,
                           (const u16 *)x, scale, shift, (u16 *)y, B, H, W, C, Ho, Wo);
    else
        hipLaunchKernelGGL(conv::maxpool_kernel<u16>, dim3(ew_blocks(n4)), dim3(256), 0, (hipStream_t)stream,
                           (const u16 *)x, scale, shift, (u16 *)y, B, H, W, C, Ho, Wo);
    return check_launch("maxpool_kernel");
}

extern "C" int mhe_avgpool_nhwc(const void *x, float *y, int B, int HW, int C, int dtype, void *stream) {
    MHE_REQUIRE(x && y && B > 0 && HW > 0 && C > 0, "mhe_avgpool_nhwc: bad arguments");
    dim3 grid((C + 63) / 64, B);
    if (dtype == MHE_F32)
        hipLaunchKernelGGL(conv::avgpool_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float *)x, y, HW, C);
    else
        hipLaunchKernelGGL(conv::avgpool_kernel<u16>, grid, dim3(256), 0, (hipStream_t)stream, (const u16 *)x, y, HW, C);
    return check_launch("avgpool_kernel");
}

extern "C" int mhe_bn_act_avgpool_nhwc(const void *x, const float *scale, const float *shift, const void *res, const float *rscale,
                                       const float *rshift, float *y, int B, int HW, int C, int relu, int dtype, void *stream) {
    MHE_REQUIRE(x && scale && shift && y && B > 0 && HW > 0 && C > 0 && C % 4 == 0, "mhe_bn_act_avgpool_nhwc: bad arguments (C=%d must be a multiple of 4)", C);
    MHE_REQUIRE((rscale == nullptr) == (rshift == nullptr) && (res || !rscale), "mhe_bn_act_avgpool_nhwc: the residual's affine needs both tables and the residual");
    MHE_REQUIRE(dtype == MHE_F32 || dtype == MHE_BF16, "mhe_bn_act_avgpool_nhwc: dtype=%d", dtype);
    const dim3 grid((C + 255) / 256, B);
    if (dtype == MHE_F32)
        hipLaunchKernelGGL(conv::bn_act_avgpool_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float *)x, scale, shift, (const float *)res,
                           rscale, rshift, y, HW, C, relu);
    else
        hipLaunchKernelGGL(conv::bn_act_avgpool_kernel<u16>, grid, dim3(256), 0, (hipStream_t)stream, (const u16 *)x, scale, shift, (const u16 *)res,
                           rscale, rshift, y, HW, C, relu);
    return check_launch("bn_act_avgpool_kernel");
}

extern "C" int mhe_nchw_to_nhwc(const float *x, void *y, int B, int C, int H, int W, int dtype, void *stream) {
    MHE_REQUIRE(x && y && B > 0 && C > 0 && H > 0 && W > 0, "mhe_nchw_to_nhwc: bad arguments");
    const int Cp = (C + elem_chunk(dtype) - 1) / elem_chunk(dtype) * elem_chunk(dtype);
    const size_t n = (size_t)B * H * W;
    if (dtype == MHE_F32)
        hipLaunchKernelGGL(conv::nchw_to_nhwc_kernel<float>, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, x,
                           (float *)y, B, C, H * W, Cp);
    else
        hipLaunchKernelGGL(conv::nchw_to_nhwc_kernel<u16>, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, x,
                           (u16 *)y, B, C, H * W, Cp);
    return check_launch("nchw_to_nhwc_kernel");
}

extern "C" int mhe_nchw_to_nhwc_pad(const float *x, void *y, int B, int C, int Cp, int H, int W, int dtype, void *stream) {
    MHE_REQUIRE(x && y && B > 0 && C > 0 && Cp >= C && H > 0 && W > 0, "mhe_nchw_to_nhwc_pad: bad arguments");
    MHE_REQUIRE(dtype == MHE_F32 || dtype == MHE_BF16, "mhe_nchw_to_nhwc_pad: dtype=%d", dtype);
    const size_t n = (size_t)B * H * W;
    if (dtype == MHE_BF16 && C == 3 && Cp == 4)
        hipLaunchKernelGGL(conv::nchw3_to_nhwc4_bf16_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, x, (uint2 *)y, B, H * W);
    else if (dtype == MHE_F32)
        hipLaunchKernelGGL(conv::nchw_to_nhwc_kernel<float>, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, x, (float *)y, B, C, H * W, Cp);
    else
        hipLaunchKernelGGL(conv::nchw_to_nhwc_kernel<u16>, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, x, (u16 *)y, B, C, H * W, Cp);
    return check_launch("nchw_to_nhwc_kernel");
}

// ---------------------------------------------------------------------------
// Skinny dense layer for the per-image heads (M = batch <= a few hundred rows):
//   Y[M,N] = act(X[M,K] W[N,K]^T + bias),  f32, exact fmaf chains on v_mfma_f32_16x16x4_f32.
// The 128x128-tile kernel gives these shapes 8-16 workgroups walking K serially (165 us for
// l1: 256x512x2048); here a workgroup owns 16 output columns x all rows (grid = N/16), its 4 waves
// split K and reduce through LDS, operands go straight from L2 into fragments (W is streamed once,
// X - at most a few MB - is L2-resident).
namespace mhe { namespace conv {
template <int MT>      // row tiles of 16 held per wave (M <= 16*MT)
__global__ __launch_bounds__(256) void skinny_linear_kernel(const float *__restrict__ X, const float *__restrict__ W,
                                                            const float *__restrict__ bias, float *__restrict__ Y,
                                                            int M, int N, int K, int relu, u16 *__restrict__ Yb) {
    __shared__ v4f red[3][MT][64];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, q = lane >> 4, l15 = lane & 15;
    const int n0 = blockIdx.x * 16;
    // blockIdx.y: slice of 16*MT rows (few output columns, e.g. l1 with N = 512: 32 column groups alone would leave 7/8 of the CUs idle)
    X += (size_t)blockIdx.y * 16 * MT * K; Y += (size_t)blockIdx.y * 16 * MT * N; M -= blockIdx.y * 16 * MT;
    if (Yb) Yb += (size_t)blockIdx.y * 16 * MT * N;
    const int kq = K / 4;                    // each wave reduces a quarter of K (K % 64 == 0)
    const int kbeg = wave * kq, kend = kbeg + kq;
    v4f acc[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) acc[t] = v4f{0.f, 0.f, 0.f, 0.f};
    const int n = n0 + l15;
    const float *wrow = W + (size_t)(n < N ? n : N - 1) * K + 4 * q;
    for (int k = kbeg; k < kend; k += 16) {
        const float4 wv = *reinterpret_cast<const float4 *>(wrow + k);
        float4 xv[MT];
#pragma unroll
        for (int t = 0; t < MT; ++t) {
            const int m = 16 * t + l15;
            xv[t] = *reinterpret_cast<const float4 *>(X + (size_t)(m < M ? m : M - 1) * K + k + 4 * q);
        }
#pragma unroll
        for (int t = 0; t < MT; ++t) {
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv.x, xv[t].x, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv.y, xv[t].y, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv.z, xv[t].z, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv.w, xv[t].w, acc[t], 0, 0, 0);
        }
    }
    if (wave > 0) {
#pragma unroll
        for (int t = 0; t < MT; ++t) red[wave - 1][t][lane] = acc[t];
    }
    __syncthreads();
    if (wave == 0) {
        // D layout: row (channel) 4q + r, column (batch row) l15
        const int nb = n0 + 4 * q;
        float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (bias && nb < N) bv = *reinterpret_cast<const float4 *>(bias + nb);
#pragma unroll
        for (int t = 0; t < MT; ++t) {
            v4f v = acc[t];
#pragma unroll
            for (int w2 = 0; w2 < 3; ++w2) { const v4f o = red[w2][t][lane]; v[0] += o[0]; v[1] += o[1]; v[2] += o[2]; v[3] += o[3]; }
            v[0] += bv.x; v[1] += bv.y; v[2] += bv.z; v[3] += bv.w;
            if (relu) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
            const int m = 16 * t + l15;
            if (m < M && nb < N) {
                *reinterpret_cast<float4 *>(Y + (size_t)m * N + nb) = make_float4(v[0], v[1], v[2], v[3]);
                if (Yb) {                    // the same values as a bf16 operand for a following bf16 product (the flow's conditioning table)
                    float t4[4] = {v[0], v[1], v[2], v[3]};
                    store4<u16>(Yb + (size_t)m * N + nb, t4);
                }
            }
        }
    }
}
}}  // namespace mhe::conv

static int skinny_launch(const float *X, const float *W, const float *bias, float *Y, mhe::u16 *Yb, int M, int N, int K, int act, void *stream) {
    using namespace mhe;
    const dim3 grid((N + 15) / 16), block(256);
    hipStream_t s = (hipStream_t)stream;
    const int relu = act == MHE_ACT_RELU;
    if (M > 64 && (N + 15) / 16 < 512) {      // few column groups: also split the rows (64 per workgroup)
        hipLaunchKernelGGL(conv::skinny_linear_kernel<4>, dim3((N + 15) / 16, (M + 63) / 64), block, 0, s, X, W, bias, Y, M, N, K, relu, Yb);
        return check_launch("skinny_linear_kernel");
    }
    if (M <= 64) hipLaunchKernelGGL(conv::skinny_linear_kernel<4>, grid, block, 0, s, X, W, bias, Y, M, N, K, relu, Yb);
    else if (M <= 128) hipLaunchKernelGGL(conv::skinny_linear_kernel<8>, grid, block, 0, s, X, W, bias, Y, M, N, K, relu, Yb);
    else hipLaunchKernelGGL(conv::skinny_linear_kernel<16>, grid, block, 0, s, X, W, bias, Y, M, N, K, relu, Yb);
    return check_launch("skinny_linear_kernel");
}

extern "C" int mhe_linear_skinny_f32(const float *X, const float *W, const float *bias, float *Y, int M, int N, int K,
                                     int act, void *stream) {
    using namespace mhe;
    MHE_REQUIRE(X && W && Y && M > 0 && M <= 256 && N > 0 && N % 4 == 0 && K > 0 && K % 64 == 0,
                "mhe_linear_skinny_f32: need M <= 256, N %% 4 == 0, K %% 64 == 0 (M=%d N=%d K=%d)", M, N, K);
    return skinny_launch(X, W, bias, Y, nullptr, M, N, K, act, stream);
}

extern "C" int mhe_linear_f32_bf16copy(const float *X, const float *W, const float *bias, float *Y, void *Y_bf16, int M, int N, int K,
                                       int act, void *stream) {
    using namespace mhe;
    MHE_REQUIRE(X && W && Y && Y_bf16 && M > 0 && M <= 256 && N > 0 && N % 4 == 0 && K > 0 && K % 64 == 0,
                "mhe_linear_f32_bf16copy: need M <= 256, N %% 4 == 0, K %% 64 == 0 (M=%d N=%d K=%d)", M, N, K);
    return skinny_launch(X, W, bias, Y, (u16 *)Y_bf16, M, N, K, act, stream);
}

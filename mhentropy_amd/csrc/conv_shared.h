// Pieces shared by the implicit-GEMM convolution kernels (conv.hip, conv_p8.hip): launch parameters, the 16-byte
// chunk helpers, the XCD-aware tile order and the common epilogue (batch statistics, output affine / residual / ReLU,
// ReLU gate and BatchNorm-reverse sums of the data-gradient form, LDS-transposed coalesced stores).
#pragma once
#include "common.h"
#include <type_traits>

namespace mhe { namespace conv {

struct Params {
    const void *x; const void *w; void *y;
    const float *in_scale, *in_shift, *out_scale, *out_shift;
    const void *residual;
    const void *mask;      // optional, shaped like y: outputs are zeroed where mask <= 0 (ReLU gate of a data gradient)
    // optional BatchNorm-reverse statistics of the (gated) outputs g: for up to two BN units whose raw outputs bn_y[u] are shaped like y,
    // bn_stats[u][shard][0][c] += sum g, [1][c] += sum g * (bn_y - mean) * invstd   (bn_mi[u] = [mean | invstd])
    const void *bn_y[2]; const float *bn_mi[2]; fx::acc_t *bn_stats[2];
    fx::acc_t *stats;      // [2][NSH][2][Cout] sharded fixed-point accumulators (common.h, namespace fx)
    // dual-input prologue (1x1, stride 1): operand = relu(x*in_scale+in_shift + (x2*x2_scale+x2_shift | x2)),
    // i.e. the tail of the previous residual block evaluated on load; a_out (optional) receives it once
    const void *x2; const float *x2_scale, *x2_shift; void *a_out;
    int B, H, W, Cin, Cout, KH, KW, stride, pad, Ho, Wo, M, Kpad, relu_in, relu_out;
    float *y32;            // optional f32 result of a bf16 convolution (plain + out_shift only), instead of y
    int force;             // mhe_conv_desc.tile - 1: kernel variant forced by the caller (tests / tuning), -1 = launcher's choice
    // output scatter of the parity-split stride-2 data gradient: output pixel (b, i, j) of the [B, Ho, Wo] grid lands at
    // (b, 2i + os_py, 2j + os_px) of a [B, 2Ho, 2Wo] tensor (y, residual, mask and bn_y are all addressed there)
    int os2, os_py, os_px;
    int res_s2;            // residual at half resolution [B, ceil(Ho/2), ceil(Wo/2), Cout], added at even output positions only
    // bottleneck tail with conv3 re-evaluated (conv_fuse.hip): w3 [Cin][Cin / 4] packed, mid_scale / mid_shift = bn3's affine [Cin]
    const void *w3; const float *mid_scale, *mid_shift;
    int stats_only;        // statistics-only launch of the streaming 1x1 kernel (y = NULL): any pixel count
    // K-concatenated operand of a 1x1 data-gradient launch (MODE 3, conv.hip): channels Cin .. Cin + Cin2 - 1 of a pixel come from xcat
    // [M][Cin2]; the weight rows are [Cout][Cin + Cin2]
    const void *xcat; int Cin2;
    // the ReLU gate as bits: byte [pixel][channel / 8], bit i = (value of channel 8 j + i) > 0.  a_bits: written next to a_out by the fused
    // bottleneck tail (conv_fuse.hip); mask_bits: read INSTEAD of mask by the streaming 1x1 data-gradient kernel (conv_stream.hip) - a
    // sixteenth of the block-wide tensor's bytes; every other kernel reads mask
    unsigned char *a_bits; const unsigned char *mask_bits;
    // conv_halo.hip, data-gradient form with the BatchNorm reverse of the convolution's own output gradient on the operand load:
    // operand = k2 x + k1 x2 + k0 per channel (rev_coef = k2 | k1 | k0, [3][Cin]; x2 = the raw output that BatchNorm normalised), written to a_out
    const float *rev_coef;
};

// tile row -> global output pixel index the epilogue addresses (or -1 outside the problem)
__device__ __forceinline__ long out_pixel(const Params &p, int m) {
    if (m >= p.M) return -1l;
    if (!p.os2) return (long)m;
    const int j = m % p.Wo, t = m / p.Wo, i = t % p.Ho, b = t / p.Ho;
    return ((long)(b * 2 * p.Ho + 2 * i + p.os_py)) * (2 * p.Wo) + 2 * j + p.os_px;
}

constexpr int NSH = fx::NSH;     // statistic shards: block b adds into shard b % NSH
constexpr int MAXC = 2048;       // largest Cin whose BatchNorm affine is staged in LDS

template <typename T> struct El;
template <> struct El<float> { static constexpr int CE = 4; };
template <> struct El<u16>   { static constexpr int CE = 8; };

__device__ __forceinline__ int swz(int row, int slot) { return row * 8 + (slot ^ ((row >> 1) & 7)); }

// one 16-byte chunk of activations -> floats and back
template <typename T> struct Chunk;
template <> struct Chunk<float> {
    static constexpr int N = 4;
    static __device__ __forceinline__ void unpack(uint4 r, float *v) {
        v[0] = __uint_as_float(r.x); v[1] = __uint_as_float(r.y); v[2] = __uint_as_float(r.z); v[3] = __uint_as_float(r.w);
    }
    static __device__ __forceinline__ uint4 pack(const float *v) {
        return make_uint4(__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3]));
    }
};
template <> struct Chunk<u16> {
    static constexpr int N = 8;
    static __device__ __forceinline__ void unpack(uint4 r, float *v) {
        const unsigned in[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) { v[2 * i] = __uint_as_float(in[i] << 16); v[2 * i + 1] = __uint_as_float(in[i] & 0xffff0000u); }
    }
    static __device__ __forceinline__ uint4 pack(const float *v) {
        unsigned o[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = (unsigned)f32_to_bf16(v[2 * i]) | ((unsigned)f32_to_bf16(v[2 * i + 1]) << 16);
        return make_uint4(o[0], o[1], o[2], o[3]);
    }
};

// relu?(x*scale+shift [+ x2*scale2+shift2 | + x2]); sc/sh point into LDS, sc2/sh2 (optional) into global memory
template <typename T>
__device__ __forceinline__ uint4 in_transform(uint4 raw, const float *sc, const float *sh, int c, int relu,
                                              bool dual, uint4 raw2, const float *sc2, const float *sh2) {
    constexpr int N = Chunk<T>::N;
    float v[N], ss[N], tt[N];
    Chunk<T>::unpack(raw, v);
#pragma unroll
    for (int i = 0; i < N; i += 4) {
        const float4 a = *reinterpret_cast<const float4 *>(sc + c + i), b = *reinterpret_cast<const float4 *>(sh + c + i);
        ss[i] = a.x; ss[i + 1] = a.y; ss[i + 2] = a.z; ss[i + 3] = a.w;
        tt[i] = b.x; tt[i + 1] = b.y; tt[i + 2] = b.z; tt[i + 3] = b.w;
    }
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = fmaf(v[i], ss[i], tt[i]);
    if (dual) {
        float w[N];
        Chunk<T>::unpack(raw2, w);
        if (sc2) {
#pragma unroll
            for (int i = 0; i < N; i += 4) {
                const float4 a = *reinterpret_cast<const float4 *>(sc2 + c + i), b = *reinterpret_cast<const float4 *>(sh2 + c + i);
                w[i] = fmaf(w[i], a.x, b.x); w[i + 1] = fmaf(w[i + 1], a.y, b.y);
                w[i + 2] = fmaf(w[i + 2], a.z, b.z); w[i + 3] = fmaf(w[i + 3], a.w, b.w);
            }
        }
#pragma unroll
        for (int i = 0; i < N; ++i) v[i] += w[i];
    }
    if (relu) {
#pragma unroll
        for (int i = 0; i < N; ++i) v[i] = fmaxf(v[i], 0.f);
    }
    return Chunk<T>::pack(v);
}

// XCD-aware tile order.  Workgroups are dealt round-robin to the 8 XCDs (linear id % 8), each with its own L2.  With the
// plain (x = M tile, y = N tile) order the N tiles that re-read one activation tile run gridDim.x launches apart on
// arbitrary XCDs; here XCD x walks M tiles x, x+8, ... and takes all N tiles of an M tile back to back, so the
// activation tile is fetched from HBM once and re-read from that XCD's L2 (the weights are small enough to sit in every L2).
#ifndef MHE_CONV_XCD_ORDER
#define MHE_CONV_XCD_ORDER 1
#endif
constexpr bool XCD_ORDER = MHE_CONV_XCD_ORDER;
#ifndef MHE_CONV_BN_EPILOGUE
#define MHE_CONV_BN_EPILOGUE 1
#endif
constexpr bool BN_EPILOGUE = MHE_CONV_BN_EPILOGUE;      // measurement switch for the BatchNorm-reverse sums in the epilogue
__device__ __forceinline__ void tile_of_block(int &mt, int &nt) {
    const int gm = gridDim.x, gn = gridDim.y;
    mt = blockIdx.x; nt = blockIdx.y;
    if (XCD_ORDER && gn > 1 && (gm & 7) == 0) {
        const int L = blockIdx.x + gm * blockIdx.y, slot = L >> 3;
        nt = slot % gn;
        mt = (slot / gn) * 8 + (L & 7);
    }
}

// element offset (without the channel) of the half-resolution residual row of output pixel m, or -1 at odd positions
__device__ __forceinline__ long res_half_row(const Params &p, long m) {
    const int j = (int)(m % p.Wo), t = (int)(m / p.Wo), i = t % p.Ho, b = t / p.Ho;
    if ((i | j) & 1) return -1l;
    const int Hh = (p.Ho + 1) >> 1, Wh = (p.Wo + 1) >> 1;
    return ((long)(b * Hh + (i >> 1))) * Wh + (j >> 1);
}

// ---- epilogue shared by the conv kernels.  The accumulator holds y^T: lane (l15, q) owns channels
// 4q..4q+3 of tile nt for tile row 16mt + l15; pix(row) maps a tile row to the global output pixel
// index (or -1 when the row is outside the image / batch).
// DG: data-gradient form - ReLU gate by p.mask and (optionally) BatchNorm-reverse sums; compiled out of the forward kernels
// (with the code present behind run-time flags the forward step was 1.5 % slower: register pressure in the store loop).
// the store half of the epilogue: the BM x BN tile lies staged in `ot` (row = pixel, 16-byte chunks XOR-swizzled by row); NTH threads
// (tid 0 .. NTH - 1, all of them and only them) walk it.  Takes 2 workgroup barriers when p.stats is set, 2 more per BatchNorm unit of the
// data-gradient form: waves of the workgroup that do not take part in the walk must execute as many (conv_halo.hip).
template <typename T, int BM, int BN, int NTH, bool DG, int GROWS = (NTH >= 512 ? 4 : 2), int MAXU = 2, typename PixF>
__device__ __forceinline__ void epilogue_store(const Params &p, unsigned char *ot, int shard, int n0, PixF pix) {
    const int tid = threadIdx.x;
    constexpr int CPR = BN * (int)sizeof(T) / 16;          // 16-byte chunks per output row
    constexpr int CMASK = (CPR - 1) & 15;
    constexpr int EPC = 16 / (int)sizeof(T);               // elements per chunk
    float *const lds = reinterpret_cast<float *>(ot);
    {
        T *yg = reinterpret_cast<T *>(p.y);
        const T *rg = reinterpret_cast<const T *>(p.residual);
        const T *mk = DG ? reinterpret_cast<const T *>(p.mask) : nullptr;
        const bool bnr = DG && BN_EPILOGUE && p.bn_y[0] != nullptr;
        const bool plain = !p.out_scale && !p.out_shift && !rg && !p.relu_out && !mk && !bnr;
        static_assert(NTH % CPR == 0, "a thread keeps one 16-byte column chunk across its rows");
        float bs1[DG ? MAXU : 1][EPC], bs2[DG ? MAXU : 1][EPC], bmu[DG ? MAXU : 1][EPC], biv[DG ? MAXU : 1][EPC];      // (MAXU: BatchNorm units the caller can pass)
        const int cfix = tid % CPR;
        // per-channel constant of the data-gradient form (out_shift without out_scale: mhe_conv2d_masked_bias_nhwc), added before the gate
        float bsh[DG ? EPC : 1];
        if constexpr (DG) {
#pragma unroll
            for (int i = 0; i < EPC; ++i) bsh[i] = p.out_shift && n0 + cfix * EPC + i < p.Cout ? p.out_shift[n0 + cfix * EPC + i] : 0.f;
        }
        if constexpr (DG) if (bnr) {
#pragma unroll
            for (int u = 0; u < MAXU; ++u)
#pragma unroll
                for (int i = 0; i < EPC; ++i) {
                    const int n = n0 + cfix * EPC + i;
                    const bool ok = p.bn_y[u] && n < p.Cout;
                    bs1[u][i] = bs2[u][i] = 0.f;
                    bmu[u][i] = ok ? p.bn_mi[u][n] : 0.f;
                    biv[u][i] = ok ? p.bn_mi[u][p.Cout + n] : 0.f;
                }
        }
        // Data-gradient fast path (whole tile inside the problem - every launch of the trunk at the bench batch): the generic loop
        // below guards each row with a branch, which keeps hipcc from issuing one row's loads before the previous row's stores
        // (measured 2.7 TB/s on the layers whose whole cost is this epilogue: up to four tensors read per output chunk).  Here
        // the loads of G rows x (gate, residual, BatchNorm operands) are issued back to back, then consumed.
        // batch statistics (forward form): per-channel sum and sum of squares of the tile AS STORED (bf16 storage: of the rounded values,
        // the ones the consumer will normalise), accumulated per thread while it walks its rows in the store loop - the memory
        // operations of that loop hide the arithmetic.  (The first version summed the f32 accumulators in a phase of its own before the
        // transpose: two more barriers and an LDS round trip, 23 % of a write-bound 1x1 launch.)
        const bool st_on = !DG && p.stats != nullptr;
        float ss1[EPC], ss2[EPC];
#pragma unroll
        for (int i = 0; i < EPC; ++i) ss1[i] = ss2[i] = 0.f;
        bool done = false;
        if constexpr (DG) {
            if (mk && pix(0) >= 0 && pix(BM - 1) >= 0 && n0 + BN <= p.Cout && !p.out_scale && !p.relu_out) {
                done = true;
                const int nbn = bnr ? (MAXU == 2 && p.bn_y[1] ? 2 : 1) : 0;
                const T *y0g = reinterpret_cast<const T *>(p.bn_y[0]), *y1g = reinterpret_cast<const T *>(MAXU == 2 ? p.bn_y[1] : nullptr);
                auto run = [&](auto RES, auto NB) {
                    constexpr bool HAS_RES = decltype(RES)::value;
                    constexpr int NBN = decltype(NB)::value, NJ = BM * CPR / NTH;
                    // rows in flight per thread: 4 on the one-workgroup-per-CU 256x256 tiles; 2 on the 256-thread tiles, which must
                    // stay within 256 registers to keep two workgroups per CU (4 took 128x128 to one wave per SIMD: 166 -> 210 us)
                    constexpr int G = GROWS;
                    static_assert(NJ % G == 0, "rows per thread");
#pragma unroll 1                  // one group of G rows at a time: fully unrolled, hipcc hoists the next groups' loads and spills (179 VGPRs on the 256x256 tile)
                    for (int j0 = 0; j0 < NJ; j0 += G) {
                        uint4 raw[G], gm[G], rr[HAS_RES ? G : 1], ya[NBN >= 1 ? G : 1], yb[NBN == 2 ? G : 1];
                        size_t off[G];
#pragma unroll
                        for (int g = 0; g < G; ++g) {
                            const int id = tid + NTH * (j0 + g);
                            const int row = id / CPR, c = id % CPR;
                            off[g] = (size_t)pix(row) * p.Cout + n0 + c * EPC;
                            gm[g] = *reinterpret_cast<const uint4 *>(mk + off[g]);
                            if constexpr (HAS_RES) {
                                if (p.res_s2) {
                                    const long hr = res_half_row(p, pix(row));
                                    rr[g] = hr >= 0 ? *reinterpret_cast<const uint4 *>(rg + (size_t)hr * p.Cout + n0 + c * EPC) : make_uint4(0u, 0u, 0u, 0u);
                                } else rr[g] = *reinterpret_cast<const uint4 *>(rg + off[g]);
                            }
                            if constexpr (NBN >= 1) ya[g] = *reinterpret_cast<const uint4 *>(y0g + off[g]);
                            if constexpr (NBN == 2) yb[g] = *reinterpret_cast<const uint4 *>(y1g + off[g]);
                            raw[g] = *reinterpret_cast<const uint4 *>(ot + ((size_t)row * CPR + (c ^ (row & CMASK))) * 16);
                        }
#pragma unroll
                        for (int g = 0; g < G; ++g) {
                            float v[EPC], t[EPC];
                            Chunk<T>::unpack(raw[g], v);
#pragma unroll
                            for (int i = 0; i < EPC; ++i) v[i] += bsh[i];
                            if constexpr (HAS_RES) {
                                Chunk<T>::unpack(rr[g], t);
#pragma unroll
                                for (int i = 0; i < EPC; ++i) v[i] += t[i];
                            }
                            Chunk<T>::unpack(gm[g], t);
#pragma unroll
                            for (int i = 0; i < EPC; ++i) v[i] = t[i] > 0.f ? v[i] : 0.f;
                            if constexpr (NBN >= 1) {
                                Chunk<T>::unpack(ya[g], t);
#pragma unroll
                                for (int i = 0; i < EPC; ++i) { bs1[0][i] += v[i]; bs2[0][i] = fmaf(v[i], (t[i] - bmu[0][i]) * biv[0][i], bs2[0][i]); }
                            }
                            if constexpr (NBN == 2) {
                                Chunk<T>::unpack(yb[g], t);
#pragma unroll
                                for (int i = 0; i < EPC; ++i) { bs1[1][i] += v[i]; bs2[1][i] = fmaf(v[i], (t[i] - bmu[1][i]) * biv[1][i], bs2[1][i]); }
                            }
                            *reinterpret_cast<uint4 *>(yg + off[g]) = Chunk<T>::pack(v);
                        }
                    }
                };
                using std::integral_constant;
                if (rg) {
                    if (nbn == 2) { if constexpr (MAXU == 2) run(integral_constant<bool, true>{}, integral_constant<int, 2>{}); }
                    else if (nbn == 1) run(integral_constant<bool, true>{}, integral_constant<int, 1>{});
                    else run(integral_constant<bool, true>{}, integral_constant<int, 0>{});
                } else {
                    if (nbn == 2) { if constexpr (MAXU == 2) run(integral_constant<bool, false>{}, integral_constant<int, 2>{}); }
                    else if (nbn == 1) run(integral_constant<bool, false>{}, integral_constant<int, 1>{});
                    else run(integral_constant<bool, false>{}, integral_constant<int, 0>{});
                }
            }
        }
#pragma unroll
        for (int j = 0; j < BM * CPR / NTH; ++j) {
            if (done) break;
            const int id = tid + NTH * j;
            const int row = id / CPR, c = id % CPR;
            const long m = pix(row);
            const int n = n0 + c * EPC;
            if (m < 0 || n >= p.Cout) continue;
            uint4 raw = *reinterpret_cast<const uint4 *>(ot + ((size_t)row * CPR + (c ^ (row & CMASK))) * 16);
            const size_t off = (size_t)m * p.Cout + n;
            if (st_on) {
                float sv[EPC];
                Chunk<T>::unpack(raw, sv);
#pragma unroll
                for (int i = 0; i < EPC; ++i) { ss1[i] += sv[i]; ss2[i] = fmaf(sv[i], sv[i], ss2[i]); }
            }
            if (!plain) {
                float v[EPC];
                Chunk<T>::unpack(raw, v);
#pragma unroll
                for (int i = 0; i < EPC; ++i) {
                    const float sc = p.out_scale ? p.out_scale[n + i] : 1.f, sh = p.out_shift ? p.out_shift[n + i] : 0.f;
                    v[i] = fmaf(v[i], sc, sh);
                }
                if (rg) {
                    const long hr = p.res_s2 ? res_half_row(p, m) : 0l;
                    if (hr >= 0) {
                        float r2[EPC];
                        Chunk<T>::unpack(*reinterpret_cast<const uint4 *>(rg + (p.res_s2 ? (size_t)hr * p.Cout + n : off)), r2);
#pragma unroll
                        for (int i = 0; i < EPC; ++i) v[i] += r2[i];
                    }
                }
                if (p.relu_out) {
#pragma unroll
                    for (int i = 0; i < EPC; ++i) v[i] = fmaxf(v[i], 0.f);
                }
                if constexpr (DG) if (mk) {
                    float g2[EPC];
                    Chunk<T>::unpack(*reinterpret_cast<const uint4 *>(mk + off), g2);
#pragma unroll
                    for (int i = 0; i < EPC; ++i) v[i] = g2[i] > 0.f ? v[i] : 0.f;
                }
                if constexpr (DG) if (bnr) {
#pragma unroll
                    for (int u = 0; u < MAXU; ++u) {
                        if (!p.bn_y[u]) continue;
                        float yv[EPC];
                        Chunk<T>::unpack(*reinterpret_cast<const uint4 *>(reinterpret_cast<const T *>(p.bn_y[u]) + off), yv);
#pragma unroll
                        for (int i = 0; i < EPC; ++i) {
                            bs1[u][i] += v[i];
                            bs2[u][i] = fmaf(v[i], (yv[i] - bmu[u][i]) * biv[u][i], bs2[u][i]);
                        }
                    }
                }
                raw = Chunk<T>::pack(v);
            }
            *reinterpret_cast<uint4 *>(yg + off) = raw;
        }
        if (st_on) {
            // threads sharing a column chunk (NTH / CPR of them) fold their partial sums through LDS; one thread per channel adds the
            // tile's two sums to this workgroup's statistic shard
            float *red = reinterpret_cast<float *>(lds);
            __syncthreads();
#pragma unroll
            for (int i = 0; i < EPC; ++i) { red[tid * (2 * EPC) + i] = ss1[i]; red[tid * (2 * EPC) + EPC + i] = ss2[i]; }
            __syncthreads();
            if (tid < CPR * EPC) {
                const int c = tid / EPC, e = tid % EPC, n = n0 + tid;
                float a = 0.f, b = 0.f;
                for (int k = 0; k < NTH / CPR; ++k) { a += red[(c + CPR * k) * (2 * EPC) + e]; b += red[(c + CPR * k) * (2 * EPC) + EPC + e]; }
                if (n < p.Cout) {
                    fx::add(p.stats, shard, 0, p.Cout, n, a);
                    fx::add(p.stats, shard, 1, p.Cout, n, b);
                }
            }
        }
        if constexpr (DG) if (bnr) {
            // threads sharing a column chunk (NTH / CPR of them) fold their partial sums through LDS; one thread per channel adds
            // the tile's two sums to this block's statistic shard
            float *red = reinterpret_cast<float *>(lds);
#pragma unroll
            for (int u = 0; u < MAXU; ++u) {
                if (!p.bn_y[u]) continue;
                __syncthreads();
#pragma unroll
                for (int i = 0; i < EPC; ++i) { red[tid * (2 * EPC) + i] = bs1[u][i]; red[tid * (2 * EPC) + EPC + i] = bs2[u][i]; }
                __syncthreads();
                if (tid < CPR * EPC) {
                    const int c = tid / EPC, e = tid % EPC, n = n0 + tid;
                    float a = 0.f, b = 0.f;
                    for (int k = 0; k < NTH / CPR; ++k) { a += red[(c + CPR * k) * (2 * EPC) + e]; b += red[(c + CPR * k) * (2 * EPC) + EPC + e]; }
                    if (n < p.Cout) {
                        fx::add(p.bn_stats[u], shard, 0, p.Cout, n, a);
                        fx::add(p.bn_stats[u], shard, 1, p.Cout, n, b);
                    }
                }
            }
        }
    }
}

template <typename T, int BM, int BN, int WM, int WN, bool DG, typename LdsT, typename AccT, typename PixF>
__device__ __forceinline__ void epilogue(const Params &p, AccT &acc, LdsT &lds, int shard, int n0, PixF pix) {
    constexpr int NTH = 64 * WM * WN;
    constexpr int MTW = BM / WM / 16, NTW = BN / WN / 16;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int q = lane >> 4, l15 = lane & 15;
    const int wr = wave / WN, wc = wave % WN;
    // ---- epilogue.  The accumulator holds y^T: lane (l15, q) owns channels 4q..4q+3 of tile nt
    // for pixel 16mt + l15.  (1) batch statistics: per-thread partials -> LDS -> one thread per
    // (statistic, channel) -> one order-independent fixed-point add (fx::add: two 64-bit integer atomics) per thread into this block's shard.
    // (2) the tile is transposed through LDS (16-byte chunks XOR-swizzled by row) so that global
    // stores are whole 16-byte pieces of contiguous channel rows instead of 8-byte row-strided ones.
    constexpr int CPR = BN * (int)sizeof(T) / 16;          // 16-byte chunks per output row
    constexpr int CMASK = (CPR - 1) & 15;
    static_assert((size_t)BM * BN * sizeof(T) <= sizeof(lds), "output tile must fit the staging buffers");
    if constexpr (sizeof(T) == 2 && !DG) {
        // f32 result from bf16 operands (mhe_conv2d_f32out_nhwc): the accumulator's own layout gives every lane 4 consecutive
        // channels of one pixel = one 16-byte store; used by the narrow (<= 64 output channels) products of the flow's reverse
        // pass whose results feed exp / tanh or the flow variable's gradient chain
        if (p.y32) {
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
                for (int mt = 0; mt < MTW; ++mt) {
                    const long m = pix(wr * (BM / WM) + mt * 16 + l15);
                    const int n = n0 + wc * (BN / WN) + nt * 16 + 4 * q;
                    if (m < 0 || n >= p.Cout) continue;
                    v4f v = acc[nt][mt];
                    if (p.out_shift) { const float4 b = *reinterpret_cast<const float4 *>(p.out_shift + n); v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w; }
                    *reinterpret_cast<float4 *>(p.y32 + (size_t)m * p.Cout + n) = make_float4(v[0], v[1], v[2], v[3]);
                }
            return;
        }
    }
    unsigned char *ot = reinterpret_cast<unsigned char *>(lds);
    {
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
            for (int mt = 0; mt < MTW; ++mt) {
                const int row = wr * (BM / WM) + mt * 16 + l15;
                const int e0 = wc * (BN / WN) + nt * 16 + 4 * q;             // first of 4 channels within the tile
                const int boff = e0 * (int)sizeof(T);
                const int chunk = (boff >> 4) ^ (row & CMASK);
                unsigned char *dst = ot + ((size_t)row * CPR + chunk) * 16 + (boff & 15);
                const v4f v = acc[nt][mt];
                if constexpr (sizeof(T) == 4) {
                    *reinterpret_cast<float4 *>(dst) = make_float4(v[0], v[1], v[2], v[3]);
                } else {
                    uint2 o;
                    o.x = (unsigned)f32_to_bf16(v[0]) | ((unsigned)f32_to_bf16(v[1]) << 16);
                    o.y = (unsigned)f32_to_bf16(v[2]) | ((unsigned)f32_to_bf16(v[3]) << 16);
                    *reinterpret_cast<uint2 *>(dst) = o;
                }
            }
    }
    __syncthreads();
    epilogue_store<T, BM, BN, NTH, DG>(p, ot, shard, n0, pix);
}

}}  // namespace mhe::conv

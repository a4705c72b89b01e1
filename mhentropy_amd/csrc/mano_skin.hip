// Full-mesh linear-blend skinning of the MANO hand on the matrix cores (round 5) - the largest line of the reference iteration's metrics pass
// (sample(N = [200, 200], mods = {uv, xyz, verts}), hand/CrossModalHand.py:357-361: 51,200 hypotheses x 778 vertices per iteration at B = 256).
//
// Reference arithmetic being replaced (all fp32): hand/manopth/manolayer.py:181-188 (shape + pose-corrective blend shapes), :236-246
// (per-vertex transform T = sum_j w_j G_j, v' = T [v; 1]), :262-273 and hand/network.py:480 (centre, millimetres, root / bone normalisation).
//
// First version (mano_skin_kernel<16>, mano.hip): one thread per vertex, every per-hypothesis value a scalar operand of an FMA - 640 FMAs per
// (hypothesis, vertex), 39 TFLOP/s = a quarter of the f32 vector peak, 1.40 ms at 51,200 hypotheses for 478 MB of output (0.34 TB/s).
//
// This version: both products are GEMMs with the HYPOTHESIS on the column index of v_mfma_f32_32x32x16_bf16 and the VERTEX on the row index:
//     X_c[v][h] = sum_k PD_c[v][k] PM[k][h]      c = 0..2, k = 135 pose-map entries + 10 betas (K = 160 with zero padding), + template (f32)
//     T_e[v][h] = sum_j  W[v][j]  G_e[j][h]      e = 0..11 (3 x 4 transform entries), K = 16 joints
// so that a lane ends up with X_0..2 and T_0..11 of ITS hypothesis for 16 vertices in registers, and v' = T [X; 1] needs no lane movement.
// f32 ACCURACY ON bf16 MATRIX CORES: every f32 operand is split into bf16 pieces (x = h + m + l, 8 mantissa bits each: h + m + l IS x) and
// the product is the sum of the piece products, accumulated in f32 -
//     T : 3 x 3 pieces, the 6 products hh, hm, mh, hl, lh, mm: terms below 2^-24 of the product dropped = an f32 FMA chain's accuracy
//     X : 2 x 2 pieces, 3 products (hh, hm, mh): 2^-16 of the blend OFFSET (<= a few mm against a template of ~100 mm: < 1e-6 of the vertex)
// 162 MFMAs of 32 cycles per (32 vertices x 32 hypotheses) = 5.2 k cycles against ~20 k for the f32 MFMA form and ~80 k VALU cycles before.
// A workgroup = 32 hypotheses x all vertices: the hypotheses' pieces (B operands) live in LDS, a wave walks every fourth vertex tile with the
// table pieces (A operands, fragment-major, 1 KiB per fragment, made from the f32 table by split_tables_kernel on every call: 3 us) streamed
// L2 -> registers; four tiles' results are transposed through LDS so that a hypothesis' 128 vertices leave as one 1.5 KB run.
#include "common.h"
#include "mano_layout.h"

namespace mhe { namespace mano {

typedef __attribute__((ext_vector_type(8))) __bf16 bf8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

constexpr int VT = VP / 32;                  // 26 vertex tiles
constexpr int KS = 10;                       // k-steps of 16: 135 pose-map entries, 10 betas, 15 zeros
constexpr size_t SPLIT_PD = 0;               // u16 [VT][KS][3 coordinates][2 pieces][64 lanes][8]
constexpr size_t SPLIT_W = (size_t)VT * KS * 3 * 2 * 512;       // u16 [VT][3 pieces][64 lanes][8]
constexpr size_t SPLIT_ELEMS = SPLIT_W + (size_t)VT * 3 * 512;
constexpr int HT = 32;                       // hypotheses per workgroup
constexpr int TP = 386;                      // pitch (floats) of a hypothesis' row in the transposition buffer: 8-byte rows, two-way conflicts on the epilogue's writes
constexpr int LDS_BYTES = KS * 2 * 1024 + 12 * 3 * 1024 + HT * 8 * 4 + 3 * VP * 4 + HT * TP * 4;

__device__ __forceinline__ void split2(float x, u16 &h, u16 &m) {
    h = f32_to_bf16(x);
    m = f32_to_bf16(x - bf16_to_f32(h));
}
__device__ __forceinline__ void split3(float x, u16 &h, u16 &m, u16 &l) {
    h = f32_to_bf16(x);
    const float r = x - bf16_to_f32(h);
    m = f32_to_bf16(r);
    l = f32_to_bf16(r - bf16_to_f32(m));
}

// the f32 vertex tables -> bf16 pieces in MFMA A-operand order (lane l: vertex l & 31, k = 8 (l >> 5) + 0..7)
__global__ __launch_bounds__(256) void split_tables_kernel(const float *__restrict__ tables, u16 *__restrict__ out) {
    const int t = blockIdx.x * 256 + threadIdx.x, lane = t & 63, f = t >> 6;
    const int v = lane & 31, kh = lane >> 5;
    if (f < VT * KS * 3) {
        const int c = f % 3, ks = (f / 3) % KS, vt = f / (3 * KS);
        u16 h[8], m[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = ks * 16 + kh * 8 + j;
            const float x = k < 135 ? tables[V_PD + (k * 3 + c) * VP + vt * 32 + v]
                          : k < 145 ? tables[V_SD + ((k - 135) * 3 + c) * VP + vt * 32 + v] : 0.f;
            split2(x, h[j], m[j]);
        }
        u16 *o = out + SPLIT_PD + ((size_t)f * 2 * 64 + lane) * 8;
#pragma unroll
        for (int j = 0; j < 8; ++j) { o[j] = h[j]; o[512 + j] = m[j]; }
    } else if (f < VT * KS * 3 + VT) {
        const int vt = f - VT * KS * 3;
        u16 *o = out + SPLIT_W + ((size_t)vt * 3 * 64 + lane) * 8;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            u16 h, m, l;
            split3(tables[V_W + (kh * 8 + j) * VP + vt * 32 + v], h, m, l);
            o[j] = h; o[512 + j] = m; o[1024 + j] = l;
        }
    }
}

#define MFMA(a, b, c) c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8, (a)), __builtin_bit_cast(bf8, (b)), (c), 0, 0, 0)

__global__ __launch_bounds__(256) void mano_skin_mfma_kernel(const float *__restrict__ ws, const float *__restrict__ tables,
                                                             const u16 *__restrict__ split, float *__restrict__ verts_o, int R, int mm_mode) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    u16 *PMb = reinterpret_cast<u16 *>(smem);                    // [KS][2 pieces][2 k halves][32 hypotheses][8]: B-operand order (lane = half * 32 + hypothesis)
    u16 *Gb = PMb + KS * 2 * 512;                                // [12 e][3 pieces][2 joint halves][32][8]
    float *nrm = reinterpret_cast<float *>(Gb + 12 * 3 * 512);   // [32][8]: centre (3), root (3), bone
    float *tmpl = nrm + HT * 8;                                  // [3][VP]: the template
    float *trans = tmpl + 3 * VP;                                // [32][TP]; first the staging area of the workspace rows
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int r0 = blockIdx.x * HT;
    // ---- the workgroup's hypotheses: workspace rows (mano_pose_kernel) -> LDS by coalesced 16-byte loads -> bf16 pieces in B-operand order
    {
        float4 st[11];
#pragma unroll
        for (int q = 0; q < 11; ++q) {
            const int i = tid + q * 256, h = i / 88, c4 = i - h * 88;             // 88 float4 per row of 352 floats
            st[q] = reinterpret_cast<const float4 *>(ws + (size_t)(r0 + h < R ? r0 + h : R - 1) * WS_STRIDE)[c4];
        }
        float tm[10];
#pragma unroll
        for (int q = 0; q < 10; ++q) tm[q] = tid + q * 256 < 3 * VP ? tables[V_T + tid + q * 256] : 0.f;
#pragma unroll
        for (int q = 0; q < 11; ++q) reinterpret_cast<float4 *>(trans)[tid + q * 256] = st[q];
#pragma unroll
        for (int q = 0; q < 10; ++q) if (tid + q * 256 < 3 * VP) tmpl[tid + q * 256] = tm[q];
    }
    __syncthreads();
    for (int i = tid; i < HT * 160; i += 256) {
        const int h = i & 31;                                  // consecutive lanes: consecutive hypotheses (row pitch 352 floats: 32-way on the
        int kk = (i >> 5) + h;                                 // read - spread by rotating k per hypothesis)
        kk = kk >= 160 ? kk - 160 : kk;
        const float *w = trans + h * WS_STRIDE;
        const float x = kk < 135 ? w[WS_PM + kk] : kk < 145 ? w[WS_BT + kk - 135] : 0.f;
        u16 hi, mi;
        split2(x, hi, mi);
        u16 *o = PMb + ((kk >> 4) * 4 + ((kk >> 3) & 1)) * 256 + h * 8 + (kk & 7);
        o[0] = hi; o[512] = mi;
    }
    for (int i = tid; i < HT * 192; i += 256) {
        const int h = i & 31;
        int q = (i >> 5) + h;
        q = q >= 192 ? q - 192 : q;
        const int j = (q * 683) >> 13, e = q - j * 12;        // q / 12 for q < 192
        u16 hi, mi, lo;
        split3(trans[h * WS_STRIDE + WS_GR + q], hi, mi, lo);
        u16 *o = Gb + (e * 6 + (j >> 3)) * 256 + h * 8 + (j & 7);
        o[0] = hi; o[512] = mi; o[1024] = lo;
    }
    if (tid < HT * 8) {
        const int h = tid >> 3, i = tid & 7;
        nrm[tid] = i < 7 ? trans[h * WS_STRIDE + WS_NRM + i] : 0.f;
    }
    __syncthreads();
    const int hyp = lane & 31, half = lane >> 5;
    // output = (vp - centre) * 1000 [millimetre mode], then (. - root) / bone: one FMA per coordinate
    const float osc = mm_mode ? 1000.f : 1000.f / nrm[hyp * 8 + 6];
    float ooff[3];
#pragma unroll
    for (int c = 0; c < 3; ++c)
        ooff[c] = mm_mode ? -1000.f * nrm[hyp * 8 + c] : -(1000.f * nrm[hyp * 8 + c] + nrm[hyp * 8 + 3 + c]) / nrm[hyp * 8 + 6];
    const uint4 *Bpm = reinterpret_cast<const uint4 *>(PMb) + lane;          // + (ks * 2 + p) * 64
    const uint4 *Bg = reinterpret_cast<const uint4 *>(Gb) + lane;            // + (e * 3 + p) * 64
    const uint4 *Apd = reinterpret_cast<const uint4 *>(split + SPLIT_PD) + lane;
    const uint4 *Aw = reinterpret_cast<const uint4 *>(split + SPLIT_W) + lane;

    // Table pieces: the first KE k-steps of the wave's NEXT vertex tile are requested before this tile's transforms and epilogue (KE x 24 VGPRs live
    // under them), the rest when a tile starts - they arrive under its first KE x 9 MFMAs.  (All 60 fragments resident next to the accumulators
    // do not fit the 256 + 256 register split: the compiler answers with scratch traffic.)
    constexpr int KE = 6;
    uint4 Ae[KE][6], Wp[3];
    auto fetch_early = [&](int vt) {
        const uint4 *a = Apd + (size_t)vt * KS * 6 * 64;
#pragma unroll
        for (int ks = 0; ks < KE; ++ks)
#pragma unroll
            for (int q = 0; q < 6; ++q) Ae[ks][q] = a[(ks * 6 + q) * 64];
#pragma unroll
        for (int p = 0; p < 3; ++p) Wp[p] = Aw[(vt * 3 + p) * 64];
    };
    fetch_early(wave);
    for (int g = 0; g < (VT + 3) / 4; ++g) {
        const int vt = g * 4 + wave;
        if (vt < VT) {
            uint4 Al[KS - KE][6];
            {
                const uint4 *a = Apd + (size_t)vt * KS * 6 * 64;
#pragma unroll
                for (int ks = KE; ks < KS; ++ks)
#pragma unroll
                    for (int q = 0; q < 6; ++q) Al[ks - KE][q] = a[(ks * 6 + q) * 64];
            }
            f32x16 X[3];
#pragma unroll
            for (int c = 0; c < 3; ++c)
#pragma unroll
                for (int i = 0; i < 16; ++i) X[c][i] = 0.f;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const uint4 bh = Bpm[(ks * 2 + 0) * 64], bm = Bpm[(ks * 2 + 1) * 64];
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const uint4 ah = ks < KE ? Ae[ks < KE ? ks : 0][2 * c] : Al[ks < KE ? 0 : ks - KE][2 * c];
                    const uint4 am = ks < KE ? Ae[ks < KE ? ks : 0][2 * c + 1] : Al[ks < KE ? 0 : ks - KE][2 * c + 1];
                    MFMA(ah, bm, X[c]); MFMA(am, bh, X[c]); MFMA(ah, bh, X[c]);
                }
            }
            uint4 wp[3];
#pragma unroll
            for (int p = 0; p < 3; ++p) wp[p] = Wp[p];
            __builtin_amdgcn_sched_barrier(0);          // the loads below take the registers the products above have just released
            if (vt + 4 < VT) fetch_early(vt + 4);
            __builtin_amdgcn_sched_barrier(0);
            float *o = trans + hyp * TP + wave * 96 + 12 * half;
            const float *tm = tmpl + vt * 32 + 4 * half;
            float x[3][16];
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int vr = (i & 3) + 8 * (i >> 2);                     // (+ 4 * half) the accumulator's row = vertex within the tile
#pragma unroll
                for (int c = 0; c < 3; ++c) x[c][i] = X[c][i] + tm[c * VP + vr];
            }
#pragma unroll
            for (int c = 0; c < 3; ++c) {            // one output coordinate at a time: four transform entries live instead of twelve
                f32x16 T[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int e = q < 3 ? 3 * c + q : 9 + c;
#pragma unroll
                    for (int i = 0; i < 16; ++i) T[q][i] = 0.f;
                    const uint4 gh = Bg[(e * 3 + 0) * 64], gm = Bg[(e * 3 + 1) * 64], gl = Bg[(e * 3 + 2) * 64];
                    MFMA(wp[2], gh, T[q]); MFMA(wp[0], gl, T[q]); MFMA(wp[1], gm, T[q]);           // small terms first
                    MFMA(wp[1], gh, T[q]); MFMA(wp[0], gm, T[q]); MFMA(wp[0], gh, T[q]);
                }
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int vr = (i & 3) + 8 * (i >> 2);
                    const float vp = T[0][i] * x[0][i] + T[1][i] * x[1][i] + T[2][i] * x[2][i] + T[3][i];      // manolayer.py:236-246
                    o[vr * 3 + c] = fmaf(vp, osc, ooff[c]);     // (vp - centre) * 1000 (manolayer.py:262-273), then (. - root) / bone (network.py:480)
                }
            }
        }
        __syncthreads();
        // a hypothesis' 128 vertices of this group: one 1.5 KB run (8-byte pieces: a row of 778 x 3 floats starts on an 8-byte boundary)
        if (g * 384 + 384 <= NV * 3) {
#pragma unroll
            for (int hh = 0; hh < HT / 4; hh += 2) {
                float2 v[2][3];
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int q = 0; q < 3; ++q) v[u][q] = reinterpret_cast<const float2 *>(trans + (wave + 4 * (hh + u)) * TP)[lane + 64 * q];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int r = r0 + wave + 4 * (hh + u);
                    if (r < R) {
                        float2 *dst = reinterpret_cast<float2 *>(verts_o + (size_t)r * NV * 3 + g * 384);
#pragma unroll
                        for (int q = 0; q < 3; ++q) dst[lane + 64 * q] = v[u][q];
                    }
                }
            }
        } else {
            const int cols = NV * 3 - g * 384;
            for (int h = wave; h < HT && r0 + h < R; h += 4)
                for (int j = lane; j < cols; j += 64) verts_o[(size_t)(r0 + h) * NV * 3 + g * 384 + j] = trans[h * TP + j];
        }
        __syncthreads();                 // the rows have left the transposition buffer
    }
}

}}  // namespace mhe::mano

using namespace mhe;

size_t mhe_mano_skin_split_floats() { return (mano::SPLIT_ELEMS * sizeof(u16) + 3) / 4; }

int mhe_mano_skin_mfma(const float *ws_rows, const float *tables, float *split, float *verts, int R, int mm_mode, hipStream_t stream) {
    static const bool set = [] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(mano::mano_skin_mfma_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, mano::LDS_BYTES);
        return true;
    }();
    (void)set;
    const int frags = mano::VT * mano::KS * 3 + mano::VT;
    hipLaunchKernelGGL(mano::split_tables_kernel, dim3((frags * 64 + 255) / 256), dim3(256), 0, stream, tables, reinterpret_cast<u16 *>(split));
    if (int rc = check_launch("split_tables_kernel")) return rc;
    hipLaunchKernelGGL(mano::mano_skin_mfma_kernel, dim3((R + mano::HT - 1) / mano::HT), dim3(256), mano::LDS_BYTES, stream, ws_rows, tables,
                       reinterpret_cast<const u16 *>(split), verts, R, mm_mode);
    return check_launch("mano_skin_mfma_kernel");
}

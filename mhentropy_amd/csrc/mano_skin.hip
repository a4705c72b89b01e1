// Full-mesh linear-blend skinning of the MANO hand on the matrix cores (round 5) - the largest line of the reference iteration's metrics pass
// (sample(N = [200, 200], mods = {uv, xyz, verts}), hand/CrossModalHand.py:357-361: 51,200 hypotheses x 778 vertices per iteration at B = 256).
//
// Reference arithmetic being replaced (all fp32): hand/manopth/manolayer.py:181-188 (shape + pose-corrective blend shapes), :236-246
// (per-vertex transform T = sum_j w_j G_j, v' = T [v; 1]), :262-273 and hand/network.py:480 (centre, millimetres, root / bone normalisation).
//
// First version (mano_skin_kernel<16>, mano.hip): one thread per vertex, every per-hypothesis value a scalar operand of an FMA - 640 FMAs per
// (hypothesis, vertex), 39 TFLOP/s = a quarter of the f32 vector peak, 1.40 ms at 51,200 hypotheses for 478 MB of output (0.34 TB/s).
//
// This version: both products are GEMMs with the HYPOTHESIS on the row index of v_mfma_f32_32x32x16_bf16 and the VERTEX on the column (lane):
//     X_c[h][v] = sum_k PM[h][k] PD_c[k][v]      c = 0..2, k = 135 pose-map entries + 10 betas + 2 template slots (K = 160 with zero padding)
//     T_e[h][v] = sum_j G_e[h][j]  W[j][v]       e = 0..11 (3 x 4 transform entries), K = 16 joints
// so that a lane ends up with X_0..2 and T_0..11 of ITS vertex for 16 hypotheses in registers, and v' = T [X; 1] needs no lane movement.
// f32 ACCURACY ON bf16 MATRIX CORES: every f32 operand is split into bf16 pieces (x = h + m + l, 8 mantissa bits each: h + m + l IS x) and
// the product is the sum of the piece products, accumulated in f32 -
//     T : 3 x 3 pieces, the 6 products hh, hm, mh, hl, lh, mm: terms below 2^-24 of the product dropped = an f32 FMA chain's accuracy
//     X : 2 x 2 pieces, 3 products (hh, hm, mh): 2^-16 of the blend OFFSET (<= a few mm against a template of ~100 mm: < 1e-6 of the vertex);
//         the template itself rides in two K slots against a pose-map entry of 1.0 - slot 145 = its (h, m) pieces, slot 146 = (l, 0): exact
// 162 MFMAs of 32 cycles per (32 hypotheses x 32 vertices) = 5.2 k cycles against ~20 k for the f32 MFMA form and ~80 k VALU cycles before.
// Measured against the f64 oracle: 5-8e-7 of the mesh's extent, the f32 oracle's own 2-6e-7 (tests/test_gpu_kernels.py).
// A workgroup = 32 hypotheses x all vertices, two workgroups per CU (72.5 KiB of LDS, 214 registers): the hypotheses' pieces live in LDS (A
// operands), a wave walks every fourth vertex tile with the table pieces (B operands; fragment-major, 1 KiB per fragment, made from the f32
// table by split_tables_kernel on every call: 5 us) streamed L2 -> registers two k-steps ahead.  A lane stores the 12 contiguous bytes of its
// vertex per hypothesis as 8 + 4 (the first coordinate waits in a wave-private 4 KiB of LDS; holding it in 16 more registers tips the
// allocator into 160 spills), 32 lanes = 384 contiguous bytes of an output row - no transposition, no barrier after the prologue.
// Trajectory at 51,200 hypotheses (tools/skin_bench.py, kernel alone): 1,400 us (one thread per vertex) -> 497 (first matrix-core version:
// hypothesis on the lane, 48 KiB transposition buffer, one workgroup per CU, two barriers per four tiles, ~150 us of prologue with nothing to
// overlap it) -> 412 (this orientation, 4-byte stores 12 bytes apart: the stores alone 184 us) -> 370.  By elimination: products 143 us
// (MFMA bound at two workgroups per CU and 3.125 -> 4 rounds: 130), prologue 29, table stream +55, stores +143 (478 MB: ~100 at the write rate).
#include "common.h"
#include "mano_layout.h"

namespace mhe { namespace mano {

typedef __attribute__((ext_vector_type(8))) __bf16 bf8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

constexpr int VT = VP / 32;                  // 26 vertex tiles in the tables (padded)
constexpr int VTL = (NV + 31) / 32;          // 25 of them hold vertices
constexpr int KS = 10;                       // k-steps of 16: 135 pose-map entries, 10 betas, 15 zeros
constexpr size_t SPLIT_PD = 0;               // u16 [VT][KS][3 coordinates][2 pieces][64 lanes][8]
constexpr size_t SPLIT_W = (size_t)VT * KS * 3 * 2 * 512;       // u16 [VT][3 pieces][64 lanes][8]
constexpr size_t SPLIT_ELEMS = SPLIT_W + (size_t)VT * 3 * 512;
constexpr int HT = 32;                       // hypotheses per workgroup
constexpr int STAGE_BYTES = HT * WS_STRIDE * 4;                                  // the workspace rows, staged (44 KB); then: the G pieces (36 KB) and
constexpr int TAIL_BYTES = 12 * 3 * 1024 + 4 * 4096;                             // 4 KB per wave where a coordinate waits for its neighbour
constexpr int LDS_BYTES = KS * 2 * 1024 + HT * 16 + (TAIL_BYTES > STAGE_BYTES ? TAIL_BYTES : STAGE_BYTES);

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
            if (k == 145 || k == 146) {              // the template against a pose-map entry of 1.0: slot 145 = (h, m), slot 146 = (l, 0)
                u16 th, tm, tl;
                split3(tables[V_T + c * VP + vt * 32 + v], th, tm, tl);
                h[j] = k == 145 ? th : tl; m[j] = k == 145 ? tm : (u16)0;
                continue;
            }
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

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))
void mano_skin_mfma_kernel(const float *__restrict__ ws, const u16 *__restrict__ split, float *__restrict__ verts_o, int R, int mm_mode) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    u16 *PMb = reinterpret_cast<u16 *>(smem);                    // [KS][2 pieces][2 k halves][32 hypotheses][8]: operand order (lane = half * 32 + hypothesis)
    float4 *nrm = reinterpret_cast<float4 *>(PMb + KS * 2 * 512);   // [32]: output scale, 3 offsets
    float *stage = reinterpret_cast<float *>(nrm + HT);          // [32][352] workspace rows, then
    u16 *Gb = reinterpret_cast<u16 *>(stage);                    // [12 e][3 pieces][2 joint halves][32][8]  (written after the rows have been read)
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int r0 = blockIdx.x * HT;
    // ---- the workgroup's hypotheses: workspace rows -> LDS by coalesced 16-byte loads -> bf16 pieces in operand order
    {
        float4 st[11];
#pragma unroll
        for (int q = 0; q < 11; ++q) {
            const int i = tid + q * 256, h = i / 88, c4 = i - h * 88;             // 88 float4 per row of 352 floats
            st[q] = reinterpret_cast<const float4 *>(ws + (size_t)(r0 + h < R ? r0 + h : R - 1) * WS_STRIDE)[c4];
        }
#pragma unroll
        for (int q = 0; q < 11; ++q) reinterpret_cast<float4 *>(stage)[tid + q * 256] = st[q];
    }
    __syncthreads();
    for (int i = tid; i < HT * 160; i += 256) {
        const int h = i & 31;                                  // consecutive lanes: consecutive hypotheses (row pitch 352 floats: 32-way on the
        int kk = (i >> 5) + h;                                 // read - spread by rotating k per hypothesis)
        kk = kk >= 160 ? kk - 160 : kk;
        const float *w = stage + h * WS_STRIDE;
        const float x = kk < 135 ? w[WS_PM + kk] : kk < 145 ? w[WS_BT + kk - 135] : kk < 147 ? 1.f : 0.f;      // 145, 146: the template's slots
        u16 hi, mi;
        split2(x, hi, mi);
        u16 *o = PMb + ((kk >> 4) * 4 + ((kk >> 3) & 1)) * 256 + h * 8 + (kk & 7);
        o[0] = hi; o[512] = mi;
    }
    if (tid < HT) {
        // output = (vp - centre) * 1000 [millimetre mode], then (. - root) / bone: one FMA per coordinate
        const float *w = stage + tid * WS_STRIDE + WS_NRM;
        const float sc = mm_mode ? 1000.f : 1000.f / w[6];
        nrm[tid] = mm_mode ? make_float4(sc, -1000.f * w[0], -1000.f * w[1], -1000.f * w[2])
                           : make_float4(sc, -(1000.f * w[0] + w[3]) / w[6], -(1000.f * w[1] + w[4]) / w[6], -(1000.f * w[2] + w[5]) / w[6]);
    }
    // the transforms: read all of them before their pieces overwrite the staged rows
    {
        float g[24];
#pragma unroll
        for (int q = 0; q < 24; ++q) {
            const int i = tid + q * 256, h = i & 31;
            int qq = (i >> 5) + h;
            qq = qq >= 192 ? qq - 192 : qq;
            g[q] = stage[h * WS_STRIDE + WS_GR + qq];
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 24; ++q) {
            const int i = tid + q * 256, h = i & 31;
            int qq = (i >> 5) + h;
            qq = qq >= 192 ? qq - 192 : qq;
            const int j = (qq * 683) >> 13, e = qq - j * 12;      // qq / 12 for qq < 192
            u16 hi, mi, lo;
            split3(g[q], hi, mi, lo);
            u16 *o = Gb + (e * 6 + (j >> 3)) * 256 + h * 8 + (j & 7);
            o[0] = hi; o[512] = mi; o[1024] = lo;
        }
    }
    __syncthreads();
    const int vl = lane & 31, half = lane >> 5;
    float *const o0 = reinterpret_cast<float *>(smem + KS * 2 * 1024 + HT * 16 + 12 * 3 * 1024) + wave * 1024 + lane;      // [16][64] per wave: the first coordinate waits here
    const int hlim = R - r0 - 4 * half;                         // rows of this lane's half past the end are not stored
    const __amdgpu_buffer_rsrc_t vout = __builtin_amdgcn_make_buffer_rsrc(verts_o, 0, (int)((unsigned)R * (unsigned)(NV * 12)), 0x00020000);
    const unsigned row0 = (unsigned)r0 * (unsigned)(NV * 12);
    const uint4 *Apm = reinterpret_cast<const uint4 *>(PMb) + lane;          // + (ks * 2 + p) * 64
    const uint4 *Ag = reinterpret_cast<const uint4 *>(Gb) + lane;            // + (e * 3 + p) * 64
    const uint4 *Bpd = reinterpret_cast<const uint4 *>(split + SPLIT_PD) + lane;
    const uint4 *Bw = reinterpret_cast<const uint4 *>(split + SPLIT_W) + lane;

    // table pieces: a ring of DEPTH k-steps (6 fragments each) in flight ahead of the products, running on into the wave's next tile
    constexpr int DEPTH = 2;            // (KS % DEPTH == 0: the ring runs on into the next tile at slot 0)
    uint4 Bf[DEPTH][6], Wp[3];
    auto fetch = [&](uint4 (&f)[6], int vt, int ks) {
        const uint4 *b = Bpd + ((size_t)vt * KS + ks) * 6 * 64;
#pragma unroll
        for (int q = 0; q < 6; ++q) f[q] = b[q * 64];
    };
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) fetch(Bf[d], wave, d);
    for (int vt = wave; vt < VTL; vt += 4) {
#pragma unroll
        for (int p = 0; p < 3; ++p) Wp[p] = Bw[(vt * 3 + p) * 64];
        f32x16 X[3];
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int i = 0; i < 16; ++i) X[c][i] = 0.f;
        const int vn = vt + 4 < VTL ? vt + 4 : vt;                // (the last tile re-fetches its own first k-steps: nothing reads them)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const uint4 ah = Apm[(ks * 2 + 0) * 64], am = Apm[(ks * 2 + 1) * 64];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                MFMA(am, Bf[ks % DEPTH][2 * c], X[c]); MFMA(ah, Bf[ks % DEPTH][2 * c + 1], X[c]); MFMA(ah, Bf[ks % DEPTH][2 * c], X[c]);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (ks + DEPTH < KS) fetch(Bf[ks % DEPTH], vt, ks + DEPTH);
            else fetch(Bf[ks % DEPTH], vn, ks + DEPTH - KS);
            __builtin_amdgcn_sched_barrier(0);
        }
        const int v = vt * 32 + vl;
        // the lane's part of the address (the rest is wave-uniform: SGPR offset).  Lanes without a vertex, and below rows past R, get an offset the
        // buffer's range check rejects (it looks at the VGPR offset: num_records below is the whole tensor, SGPR offsets stay inside it)
        const unsigned voff = v < NV ? (unsigned)((4 * half * NV + v) * 12) : 0xffffffffu;
#pragma unroll
        for (int c = 0; c < 3; ++c) {            // one output coordinate at a time: four transform entries live instead of twelve
            f32x16 T[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int e = q < 3 ? 3 * c + q : 9 + c;
#pragma unroll
                for (int i = 0; i < 16; ++i) T[q][i] = 0.f;
                const uint4 gh = Ag[(e * 3 + 0) * 64], gm = Ag[(e * 3 + 1) * 64], gl = Ag[(e * 3 + 2) * 64];
                MFMA(gh, Wp[2], T[q]); MFMA(gl, Wp[0], T[q]); MFMA(gm, Wp[1], T[q]);           // small terms first
                MFMA(gh, Wp[1], T[q]); MFMA(gm, Wp[0], T[q]); MFMA(gh, Wp[0], T[q]);
                __builtin_amdgcn_sched_barrier(0);
            }
            // v' = T [X; 1] (manolayer.py:236-246); (v' - centre) * 1000 (:262-273), then (. - root) / bone (network.py:480).  A lane's three
            // coordinates of one hypothesis leave as an 8-byte and a 4-byte store, 32 lanes = 384 contiguous bytes of the hypothesis' row
            const float *np = reinterpret_cast<const float *>(nrm + 4 * half);
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int hr = (i & 3) + 8 * (i >> 2);                    // (+ 4 * half) the accumulator's row = hypothesis within the workgroup
                const float vp = T[0][i] * X[0][i] + T[1][i] * X[1][i] + T[2][i] * X[2][i] + T[3][i];
                const float r = fmaf(vp, np[hr * 4], np[hr * 4 + 1 + c]);
                const unsigned soff = row0 + (unsigned)(hr * NV * 12);
                if (c == 0) o0[i * 64] = r;
                else if (c == 1) {
                    typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
                    const u32x2 o = {__float_as_uint(o0[i * 64]), __float_as_uint(r)};
                    __builtin_amdgcn_raw_buffer_store_b64(o, vout, (int)(hr < hlim ? voff : 0xffffffffu), (int)soff, 0);
                } else __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(r), vout, (int)(hr < hlim ? voff : 0xffffffffu), (int)(soff + 8u), 0);
            }
        }
    }
}

}}  // namespace mhe::mano

using namespace mhe;

size_t mhe_mano_skin_split_floats() { return (mano::SPLIT_ELEMS * sizeof(u16) + 3) / 4; }

int mhe_mano_skin_mfma(const float *ws_rows, const float *tables, float *split, float *verts, int R, int mm_mode, hipStream_t stream) {
    MHE_REQUIRE((size_t)(R + 64) * mano::NV * 12 < (1ull << 32), "full-mesh skinning: R=%d beyond the 32-bit byte offsets of the output", R);
    const int frags = mano::VT * mano::KS * 3 + mano::VT;
    hipLaunchKernelGGL(mano::split_tables_kernel, dim3((frags * 64 + 255) / 256), dim3(256), 0, stream, tables, reinterpret_cast<u16 *>(split));
    if (int rc = check_launch("split_tables_kernel")) return rc;
    hipLaunchKernelGGL(mano::mano_skin_mfma_kernel, dim3((R + mano::HT - 1) / mano::HT), dim3(256), mano::LDS_BYTES, stream, ws_rows,
                       reinterpret_cast<const u16 *>(split), verts, R, mm_mode);
    return check_launch("mano_skin_mfma_kernel");
}

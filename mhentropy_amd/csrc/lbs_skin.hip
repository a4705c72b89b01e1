// Linear-blend skinning of a body model of RUNTIME size on the matrix cores (round 5): the scheme of mano_skin.hip - both products as GEMMs on
// v_mfma_f32_32x32x16_bf16 with every f32 operand split into bf16 pieces whose products are summed in f32 (blend shapes: 2 x 2 pieces, 3 products;
// per-vertex transforms: 3 x 3 pieces, 6 products; the template exact in two K slots against a coefficient of 1.0), hypothesis = MFMA row (pieces in
// LDS), vertex = column = lane (table pieces fragment-major from L2 / the Infinity Cache), 12 contiguous bytes per lane and hypothesis - for
// J <= 32 joints, any vertex count and nb shape coefficients: SMPL (24 joints, 6,890 vertices, 10 + 207 blend coefficients), the body path of
// SURVEY.md section 8 row f1 / BASELINE.json configs[4].  Reference arithmetic: hand/manopth/manolayer.py:181-188,236-246 at other sizes.
// Differences from the hand kernel: the sizes are kernel arguments (K = 9 (J - 1) + nb + 2 in KS k-steps of 16, KS even; the joints in JS = 1 or 2
// k-steps); the table pieces are made ONCE per model (mhe_lbs_split_tables_f32: they are buffers of the module, 19 MB for SMPL), not per call;
// the hypotheses' pieces take 100 KiB of LDS at SMPL's sizes, so one workgroup per CU (216 vertex tiles per workgroup: the prologue is 2 % of it).
// The scalar-operand kernel (body.hip: lbs_skin_kernel<8>, 20 TFMA/s) stays for outputs beyond 4 GiB and as MHE_LBS_MFMA=0.
#include "common.h"

namespace mhe { namespace body {

// workspace row of lbs_pose_kernel (body.hip): pose map [9(J-1)] | betas [nb] | skinning transforms [J][12] | posed joints [J][3]
__host__ __device__ inline int lws_bt(int J, int) { return 9 * (J - 1); }
__host__ __device__ inline int lws_a(int J, int nb) { return 9 * (J - 1) + nb; }
__host__ __device__ inline int lws_stride(int J, int nb) { return (9 * (J - 1) + nb + 15 * J + 15) / 16 * 16; }

typedef __attribute__((ext_vector_type(8))) __bf16 bf8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
constexpr int HT = 32;

__host__ __device__ inline int lbs_ks(int J, int nb) { return ((9 * (J - 1) + nb + 2 + 15) / 16 + 1) / 2 * 2; }      // k-steps of 16, even
__host__ __device__ inline int lbs_js(int J) { return (J + 15) / 16; }
// u16 elements: PD [VT][KS][3][2][512], then W [VT][JS][3][512]
__host__ __device__ inline size_t lbs_split_pd(int VT, int KS) { return (size_t)VT * KS * 6 * 512; }
__host__ __device__ inline size_t lbs_split_elems(int VT, int KS, int JS) { return lbs_split_pd(VT, KS) + (size_t)VT * JS * 3 * 512; }

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

// the f32 vertex tables (vertex-fastest, pitch VP: body.py) -> bf16 pieces in MFMA operand order (lane l: vertex l & 31, k = 8 (l >> 5) + 0..7)
__global__ __launch_bounds__(256) void lbs_split_tables_kernel(const float *__restrict__ Vt, const float *__restrict__ Vsd, const float *__restrict__ Vpd,
                                                               const float *__restrict__ Vw, u16 *__restrict__ out, int J, int nb, int VP, int KS, int JS) {
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    const int lane = (int)(t & 63), v = lane & 31, kh = lane >> 5;
    const long f = t >> 6;
    const int VT = VP / 32, NP = 9 * (J - 1);
    if (f < (long)VT * KS * 3) {
        const int c = (int)(f % 3), ks = (int)((f / 3) % KS), vt = (int)(f / (3 * KS));
        u16 *o = out + ((size_t)f * 2 * 64 + lane) * 8;
        for (int j = 0; j < 8; ++j) {
            const int k = ks * 16 + kh * 8 + j;
            u16 h = 0, m = 0;
            if (k == NP + nb || k == NP + nb + 1) {          // the template against a coefficient of 1.0: (h, m), then (l, 0)
                u16 th, tm, tl;
                split3(Vt[(size_t)c * VP + vt * 32 + v], th, tm, tl);
                h = k == NP + nb ? th : tl; m = k == NP + nb ? tm : (u16)0;
            } else if (k < NP + nb) {
                const float x = k < NP ? Vpd[((size_t)k * 3 + c) * VP + vt * 32 + v] : Vsd[((size_t)(k - NP) * 3 + c) * VP + vt * 32 + v];
                split2(x, h, m);
            }
            o[j] = h; o[512 + j] = m;
        }
    } else if (f < (long)VT * KS * 3 + (long)VT * JS) {
        const long g = f - (long)VT * KS * 3;
        const int js = (int)(g % JS), vt = (int)(g / JS);
        u16 *o = out + lbs_split_pd(VT, KS) + ((size_t)g * 3 * 64 + lane) * 8;
        for (int j = 0; j < 8; ++j) {
            const int jj = js * 16 + kh * 8 + j;
            u16 h = 0, m = 0, l = 0;
            if (jj < J) split3(Vw[(size_t)jj * VP + vt * 32 + v], h, m, l);
            o[j] = h; o[512 + j] = m; o[1024 + j] = l;
        }
    }
}

#define MFMA(a, b, c) c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8, (a)), __builtin_bit_cast(bf8, (b)), (c), 0, 0, 0)

template <int JS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))
void lbs_skin_mfma_kernel(const float *__restrict__ ws, const u16 *__restrict__ split, float *__restrict__ verts_o, int R, int J, int nb, int NV, int VP,
                          int KS, float scale) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    u16 *PMb = reinterpret_cast<u16 *>(smem);                    // [KS][2 pieces][2 k halves][32 hypotheses][8]
    u16 *Gb = PMb + (size_t)KS * 2 * 512;                        // [12 e][JS][3 pieces][2 joint halves][32][8]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int r0 = blockIdx.x * HT;
    const int NP = 9 * (J - 1), KT = KS * 16, stride = lws_stride(J, nb), oa = lws_a(J, nb), ob = lws_bt(J, nb);
    // ---- the workgroup's hypotheses: workspace rows -> bf16 pieces in operand order (consecutive lanes: consecutive hypotheses, k rotated per
    // hypothesis so that the 32 rows' reads do not all start in one memory channel)
    for (int i = tid; i < HT * KT; i += 256) {
        const int h = i & 31;
        int kk = (i >> 5) + h;
        kk = kk >= KT ? kk - KT : kk;
        const float *w = ws + (size_t)(r0 + h < R ? r0 + h : R - 1) * stride;
        const float x = kk < NP ? w[kk] : kk < NP + nb ? w[ob + kk - NP] : kk < NP + nb + 2 ? 1.f : 0.f;
        u16 hi, mi;
        split2(x, hi, mi);
        u16 *o = PMb + ((kk >> 4) * 4 + ((kk >> 3) & 1)) * 256 + h * 8 + (kk & 7);
        o[0] = hi; o[512] = mi;
    }
    for (int i = tid; i < HT * JS * 16 * 12; i += 256) {
        const int h = i & 31, q = i >> 5, j = q / 12, e = q - j * 12;          // j < JS * 16 (joints past J: zero pieces)
        const float *w = ws + (size_t)(r0 + h < R ? r0 + h : R - 1) * stride;
        u16 hi = 0, mi = 0, lo = 0;
        if (j < J) split3(w[oa + j * 12 + e], hi, mi, lo);
        u16 *o = Gb + (((e * JS + (j >> 4)) * 3) * 2 + ((j >> 3) & 1)) * 256 + h * 8 + (j & 7);
        o[0] = hi; o[512] = mi; o[1024] = lo;
    }
    __syncthreads();
    const int vl = lane & 31, half = lane >> 5;
    float *const o0 = reinterpret_cast<float *>(smem + (size_t)KS * 2 * 1024 + 12 * JS * 3 * 1024) + wave * 1024 + lane;      // [16][64] per wave
    const int hlim = R - r0 - 4 * half;                         // rows of this lane's half past the end are not stored
    const __amdgpu_buffer_rsrc_t vout = __builtin_amdgcn_make_buffer_rsrc(verts_o, 0, (int)((unsigned)R * (unsigned)(NV * 12)), 0x00020000);
    const unsigned row0 = (unsigned)r0 * (unsigned)(NV * 12);
    const int VT = VP / 32, VTL = (NV + 31) / 32;
    const uint4 *Apm = reinterpret_cast<const uint4 *>(PMb) + lane;          // + (ks * 2 + p) * 64
    const uint4 *Ag = reinterpret_cast<const uint4 *>(Gb) + lane;            // + ((e * JS + js) * 3 + p) * 64
    const uint4 *Bpd = reinterpret_cast<const uint4 *>(split) + lane;
    const uint4 *Bw = reinterpret_cast<const uint4 *>(split + lbs_split_pd(VT, KS)) + lane;

    // table pieces: two k-steps (6 fragments each) in flight ahead of the products, running on into the wave's next tile (KS is even)
    uint4 Bf[2][6], Wp[JS][3];
    auto fetch = [&](uint4 (&f)[6], int vt, int ks) {
        const uint4 *b = Bpd + ((size_t)vt * KS + ks) * 6 * 64;
#pragma unroll
        for (int q = 0; q < 6; ++q) f[q] = b[q * 64];
    };
    if (wave < VTL) { fetch(Bf[0], wave, 0); fetch(Bf[1], wave, 1); }
    for (int vt = wave; vt < VTL; vt += 4) {
#pragma unroll
        for (int js = 0; js < JS; ++js)
#pragma unroll
            for (int p = 0; p < 3; ++p) Wp[js][p] = Bw[((vt * JS + js) * 3 + p) * 64];
        f32x16 X[3];
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int i = 0; i < 16; ++i) X[c][i] = 0.f;
        const int vn = vt + 4 < VTL ? vt + 4 : vt;                // (the last tile re-fetches its own first k-steps: nothing reads them)
        for (int ks = 0; ks < KS; ks += 2) {
#pragma unroll
            for (int d = 0; d < 2; ++d) {
                const uint4 ah = Apm[((ks + d) * 2 + 0) * 64], am = Apm[((ks + d) * 2 + 1) * 64];
#pragma unroll
                for (int c = 0; c < 3; ++c) { MFMA(am, Bf[d][2 * c], X[c]); MFMA(ah, Bf[d][2 * c + 1], X[c]); MFMA(ah, Bf[d][2 * c], X[c]); }
                __builtin_amdgcn_sched_barrier(0);
                if (ks + d + 2 < KS) fetch(Bf[d], vt, ks + d + 2);
                else fetch(Bf[d], vn, ks + d + 2 - KS);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        const int v = vt * 32 + vl;
        // the lane's part of the address (the rest is wave-uniform: SGPR offset); lanes without a vertex and rows past R get an offset the
        // buffer's range check rejects (it looks at the VGPR offset)
        const unsigned voff = v < NV ? (unsigned)((4 * half * NV + v) * 12) : 0xffffffffu;
#pragma unroll
        for (int c = 0; c < 3; ++c) {            // one output coordinate at a time: four transform entries live instead of twelve
            f32x16 T[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int e = q < 3 ? 3 * c + q : 9 + c;
#pragma unroll
                for (int i = 0; i < 16; ++i) T[q][i] = 0.f;
#pragma unroll
                for (int js = 0; js < JS; ++js) {
                    const uint4 gh = Ag[((e * JS + js) * 3 + 0) * 64], gm = Ag[((e * JS + js) * 3 + 1) * 64], gl = Ag[((e * JS + js) * 3 + 2) * 64];
                    MFMA(gh, Wp[js][2], T[q]); MFMA(gl, Wp[js][0], T[q]); MFMA(gm, Wp[js][1], T[q]);           // small terms first
                    MFMA(gh, Wp[js][1], T[q]); MFMA(gm, Wp[js][0], T[q]); MFMA(gh, Wp[js][0], T[q]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            // v' = T [X; 1] (manolayer.py:236-246), x scale; 8 + 4 bytes of the lane's 12 per hypothesis (the first coordinate waits in LDS)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int hr = (i & 3) + 8 * (i >> 2);                    // (+ 4 * half) the accumulator's row = hypothesis within the workgroup
                const float r = (T[0][i] * X[0][i] + T[1][i] * X[1][i] + T[2][i] * X[2][i] + T[3][i]) * scale;
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

}}  // namespace mhe::body

using namespace mhe;

extern "C" size_t mhe_lbs_split_floats(int J, int nb, int VP) {
    if (J <= 0 || J > 32 || nb < 0 || VP <= 0 || VP % 32) return 0;
    return (body::lbs_split_elems(VP / 32, body::lbs_ks(J, nb), body::lbs_js(J)) * sizeof(u16) + 3) / 4;
}

extern "C" int mhe_lbs_split_tables_f32(const float *v_template, const float *v_shapedirs, const float *v_posedirs, const float *v_weights,
                                        float *split, int J, int nb, int VP, void *stream) {
    MHE_REQUIRE(v_template && v_shapedirs && v_posedirs && v_weights && split, "mhe_lbs_split_tables_f32: null pointer");
    MHE_REQUIRE(J > 0 && J <= 32 && nb > 0 && VP > 0 && VP % 32 == 0, "mhe_lbs_split_tables_f32: J=%d nb=%d VP=%d (VP a multiple of 32)", J, nb, VP);
    const int KS = body::lbs_ks(J, nb), JS = body::lbs_js(J), VT = VP / 32;
    const long frags = (long)VT * KS * 3 + (long)VT * JS;
    hipLaunchKernelGGL(body::lbs_split_tables_kernel, dim3((unsigned)((frags * 64 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, v_template,
                       v_shapedirs, v_posedirs, v_weights, reinterpret_cast<u16 *>(split), J, nb, VP, KS, JS);
    return check_launch("lbs_split_tables_kernel");
}

extern "C" int mhe_lbs_skin_mfma_supported(int R, int J, int nb, int NV, int VP) {
    if (R <= 0 || J <= 0 || J > 32 || nb <= 0 || NV <= 0 || VP < NV || VP % 32) return 0;
    if ((size_t)(R + 64) * NV * 12 >= (1ull << 32)) return 0;                                   // 32-bit byte offsets of the output
    const size_t lds = (size_t)body::lbs_ks(J, nb) * 2048 + (size_t)12 * body::lbs_js(J) * 3072 + 4 * 4096;
    return lds <= 160 * 1024;
}

extern "C" int mhe_lbs_skin_mfma_f32(const float *workspace, const float *split, float *verts, int R, int J, int nb, int NV, int VP, float scale,
                                     void *stream) {
    MHE_REQUIRE(workspace && split && verts, "mhe_lbs_skin_mfma_f32: null pointer");
    MHE_REQUIRE(mhe_lbs_skin_mfma_supported(R, J, nb, NV, VP), "mhe_lbs_skin_mfma_f32: R=%d J=%d nb=%d NV=%d VP=%d not supported (see mhe_lbs_skin_mfma_supported)",
                R, J, nb, NV, VP);
    const int KS = body::lbs_ks(J, nb), JS = body::lbs_js(J);
    const int lds = KS * 2048 + 12 * JS * 3072 + 4 * 4096;
    const dim3 grid((unsigned)((R + body::HT - 1) / body::HT));
    if (JS == 1) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(body::lbs_skin_mfma_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        hipLaunchKernelGGL(body::lbs_skin_mfma_kernel<1>, grid, dim3(256), lds, (hipStream_t)stream, workspace, reinterpret_cast<const u16 *>(split), verts,
                           R, J, nb, NV, VP, KS, scale);
    } else {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(body::lbs_skin_mfma_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        hipLaunchKernelGGL(body::lbs_skin_mfma_kernel<2>, grid, dim3(256), lds, (hipStream_t)stream, workspace, reinterpret_cast<const u16 *>(split), verts,
                           R, J, nb, NV, VP, KS, scale);
    }
    return check_launch("lbs_skin_mfma_kernel");
}

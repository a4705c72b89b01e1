// ResNet stem for the forward-only path: conv1 7x7 / stride 2 (bf16 MFMA) + batch statistics + the 3x3 / stride-2 max pool in ONE kernel
// (reference hand/network.py:54-61,110 = torchvision ResNet.conv1 -> bn1 -> relu -> maxpool in training mode).
//
// Train-mode BatchNorm needs the batch statistics of conv1's raw output before anything can be normalised, so the library wrote that
// output (537 MB at config C2), and a second kernel read it back (816 MB of traffic) to pool it: 0.46 ms of the forward for 0.94 GB of
// algorithmic traffic.  But relu(s * y + t) is monotone in y for either sign of s, and sign(s) = sign(gamma) is known before the statistics
// are: max over the window of relu(s y + t) = relu(s * (s >= 0 ? max y : min y) + t).  So this kernel keeps, per channel, the window
// maximum (gamma >= 0) or minimum (gamma < 0) of the RAW bf16-rounded outputs - a quarter of the output's size - and the consumers
// (layer1.0's conv1 and shortcut, both with BatchNorm + ReLU on their operand load) apply the affine once the statistics are known.
// The full-resolution output is never written.  Same values as the two-kernel path bit for bit (monotone maps commute with max).
//
// Persistent workgroups of 8 waves walk strips of 4 pooled rows x the full 64-column width = 9 conv rows x 128 columns (the first row is
// the previous strip's last: 12.5 % recomputed, its statistics not counted).  Per strip: the 23 x 261 x 3 input patch goes NCHW f32 ->
// channel-interleaved bf16 in LDS (the next strip's patch is already in flight in registers); then 9 row steps: wave w computes columns
// 16 w .. 16 w + 15 of the row for all 64 channels (24 MFMAs; the weight fragments live in registers for the workgroup's whole life, the
// activation fragments come straight from the patch as in stem_kernel), rounds to bf16 into a 4-slot ring of row images and adds the
// row's batch statistics; after every second row all threads pool the three newest rows (3 x 3 window, stride 2) and store one pooled
// row with 16-byte lanes.  One barrier per row.
#include "conv_shared.h"

namespace mhe { namespace conv {

namespace {
constexpr int SP_PH = 4;                       // pooled rows per strip
constexpr int SP_ROWS = 2 * SP_PH + 1;         // conv rows per strip
constexpr int SP_W = 128, SP_PW = 64;          // conv / pooled columns (full width)
constexpr int SP_IROWS = 2 * SP_ROWS + 5;      // input rows of a strip's patch (23)
constexpr int SP_ICOLS = 2 * SP_W + 5;         // 261
constexpr int SP_PSI = 3 * SP_ICOLS + 1;       // 784 elements per interleaved patch row (even: fragments are 4-byte aligned)
constexpr int SP_PROWS = SP_IROWS + 1;         // + one zero row: the k-slots of the K padding (kh = 7) of the last conv row land there
constexpr int SP_NV = (3 * SP_IROWS * (2 * SP_W / 4) + 511) / 512;     // float4 loads per thread and strip (9): 69 input rows x 64 float4
constexpr int SP_RING = 4;
// The ring holds the conv outputs as SORTABLE KEYS: bf16 bits (sign-flipped first where gamma < 0, so that the window maximum is always
// the value wanted) mapped so that unsigned 16-bit order = float order (negative: all bits inverted, else: sign bit set).  The 3 x 3
// window maximum is then nine packed v_pk_max_u16 per register pair instead of unpack + nine f32 max per element: the pooling phase was
// a third of the kernel's VALU time.
typedef float f2v __attribute__((ext_vector_type(2)));          // statistics as packed-f32 pairs (v_pk_add_f32 / v_pk_fma_f32)
typedef unsigned short us2 __attribute__((ext_vector_type(2)));
typedef short ss2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned to_key(unsigned x) {
    const unsigned neg = __builtin_bit_cast(unsigned, __builtin_bit_cast(ss2, x) >> (ss2){15, 15});     // 0xffff per negative half
    return x ^ (neg | 0x80008000u);
}
__device__ __forceinline__ unsigned from_key(unsigned k) {
    const unsigned pos = __builtin_bit_cast(unsigned, __builtin_bit_cast(ss2, k) >> (ss2){15, 15});     // 0xffff where the value was >= 0
    return k ^ (~pos | 0x80008000u);
}
__device__ __forceinline__ unsigned max_key(unsigned a, unsigned b) {
    return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(us2, a), __builtin_bit_cast(us2, b)));
}
}

__global__ __launch_bounds__(512) void stem_pool_kernel(const float *__restrict__ x, const u16 *__restrict__ wg, const float *__restrict__ gamma,
                                                        u16 *__restrict__ pooled, mhe_stat_t *__restrict__ stats, int B, int H, int W) {
    __shared__ __attribute__((aligned(16))) u16 patch[SP_PROWS * SP_PSI + 8];      // 37.6 KiB (+ the zero-weight k-slots past the last row's end)
    __shared__ uint4 ring[SP_RING][SP_W * 8];                                      // 4 x 16 KiB: conv rows as [pixel][64 channels], chunks swizzled
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, q = lane >> 4, l15 = lane & 15;
    const int strips_per_img = SP_PW / SP_PH, nstrips = B * strips_per_img;
    // weight fragments: chunk 4 ks + q of row 16 nt + l15, for the kernel's whole life
    uint4 fb[6][4];
#pragma unroll
    for (int ks = 0; ks < 6; ++ks)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) fb[ks][nt] = *reinterpret_cast<const uint4 *>(wg + (size_t)(nt * 16 + l15) * 192 + (4 * ks + q) * 8);
    // pooling role: thread = pooled column tid >> 3, 16-byte channel chunk tid & 7; sign mask of its 8 channels (gamma < 0: min = -max(-v))
    const int pcol = tid >> 3, pch = tid & 7;
    uint4 sgn;
    {
        unsigned m[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
            m[i] = (gamma[pch * 8 + 2 * i] < 0.f ? 0x8000u : 0u) | (gamma[pch * 8 + 2 * i + 1] < 0.f ? 0x80000000u : 0u);
        sgn = make_uint4(m[0], m[1], m[2], m[3]);
    }
    // zero for good: the zero row (and the slack behind it), and in every real row the 3 + 2 padding columns (the strip spans the full image
    // width, so columns -3..-1 and 256..257 are always outside) and the pad element
    for (int i = tid; i < SP_PSI + 8; i += 512) patch[SP_IROWS * SP_PSI + i] = 0;
    for (int i = tid; i < SP_IROWS * 16; i += 512) {
        const int r = i >> 4, e = i & 15;                      // elements 0..8 (columns -3..-1) and 777..783 (columns 256, 257 and the pad)
        patch[r * SP_PSI + (e < 9 ? e : 768 + e)] = 0;
    }
    // sign masks of the channels this lane holds in the accumulator layout (16 nt + 4 q .. + 3)
    unsigned sgl[4][2];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int c0 = nt * 16 + 4 * q + 2 * h;
            sgl[nt][h] = (gamma[c0] < 0.f ? 0x8000u : 0u) | (gamma[c0 + 1] < 0.f ? 0x80000000u : 0u);
        }
    f2v ss1[4][2], ss2[4][2];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int c = 0; c < 2; ++c) { ss1[a][c] = f2v{0.f, 0.f}; ss2[a][c] = f2v{0.f, 0.f}; }

    // the patch's 3 x 23 input rows are whole image rows (256 contiguous floats): float4 load L = tid + 512 j covers row L / 64 (channel
    // row / 23, patch row row % 23), columns 4 (L % 64) .. + 3
    float4 pv[SP_NV];
    auto load_patch = [&](int strip) __attribute__((always_inline)) {
        const int sc = strip < nstrips ? strip : nstrips - 1;
        const int b = sc / strips_per_img, i0 = (sc % strips_per_img) * SP_PH;
        const int iy0 = 2 * (2 * i0 - 1) - 3;
        const float *xb = x + (size_t)b * 3 * H * W;
#pragma unroll
        for (int j = 0; j < SP_NV; ++j) {
            const int L = tid + 512 * j, row = L >> 6, c = row / SP_IROWS, r = row - c * SP_IROWS;
            const int iy = iy0 + r;
            const bool ok = row < 3 * SP_IROWS && (unsigned)iy < (unsigned)H;
            const float4 v = *reinterpret_cast<const float4 *>(xb + ((size_t)(ok ? c : 0) * H + (ok ? iy : 0)) * W + 4 * (L & 63));
            pv[j] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto store_patch = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < SP_NV; ++j) {
            const int L = tid + 512 * j, row = L >> 6, c = row / SP_IROWS, r = row - c * SP_IROWS;
            if (row < 3 * SP_IROWS) {
                u16 *dst = patch + r * SP_PSI + 3 * (4 * (L & 63) + 3) + c;
                dst[0] = f32_to_bf16(pv[j].x); dst[3] = f32_to_bf16(pv[j].y); dst[6] = f32_to_bf16(pv[j].z); dst[9] = f32_to_bf16(pv[j].w);
            }
        }
    };
    const unsigned *pw = reinterpret_cast<const unsigned *>(patch);

    load_patch(blockIdx.x);
    for (int strip = blockIdx.x; strip < nstrips; strip += (int)gridDim.x) {
        const int b = strip / strips_per_img, i0 = (strip % strips_per_img) * SP_PH;
        __syncthreads();                                   // the previous strip's fragment reads of the patch are done
        store_patch();
        load_patch(strip + (int)gridDim.x);                // in flight during this strip's nine rows
        __syncthreads();
#pragma unroll 1
        for (int rho = 0; rho < SP_ROWS; ++rho) {
            const int crow = 2 * i0 - 1 + rho;             // conv row (-1 for the first strip's first step: computed on padding, never used)
            v4f acc[4];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) acc[nt] = v4f{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 6; ++ks) {
                const int chunk = 4 * ks + q, kh = chunk / 3, jj = chunk - kh * 3;
                const int e = (2 * rho + kh) * SP_PSI + 6 * (16 * wave + l15) + 8 * jj;
                const unsigned *src = pw + e / 2;
                const uint4 fa = make_uint4(src[0], src[1], src[2], src[3]);
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
                    acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(__attribute__((ext_vector_type(8))) __bf16, fb[ks][nt]),
                                                                      __builtin_bit_cast(__attribute__((ext_vector_type(8))) __bf16, fa), acc[nt], 0, 0, 0);
            }
            // lane (l15, q): channels 16 nt + 4 q .. + 3 of pixel 16 wave + l15 -> bf16 -> ring row image; statistics of the rounded values
            // (rows this strip owns: all but its first, which is the previous strip's last)
            unsigned char *rb = reinterpret_cast<unsigned char *>(ring[rho & (SP_RING - 1)]);
            const bool own = rho > 0;
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const int px = 16 * wave + l15, chunk = nt * 2 + (q >> 1);
                const v4f v = acc[nt];
                uint2 o;
                o.x = (unsigned)f32_to_bf16(v[0]) | ((unsigned)f32_to_bf16(v[1]) << 16);
                o.y = (unsigned)f32_to_bf16(v[2]) | ((unsigned)f32_to_bf16(v[3]) << 16);
                *reinterpret_cast<uint2 *>(rb + (size_t)swz(px, chunk) * 16 + (q & 1) * 8) = make_uint2(to_key(o.x ^ sgl[nt][0]), to_key(o.y ^ sgl[nt][1]));
                if (own) {
                    const f2v fa2 = {__uint_as_float(o.x << 16), __uint_as_float(o.x & 0xffff0000u)};
                    const f2v fb2 = {__uint_as_float(o.y << 16), __uint_as_float(o.y & 0xffff0000u)};
                    ss1[nt][0] += fa2; ss1[nt][1] += fb2;
                    ss2[nt][0] = __builtin_elementwise_fma(fa2, fa2, ss2[nt][0]); ss2[nt][1] = __builtin_elementwise_fma(fb2, fb2, ss2[nt][1]);
                }
            }
            __syncthreads();                               // row rho is in the ring (and the slot written next was last read two steps ago)
            if (rho >= 2 && (rho & 1) == 0) {
                // pooled row i0 + rho / 2 - 1 from conv rows rho - 2, rho - 1, rho: columns 2 j - 1, 2 j, 2 j + 1 (column -1 / row -1 = padding: skipped)
                const bool top_ok = crow - 2 >= 0;
                uint4 best = make_uint4(0u, 0u, 0u, 0u);                              // key 0 sorts below every value
#pragma unroll
                for (int dr = 0; dr < 3; ++dr) {
                    if (dr == 0 && !top_ok) continue;
                    const uint4 *rr = ring[(rho - 2 + dr) & (SP_RING - 1)];
#pragma unroll
                    for (int dc = -1; dc <= 1; ++dc) {
                        const int col = 2 * pcol + dc;
                        if (col < 0) continue;
                        const uint4 v = rr[swz(col, pch)];
                        best.x = max_key(best.x, v.x); best.y = max_key(best.y, v.y); best.z = max_key(best.z, v.z); best.w = max_key(best.w, v.w);
                    }
                }
                uint4 o = make_uint4(from_key(best.x) ^ sgn.x, from_key(best.y) ^ sgn.y, from_key(best.z) ^ sgn.z, from_key(best.w) ^ sgn.w);
                const int prow = i0 + rho / 2 - 1;
                *reinterpret_cast<uint4 *>(pooled + (((size_t)b * SP_PW + prow) * SP_PW + pcol) * 64 + pch * 8) = o;
            }
        }
    }
    // batch statistics: fold the 16 pixel lanes of every (q, nt) group, then the 8 waves, one atomic per channel and workgroup
    if (stats) {
        float *red = reinterpret_cast<float *>(ring);          // [wave][2][64]
        __syncthreads();
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float a = ss1[nt][c >> 1][c & 1], s2 = ss2[nt][c >> 1][c & 1];
#pragma unroll
                for (int o = 8; o > 0; o >>= 1) { a += __shfl_xor(a, o, 64); s2 += __shfl_xor(s2, o, 64); }
                if (l15 == 0) { red[(wave * 2) * 64 + nt * 16 + 4 * q + c] = a; red[(wave * 2 + 1) * 64 + nt * 16 + 4 * q + c] = s2; }
            }
        __syncthreads();
        if (tid < 128) {
            const int which = tid >> 6, ch = tid & 63;
            float a = 0.f;
#pragma unroll
            for (int w2 = 0; w2 < 8; ++w2) a += red[(w2 * 2 + which) * 64 + ch];
            fx::add(stats, (int)blockIdx.x % NSH, which, 64, ch, a);
        }
    }
}

}}  // namespace mhe::conv

using namespace mhe;

extern "C" int mhe_stem_pool_supported(int B, int H, int W, int dtype) { return dtype == MHE_BF16 && B > 0 && H == 256 && W == 256; }

extern "C" int mhe_stem_conv7x7s2_pool(const float *x_nchw, const void *w, const float *bn_gamma, void *pooled, mhe_stat_t *stats, int B, int H, int W,
                                       void *stream) {
    MHE_REQUIRE(x_nchw && w && bn_gamma && pooled, "mhe_stem_conv7x7s2_pool: null pointer");
    MHE_REQUIRE(mhe_stem_pool_supported(B, H, W, MHE_BF16), "mhe_stem_conv7x7s2_pool: bf16, 256 x 256 images (B=%d H=%d W=%d)", B, H, W);
    const int nstrips = B * (conv::SP_PW / conv::SP_PH);
    const dim3 grid((unsigned)(nstrips < 256 ? nstrips : 256));
    hipLaunchKernelGGL(conv::stem_pool_kernel, grid, dim3(512), 0, (hipStream_t)stream, x_nchw, (const u16 *)w, bn_gamma, (u16 *)pooled, stats, B, H, W);
    return check_launch("stem_pool_kernel");
}

// Residual-block tail + next conv1 for the wide layers (variant 10): y = relu(bn3(x) + identity) W^T with 1024 / 2048 (or 512) input channels
// and 256 / 512 output channels - layer3 / layer4 of ResNet-50 and the transitions into them, reference hand/network.py (torchvision
// Bottleneck.forward: `out += identity; out = relu(out)` followed by the next block's conv1).
//
// Why its own kernel.  On these shapes the 128x128 register-staged kernel (conv.hip, MODE 2) is bound by L2 -> CU traffic, not by HBM:
// every 128-pixel tile re-reads its weight slab and every one of the 2 - 4 column tiles re-reads (and re-evaluates) both activation
// tensors - 940 MB through L2 for 436 MB of compulsory traffic at layer3, 8.2 TB/s (profiles/EXPERIMENTS.md).  A 128 x 256 tile halves /
// quarters the activation re-reads; the accumulators of such a tile (128 VGPRs per wave over four waves) leave no room for a second workgroup
// on the CU to cover the load -> transform -> LDS latency, so that work moves to waves of its own, as in conv_stream.hip's 3x3 kernel:
//   multiply waves 0-3 : 64 MFMAs per 64-deep K tile (128 pixels x 64 of the 256 channels each), fragments from the stage of this tick;
//   transfer waves 4-7 : the K tile three ticks ahead from global memory into registers (two tiles in flight) (both activation tensors + the 256 weight rows), the
//                        tail arithmetic, the evaluated tile into the other LDS stage and - column tile 0 only - out to a_out.
// Workgroups are persistent (one per CU): the transfer waves load the next pixel tile's first K tiles while this tile's epilogue runs.
// One barrier per K tile; two LDS stages of 48 KiB; the epilogue stages the 128 x 256 outputs through the same LDS, stores them with
// 16-byte lanes and takes the batch statistics of the stored (bf16-rounded) values in the store loop.
#include "conv_shared.h"

namespace mhe { namespace conv {

namespace {

constexpr int TBM = 128, TBN = 256, TBK = 64, TMAXK = 2048;
constexpr int T_A = TBM * 8, T_W = TBN * 8, T_STAGE = T_A + T_W;        // uint4 per stage: 16 KiB activations + 32 KiB weights

}  // namespace

// DG: data-gradient form - the operand is gy = k2 g + k1 y + k0 (the BatchNorm reverse of the convolution's own output gradient, written out
// once for the weight gradient); the outputs are ReLU-gated by p.mask (+ residual) and feed the BatchNorm-reverse sums of up to two consumers
template <bool DG>
__global__ __launch_bounds__(512) void conv_tail_kernel(const Params p) {
    using T = u16;
    __shared__ uint4 lds[2 * T_STAGE];                                   // 96 KiB; the epilogue's 64 KiB staging buffer lies over it
    __shared__ float aff[4 * TMAXK];                                      // scale | shift of x, scale | shift of x2 (1 | 0 without its affine)   32 KiB
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, q = lane >> 4, l15 = lane & 15;
    const bool mult = wave < 4;
    const int t2 = tid & 255, s = t2 & 7, rbase = t2 >> 3;
    const int K = p.Cin, nk = K / TBK;
    const int gm = p.M / TBM, gn = p.Cout / TBN, ntiles = gm * gn;
    // persistent workgroups, XCD-aware order (conv_shared.h tile_of_block): the column tiles of one pixel tile run on one XCD at the same time
    auto tile_at = [&](int L, int &mt, int &nt) __attribute__((always_inline)) {
        if (gn > 1 && (gm & 7) == 0) {
            const int slot = L >> 3;
            nt = slot % gn;
            mt = (slot / gn) * 8 + (L & 7);
        } else { mt = L % gm; nt = L / gm; }
    };
    const bool aff2 = p.x2_scale != nullptr;
    for (int i = tid; i < K; i += 512) {
        aff[i] = p.in_scale[i]; aff[K + i] = p.in_shift[i];
        aff[2 * K + i] = aff2 ? p.x2_scale[i] : 1.f; aff[3 * K + i] = aff2 ? p.x2_shift[i] : 0.f;
    }
    const T *xg = reinterpret_cast<const T *>(p.x), *x2g = reinterpret_cast<const T *>(p.x2), *wg = reinterpret_cast<const T *>(p.w);
    T *yg = reinterpret_cast<T *>(p.y);
    // transfer role.  This thread: 16-byte chunk s of the K tile, activation rows rbase + 32 j (j < 4), weight rows rbase + 32 j (j < 8).
    // Two K tiles in flight in registers: a tile is loaded two ticks before it is written to LDS (with one tile in flight a tick
    // lasted one memory latency, 4.6 us for 64 KiB per CU: the first version ran 148 us against the tiled kernel's 119; three sets spill)
    struct Set { uint4 a[4], b[4], w[8]; };
    Set st0, st1;
    auto load_tile = [&](int t, Set &st, int m0, int n0) __attribute__((always_inline)) {
        const int kc = t * TBK + s * 8;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const size_t off = (size_t)(m0 + rbase + 32 * j) * K + kc;
            st.a[j] = *reinterpret_cast<const uint4 *>(xg + off);
            st.b[j] = *reinterpret_cast<const uint4 *>(x2g + off);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) st.w[j] = *reinterpret_cast<const uint4 *>(wg + (size_t)(n0 + rbase + 32 * j) * p.Kpad + kc);
    };
    auto store_tile = [&](int t, const Set &st, int m0, T *ag) __attribute__((always_inline)) {
        uint4 *At = lds + (t & 1) * T_STAGE, *Wt = At + T_A;
        const int kc = t * TBK + s * 8;
        float sc[8], sh[8], s2[8], h2[8];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const float4 a0 = *reinterpret_cast<const float4 *>(aff + kc + 4 * h), a1 = *reinterpret_cast<const float4 *>(aff + K + kc + 4 * h);
            const float4 b0 = *reinterpret_cast<const float4 *>(aff + 2 * K + kc + 4 * h), b1 = *reinterpret_cast<const float4 *>(aff + 3 * K + kc + 4 * h);
            sc[4 * h] = a0.x; sc[4 * h + 1] = a0.y; sc[4 * h + 2] = a0.z; sc[4 * h + 3] = a0.w;
            sh[4 * h] = a1.x; sh[4 * h + 1] = a1.y; sh[4 * h + 2] = a1.z; sh[4 * h + 3] = a1.w;
            s2[4 * h] = b0.x; s2[4 * h + 1] = b0.y; s2[4 * h + 2] = b0.z; s2[4 * h + 3] = b0.w;
            h2[4 * h] = b1.x; h2[4 * h + 1] = b1.y; h2[4 * h + 2] = b1.z; h2[4 * h + 3] = b1.w;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            // the arithmetic of conv_shared.h's in_transform, in its order: x * scale + shift, + (x2 * scale2 + shift2 | x2), relu, round
            float v[8], w[8];
            Chunk<T>::unpack(st.a[j], v);
            Chunk<T>::unpack(st.b[j], w);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                v[i] = fmaf(v[i], sc[i], sh[i]);
                if (aff2) w[i] = fmaf(w[i], s2[i], h2[i]);
                v[i] += w[i];
                if (p.relu_in) v[i] = fmaxf(v[i], 0.f);
            }
            const uint4 o = Chunk<T>::pack(v);
            At[swz(rbase + 32 * j, s)] = o;
            if (ag) *reinterpret_cast<uint4 *>(ag + (size_t)(m0 + rbase + 32 * j) * K + kc) = o;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) Wt[swz(rbase + 32 * j, s)] = st.w[j];
    };
    const bool st_on = p.stats != nullptr;
    // the epilogue both roles run: 128 pixels x 32 chunks of 16 B from the staging buffer, thread = chunk tid % 32 of rows tid / 32 + 16 j
    auto store_outputs = [&](int mt, int m0, int n0) __attribute__((always_inline)) {
        const unsigned char *ot = reinterpret_cast<const unsigned char *>(lds);
        const int cc = tid & 31, r0 = tid >> 5;
        float *red = reinterpret_cast<float *>(lds);
        // fold 16 threads' partial sums per column chunk and add them to this pixel tile's shard of dst
        auto fold = [&](const float *a8, const float *b8, fx::acc_t *dst) __attribute__((always_inline)) {
            __syncthreads();
#pragma unroll
            for (int i = 0; i < 8; ++i) { red[tid * 16 + i] = a8[i]; red[tid * 16 + 8 + i] = b8[i]; }
            __syncthreads();
            if (tid < 256) {
                const int ch = tid >> 3, e = tid & 7;
                float a = 0.f, b = 0.f;
                for (int k = 0; k < 16; ++k) { a += red[(ch + 32 * k) * 16 + e]; b += red[(ch + 32 * k) * 16 + 8 + e]; }
                fx::add(dst, mt % NSH, 0, p.Cout, n0 + tid, a);
                fx::add(dst, mt % NSH, 1, p.Cout, n0 + tid, b);
            }
        };
        if constexpr (DG) {
            const T *mk = reinterpret_cast<const T *>(p.mask), *rg = reinterpret_cast<const T *>(p.residual);
            const T *y0g = reinterpret_cast<const T *>(p.bn_y[0]), *y1g = reinterpret_cast<const T *>(p.bn_y[1]);
            const int ccol = n0 + cc * 8;
            float bs1[2][8], bs2[2][8], bmu[2][8], biv[2][8];
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const bool ok = p.bn_y[u] != nullptr;
                    bs1[u][i] = bs2[u][i] = 0.f;
                    bmu[u][i] = ok ? p.bn_mi[u][ccol + i] : 0.f;
                    biv[u][i] = ok ? p.bn_mi[u][p.Cout + ccol + i] : 0.f;
                }
#pragma unroll 1
            for (int j0 = 0; j0 < 8; j0 += 4) {                          // four rows at a time: their operand loads in flight together
                uint4 raw[4], gm[4], rr[4], ya[4], yb[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int row = r0 + 16 * (j0 + j);
                    const size_t off = (size_t)(m0 + row) * p.Cout + ccol;
                    gm[j] = *reinterpret_cast<const uint4 *>(mk + off);
                    rr[j] = rg ? *reinterpret_cast<const uint4 *>(rg + off) : make_uint4(0, 0, 0, 0);
                    if (y0g) ya[j] = *reinterpret_cast<const uint4 *>(y0g + off);
                    if (y1g) yb[j] = *reinterpret_cast<const uint4 *>(y1g + off);
                    raw[j] = *reinterpret_cast<const uint4 *>(ot + ((size_t)row * 32 + (cc ^ (row & 31))) * 16);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int row = r0 + 16 * (j0 + j);
                    float v[8], t[8];
                    Chunk<T>::unpack(raw[j], v);
                    if (rg) {
                        Chunk<T>::unpack(rr[j], t);
#pragma unroll
                        for (int i = 0; i < 8; ++i) v[i] += t[i];
                    }
                    Chunk<T>::unpack(gm[j], t);
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[i] = t[i] > 0.f ? v[i] : 0.f;
                    if (y0g) {
                        Chunk<T>::unpack(ya[j], t);
#pragma unroll
                        for (int i = 0; i < 8; ++i) { bs1[0][i] += v[i]; bs2[0][i] = fmaf(v[i], (t[i] - bmu[0][i]) * biv[0][i], bs2[0][i]); }
                    }
                    if (y1g) {
                        Chunk<T>::unpack(yb[j], t);
#pragma unroll
                        for (int i = 0; i < 8; ++i) { bs1[1][i] += v[i]; bs2[1][i] = fmaf(v[i], (t[i] - bmu[1][i]) * biv[1][i], bs2[1][i]); }
                    }
                    *reinterpret_cast<uint4 *>(yg + (size_t)(m0 + row) * p.Cout + ccol) = Chunk<T>::pack(v);
                }
            }
#pragma unroll
            for (int u = 0; u < 2; ++u)
                if (p.bn_y[u]) fold(bs1[u], bs2[u], p.bn_stats[u]);
        } else {
            float ss1[8], ss2[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) ss1[i] = ss2[i] = 0.f;
            uint4 raw[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int row = r0 + 16 * j;
                raw[j] = *reinterpret_cast<const uint4 *>(ot + ((size_t)row * 32 + (cc ^ (row & 31))) * 16);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int row = r0 + 16 * j;
                if (st_on) {
                    float f[8];
                    Chunk<T>::unpack(raw[j], f);
#pragma unroll
                    for (int i = 0; i < 8; ++i) { ss1[i] += f[i]; ss2[i] = fmaf(f[i], f[i], ss2[i]); }
                }
                *reinterpret_cast<uint4 *>(yg + (size_t)(m0 + row) * p.Cout + n0 + cc * 8) = raw[j];
            }
            if (st_on) fold(ss1, ss2, p.stats);
        }
    };
    // Each role has its own loop over the workgroup's tiles (the transfer waves' register sets must not be live in the multiply waves'
    // code); both take the same barriers: one at the top of a tile, one per K tile, one when the stages are free, one when the outputs
    // are staged, two in the statistics fold.
    if (mult) {
        for (int L = blockIdx.x; L < ntiles; L += (int)gridDim.x) {
            int mt, ntile;
            tile_at(L, mt, ntile);
            __syncthreads();                                              // affine tables written; the previous tile's staging / fold reads done
            v4f acc[4][8];                                                // [channel tile of this wave's 64][pixel tile of the 128]
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 8; ++b) acc[a][b] = v4f{0.f, 0.f, 0.f, 0.f};
            for (int t = 0; t < nk; ++t) {
                __syncthreads();
                const uint4 *At = lds + (t & 1) * T_STAGE, *Wt = At + T_A + wave * 64 * 8;
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    uint4 fa[8], fb[4];
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) fb[nt] = Wt[swz(nt * 16 + l15, kk * 4 + q)];
#pragma unroll
                    for (int m = 0; m < 8; ++m) fa[m] = At[swz(m * 16 + l15, kk * 4 + q)];
#pragma unroll
                    for (int m = 0; m < 8; ++m)
#pragma unroll
                        for (int nt = 0; nt < 4; ++nt)
                            acc[nt][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                                __builtin_bit_cast(__attribute__((ext_vector_type(8))) __bf16, fb[nt]),
                                __builtin_bit_cast(__attribute__((ext_vector_type(8))) __bf16, fa[m]), acc[nt][m], 0, 0, 0);
                }
            }
            __syncthreads();                                              // every wave is done with the stages: they become the staging buffer
            unsigned char *ot = reinterpret_cast<unsigned char *>(lds);
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int m = 0; m < 8; ++m) {
                    // D layout: lane (l15, q) holds channels 4q..4q+3 of tile nt for pixel 16m + l15; staging row = pixel, 32 chunks of 16 B
                    const int row = m * 16 + l15, boff = (wave * 64 + nt * 16 + 4 * q) * 2;
                    const int chunk = (boff >> 4) ^ (row & 31);
                    const v4f v = acc[nt][m];
                    uint2 o;
                    o.x = (unsigned)f32_to_bf16(v[0]) | ((unsigned)f32_to_bf16(v[1]) << 16);
                    o.y = (unsigned)f32_to_bf16(v[2]) | ((unsigned)f32_to_bf16(v[3]) << 16);
                    *reinterpret_cast<uint2 *>(ot + ((size_t)row * 32 + chunk) * 16 + (boff & 15)) = o;
                }
            __syncthreads();                                              // the staged outputs are visible
            store_outputs(mt, mt * TBM, ntile * TBN);
        }
    } else {
        int mt, ntile;
        tile_at(blockIdx.x, mt, ntile);
        load_tile(0, st0, mt * TBM, ntile * TBN);                         // the first tile's first two K tiles: the one exposed load of the workgroup
        if (1 < nk) load_tile(1, st1, mt * TBM, ntile * TBN);
        for (int L = blockIdx.x; L < ntiles; L += (int)gridDim.x) {
            tile_at(L, mt, ntile);
            const int m0 = mt * TBM, n0 = ntile * TBN;
            T *ag = p.a_out && ntile == 0 ? reinterpret_cast<T *>(p.a_out) : nullptr;
            __syncthreads();
            // K tile u lives in set u & 1; tiles 0 and 1 are in flight (loaded during the previous pixel tile's epilogue)
            store_tile(0, st0, m0, ag);
            if (2 < nk) load_tile(2, st0, m0, n0);
            // tick t: write tile t + 1 (set (t + 1) & 1) into the stage the multiply waves are not reading, then reload that set with tile t + 3
            auto tick = [&](int t, Set &st) __attribute__((always_inline)) {
                __syncthreads();                                          // stage (t + 1) & 1 was last read during tick t - 1
                if (t + 1 < nk) store_tile(t + 1, st, m0, ag);            // first everything that consumes loaded registers ...
                if (t + 3 < nk) load_tile(t + 3, st, m0, n0);             // ... then this tick's loads
            };
            for (int t = 0; t < nk; t += 2) {
                tick(t, st1);
                if (t + 1 < nk) tick(t + 1, st0);
            }
            __syncthreads();
            if constexpr (!DG) {
                if (L + (int)gridDim.x < ntiles) {                        // the next pixel tile's first K tiles, in flight across this tile's epilogue
                    int mt2, nt2;
                    tile_at(L + (int)gridDim.x, mt2, nt2);
                    load_tile(0, st0, mt2 * TBM, nt2 * TBN);
                    if (1 < nk) load_tile(1, st1, mt2 * TBM, nt2 * TBN);
                }
            }
            __syncthreads();
            store_outputs(mt, m0, n0);
            if constexpr (DG) {                                           // (the epilogue's operand registers and the two sets do not fit together)
                if (L + (int)gridDim.x < ntiles) {
                    int mt2, nt2;
                    tile_at(L + (int)gridDim.x, mt2, nt2);
                    load_tile(0, st0, mt2 * TBM, nt2 * TBN);
                    if (1 < nk) load_tile(1, st1, mt2 * TBM, nt2 * TBN);
                }
            }
        }
    }
}


// ---------------------------------------------------------------------------------------------------------------------------------
// Variant 15 (round 4): the same tile and arithmetic with every operand brought in by LDS-DMA rings instead of register sets.
//
// Why.  conv_tail_kernel keeps two 64-deep K tiles in flight in the transfer waves' REGISTERS (a third set spills next to the multiply
// role's 255 VGPRs, and hipcc's wait counts drain loop-carried sets: profiles/EXPERIMENTS.md, two negative results): one 64 KiB set in
// flight per CU, a tick = one loaded memory latency (2.7 us), 3.7 TB/s of compulsory bytes.  `buffer_load ... lds` needs no registers, so
// the bytes in flight are bounded by LDS only - and counted `vmcnt` waits written in asm are the ones that execute:
//   * K is walked in 32-deep ticks; per tick a W stage (256 rows x 64 B = 16 KiB) and a raw stage (x and x2: 128 rows x 64 B each = 16 KiB);
//   * two rings of FOUR stages: three ticks ahead in flight = 96 KiB per CU (the guide's gather measurements hide an HBM miss with 72);
//   * waves 0-3 multiply (32 MFMAs per tick) and issue the W ring's DMA (4 pieces each per tick, `vmcnt(8)`: two stages stay in flight);
//     they also store the tile's outputs (+ batch statistics) - their own VMEM queue is empty then;
//   * waves 4-5 issue the raw ring's DMA and nothing else (8 pieces each per tick, `vmcnt(16)`);
//   * waves 6-7 evaluate relu(bn3(x) + identity) LDS -> LDS (raw stage g + 1 -> A stage (g + 1) & 1, the layout the fragments are read
//     from) and write the block output a_out on the way (column tile 0 only);
//   * the W image is written already bank-swizzled (the XOR moves into the DMA's per-lane SOURCE offset); 64-byte rows take the swizzle
//     slot ^ (-(row >> 2) & 3), conflict-free for the four 16-lane groups of ds_read_b128 (MI355X_MICROARCH.md, LDS table);
//   * the f32 affine tables stay resident (16 KiB: K <= 2048 without an identity affine, K <= 1024 with one);
//   * one barrier per tick; the epilogue stages the outputs over the W ring only, so the raw ring keeps prefetching the next pixel tile.
// Same products in the same order as conv_tail_kernel / MODE 2 of conv.hip: results equal bit for bit.
namespace {
typedef unsigned u4t __attribute__((ext_vector_type(4)));
constexpr int T2K = 32, T2_WS = TBN * 64, T2_RS = 2 * TBM * 64, T2_AS = TBM * 64;      // bytes: W stage, raw stage (x | x2), A stage
constexpr int T2_RING = 4, T2_AFF = 4096;
constexpr unsigned T2_OOB = 0x80000000u;
__device__ __forceinline__ int swz64(int row, int slot) { return row * 4 + (slot ^ ((0 - (row >> 2)) & 3)); }
// one LDS-DMA piece: 64 lanes x 16 B from (resource, per-lane byte offset) to LDS bytes [lds_base, lds_base + 1024)
__device__ __forceinline__ void dma_piece(u4t rs, unsigned off, unsigned lds_base) {
    unsigned keep;
    asm volatile("s_mov_b32 %[keep], m0\n\ts_mov_b32 m0, %[lb]\n\ts_nop 0\n\tbuffer_load_dwordx4 %[v0], %[rs], 0 offen lds\n\ts_mov_b32 m0, %[keep]"
                 : [keep] "=&s"(keep) : [v0] "v"(off), [rs] "s"(rs), [lb] "s"(lds_base) : "memory");
}
}  // namespace

__global__ __launch_bounds__(512) void conv_tail2_kernel(const Params p) {
    using T = u16;
    __shared__ uint4 lds[(T2_RING * T2_WS + T2_RING * T2_RS + 2 * T2_AS) / 16];            // 64 + 64 + 16 KiB
    __shared__ float aff[T2_AFF];                                                          // scale | shift (| scale2 | shift2)      16 KiB
    unsigned char *const wr = reinterpret_cast<unsigned char *>(lds), *const rr = wr + T2_RING * T2_WS, *const as = rr + T2_RING * T2_RS;
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, q = lane >> 4, l15 = lane & 15;
    const int K = p.Cin, nk = K / T2K;
    const int gm = p.M / TBM, gn = p.Cout / TBN, ntiles = gm * gn, grid = (int)gridDim.x;
    auto tile_at = [&](int L, int &mt, int &nt) __attribute__((always_inline)) {
        if (gn > 1 && (gm & 7) == 0) {
            const int slot = L >> 3;
            nt = slot % gn;
            mt = (slot / gn) * 8 + (L & 7);
        } else { mt = L % gm; nt = L / gm; }
    };
    const bool aff2 = p.x2_scale != nullptr;
    for (int i = tid; i < K; i += 512) {
        aff[i] = p.in_scale[i]; aff[K + i] = p.in_shift[i];
        if (aff2) { aff[2 * K + i] = p.x2_scale[i]; aff[3 * K + i] = p.x2_shift[i]; }
    }
    const int nloc = (ntiles - (int)blockIdx.x + grid - 1) / grid;                         // this workgroup's tiles: blockIdx.x + j * grid
    const int G = nloc * nk;                                                               // ... and its ticks, numbered through
    const unsigned wr_lds = (unsigned)(size_t)((__attribute__((address_space(3))) unsigned char *)wr);
    const unsigned rr_lds = wr_lds + T2_RING * T2_WS;
    __syncthreads();                                                                       // tables written

    if (wave < 4) {
        // ------------------------------------------------------------------ multiply role (+ W ring, + the output epilogue)
        const size_t wbytes = (size_t)p.Cout * p.Kpad * 2;
        const u4t rsw = {(unsigned)(size_t)p.w, (unsigned)((size_t)p.w >> 32) & 0xffffu, (unsigned)wbytes, 0x00020000u};
        unsigned woff[4];                                                                  // this lane's four pieces of a W stage
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int I = (wave * 4 + i) * 64 + lane, row = I >> 2, sp = I & 3, slot = sp ^ ((0 - (row >> 2)) & 3);
            woff[i] = (unsigned)(row * p.Kpad + slot * 8) * 2u;
        }
        auto issue_w = [&](bool ok, unsigned ub, int g) __attribute__((always_inline)) {    // W stage of tick g (ub: the tile's / K tile's byte offset)
            const unsigned lb = wr_lds + (unsigned)((g & 3) * T2_WS + wave * 4096);
#pragma unroll
            for (int i = 0; i < 4; ++i) dma_piece(rsw, ok ? woff[i] + ub : T2_OOB, lb + i * 1024);
        };
        const int f = (0 - (l15 >> 2)) & 3;
        const int offA = (l15 * 4 + (q ^ f)) * 16, offW = offA + wave * 4096;              // + m * 1024 / + nt * 1024
        const bool st_on = p.stats != nullptr;
        const int cc = tid & 31, r0 = tid >> 5;                                            // epilogue: 16-byte chunk cc of rows r0 + 8 j
        float ss1[8], ss2[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) ss1[i] = ss2[i] = 0.f;
        int n0 = 0;
        __builtin_amdgcn_s_barrier();                                                      // (A: pairs with the raw ring's first stage)
        for (int j = 0; j < nloc; ++j) {
            int mt, ntile;
            tile_at((int)blockIdx.x + j * grid, mt, ntile);
            const int m0 = mt * TBM;
            n0 = ntile * TBN;
            const unsigned ubw = (unsigned)(n0 * p.Kpad) * 2u;
            issue_w(true, ubw, j * nk); issue_w(true, ubw + 64u, j * nk + 1); issue_w(true, ubw + 128u, j * nk + 2);
            v4f acc[4][8];                                                                 // [channel tile of this wave's 64][pixel tile of the 128]
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 8; ++b) acc[a][b] = v4f{0.f, 0.f, 0.f, 0.f};
            for (int t = 0; t < nk; ++t) {
                const int g = j * nk + t;
                __builtin_amdgcn_sched_barrier(0);                                         // (tick g - 1's MFMAs - and the waits of their fragment reads - stay above)
                asm volatile("s_waitcnt vmcnt(8)" ::: "memory");                           // this wave's share of W(g); W(g + 1), W(g + 2) stay in flight
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                issue_w(t + 3 < nk, ubw + (unsigned)(t + 3) * 64u, g + 3);                 // into the stage read during tick g - 1 (past the tile: zero fill)
                const unsigned char *At = as + (g & 1) * T2_AS + offA, *Wt = wr + (g & 3) * T2_WS + offW;
                uint4 fa[8], fb[4];
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) fb[nt] = *reinterpret_cast<const uint4 *>(Wt + nt * 1024);
#pragma unroll
                for (int m = 0; m < 8; ++m) fa[m] = *reinterpret_cast<const uint4 *>(At + m * 1024);
#pragma unroll
                for (int m = 0; m < 8; ++m)
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt)
                        acc[nt][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                            __builtin_bit_cast(__attribute__((ext_vector_type(8))) __bf16, fb[nt]),
                            __builtin_bit_cast(__attribute__((ext_vector_type(8))) __bf16, fa[m]), acc[nt][m], 0, 0, 0);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                               // the past-the-end zero fills must not land on the staged outputs
            __builtin_amdgcn_s_barrier();                                                  // E1: every wave is done with the W ring: it becomes the staging buffer
            unsigned char *ot = wr;
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int m = 0; m < 8; ++m) {
                    // D layout: lane (l15, q) holds channels 4q..4q+3 of tile nt for pixel 16m + l15; staging row = pixel, 32 chunks of 16 B
                    const int row = m * 16 + l15, boff = (wave * 64 + nt * 16 + 4 * q) * 2;
                    const int chunk = (boff >> 4) ^ (row & 31);
                    const v4f v = acc[nt][m];
                    uint2 o;
                    o.x = (unsigned)f32_to_bf16(v[0]) | ((unsigned)f32_to_bf16(v[1]) << 16);
                    o.y = (unsigned)f32_to_bf16(v[2]) | ((unsigned)f32_to_bf16(v[3]) << 16);
                    *reinterpret_cast<uint2 *>(ot + ((size_t)row * 32 + chunk) * 16 + (boff & 15)) = o;
                }
            __syncthreads();                                                               // E2: the staged outputs are visible
            T *yg = reinterpret_cast<T *>(p.y);
#pragma unroll
            for (int j0 = 0; j0 < 16; j0 += 8) {
                uint4 raw[8];
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) {
                    const int row = r0 + 8 * (j0 + jj);
                    raw[jj] = *reinterpret_cast<const uint4 *>(ot + ((size_t)row * 32 + (cc ^ (row & 31))) * 16);
                }
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) {
                    const int row = r0 + 8 * (j0 + jj);
                    if (st_on) {
                        float fv[8];
                        Chunk<T>::unpack(raw[jj], fv);
#pragma unroll
                        for (int i = 0; i < 8; ++i) { ss1[i] += fv[i]; ss2[i] = fmaf(fv[i], fv[i], ss2[i]); }
                    }
                    *reinterpret_cast<uint4 *>(yg + (size_t)(m0 + row) * p.Cout + n0 + cc * 8) = raw[jj];
                }
            }
            __syncthreads();                                                               // E3: the staging reads are done: the ring is free again
        }
        if (st_on) {                                                                       // (every tile of a workgroup has the same column tile: supports())
            float *red = reinterpret_cast<float *>(wr);
#pragma unroll
            for (int i = 0; i < 8; ++i) { red[tid * 16 + i] = ss1[i]; red[tid * 16 + 8 + i] = ss2[i]; }
            __syncthreads();
            const int ch = tid >> 3, e = tid & 7;
            float a = 0.f, b = 0.f;
            for (int k = 0; k < 8; ++k) { a += red[(ch + 32 * k) * 16 + e]; b += red[(ch + 32 * k) * 16 + 8 + e]; }
            fx::add(p.stats, (int)(blockIdx.x % NSH), 0, p.Cout, n0 + tid, a);
            fx::add(p.stats, (int)(blockIdx.x % NSH), 1, p.Cout, n0 + tid, b);
        }
    } else if (wave < 6) {
        // ------------------------------------------------------------------ raw ring: x and x2 (identity) K tiles by DMA, nothing else
        const int w2 = wave - 4;
        const size_t abytes = (size_t)p.M * K * 2;
        const u4t rsx = {(unsigned)(size_t)p.x, (unsigned)((size_t)p.x >> 32) & 0xffffu, (unsigned)abytes, 0x00020000u};
        const u4t rsy = {(unsigned)(size_t)p.x2, (unsigned)((size_t)p.x2 >> 32) & 0xffffu, (unsigned)abytes, 0x00020000u};
        unsigned aoff[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int I = (w2 * 4 + i) * 64 + lane, row = I >> 2, slot = I & 3;            // (linear image: the transform reads whole 16-byte pieces)
            aoff[i] = (unsigned)(row * K + slot * 8) * 2u;
        }
        auto issue_raw = [&](int g) __attribute__((always_inline)) {
            unsigned ub = T2_OOB;
            if (g < G) {
                int mt, ntile;
                tile_at((int)blockIdx.x + (g / nk) * grid, mt, ntile);
                ub = ((unsigned)(mt * TBM) * (unsigned)K + (unsigned)(g % nk) * T2K) * 2u;
            }
            const unsigned lb = rr_lds + (unsigned)((g & 3) * T2_RS + w2 * 4096);
#pragma unroll
            for (int i = 0; i < 4; ++i) dma_piece(rsx, g < G ? aoff[i] + ub : T2_OOB, lb + i * 1024);
#pragma unroll
            for (int i = 0; i < 4; ++i) dma_piece(rsy, g < G ? aoff[i] + ub : T2_OOB, lb + T2_RS / 2 + i * 1024);
        };
        issue_raw(0); issue_raw(1); issue_raw(2); issue_raw(3);
        asm volatile("s_waitcnt vmcnt(24)" ::: "memory");                                  // stage 0 has landed
        __builtin_amdgcn_s_barrier();                                                      // A
        for (int g = 0; g < G; ++g) {
            asm volatile("s_waitcnt vmcnt(16)" ::: "memory");                              // stage g + 1 (transformed during this tick); g + 2, g + 3 in flight
            __builtin_amdgcn_s_barrier();
            issue_raw(g + 4);                                                              // into the stage transformed during tick g - 1
            if (g % nk == nk - 1) { __builtin_amdgcn_s_barrier(); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_s_barrier(); }      // E1 - E3
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                   // (the past-the-end zero fills land before the workgroup's LDS is handed on)
        if (p.stats) __builtin_amdgcn_s_barrier();
    } else {
        // ------------------------------------------------------------------ transform: raw stage -> relu(x * scale + shift + identity) -> A stage (+ a_out)
        const int u = tid - 384, slot = u & 3, r = u >> 2;                                 // 16-byte piece `slot` of rows r + 32 j
        const size_t abytes = (size_t)p.M * K * 2;
        const __amdgpu_buffer_rsrc_t ag = __builtin_amdgcn_make_buffer_rsrc(p.a_out ? p.a_out : const_cast<void *>(p.x), 0, p.a_out ? (int)abytes : 0, 0x00020000);
        auto xform = [&](int g) __attribute__((always_inline)) {
            if (g >= G) return;
            int mt, ntile;
            tile_at((int)blockIdx.x + (g / nk) * grid, mt, ntile);
            const int t = g % nk, kc = t * T2K + slot * 8;
            const unsigned char *rx = rr + (g & 3) * T2_RS, *ry = rx + T2_RS / 2;
            unsigned char *At = as + (g & 1) * T2_AS;
            float sc[8], sh[8], s2[8], h2[8];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const float4 a0 = *reinterpret_cast<const float4 *>(aff + kc + 4 * h), a1 = *reinterpret_cast<const float4 *>(aff + K + kc + 4 * h);
                sc[4 * h] = a0.x; sc[4 * h + 1] = a0.y; sc[4 * h + 2] = a0.z; sc[4 * h + 3] = a0.w;
                sh[4 * h] = a1.x; sh[4 * h + 1] = a1.y; sh[4 * h + 2] = a1.z; sh[4 * h + 3] = a1.w;
                if (aff2) {
                    const float4 b0 = *reinterpret_cast<const float4 *>(aff + 2 * K + kc + 4 * h), b1 = *reinterpret_cast<const float4 *>(aff + 3 * K + kc + 4 * h);
                    s2[4 * h] = b0.x; s2[4 * h + 1] = b0.y; s2[4 * h + 2] = b0.z; s2[4 * h + 3] = b0.w;
                    h2[4 * h] = b1.x; h2[4 * h + 1] = b1.y; h2[4 * h + 2] = b1.z; h2[4 * h + 3] = b1.w;
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i) { s2[4 * h + i] = 1.f; h2[4 * h + i] = 0.f; }
                }
            }
            const bool write_a = p.a_out != nullptr && ntile == 0;
            uint4 xa[4], xb[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                xa[j] = *reinterpret_cast<const uint4 *>(rx + ((r + 32 * j) * 4 + slot) * 16);
                xb[j] = *reinterpret_cast<const uint4 *>(ry + ((r + 32 * j) * 4 + slot) * 16);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                // the arithmetic of conv_shared.h's in_transform, in its order: x * scale + shift, + (x2 * scale2 + shift2 | x2), relu, round
                const int row = r + 32 * j;
                float v[8], w[8];
                Chunk<T>::unpack(xa[j], v);
                Chunk<T>::unpack(xb[j], w);
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    v[i] = fmaf(v[i], sc[i], sh[i]);
                    if (aff2) w[i] = fmaf(w[i], s2[i], h2[i]);
                    v[i] += w[i];
                    if (p.relu_in) v[i] = fmaxf(v[i], 0.f);
                }
                const uint4 o = Chunk<T>::pack(v);
                *reinterpret_cast<uint4 *>(At + swz64(row, slot) * 16) = o;
                const u4t ov = {o.x, o.y, o.z, o.w};
                // (not this column tile's to write: an out-of-range offset, the store is dropped - no branch around a memory operation)
                __builtin_amdgcn_raw_buffer_store_b128(ov, ag, write_a ? (int)(((unsigned)(mt * TBM + row) * (unsigned)K + (unsigned)kc) * 2u) : (int)T2_OOB, 0, 0);
            }
        };
        __builtin_amdgcn_s_barrier();                                                      // A: raw stage 0 has landed
        xform(0);
        for (int g = 0; g < G; ++g) {
            __syncthreads();                                                               // (its fence: this wave's A-stage writes have reached LDS)
            xform(g + 1);
            if (g % nk == nk - 1) { __builtin_amdgcn_s_barrier(); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_s_barrier(); }      // E1 - E3
        }
        if (p.stats) __builtin_amdgcn_s_barrier();
    }
}

bool tail_supports(const Params &p);
// the DMA-ring form: forward form only; the resident tables bound K; every tile of a workgroup must have the same column tile
bool tail2_supports(const Params &p) {
    if (!tail_supports(p) || p.mask) return false;
    const int gm = p.M / TBM, gn = p.Cout / TBN;
    if ((p.x2_scale ? 4 : 2) * p.Cin > T2_AFF || p.Cin % T2K || p.Cin / T2K < 4) return false;
    if (gn > 1 && !((gm & 7) == 0 && 32 % gn == 0)) return false;
    const size_t abytes = (size_t)p.M * p.Cin * 2, wbytes = (size_t)p.Cout * p.Kpad * 2;
    return abytes < 0x7fff0000ull && wbytes < 0x7fff0000ull;
}

int launch_tail2(const Params &p, hipStream_t s) {
    const int ntiles = (p.M / TBM) * (p.Cout / TBN);
    const dim3 grid((unsigned)(ntiles < 256 ? ntiles : 256));
    hipLaunchKernelGGL(conv_tail2_kernel, grid, dim3(512), 0, s, p);
    return check_launch("conv_tail2_kernel");
}

bool tail_supports(const Params &p) {
    if (p.mask ? (p.stats != nullptr) : (p.residual != nullptr)) return false;      // data-gradient form: gate (+ residual) + BatchNorm-reverse sums, no statistics
    return p.x2 && p.in_scale && !p.out_scale && !p.out_shift && !p.relu_out && !p.y32 && !p.os2 && !p.res_s2 &&
           p.KH == 1 && p.KW == 1 && p.stride == 1 && p.pad == 0 && p.Kpad == p.Cin && p.Cin % TBK == 0 && p.Cin >= 2 * TBK && p.Cin <= TMAXK &&
           p.Cout % TBN == 0 && p.M % TBM == 0;
}

int launch_tail(const Params &p, hipStream_t s) {
    const int ntiles = (p.M / TBM) * (p.Cout / TBN);
    const dim3 grid((unsigned)(ntiles < 256 ? ntiles : 256));
    if (p.mask) hipLaunchKernelGGL(conv_tail_kernel<true>, grid, dim3(512), 0, s, p);
    else hipLaunchKernelGGL(conv_tail_kernel<false>, grid, dim3(512), 0, s, p);
    return check_launch("conv_tail_kernel");
}

}}  // namespace mhe::conv

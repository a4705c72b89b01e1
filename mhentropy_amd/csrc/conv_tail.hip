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

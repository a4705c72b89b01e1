// Streaming 1x1 convolution for the write-bound layers (bf16, K = 64 or 128 input channels, stride 1: the conv3 / shortcut of
// ResNet-50's layer1 and layer2 - 64 -> 256 and 128 -> 512 at 64x64 / 32x32): every output pixel reads 128-256 bytes and writes
// 512-1024, so the launch is a store stream with a small matrix product attached.  The tiled kernels (conv.hip, conv_p8.hip)
// run load -> MFMA -> store-epilogue back to back per 256-pixel tile, one or two tiles per CU at a time (phase-pipelined kernel,
// ablation builds: 63 of 178 us are load latency that nothing overlaps).  Here a workgroup keeps its 256 x K weight tile in LDS
// for its whole life and walks 64-pixel chunks: the next chunk's activations are in flight (registers) while the current one is
// multiplied and its 32 KiB of outputs are staged through LDS and stored; two workgroups per CU (K = 64) interleave their phases.
// Optional producer BatchNorm + ReLU on the operand load (the forward's "on load" form) and the batch statistics of the output
// as stored (accumulated in the store loop, folded once per workgroup) - same contracts as mhe_conv2d_nhwc.
#include "conv_shared.h"

namespace mhe { namespace conv {

constexpr int ST_PIX = 64, ST_BN = 256;

template <int KT, bool BNLOAD>      // KT = Cin / 64
__global__ __launch_bounds__(256, KT == 1 ? 2 : 1) void conv1x1_stream_kernel(const Params p, int nchunks, int ntiles_n) {
    using T = u16;
    constexpr int NTH = 256, CPR = ST_BN * 2 / 16;                // 32 sixteen-byte chunks per output row
    __shared__ uint4 Wl[ST_BN * 8 * KT];                          // weights: 256 rows x (KT x 128 B), XOR-swizzled like the tiled kernels
    __shared__ uint4 Al[2][ST_PIX * 8 * KT];                      // activations of a chunk, double-buffered
    __shared__ uint4 Ol[ST_PIX * CPR];                            // output staging (32 KiB); statistic partials at the end
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, q = lane >> 4, l15 = lane & 15;
    const int nt_id = blockIdx.x % ntiles_n, wg = blockIdx.x / ntiles_n, nwg = gridDim.x / ntiles_n;
    const int n0 = nt_id * ST_BN;
    const int s = tid & 7, rbase = tid >> 3;                      // this thread's 16-byte slot and first row of a 32-row pass
    const T *xg = reinterpret_cast<const T *>(p.x);
    const T *wg_ = reinterpret_cast<const T *>(p.w);
    T *yg = reinterpret_cast<T *>(p.y);

    // ---- weights -> LDS, once
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int j = 0; j < ST_BN / 32; ++j) {
            const int r = rbase + 32 * j, n = n0 + r;
            const uint4 v = n < p.Cout ? *reinterpret_cast<const uint4 *>(wg_ + (size_t)n * p.Kpad + kt * 64 + s * 8) : make_uint4(0, 0, 0, 0);
            Wl[kt * ST_BN * 8 + swz(r, s)] = v;
        }
    float sc[KT][8], sh[KT][8];
    if constexpr (BNLOAD) {
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
            for (int i = 0; i < 8; ++i) { sc[kt][i] = p.in_scale[kt * 64 + s * 8 + i]; sh[kt][i] = p.in_shift[kt * 64 + s * 8 + i]; }
    }
    // ---- chunk c covers pixels [64 c, 64 c + 64); this workgroup takes c = wg, wg + nwg, ...
    uint4 ra[KT][2];
    auto load_chunk = [&](int c) {
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const long m = (long)c * ST_PIX + rbase + 32 * j;
                ra[kt][j] = (c < nchunks && m < p.M) ? *reinterpret_cast<const uint4 *>(xg + (size_t)m * p.Cin + kt * 64 + s * 8) : make_uint4(0, 0, 0, 0);
            }
    };
    auto store_chunk = [&](int buf, int c) {
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                uint4 v = ra[kt][j];
                if constexpr (BNLOAD) {
                    const long m = (long)c * ST_PIX + rbase + 32 * j;
                    if (m < p.M) {                               // rows past the end stay zero
                        float f[8];
                        Chunk<T>::unpack(v, f);
#pragma unroll
                        for (int i = 0; i < 8; ++i) {
                            f[i] = fmaf(f[i], sc[kt][i], sh[kt][i]);
                            if (p.relu_in) f[i] = fmaxf(f[i], 0.f);
                        }
                        v = Chunk<T>::pack(f);
                    }
                }
                Al[buf][kt * ST_PIX * 8 + swz(rbase + 32 * j, s)] = v;
            }
    };
    const bool st_on = p.stats != nullptr;
    float ss1[8], ss2[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) ss1[i] = ss2[i] = 0.f;

    load_chunk(wg);
    int it = 0;
    for (int c = wg; c < nchunks; c += nwg, ++it) {
        const int buf = it & 1;
        store_chunk(buf, c);
        load_chunk(c + nwg);                                     // in flight during this chunk's product and stores
        __syncthreads();
        v4f acc[4][4];                                           // [channel tile of this wave's 64][pixel tile]
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) acc[a][b] = v4f{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                uint4 fa[4], fb[4];
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) fa[mt] = Al[buf][kt * ST_PIX * 8 + swz(mt * 16 + l15, kk * 4 + q)];
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) fb[nt] = Wl[kt * ST_BN * 8 + swz(wave * 64 + nt * 16 + l15, kk * 4 + q)];
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt)
                        acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                            __builtin_bit_cast(__attribute__((ext_vector_type(8))) __bf16, fb[nt]),
                            __builtin_bit_cast(__attribute__((ext_vector_type(8))) __bf16, fa[mt]), acc[nt][mt], 0, 0, 0);
            }
        // accumulators (lane (l15, q): channels 4q..4q+3 of tile nt, pixel l15 of tile mt) -> staging rows, 16-byte chunks XOR-swizzled by row
        {
            unsigned char *ot = reinterpret_cast<unsigned char *>(Ol);
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    const int row = mt * 16 + l15, boff = (wave * 64 + nt * 16 + 4 * q) * 2;
                    const int chunk = (boff >> 4) ^ (row & 15);
                    const v4f v = acc[nt][mt];
                    uint2 o;
                    o.x = (unsigned)f32_to_bf16(v[0]) | ((unsigned)f32_to_bf16(v[1]) << 16);
                    o.y = (unsigned)f32_to_bf16(v[2]) | ((unsigned)f32_to_bf16(v[3]) << 16);
                    *reinterpret_cast<uint2 *>(ot + ((size_t)row * CPR + chunk) * 16 + (boff & 15)) = o;
                }
        }
        __syncthreads();
        {
            const unsigned char *ot = reinterpret_cast<const unsigned char *>(Ol);
            uint4 raw[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int id = tid + NTH * j, row = id / CPR, cc = id % CPR;
                raw[j] = *reinterpret_cast<const uint4 *>(ot + ((size_t)row * CPR + (cc ^ (row & 15))) * 16);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int id = tid + NTH * j, row = id / CPR, cc = id % CPR;
                const long m = (long)c * ST_PIX + row;
                const int n = n0 + cc * 8;
                if (m < p.M && n < p.Cout) {
                    if (st_on) {
                        float f[8];
                        Chunk<T>::unpack(raw[j], f);
#pragma unroll
                        for (int i = 0; i < 8; ++i) { ss1[i] += f[i]; ss2[i] = fmaf(f[i], f[i], ss2[i]); }
                    }
                    *reinterpret_cast<uint4 *>(yg + (size_t)m * p.Cout + n) = raw[j];
                }
            }
        }
        // (the next iteration's barrier separates these staging reads from its staging writes)
    }
    if (st_on) {
        float *red = reinterpret_cast<float *>(Ol);
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 8; ++i) { red[tid * 16 + i] = ss1[i]; red[tid * 16 + 8 + i] = ss2[i]; }
        __syncthreads();
        const int cch = tid / 8, e = tid % 8, n = n0 + tid;      // channel tid = chunk column cch, element e
        float a = 0.f, b = 0.f;
        for (int k = 0; k < NTH / CPR; ++k) { a += red[(cch + CPR * k) * 16 + e]; b += red[(cch + CPR * k) * 16 + 8 + e]; }
        if (n < p.Cout) {
            const int shard = wg % NSH;
            atomicAdd(p.stats + ((size_t)shard * 2) * p.Cout + n, a);
            atomicAdd(p.stats + ((size_t)shard * 2 + 1) * p.Cout + n, b);
        }
    }
}

// geometry this kernel takes
bool stream_supports(const Params &p) {
    return p.KH == 1 && p.KW == 1 && p.stride == 1 && p.pad == 0 && (p.Cin == 64 || p.Cin == 128) && p.Kpad == p.Cin && p.Cout % ST_BN == 0 &&
           !p.x2 && !p.mask && !p.residual && !p.out_scale && !p.out_shift && !p.relu_out && !p.y32 && !p.os2 && !p.a_out &&
           p.M >= 64 * 1024;
}

int launch_stream(const Params &p, hipStream_t s) {
    const int nchunks = (p.M + ST_PIX - 1) / ST_PIX, ntn = p.Cout / ST_BN;
    const int per_cu = p.Cin == 64 ? 2 : 1;
    int nwg = 256 * per_cu / ntn;                                  // resident workgroups: pixel groups x N tiles
    if (nwg < 1) nwg = 1;
    if (nwg > nchunks) nwg = nchunks;
    const dim3 grid((unsigned)(nwg * ntn)), block(256);
    if (p.Cin == 64) {
        if (p.in_scale) hipLaunchKernelGGL((conv1x1_stream_kernel<1, true>), grid, block, 0, s, p, nchunks, ntn);
        else hipLaunchKernelGGL((conv1x1_stream_kernel<1, false>), grid, block, 0, s, p, nchunks, ntn);
    } else {
        if (p.in_scale) hipLaunchKernelGGL((conv1x1_stream_kernel<2, true>), grid, block, 0, s, p, nchunks, ntn);
        else hipLaunchKernelGGL((conv1x1_stream_kernel<2, false>), grid, block, 0, s, p, nchunks, ntn);
    }
    return check_launch("conv1x1_stream_kernel");
}

}}  // namespace mhe::conv

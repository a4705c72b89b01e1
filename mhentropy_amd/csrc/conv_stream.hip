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

constexpr int ST_PIX = 64;

// DG: the data-gradient epilogue of conv_shared.h (residual - full or half resolution -, ReLU gate by p.mask, BatchNorm-reverse sums of up to
// two units) applied in the store loop; used for the data gradient of a bottleneck's conv1 (K = 64 / 128 -> 256 / 512)
// ST_BN = output channels per workgroup (256, or 128 where the weight tile of 256 would not leave room: K = 256, or to run two workgroups per
// CU at K = 128)
template <int KT, int ST_BN, bool BNLOAD, bool DG = false>      // KT = Cin / 64
__global__ __launch_bounds__(256, (ST_BN * KT + 2 * ST_PIX * KT) * 128 + ST_PIX * ST_BN * 2 <= 80 * 1024 ? 2 : 1)
void conv1x1_stream_kernel(const Params p, int nchunks, int ntiles_n) {
    using T = u16;
    constexpr int NTH = 256, CPR = ST_BN * 2 / 16;                // sixteen-byte chunks per output row
    constexpr int NTW = ST_BN / 64, NJ = ST_PIX * CPR / NTH;      // 16-channel tiles per wave; staged chunks per thread
    constexpr int SWM = CPR < 16 ? CPR - 1 : 15;                  // staging rows: 16-byte chunks XOR-swizzled by the row within the row's own width
    __shared__ uint4 Wl[ST_BN * 8 * KT];                          // weights: 256 rows x (KT x 128 B), XOR-swizzled like the tiled kernels
    __shared__ uint4 Al[2][ST_PIX * 8 * KT];                      // activations of a chunk, double-buffered
    __shared__ uint4 Ol[ST_PIX * CPR > NTH * 4 ? ST_PIX * CPR : NTH * 4];    // output staging; statistic partials at the end (256 threads x 16 floats)
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
    const bool st_on = !DG && p.stats != nullptr;
    float ss1[8], ss2[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) ss1[i] = ss2[i] = 0.f;
    // data-gradient form: this thread's 8 channels are fixed (chunk column tid % 32) - BatchNorm-reverse operands and partial sums
    const int ccol = n0 + (tid % CPR) * 8;
    const T *mk = reinterpret_cast<const T *>(p.mask), *rg = reinterpret_cast<const T *>(p.residual);
    const T *y0g = reinterpret_cast<const T *>(p.bn_y[0]), *y1g = reinterpret_cast<const T *>(p.bn_y[1]);
    // gate bits instead of the gate tensor (p.mask_bits); a consumer handed the gate tensor as its raw output wants sum g only (train step:
    // conv3 + bn3 reversed on Gram statistics) - its second sum is not taken and the tensor is not read for it
    const unsigned char *mkb = DG ? p.mask_bits : nullptr;
    const bool y0sum = DG && y0g != nullptr && y0g == mk, y1sum = DG && y1g != nullptr && y1g == mk;
    float bs1[DG ? 2 : 1][8], bs2[DG ? 2 : 1][8], bmu[DG ? 2 : 1][8], biv[DG ? 2 : 1][8];
    if constexpr (DG) {
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const bool ok = p.bn_y[u] != nullptr && ccol + i < p.Cout;
                bs1[u][i] = bs2[u][i] = 0.f;
                bmu[u][i] = ok ? p.bn_mi[u][ccol + i] : 0.f;
                biv[u][i] = ok ? p.bn_mi[u][p.Cout + ccol + i] : 0.f;
            }
    }

    // data-gradient form: the epilogue's operands of a group of four rows (gate, residual, the consumers' raw tensors) do not depend on the
    // product - group 0 of a chunk is requested BEFORE the chunk's product and staging, so that its latency runs under them (round 4: the
    // store phase used to start with these loads: a chunk iteration took 5.5 us for 80 KB)
    uint4 gm[DG ? 4 : 1], rr[DG ? 4 : 1], ya[DG ? 4 : 1], yb[DG ? 4 : 1];
    unsigned gmb[DG ? 4 : 1];
    size_t off[DG ? 4 : 1];
    bool okr[DG ? 4 : 1];
    auto dg_load = [&](int j0, int c) __attribute__((always_inline)) {
        if constexpr (DG) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int id = tid + NTH * (j0 + g), row = id / CPR;
                const long m = (long)c * ST_PIX + row;
                okr[g] = m < p.M && ccol < p.Cout;
                const long mc = okr[g] ? m : 0;
                off[g] = (size_t)mc * p.Cout + ccol;
                if (mkb) gmb[g] = mkb[(size_t)mc * (p.Cout / 8) + (ccol >> 3)];
                else gm[g] = *reinterpret_cast<const uint4 *>(mk + off[g]);
                rr[g] = make_uint4(0u, 0u, 0u, 0u);
                if (rg) {
                    if (p.res_s2) {
                        const long hr = res_half_row(p, mc);
                        if (hr >= 0) rr[g] = *reinterpret_cast<const uint4 *>(rg + (size_t)hr * p.Cout + ccol);
                    } else rr[g] = *reinterpret_cast<const uint4 *>(rg + off[g]);
                }
                if (y0g && !y0sum) ya[g] = *reinterpret_cast<const uint4 *>(y0g + off[g]);
                if (y1g && !y1sum) yb[g] = *reinterpret_cast<const uint4 *>(y1g + off[g]);
            }
        }
    };
    load_chunk(wg);
    int it = 0;
    for (int c = wg; c < nchunks; c += nwg, ++it) {
        const int buf = it & 1;
        store_chunk(buf, c);
        load_chunk(c + nwg);                                     // in flight during this chunk's product and stores
        if constexpr (ST_BN <= 128) dg_load(0, c);               // (the 256-wide tile has no registers for it: 20 VGPRs spilled)
        __syncthreads();
        v4f acc[NTW][4];                                         // [channel tile of this wave's ST_BN / 4][pixel tile]
#pragma unroll
        for (int a = 0; a < NTW; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) acc[a][b] = v4f{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                uint4 fa[4], fb[NTW];
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) fa[mt] = Al[buf][kt * ST_PIX * 8 + swz(mt * 16 + l15, kk * 4 + q)];
#pragma unroll
                for (int nt = 0; nt < NTW; ++nt) fb[nt] = Wl[kt * ST_BN * 8 + swz(wave * (ST_BN / 4) + nt * 16 + l15, kk * 4 + q)];
#pragma unroll
                for (int nt = 0; nt < NTW; ++nt)
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
            for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    const int row = mt * 16 + l15, boff = (wave * (ST_BN / 4) + nt * 16 + 4 * q) * 2;
                    const int chunk = (boff >> 4) ^ (row & SWM);
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
            if constexpr (DG) {
#pragma unroll 1
                for (int j0 = 0; j0 < NJ; j0 += 4) {                 // 4 rows at a time: their operand loads are issued together
                    if (j0 || ST_BN > 128) dg_load(j0, c);           // (128-wide tile: group 0 was requested before this chunk's product, see the loop head)
                    uint4 raw[4];
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int id = tid + NTH * (j0 + g), row = id / CPR, cc = id % CPR;
                        raw[g] = *reinterpret_cast<const uint4 *>(ot + ((size_t)row * CPR + (cc ^ (row & SWM))) * 16);
                    }
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        if (!okr[g]) continue;
                        float v[8], t[8];
                        Chunk<T>::unpack(raw[g], v);
                        if (rg) {
                            Chunk<T>::unpack(rr[g], t);
#pragma unroll
                            for (int i = 0; i < 8; ++i) v[i] += t[i];
                        }
                        if (mkb) {
#pragma unroll
                            for (int i = 0; i < 8; ++i) v[i] = ((gmb[g] >> i) & 1u) ? v[i] : 0.f;
                        } else {
                            Chunk<T>::unpack(gm[g], t);
#pragma unroll
                            for (int i = 0; i < 8; ++i) v[i] = t[i] > 0.f ? v[i] : 0.f;
                        }
                        if (y0sum) {
#pragma unroll
                            for (int i = 0; i < 8; ++i) bs1[0][i] += v[i];
                        } else if (y0g) {
                            Chunk<T>::unpack(ya[g], t);
#pragma unroll
                            for (int i = 0; i < 8; ++i) { bs1[0][i] += v[i]; bs2[0][i] = fmaf(v[i], (t[i] - bmu[0][i]) * biv[0][i], bs2[0][i]); }
                        }
                        if (y1sum) {
#pragma unroll
                            for (int i = 0; i < 8; ++i) bs1[1][i] += v[i];
                        } else if (y1g) {
                            Chunk<T>::unpack(yb[g], t);
#pragma unroll
                            for (int i = 0; i < 8; ++i) { bs1[1][i] += v[i]; bs2[1][i] = fmaf(v[i], (t[i] - bmu[1][i]) * biv[1][i], bs2[1][i]); }
                        }
                        *reinterpret_cast<uint4 *>(yg + off[g]) = Chunk<T>::pack(v);
                    }
                }
                continue;
            }
            uint4 raw[NJ];
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int id = tid + NTH * j, row = id / CPR, cc = id % CPR;
                raw[j] = *reinterpret_cast<const uint4 *>(ot + ((size_t)row * CPR + (cc ^ (row & SWM))) * 16);
            }
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
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
                    if (yg) *reinterpret_cast<uint4 *>(yg + (size_t)m * p.Cout + n) = raw[j];      // (statistics-only launch: y = NULL)
                }
            }
        }
        // (the next iteration's barrier separates these staging reads from its staging writes)
    }
    if constexpr (DG) {
        float *red = reinterpret_cast<float *>(Ol);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (!p.bn_y[u]) continue;
            __syncthreads();
#pragma unroll
            for (int i = 0; i < 8; ++i) { red[tid * 16 + i] = bs1[u][i]; red[tid * 16 + 8 + i] = bs2[u][i]; }
            __syncthreads();
            const int cch = tid / 8, e = tid % 8, n = n0 + tid;
            float a = 0.f, b = 0.f;
            if (tid < ST_BN)
                for (int k = 0; k < NTH / CPR; ++k) { a += red[(cch + CPR * k) * 16 + e]; b += red[(cch + CPR * k) * 16 + 8 + e]; }
            if (tid < ST_BN && n < p.Cout) {
                fx::add(p.bn_stats[u], wg % NSH, 0, p.Cout, n, a);
                fx::add(p.bn_stats[u], wg % NSH, 1, p.Cout, n, b);
            }
        }
    }
    if (st_on) {
        float *red = reinterpret_cast<float *>(Ol);
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 8; ++i) { red[tid * 16 + i] = ss1[i]; red[tid * 16 + 8 + i] = ss2[i]; }
        __syncthreads();
        const int cch = tid / 8, e = tid % 8, n = n0 + tid;      // channel tid = chunk column cch, element e
        float a = 0.f, b = 0.f;
        if (tid < ST_BN)
            for (int k = 0; k < NTH / CPR; ++k) { a += red[(cch + CPR * k) * 16 + e]; b += red[(cch + CPR * k) * 16 + 8 + e]; }
        if (tid < ST_BN && n < p.Cout) {
            const int shard = wg % NSH;
            fx::add(p.stats, shard, 0, p.Cout, n, a);
            fx::add(p.stats, shard, 1, p.Cout, n, b);
        }
    }
}

// geometry this kernel takes
// output-channel tile for a geometry: 256 at K = 64 (two workgroups per CU), 128 at K = 128 (two per CU) and at K = 256 (one)
// ---------------------------------------------------------------------------------------------------------------------------------
// 3x3 / stride 1 / pad 1 at 64 -> 64 channels on 64-pixel-wide maps (ResNet-50 layer1's conv2 at 256x256 input, and its data gradient):
// 77 GFLOP per launch at C2 but only 268 MB - the tiled kernel re-stages every input pixel nine times (once per tap) through L2 -> LDS and
// runs at ~570 TF.  Here the 72 KiB of weights stay in LDS, a workgroup walks the rows of whole images keeping the three input rows of
// the current output row (plus the one in flight) in a ring of LDS row buffers with zero pad columns: every input pixel is loaded once
// and all nine taps read it from LDS.  Optional producer BatchNorm + ReLU applied once per input element on the load (the tiled on-load
// form repeats it per tap), statistics / data-gradient epilogue in the store loop as in the 1x1 kernel.
// Row images read at pixel offsets 0, 1, 2 (the three taps of a row): the ds_read_b128 lane groups are {0-3, 12-15, 20-27}, ... = 8 rows
// at chunk c and 8 rows at chunk c ^ 1, and conv_shared.h's swz() is conflict-free for them only when the first row is a multiple of
// 16.  XOR with 2 * ((row >> 1) & 3) is conflict-free for every first row (rows of one parity: the 4 + 4 rows cover all four values).
__device__ __forceinline__ int swz3(int row, int slot) { return row * 8 + (slot ^ (((row >> 1) & 3) << 1)); }

// Two roles in one 512-thread workgroup (one wave of each role per SIMD):
//   multiply waves 0-3 : per tick 144 MFMAs (two output rows x 64 pixels x 64 channels; a wave = 32 pixels of one row, all channels), then the
//                        accumulators as bf16 into the staging buffer of this tick's parity;
//   transfer waves 4-7 : global loads of the input rows two ticks ahead into registers, (BatchNorm + ReLU,) the row pair of the next tick into
//                        the ring, and the epilogue of the PREVIOUS tick's output (staging -> batch statistics or the data-gradient
//                        epilogue -> 16-byte global stores).
// One barrier per tick.  Input rows travel in pairs (2k - 1, 2k), k = 0 .. ceil(H / 2); tick t holds pair t of the flat stream over this
// workgroup's images and multiplies the step whose second pair it is (no step on an image's first pair: 1 tick in 33 idle at H = 64).
// With one role doing everything in turn (the first version) the MFMA pipe idled during loads, staging and stores: 96 us; the 16-pixel
// wave tile before that needed 320 B/clk of LDS reads per CU: 127 us.
template <bool BNLOAD, bool DG>
__global__ __launch_bounds__(512) void conv3x3_c64_stream_kernel(const Params p) {
    using T = u16;
    constexpr int NTH = 512, NT2 = 256, CPR = 8, WPX = 64, RROW = (WPX + 2) * 8;   // 16-byte chunks per output row; uint4 per row buffer
    __shared__ uint4 Wl[9 * 64 * 8];                                        // [tap][64 output channels][64 input channels]          72 KiB
    __shared__ uint4 Rl[6][RROW];                                           // three pair slots; pixel slots 0 and 65 = zero padding 49.5 KiB
    __shared__ uint4 Ol[2][1024];                                           // output staging of two rows, by tick parity            32 KiB
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, q = lane >> 4, l15 = lane & 15;
    const bool mult = wave < 4;
    const int t2 = tid & (NT2 - 1);
    const int s = t2 & 7, pbase = t2 >> 3;
    const int wrow = (wave >> 1) & 1, wpx = (wave & 1) * 32;                // multiply wave: output row y + wrow, pixels wpx .. wpx + 31
    const T *xg = reinterpret_cast<const T *>(p.x);
    const T *wg_ = reinterpret_cast<const T *>(p.w);
    T *yg = reinterpret_cast<T *>(p.y);
    const int H = p.H;
    const int npair = (H + 1) / 2 + 1;                                      // pairs per image
    const int nimg = (p.B - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    const int NT = nimg * npair;
    for (int i = tid; i < 9 * 64 * 8; i += NTH) {
        const int sl = i & 7, co = (i >> 3) & 63, tap = i >> 9;
        Wl[tap * 512 + swz3(co, sl)] = *reinterpret_cast<const uint4 *>(wg_ + (size_t)co * p.Kpad + tap * 64 + sl * 8);
    }
    if (tid < 6 * 16) {                                                      // pad columns of the row buffers: zero for good
        const int b = tid >> 4, sl = tid & 7, px = (tid & 8) ? WPX + 1 : 0;
        Rl[b][swz3(px, sl)] = make_uint4(0, 0, 0, 0);
    }
    const bool st_on = !DG && p.stats != nullptr;
    float ss1[8], ss2[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) ss1[i] = ss2[i] = 0.f;
    const int ccol = (t2 % CPR) * 8;
    float bs1[DG ? 2 : 1][8], bs2[DG ? 2 : 1][8];
#pragma unroll
    for (int u = 0; u < (DG ? 2 : 1); ++u)
#pragma unroll
        for (int i = 0; i < 8; ++i) bs1[u][i] = bs2[u][i] = 0.f;

    if (mult) {
        int ck = 0, sp = 2, sc_ = 0;                                        // pair index inside the image at this tick; slots of the previous / this pair
        for (int t = 0; t <= NT; ++t) {
            __syncthreads();
            if (t < NT && ck >= 1) {
                v4f acc[4][2];                                               // [channel tile][pixel tile of this wave's 32]
#pragma unroll
                for (int a = 0; a < 4; ++a) { acc[a][0] = v4f{0.f, 0.f, 0.f, 0.f}; acc[a][1] = v4f{0.f, 0.f, 0.f, 0.f}; }
                // 18 k-steps (tap, 32-channel half), software-pipelined: the fragments of step i + 1 are read while step i multiplies;
                // per k-step a wave reads 2 activation + 4 weight fragments for 8 MFMAs
                uint4 fa[2][2], fb[2][4];
                auto frag = [&](int i, int bi) __attribute__((always_inline)) {
                    const int kh = i / 6, kw = (i / 2) % 3, kk = i & 1;
                    const int d = wrow + kh;                                 // input row y - 1 + d: rows 0, 1 of the previous pair, then of this one
                    const uint4 *row = Rl[(d < 2 ? sp : sc_) * 2 + (d & 1)];
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt) fa[bi][mt] = row[swz3(wpx + mt * 16 + l15 + kw, kk * 4 + q)];
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) fb[bi][nt] = Wl[(kh * 3 + kw) * 512 + swz3(nt * 16 + l15, kk * 4 + q)];
                };
                frag(0, 0);
#pragma unroll
                for (int i = 0; i < 18; ++i) {
                    if (i + 1 < 18) frag(i + 1, (i + 1) & 1);
                    __builtin_amdgcn_sched_barrier(0);                       // left alone the scheduler sinks each read to just before its use
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                        for (int mt = 0; mt < 2; ++mt)
                            acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                                __builtin_bit_cast(__attribute__((ext_vector_type(8))) __bf16, fb[i & 1][nt]),
                                __builtin_bit_cast(__attribute__((ext_vector_type(8))) __bf16, fa[i & 1][mt]), acc[nt][mt], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
                unsigned char *ot = reinterpret_cast<unsigned char *>(Ol[t & 1]);
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt) {
                        const int row = wrow * WPX + wpx + mt * 16 + l15, boff = (nt * 16 + 4 * q) * 2;     // staging row = (output row, pixel)
                        const int chunk = (boff >> 4) ^ (row & 7);
                        const v4f v = acc[nt][mt];
                        uint2 o;
                        o.x = (unsigned)f32_to_bf16(v[0]) | ((unsigned)f32_to_bf16(v[1]) << 16);
                        o.y = (unsigned)f32_to_bf16(v[2]) | ((unsigned)f32_to_bf16(v[3]) << 16);
                        *reinterpret_cast<uint2 *>(ot + ((size_t)row * CPR + chunk) * 16 + (boff & 15)) = o;
                    }
            }
            sp = sc_;
            sc_ = sc_ == 2 ? 0 : sc_ + 1;
            if (++ck == npair) ck = 0;
        }
    } else {
        float sc[8], sh[8];
        if constexpr (BNLOAD) {
#pragma unroll
            for (int i = 0; i < 8; ++i) { sc[i] = p.in_scale[s * 8 + i]; sh[i] = p.in_shift[s * 8 + i]; }
        }
        const T *mk = reinterpret_cast<const T *>(p.mask), *rg = reinterpret_cast<const T *>(p.residual);
        const T *y0g = reinterpret_cast<const T *>(p.bn_y[0]), *y1g = reinterpret_cast<const T *>(p.bn_y[1]);
        float bmu[DG ? 2 : 1][8], biv[DG ? 2 : 1][8];
        if constexpr (DG) {
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const bool ok = p.bn_y[u] != nullptr;
                    bmu[u][i] = ok ? p.bn_mi[u][ccol + i] : 0.f;
                    biv[u][i] = ok ? p.bn_mi[u][64 + ccol + i] : 0.f;
                }
        }
        // two pairs in flight in registers (set = pair parity in the flat stream): a pair is loaded two ticks before it is written to LDS
        uint4 ra[2][2][2];
        auto load_pair = [&](int img, int k, auto SET) __attribute__((always_inline)) {                     // rows 2k - 1, 2k of this workgroup's img-th image; outside [0, H): zeros
            constexpr int set = decltype(SET)::value;
            const size_t b = (size_t)blockIdx.x + (size_t)img * gridDim.x;
#pragma unroll
            for (int e = 0; e < 2; ++e)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    ra[set][e][j] = (unsigned)(2 * k - 1 + e) < (unsigned)H
                                        ? *reinterpret_cast<const uint4 *>(xg + ((b * H + 2 * k - 1 + e) * WPX + pbase + 32 * j) * 64 + s * 8)
                                        : make_uint4(0, 0, 0, 0);
        };
        auto store_pair = [&](int k, int slot, auto SET) __attribute__((always_inline)) {
            constexpr int set = decltype(SET)::value;
#pragma unroll
            for (int e = 0; e < 2; ++e)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    uint4 v = ra[set][e][j];
                    if constexpr (BNLOAD) {
                        if ((unsigned)(2 * k - 1 + e) < (unsigned)H) {       // padding rows stay zero
                            float f[8];
                            Chunk<T>::unpack(v, f);
#pragma unroll
                            for (int i = 0; i < 8; ++i) {
                                f[i] = fmaf(f[i], sc[i], sh[i]);
                                if (p.relu_in) f[i] = fmaxf(f[i], 0.f);
                            }
                            v = Chunk<T>::pack(f);
                        }
                    }
                    Rl[slot * 2 + e][swz3(pbase + 32 * j + 1, s)] = v;
                }
        };
        // data-gradient operands (gate, residual, the BatchNorm units' forward outputs) of the step the multiply waves are working on:
        // loaded one tick before the epilogue that uses them (loaded inside it, their latency was the tick: 600 us per launch)
        uint4 gm[DG ? 4 : 1], rr[DG ? 4 : 1], ya[DG ? 4 : 1], yb[DG ? 4 : 1];
        auto dg_load = [&](int img, int k) __attribute__((always_inline)) {
            if constexpr (DG) {
                const int y = 2 * (k - 1);
                const size_t m0 = (((size_t)blockIdx.x + (size_t)img * gridDim.x) * H + y) * WPX;
                const int nrows = (y + 1 < H ? 2 : 1) * WPX;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int row = (t2 + NT2 * j) / CPR;
                    const size_t off = (m0 + (row < nrows ? row : 0)) * 64 + ccol;
                    gm[j] = *reinterpret_cast<const uint4 *>(mk + off);
                    rr[j] = rg ? *reinterpret_cast<const uint4 *>(rg + off) : make_uint4(0, 0, 0, 0);
                    if (y0g) ya[j] = *reinterpret_cast<const uint4 *>(y0g + off);
                    if (y1g) yb[j] = *reinterpret_cast<const uint4 *>(y1g + off);
                }
            }
        };
        // the output rows 2(k - 1), 2(k - 1) + 1 of image img, staged in Ol[par]
        auto epilogue = [&](int img, int k, int par) __attribute__((always_inline)) {
            const unsigned char *ot = reinterpret_cast<const unsigned char *>(Ol[par]);
            const int y = 2 * (k - 1);
            const size_t m0 = (((size_t)blockIdx.x + (size_t)img * gridDim.x) * H + y) * WPX;
            const int nrows = (y + 1 < H ? 2 : 1) * WPX;                     // staged pixels that exist
            uint4 raw[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int id = t2 + NT2 * j, row = id / CPR, cc = id % CPR;
                raw[j] = *reinterpret_cast<const uint4 *>(ot + ((size_t)row * CPR + (cc ^ (row & 7))) * 16);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int id = t2 + NT2 * j, row = id / CPR;
                if (row >= nrows) continue;
                const size_t off = (m0 + row) * 64 + ccol;
                if constexpr (DG) {
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
                    *reinterpret_cast<uint4 *>(yg + off) = Chunk<T>::pack(v);
                } else {
                    if (st_on) {
                        float f[8];
                        Chunk<T>::unpack(raw[j], f);
#pragma unroll
                        for (int i = 0; i < 8; ++i) { ss1[i] += f[i]; ss2[i] = fmaf(f[i], f[i], ss2[i]); }
                    }
                    *reinterpret_cast<uint4 *>(yg + off) = raw[j];
                }
            }
        };
        using S0 = std::integral_constant<int, 0>;
        using S1 = std::integral_constant<int, 1>;
        int li = 0, lk = 0;                                                  // next pair to load
        auto next_load = [&](auto SET) __attribute__((always_inline)) {
            if (li < nimg) load_pair(li, lk, SET);
            if (++lk == npair) { lk = 0; ++li; }
        };
        next_load(S0{});
        store_pair(0, 0, S0{});                                              // pair 0 (the only exposed load)
        next_load(S1{});                                                     // pair 1: written in tick 0
        next_load(S0{});                                                     // pair 2: written in tick 1
        int wk = 1, ws = 1;                                                  // the pair written in this tick (t + 1): index in its image, slot
        int ci = 0, ck = 0, pi = 0, pk = 0;                                  // pair of this tick, of the previous tick
        auto tick = [&](int t, auto SET) __attribute__((always_inline)) {
            __syncthreads();
            if (t + 1 < NT) store_pair(wk, ws, SET);
            if (++wk == npair) wk = 0;
            ws = ws == 2 ? 0 : ws + 1;
            if (t >= 1 && pk >= 1) epilogue(pi, pk, (t - 1) & 1);            // first everything that consumes loaded registers ...
            next_load(SET);                                                  // ... then this tick's loads: pair t + 3
            if (t < NT && ck >= 1) dg_load(ci, ck);
            pi = ci; pk = ck;
            if (++ck == npair) { ck = 0; ++ci; }
        };
        for (int t = 0; t <= NT; t += 2) {
            tick(t, S1{});
            if (t + 1 <= NT) tick(t + 1, S0{});
        }
    }
    // fold the transfer threads' partial sums (threads sharing a column chunk: 32 of them) and add them to this workgroup's shard
    float *red = reinterpret_cast<float *>(Ol[0]);
    auto fold = [&](const float *a8, const float *b8, fx::acc_t *dst) __attribute__((always_inline)) {
        __syncthreads();
        if (!mult) {
#pragma unroll
            for (int i = 0; i < 8; ++i) { red[t2 * 16 + i] = a8[i]; red[t2 * 16 + 8 + i] = b8[i]; }
        }
        __syncthreads();
        if (!mult && t2 < 64) {
            const int cch = t2 / 8, e = t2 % 8;
            float a = 0.f, b = 0.f;
            for (int k = 0; k < NT2 / CPR; ++k) { a += red[(cch + CPR * k) * 16 + e]; b += red[(cch + CPR * k) * 16 + 8 + e]; }
            fx::add(dst, (int)(blockIdx.x % NSH), 0, 64, t2, a);
            fx::add(dst, (int)(blockIdx.x % NSH), 1, 64, t2, b);
        }
    };
    if constexpr (DG) {
#pragma unroll
        for (int u = 0; u < 2; ++u)
            if (p.bn_y[u]) fold(bs1[u], bs2[u], p.bn_stats[u]);
    } else if (st_on) fold(ss1, ss2, p.stats);
}

bool stream3_supports(const Params &p) {
    if (!(p.KH == 3 && p.KW == 3 && p.stride == 1 && p.pad == 1 && p.Cin == 64 && p.Cout == 64 && p.W == 64 && p.Kpad == 576 && p.B >= 128 &&
          p.H >= 16 && !p.x2 && !p.out_scale && !p.out_shift && !p.relu_out && !p.y32 && !p.os2 && !p.a_out && !p.res_s2))
        return false;
    if (p.mask) return !p.in_scale && !p.stats;
    return !p.residual;
}

int launch_stream3(const Params &p, hipStream_t s) {
    const dim3 grid((unsigned)(p.B < 256 ? p.B : 256)), block(512);
    if (p.mask) hipLaunchKernelGGL((conv3x3_c64_stream_kernel<false, true>), grid, block, 0, s, p);
    else if (p.in_scale) hipLaunchKernelGGL((conv3x3_c64_stream_kernel<true, false>), grid, block, 0, s, p);
    else hipLaunchKernelGGL((conv3x3_c64_stream_kernel<false, false>), grid, block, 0, s, p);
    return check_launch("conv3x3_c64_stream_kernel");
}

// (measured, tools/conv_variants.py / dg_variants.py: at K = 128 the plain / on-load forms run 85 / 95 us on the 256 tile against 96 / 105 us
// on the 128 tile, the data-gradient form 254 against 227 us)
// (round 4: 64 -> 64 channels, forward forms - layer1.0's conv1 on the pooled stem output, 1 M pixels at config C2: 268 MB in 107 us on the
// 128 x 64 register-staged tile; a 64-wide weight tile here, four workgroups per CU)
static int stream_bn(const Params &p) { return p.Cin == 64 ? (p.Cout == 64 && !p.mask ? 64 : 256) : (p.Cin == 128 && !p.mask ? 256 : 128); }

bool stream_supports(const Params &p) {
    if (!(p.KH == 1 && p.KW == 1 && p.stride == 1 && p.pad == 0 && (p.Cin == 64 || p.Cin == 128 || p.Cin == 256) && p.Kpad == p.Cin &&
          !p.x2 && !p.out_scale && !p.out_shift && !p.relu_out && !p.y32 && !p.os2 && !p.a_out && (p.M >= 64 * 1024 || (p.stats_only && !p.mask))))
        return false;
    if (p.Cout % stream_bn(p)) return false;
    if (p.Cin == 256 && !p.mask) return false;             // K = 256: only the data-gradient form gains (147 vs 158 us); plain 85 vs 63 us on the tiled kernel
    if (p.mask) return !p.in_scale && !p.stats;            // data-gradient form
    return !p.residual;
}

template <int KT, int NB>
static void launch_stream_t(const Params &p, hipStream_t s, dim3 grid, int nchunks, int ntn) {
    const dim3 block(256);
    if constexpr (NB == 64) {                      // forward forms only (stream_bn)
        if (p.in_scale) hipLaunchKernelGGL((conv1x1_stream_kernel<KT, NB, true>), grid, block, 0, s, p, nchunks, ntn);
        else hipLaunchKernelGGL((conv1x1_stream_kernel<KT, NB, false>), grid, block, 0, s, p, nchunks, ntn);
    } else if (p.mask) hipLaunchKernelGGL((conv1x1_stream_kernel<KT, NB, false, true>), grid, block, 0, s, p, nchunks, ntn);
    else if constexpr (KT <= 2) {
        if (p.in_scale) hipLaunchKernelGGL((conv1x1_stream_kernel<KT, NB, true>), grid, block, 0, s, p, nchunks, ntn);
        else hipLaunchKernelGGL((conv1x1_stream_kernel<KT, NB, false>), grid, block, 0, s, p, nchunks, ntn);
    } else hipLaunchKernelGGL((conv1x1_stream_kernel<KT, NB, false>), grid, block, 0, s, p, nchunks, ntn);
}

int launch_stream(const Params &p, hipStream_t s) {
    const int nb = stream_bn(p);
    const int nchunks = (p.M + ST_PIX - 1) / ST_PIX, ntn = p.Cout / nb;
    const int per_cu = nb == 64 ? 4 : (p.Cin == 256 || (p.Cin == 128 && nb == 256)) ? 1 : 2;
    int nwg = 256 * per_cu / ntn;                                  // resident workgroups: pixel groups x N tiles
    if (nwg < 1) nwg = 1;
    if (nwg > nchunks) nwg = nchunks;
    const dim3 grid((unsigned)(nwg * ntn));
    if (p.Cin == 64 && nb == 64) launch_stream_t<1, 64>(p, s, grid, nchunks, ntn);
    else if (p.Cin == 64) launch_stream_t<1, 256>(p, s, grid, nchunks, ntn);
    else if (p.Cin == 128 && nb == 256) launch_stream_t<2, 256>(p, s, grid, nchunks, ntn);
    else if (p.Cin == 128) launch_stream_t<2, 128>(p, s, grid, nchunks, ntn);
    else launch_stream_t<4, 128>(p, s, grid, nchunks, ntn);
    return check_launch("conv1x1_stream_kernel");
}

}}  // namespace mhe::conv

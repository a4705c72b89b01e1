// Write-bound 1x1 convolutions with MANY output channels and 256 / 512 input channels (variant 11): conv3 of ResNet-50's layer3 / layer4
// bottlenecks (256 -> 1024 at 16x16, 512 -> 2048 at 8x8; torchvision Bottleneck.conv3, reference hand/network.py's encoder).
//
// The phase-pipelined 256x256 kernel runs them at 2.5 / 1.7 TB/s of compulsory traffic (66 / 50 us): with K = 256 a tile has four K steps,
// so its load -> MFMA -> 128 KiB store-epilogue sequence never overlaps, and every 256-pixel tile re-reads its weight slab.  Here, as in
// conv_stream.hip's kernels, the WEIGHTS stay in LDS: a persistent workgroup owns a slab of NS output channels x all K (64 KiB) and walks
// 128-pixel tiles; and, as in conv_tail.hip, in two roles:
//   multiply waves 0-3 : per tick one 64-deep K tile of the current pixel tile against the resident slab; after the tile's last K tile the
//                        accumulators go to the staging buffer as bf16;
//   transfer waves 4-7 : the activation K tiles NSET ticks ahead from global memory into registers (a K tile is only 16 KiB: six of them in
//                        flight), (producer BatchNorm + ReLU,) into the LDS stage the multiply waves are not reading; after a tile's last
//                        tick the staged outputs -> batch statistics -> 16-byte global stores.
// One barrier per tick; the activation stages are written two ticks ahead, so that the multiply waves read a tick's fragments during the tick
// before it.  The eight workgroups that run on one XCD at a time share their pixel tiles' activations through that XCD's L2
// (workgroup w: XCD w & 7; slab (w >> 3) % slabs), so the activations are read from HBM once.
#include "conv_shared.h"

namespace mhe { namespace conv {

namespace {
constexpr int WBM = 128, WBK = 64, WSLAB = 64 * 1024 / 16;              // pixel tile, K tile, uint4 of the resident weight slab
constexpr int WNSET = 8;                                                  // ticks per unrolled iteration = register sets in flight
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
}

// NS output channels per slab, NK = K / 64 K tiles per pixel tile: (128, 4) for K = 256, (64, 8) for K = 512.  The tick loop is unrolled
// over WNSET ticks = 8 / NK pixel tiles with every tile / K-tile index a compile-time constant and NO branch around a global-memory
// operation (loads from clamped addresses, stores through a buffer descriptor with an out-of-range offset when there is nothing to store):
// only then does the compiler wait for a register set with a COUNTED vmcnt and leave the younger sets in flight.  (First version: loads
// and the epilogue behind run-time conditions -> vmcnt(0) before every LDS write -> a tick lasted one memory latency: 63 / 72 us.)
template <int NS, int NK, bool BNLOAD>
__global__ __launch_bounds__(512) void conv_wide_kernel(const Params p, int nslab, int tiles_per_wg) {
    using T = u16;
    constexpr int CPR = NS / 8;                                           // 16-byte chunks per output row of the slab
    constexpr int TPI = WNSET / NK;                                       // pixel tiles per unrolled iteration
    __shared__ uint4 Wl[WSLAB];                                           // [K tile][NS rows][8 chunks], rows swizzled                   64 KiB
    __shared__ uint4 Al[3][WBM * 8];                                      // activation stages: tick t is multiplied from t % 3         48 KiB
    __shared__ uint4 Ol[WBM * CPR];                                       // output staging of one pixel tile                       32 / 16 KiB
    __shared__ float aff[BNLOAD ? 2 * 512 : 2];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, q = lane >> 4, l15 = lane & 15;
    const bool mult = wave < 4;
    const int t2 = tid & 255, s = t2 & 7, rbase = t2 >> 3;
    constexpr int K = NK * WBK;
    const int xcd = blockIdx.x & 7, a = blockIdx.x >> 3, slab = a % nslab, group = a / nslab, ngroup = 32 / nslab;
    const int n0 = slab * NS;
    const int niter = tiles_per_wg / TPI;
    // pixel tile j of this workgroup
    auto tile_m0 = [&](int j) __attribute__((always_inline)) { return (8 * (group + ngroup * j) + xcd) * WBM; };
    const T *xg = reinterpret_cast<const T *>(p.x), *wg = reinterpret_cast<const T *>(p.w);
    for (int i = tid; i < NS * NK * 8; i += 512) {                        // the slab: row n, K tile kt, chunk c
        const int c = i & 7, n = (i >> 3) % NS, kt = i / (8 * NS);
        Wl[kt * NS * 8 + swz(n, c)] = *reinterpret_cast<const uint4 *>(wg + (size_t)(n0 + n) * p.Kpad + kt * WBK + c * 8);
    }
    if constexpr (BNLOAD) {
        for (int i = tid; i < K; i += 512) { aff[i] = p.in_scale[i]; aff[512 + i] = p.in_shift[i]; }
    }
    __syncthreads();
    const bool st_on = p.stats != nullptr;
    constexpr int RSTEP = 256 / CPR, NR = WBM / RSTEP;                    // epilogue rows of a transfer thread: 16 apart, 8 of them (NS = 128); 32 apart, 4 (NS = 64)
    float ss1[8], ss2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) ss1[e] = ss2[e] = 0.f;
    if (mult) {
        // NS = 128: wave = 64 pixels x 64 channels (2 x 2 waves); NS = 64: wave = 32 pixels x 64 channels (4 x 1)
        constexpr int MT = NS == 128 ? 4 : 2;
        const int px0 = NS == 128 ? (wave >> 1) * 64 : wave * 32, ch0 = NS == 128 ? (wave & 1) * 64 : 0;
        v4f acc[4][MT];
        // fragments of tick t + 1 are read (stage (t + 1) % 3, complete since the barrier of tick t) while tick t multiplies: with the reads
        // of a tick issued after its own barrier a tick was LDS latency + MFMA time (58 us at layer3)
        uint4 fa[2][2][MT], fb[2][2][4];
        auto read_frags = [&](int buf, const uint4 *At, int kt) __attribute__((always_inline)) {
            const uint4 *Wt = Wl + kt * NS * 8;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
                for (int n = 0; n < 4; ++n) fb[buf][kk][n] = Wt[swz(ch0 + n * 16 + l15, kk * 4 + q)];
#pragma unroll
                for (int m = 0; m < MT; ++m) fa[buf][kk][m] = At[swz(px0 + m * 16 + l15, kk * 4 + q)];
            }
        };
        __syncthreads();                                                  // ticks 0 and 1 are staged
        read_frags(0, Al[0], 0);
        int sn = 1;                                                       // stage of tick t + 1
        for (int it = 0; it < niter; ++it) {
#pragma unroll
            for (int i = 0; i < WNSET; ++i) {
                const int kt = i % NK;
                if (kt == 0) {
#pragma unroll
                    for (int n = 0; n < 4; ++n)
#pragma unroll
                        for (int m = 0; m < MT; ++m) acc[n][m] = v4f{0.f, 0.f, 0.f, 0.f};
                }
                __syncthreads();
                read_frags((i + 1) & 1, Al[sn], (i + 1) % NK);
                sn = sn == 2 ? 0 : sn + 1;
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                    for (int m = 0; m < MT; ++m)
#pragma unroll
                        for (int n = 0; n < 4; ++n)
                            acc[n][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                                __builtin_bit_cast(__attribute__((ext_vector_type(8))) __bf16, fb[i & 1][kk][n]),
                                __builtin_bit_cast(__attribute__((ext_vector_type(8))) __bf16, fa[i & 1][kk][m]), acc[n][m], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (kt == NK - 1) {
                    // the transfer waves read the previous tile's staged outputs during that tile's successor's FIRST tick: free again by now
                    unsigned char *ot = reinterpret_cast<unsigned char *>(Ol);
#pragma unroll
                    for (int n = 0; n < 4; ++n)
#pragma unroll
                        for (int m = 0; m < MT; ++m) {
                            const int row = px0 + m * 16 + l15, boff = (ch0 + n * 16 + 4 * q) * 2;
                            const int chunk = (boff >> 4) ^ (row & (CPR - 1));
                            const v4f v = acc[n][m];
                            uint2 o;
                            o.x = (unsigned)f32_to_bf16(v[0]) | ((unsigned)f32_to_bf16(v[1]) << 16);
                            o.y = (unsigned)f32_to_bf16(v[2]) | ((unsigned)f32_to_bf16(v[3]) << 16);
                            *reinterpret_cast<uint2 *>(ot + ((size_t)row * CPR + chunk) * 16 + (boff & 15)) = o;
                        }
                }
            }
        }
        __syncthreads();                                                  // the tick after the last one: the transfer waves' final epilogue
    } else {
        struct Set { uint4 a[4]; };
        Set st[WNSET];
        // (statistics-only launch, p.y == NULL: a descriptor of zero records - every store is dropped by the hardware, the staged tile still
        // feeds the sums of the values that WOULD have been stored)
        const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(p.y ? p.y : const_cast<void *>(p.x), 0, p.y ? (int)((size_t)p.M * p.Cout * 2) : 0, 0x00020000);
        // flat tick u = (pixel tile j, K tile kt): this thread's chunk s of rows rbase + 32 i; j clamped (past the end: loaded, never used)
        auto load_tick = [&](int j, int kt, Set &S) __attribute__((always_inline)) {
            const size_t m0 = (size_t)tile_m0(j < tiles_per_wg ? j : tiles_per_wg - 1);
#pragma unroll
            for (int i = 0; i < 4; ++i) S.a[i] = *reinterpret_cast<const uint4 *>(xg + (m0 + rbase + 32 * i) * K + kt * WBK + s * 8);
        };
        auto store_tick = [&](int stage, int kt, const Set &S) __attribute__((always_inline)) {
            uint4 *At = Al[stage];                                         // (run-time stage index: a uniform address add)
            float sc[8], sh[8];
            if constexpr (BNLOAD) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const float4 a0 = *reinterpret_cast<const float4 *>(aff + kt * WBK + s * 8 + 4 * h);
                    const float4 a1 = *reinterpret_cast<const float4 *>(aff + 512 + kt * WBK + s * 8 + 4 * h);
                    sc[4 * h] = a0.x; sc[4 * h + 1] = a0.y; sc[4 * h + 2] = a0.z; sc[4 * h + 3] = a0.w;
                    sh[4 * h] = a1.x; sh[4 * h + 1] = a1.y; sh[4 * h + 2] = a1.z; sh[4 * h + 3] = a1.w;
                }
            }
            const float lo = p.relu_in ? 0.f : -3.0e38f;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                uint4 v = S.a[i];
                if constexpr (BNLOAD) {
                    float f[8];
                    Chunk<T>::unpack(v, f);
#pragma unroll
                    for (int e = 0; e < 8; ++e) f[e] = fmaxf(fmaf(f[e], sc[e], sh[e]), lo);
                    v = Chunk<T>::pack(f);
                }
                At[swz(rbase + 32 * i, s)] = v;
            }
        };
        const int cc = t2 % CPR, r0 = t2 / CPR;                           // epilogue: chunk cc of rows r0 + RSTEP i
        auto epilogue = [&](int j, bool valid) __attribute__((always_inline)) {
            const unsigned char *ot = reinterpret_cast<const unsigned char *>(Ol);
            const unsigned m0 = (unsigned)tile_m0(valid ? j : 0);
            uint4 raw[NR];
#pragma unroll
            for (int i = 0; i < NR; ++i) {
                const int row = r0 + RSTEP * i;
                raw[i] = *reinterpret_cast<const uint4 *>(ot + ((size_t)row * CPR + (cc ^ (row & (CPR - 1)))) * 16);
            }
#pragma unroll
            for (int i = 0; i < NR; ++i) {
                const int row = r0 + RSTEP * i;
                if (st_on && valid) {                                     // (arithmetic only inside the branch)
                    float f[8];
                    Chunk<T>::unpack(raw[i], f);
#pragma unroll
                    for (int e = 0; e < 8; ++e) { ss1[e] += f[e]; ss2[e] = fmaf(f[e], f[e], ss2[e]); }
                }
                const unsigned off = ((m0 + row) * (unsigned)p.Cout + n0 + cc * 8) * 2u;
                const u32x4 v = {raw[i].x, raw[i].y, raw[i].z, raw[i].w};
                __builtin_amdgcn_raw_buffer_store_b128(v, yrs, valid ? (int)off : (int)0x80000000u, 0, 0);     // out of range: dropped by the hardware
            }
        };
        // tick u lives in set u % 8 and is staged in Al[u % 3], TWO ticks before it is multiplied.  Prologue: ticks 0 .. 7 in flight, ticks 0
        // and 1 written at once (the exposed loads), ticks 8 and 9 issued.
#pragma unroll
        for (int u = 0; u < WNSET; ++u) load_tick(u / NK, u % NK, st[u]);
        store_tick(0, 0, st[0]);
        load_tick(WNSET / NK, 0, st[0]);
        store_tick(1, 1 % NK, st[1]);
        load_tick((WNSET + 1) / NK, 1 % NK, st[1]);
        __syncthreads();
        int sw = 2;                                                       // stage of tick t + 2
        for (int it = 0; it < niter; ++it) {
#pragma unroll
            for (int i = 0; i < WNSET; ++i) {                             // tick t = 8 it + i: pixel tile it * TPI + i / NK, K tile i % NK
                __syncthreads();                                          // stage (t + 2) % 3 was last read during tick t - 2
                store_tick(sw, (i + 2) % NK, st[(i + 2) % WNSET]);        // tick t + 2: first what consumes loaded registers ...
                sw = sw == 2 ? 0 : sw + 1;
                if (i % NK == 0) epilogue(it * TPI + i / NK - 1, it * TPI + i / NK - 1 >= 0);   // the tile whose last tick was t - 1
                load_tick(it * TPI + (i + 2 + WNSET) / NK, (i + 2) % NK, st[(i + 2) % WNSET]);  // ... then tick t + 10 into the same set
            }
        }
        __syncthreads();
        epilogue(tiles_per_wg - 1, true);                                 // the last tile's outputs (staged during the last tick)
    }
    if (st_on) {
        // fold the transfer threads that share a column chunk (256 / CPR of them) and add to this workgroup's shard
        __syncthreads();
        float *red = reinterpret_cast<float *>(Al);
        if (!mult) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { red[t2 * 16 + e] = ss1[e]; red[t2 * 16 + 8 + e] = ss2[e]; }
        }
        __syncthreads();
        if (!mult && t2 < NS) {
            const int ch = t2 >> 3, e = t2 & 7;
            float a1 = 0.f, b1 = 0.f;
            for (int k = 0; k < RSTEP; ++k) { a1 += red[(ch + CPR * k) * 16 + e]; b1 += red[(ch + CPR * k) * 16 + 8 + e]; }
            fx::add(p.stats, (int)(blockIdx.x % NSH), 0, p.Cout, n0 + t2, a1);
            fx::add(p.stats, (int)(blockIdx.x % NSH), 1, p.Cout, n0 + t2, b1);
        }
    }
}

static int wide_ns(const Params &p) { return p.Cin == 256 ? 128 : 64; }

bool wide_supports(const Params &p) {
    if (!(p.KH == 1 && p.KW == 1 && p.stride == 1 && p.pad == 0 && (p.Cin == 256 || p.Cin == 512) && p.Kpad == p.Cin && !p.x2 && !p.mask &&
          !p.residual && !p.out_scale && !p.out_shift && !p.relu_out && !p.y32 && !p.os2 && !p.res_s2 && !p.a_out && p.M % WBM == 0))
        return false;
    const int ns = wide_ns(p);
    if (p.Cout % ns) return false;
    const int nslab = p.Cout / ns, gm = p.M / WBM;
    if (nslab < 4 || nslab > 32 || (32 % nslab) || gm % (256 / nslab)) return false;      // 256 workgroups: 8 XCDs x (32 / nslab) tile groups x nslab slabs
    const int tiles = gm / (256 / nslab);
    if ((size_t)p.M * p.Cout * 2 >= (1ull << 31)) return false;                            // 32-bit byte offsets into the output
    return tiles >= 4 && tiles % (p.Cin == 256 ? 2 : 1) == 0;                              // whole unrolled iterations (8 ticks = 2 tiles at K = 256)
}

int launch_wide(const Params &p, hipStream_t s) {
    const int ns = wide_ns(p), nslab = p.Cout / ns, tiles = (p.M / WBM) / (256 / nslab);
    const dim3 grid(256), block(512);
    if (ns == 128) {
        if (p.in_scale) hipLaunchKernelGGL((conv_wide_kernel<128, 4, true>), grid, block, 0, s, p, nslab, tiles);
        else hipLaunchKernelGGL((conv_wide_kernel<128, 4, false>), grid, block, 0, s, p, nslab, tiles);
    } else {
        if (p.in_scale) hipLaunchKernelGGL((conv_wide_kernel<64, 8, true>), grid, block, 0, s, p, nslab, tiles);
        else hipLaunchKernelGGL((conv_wide_kernel<64, 8, false>), grid, block, 0, s, p, nslab, tiles);
    }
    return check_launch("conv_wide_kernel");
}

}}  // namespace mhe::conv

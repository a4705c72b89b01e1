// Phase-pipelined 256x256 bf16 implicit-GEMM convolution for gfx950 (the plain-operand form of the trunk's
// convolutions and of their data gradients; reference: torchvision ResNet via hand/network.py:54-61,110).
//
// Same GEMM view, operand layout and epilogue as conv_kernel (conv.hip): y^T[n][m] = sum_k w[n][k] xcol[m][k],
// K ordered (kh, kw, cin), 64-deep K tiles, one 512-thread workgroup (2 x 4 waves, 128 pixels x 64 channels each)
// per CU.  What differs is how the operands reach the matrix cores:
//
//  * HBM/L2 -> LDS by LDS-DMA (buffer_load_dwordx4 ... lds), no VGPR staging and no ds_write pass.  The DMA writes LDS
//    lane-linearly, so the bank-conflict swizzle (16-byte chunk ^ ((row >> 1) & 7), as in conv_kernel) is applied to the
//    per-lane SOURCE offset.  Rows outside the image (padding taps), pixels >= M, channels >= Cout and K tiles past the
//    end get an offset beyond the descriptor's num_records: the hardware writes zeros for them (probe:
//    tools/probes/lds_dma_oob.hip), so there is no predication, no zero page and no tail code.
//  * a K tile is staged as four 16 KiB half-tiles (A0 | B0 | B1 | A1: pixel rows 0-63 / 64-127 of every wave row, channel rows
//    0-31 / 32-63 of every wave column) into a two-stage ring; its 64 MFMAs per wave run as four PHASES of 16, one
//    (A-half, B-half) quadrant each: (A0,B0) (A0,B1) (A1,B1) (A1,B0).  Phase p reads only the fragments it newly needs
//    (12 / 4 / 8 / 0 ds_read_b128), issues ONE half-tile of prefetch (2 DMA instructions per wave) and retires, with a
//    COUNTED s_waitcnt vmcnt(8), the half-tile issued four phases earlier - four half-tiles (64 KiB per CU) stay in flight
//    across the barriers at all times, vmcnt never drains to 0 inside the loop;
//  * the two wave rows run STAGGERED by one barrier (waves 4-7 take one extra s_barrier before the loop, waves 0-3 one
//    after it): on every SIMD one wave is in its MFMA cluster while its partner reads LDS / issues DMA.
//
// Hazards, per half-tile buffer (g = global phase number): DMA issued in phase g-4 by every wave; each wave's vmcnt(8) in
// phase g sits before that phase's first barrier; the first ds_read of the buffer is in phase g+1 (after a barrier every wave
// passed behind its wait).  The buffer's last ds_read is waited for (lgkmcnt(0)) behind the first barrier of its phase r;
// it is re-staged in phase r+2 at the earliest (A0: read 4t+1, re-staged 4t+3; B0 4t+1 / 4t+4; B1 4t+2 / 4t+5; A1 4t+3 / 4t+6),
// i.e. after a barrier that the staggered partner row has passed behind its own lgkmcnt(0).
#include "conv_shared.h"

namespace mhe { namespace conv {

typedef unsigned u4 __attribute__((ext_vector_type(4)));
typedef __attribute__((ext_vector_type(8))) __bf16 bf8;

constexpr unsigned P8_OOB = 0x80000000u;      // an offset no descriptor of < 2 GiB admits
constexpr int P8_STAGE = 65536, P8_A0 = 0, P8_B0 = 16384, P8_B1 = 32768, P8_A1 = 49152;

// ---- the load half of a phase as ONE asm statement (hipcc neither counts nor reorders what is inside):
// fragment reads, then the two DMA pieces of the half-tile to prefetch, the counted wait that retires the half-tile issued
// four phases ago, the barrier, and the wait for the fragments.  M0 (the DMA's LDS base) is written and used inside.
#define P8_DMA2                                                                                               \
    "s_mov_b32 %[keep], m0\n\ts_mov_b32 m0, %[l0]\n\ts_nop 0\n\tbuffer_load_dwordx4 %[v0], %[rs], 0 offen lds\n\t" \
    "s_mov_b32 m0, %[l1]\n\ts_nop 0\n\tbuffer_load_dwordx4 %[v1], %[rs], 0 offen lds\n\ts_mov_b32 m0, %[keep]\n\t"
#define P8_SYNC "s_waitcnt vmcnt(8)\n\ts_barrier\n\ts_waitcnt lgkmcnt(0)"

// phase 1: A half 0 (4 row tiles x 2 k-steps) + B half 0 (2 x 2)
__device__ __forceinline__ void p8_load_a_b(u4 (&fa)[4][2], u4 (&fb)[2][2], unsigned ra0, unsigned ra1, unsigned rb0, unsigned rb1,
                                            unsigned v0, unsigned v1, u4 rs, unsigned l0, unsigned l1) {
    unsigned keep;
    asm volatile(
        "ds_read_b128 %[b00], %[rb0] offset:16384\n\tds_read_b128 %[b01], %[rb1] offset:16384\n\t"
        "ds_read_b128 %[b10], %[rb0] offset:18432\n\tds_read_b128 %[b11], %[rb1] offset:18432\n\t"
        "ds_read_b128 %[a00], %[ra0]\n\tds_read_b128 %[a01], %[ra1]\n\t"
        "ds_read_b128 %[a10], %[ra0] offset:2048\n\tds_read_b128 %[a11], %[ra1] offset:2048\n\t"
        "ds_read_b128 %[a20], %[ra0] offset:4096\n\tds_read_b128 %[a21], %[ra1] offset:4096\n\t"
        "ds_read_b128 %[a30], %[ra0] offset:6144\n\tds_read_b128 %[a31], %[ra1] offset:6144\n\t"
        P8_DMA2 P8_SYNC
        : [a00] "=&v"(fa[0][0]), [a01] "=&v"(fa[0][1]), [a10] "=&v"(fa[1][0]), [a11] "=&v"(fa[1][1]),
          [a20] "=&v"(fa[2][0]), [a21] "=&v"(fa[2][1]), [a30] "=&v"(fa[3][0]), [a31] "=&v"(fa[3][1]),
          [b00] "=&v"(fb[0][0]), [b01] "=&v"(fb[0][1]), [b10] "=&v"(fb[1][0]), [b11] "=&v"(fb[1][1]), [keep] "=&s"(keep)
        : [ra0] "v"(ra0), [ra1] "v"(ra1), [rb0] "v"(rb0), [rb1] "v"(rb1), [v0] "v"(v0), [v1] "v"(v1), [rs] "s"(rs),
          [l0] "s"(l0), [l1] "s"(l1)
        : "memory");
}
// phase 2: B half 1
__device__ __forceinline__ void p8_load_b1(u4 (&fb)[2][2], unsigned rb0, unsigned rb1, unsigned v0, unsigned v1, u4 rs,
                                           unsigned l0, unsigned l1) {
    unsigned keep;
    asm volatile(
        "ds_read_b128 %[b00], %[rb0] offset:32768\n\tds_read_b128 %[b01], %[rb1] offset:32768\n\t"
        "ds_read_b128 %[b10], %[rb0] offset:34816\n\tds_read_b128 %[b11], %[rb1] offset:34816\n\t"
        "s_nop 1\n\t"
        P8_DMA2 P8_SYNC
        : [b00] "=&v"(fb[0][0]), [b01] "=&v"(fb[0][1]), [b10] "=&v"(fb[1][0]), [b11] "=&v"(fb[1][1]), [keep] "=&s"(keep)
        : [rb0] "v"(rb0), [rb1] "v"(rb1), [v0] "v"(v0), [v1] "v"(v1), [rs] "s"(rs), [l0] "s"(l0), [l1] "s"(l1)
        : "memory");
}
// phase 3: A half 1
__device__ __forceinline__ void p8_load_a1(u4 (&fa)[4][2], unsigned ra0, unsigned ra1, unsigned v0, unsigned v1, u4 rs, unsigned l0,
                                           unsigned l1) {
    unsigned keep;
    asm volatile(
        "ds_read_b128 %[a00], %[ra0] offset:49152\n\tds_read_b128 %[a01], %[ra1] offset:49152\n\t"
        "ds_read_b128 %[a10], %[ra0] offset:51200\n\tds_read_b128 %[a11], %[ra1] offset:51200\n\t"
        "ds_read_b128 %[a20], %[ra0] offset:53248\n\tds_read_b128 %[a21], %[ra1] offset:53248\n\t"
        "ds_read_b128 %[a30], %[ra0] offset:55296\n\tds_read_b128 %[a31], %[ra1] offset:55296\n\t"
        P8_DMA2 P8_SYNC
        : [a00] "=&v"(fa[0][0]), [a01] "=&v"(fa[0][1]), [a10] "=&v"(fa[1][0]), [a11] "=&v"(fa[1][1]),
          [a20] "=&v"(fa[2][0]), [a21] "=&v"(fa[2][1]), [a30] "=&v"(fa[3][0]), [a31] "=&v"(fa[3][1]), [keep] "=&s"(keep)
        : [ra0] "v"(ra0), [ra1] "v"(ra1), [v0] "v"(v0), [v1] "v"(v1), [rs] "s"(rs), [l0] "s"(l0), [l1] "s"(l1)
        : "memory");
}
// phase 4: no new fragments
__device__ __forceinline__ void p8_load_none(unsigned v0, unsigned v1, u4 rs, unsigned l0, unsigned l1) {
    unsigned keep;
    asm volatile("s_nop 4\n\t" P8_DMA2 P8_SYNC
                 : [keep] "=&s"(keep)
                 : [v0] "v"(v0), [v1] "v"(v1), [rs] "s"(rs), [l0] "s"(l0), [l1] "s"(l1)
                 : "memory");
}
// prologue piece: issue only
__device__ __forceinline__ void p8_issue(unsigned v0, unsigned v1, u4 rs, unsigned l0, unsigned l1) {
    unsigned keep;
    asm volatile("s_nop 4\n\t" P8_DMA2 "s_nop 0"
                 : [keep] "=&s"(keep)
                 : [v0] "v"(v0), [v1] "v"(v1), [rs] "s"(rs), [l0] "s"(l0), [l1] "s"(l1)
                 : "memory");
}

// 16 MFMAs of one quadrant: channel tiles 2*HB + {0,1}, pixel tiles 4*HA + {0..3}, both 32-deep k-steps
template <int HA, int HB>
__device__ __forceinline__ void p8_mfma(v4f (&acc)[4][8], const u4 (&fa)[4][2], const u4 (&fb)[2][2]) {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
                acc[2 * HB + nt][4 * HA + mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                    __builtin_bit_cast(bf8, fb[nt][kk]), __builtin_bit_cast(bf8, fa[mt][kk]), acc[2 * HB + nt][4 * HA + mt], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
}

// position of a K tile in the (kh, kw, cin) order, advanced incrementally (everything here is wave-uniform)
struct P8Kpos {
    int tap, c0, kh, kw;
    __device__ __forceinline__ void next(const Params &p) {
        c0 += 64;
        if (c0 >= p.Cin) { c0 = 0; ++tap; if (++kw == p.KW) { kw = 0; ++kh; } }
    }
    // byte offset, relative to a pixel row's (hi0, wi0) origin, of this tile's 128 bytes
    __device__ __forceinline__ unsigned aoff(const Params &p) const { return (unsigned)(((kh * p.W + kw) * p.Cin + c0) * 2); }
};

template <bool DG, bool TAPS>
__global__ __launch_bounds__(512) void conv_p8_kernel(const Params p) {
    using T = u16;
    __shared__ uint4 lds[2 * P8_STAGE / 16];          // 128 KiB: the two-stage ring, then the epilogue's staging

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = lane >> 4, l15 = lane & 15;
    const int wr = wave >> 2, wc = wave & 3;
    int mtile, ntile;
    tile_of_block(mtile, ntile);
    const int m0 = mtile * 256, n0 = ntile * 256;
    const unsigned lds_base = (unsigned)(size_t)((__attribute__((address_space(3))) unsigned char *)lds);

    // ---- DMA role of this lane.  Wave-instruction i (0/1) of wave w fills the 1 KiB piece 2w+i of a half-tile: LDS rows
    // rho = 8(2w+i) + lane/8, physical chunk lane%8, which must hold logical chunk (lane%8) ^ ((rho>>1)&7).
    unsigned xo[2][2], wo[2][2], msk[2][2];           // [half][i]: byte offsets of the row's origin in x / w; tap validity bits
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int rho = 8 * (2 * wave + i) + (lane >> 3);
        const unsigned lc = (unsigned)((lane & 7) ^ ((rho >> 1) & 7));
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int m = m0 + (rho >> 6) * 128 + h * 64 + (rho & 63);
            const bool mv = m < p.M;
            const int mm = mv ? m : 0;
            const int wo_ = mm % p.Wo, t2 = mm / p.Wo, ho = t2 % p.Ho, b = t2 / p.Ho;
            const int hi0 = ho * p.stride - p.pad, wi0 = wo_ * p.stride - p.pad;
            unsigned bits = 0;
            if constexpr (TAPS) {
                for (int tap = 0; tap < p.KH * p.KW; ++tap) {
                    const int hi = hi0 + tap / p.KW, wi = wi0 + tap % p.KW;
                    if (mv && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W) bits |= 1u << tap;
                }
            }
            msk[h][i] = bits;
            const unsigned off = (unsigned)((((b * p.H + hi0) * p.W + wi0) * p.Cin) * 2) + lc * 16u;
            xo[h][i] = (TAPS || mv) ? off : P8_OOB;
            const int n = n0 + (rho >> 5) * 64 + h * 32 + (rho & 31);
            wo[h][i] = n < p.Cout ? (unsigned)n * (unsigned)p.Kpad * 2u + lc * 16u : P8_OOB;
        }
    }
    // buffer descriptors (base, num_records = the tensor's bytes, raw dword access)
    const size_t xbytes = (size_t)p.B * p.H * p.W * p.Cin * 2, wbytes = (size_t)p.Cout * p.Kpad * 2;
    const u4 rsx = {(unsigned)(size_t)p.x, (unsigned)((size_t)p.x >> 32) & 0xffffu, (unsigned)xbytes, 0x00020000u};
    const u4 rsw = {(unsigned)(size_t)p.w, (unsigned)((size_t)p.w >> 32) & 0xffffu, (unsigned)wbytes, 0x00020000u};
    const unsigned ldst = lds_base + (unsigned)(2 * wave) * 1024u;        // this wave's first piece within a half-tile buffer

    const int nk = p.Kpad / 64;
    auto issue_a = [&](int h, int slot, int u, const P8Kpos &kp, auto &&emit) {          // A half h of K tile u
        const unsigned so = u < nk ? kp.aoff(p) : P8_OOB;
        unsigned v0 = xo[h][0] + so, v1 = xo[h][1] + so;
        if constexpr (TAPS) {
            v0 = (msk[h][0] >> kp.tap) & 1u ? v0 : P8_OOB;
            v1 = (msk[h][1] >> kp.tap) & 1u ? v1 : P8_OOB;
        }
        const unsigned l = ldst + (unsigned)((u & 1) * P8_STAGE + slot);
        emit(v0, v1, rsx, l, l + 1024u);
    };
    auto issue_b = [&](int h, int slot, int u, auto &&emit) {                            // B half h of K tile u
        const unsigned so = u < nk ? (unsigned)u * 128u : P8_OOB;
        const unsigned l = ldst + (unsigned)((u & 1) * P8_STAGE + slot);
        emit(wo[h][0] + so, wo[h][1] + so, rsw, l, l + 1024u);
    };
    auto only_issue = [&](unsigned v0, unsigned v1, u4 rs, unsigned l0, unsigned l1) { p8_issue(v0, v1, rs, l0, l1); };

    // ---- fragment read addresses: tile row l15 of a 16-row tile, logical chunk 4kk+q -> physical (4kk+q) ^ (l15>>1)
    const unsigned sw0 = (unsigned)((q ^ (l15 >> 1)) * 16), sw1 = (unsigned)(((4 + q) ^ (l15 >> 1)) * 16);
    const unsigned fa_row = lds_base + (unsigned)(wr * 64 + l15) * 128u, fb_row = lds_base + (unsigned)(wc * 32 + l15) * 128u;

    v4f acc[4][8];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 8; ++b) acc[a][b] = v4f{0.f, 0.f, 0.f, 0.f};
    u4 fa[4][2], fb0[2][2], fb1[2][2];

    // ---- prologue: K tile 0 whole, A0 / B0 of K tile 1
    P8Kpos k1{0, 0, 0, 0};            // position of K tile t+1 (starts at tile 0 here, advanced below)
    issue_a(0, P8_A0, 0, k1, only_issue);
    issue_b(0, P8_B0, 0, only_issue);
    issue_b(1, P8_B1, 0, only_issue);
    issue_a(1, P8_A1, 0, k1, only_issue);
    k1.next(p);
    issue_a(0, P8_A0, 1, k1, only_issue);
    issue_b(0, P8_B0, 1, only_issue);
    P8Kpos k2 = k1;                   // position of K tile t+2
    k2.next(p);
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();          // stagger: waves 4-7 run one barrier behind waves 0-3
    __builtin_amdgcn_sched_barrier(0);

    for (int t = 0; t < nk; ++t) {
        const unsigned st = (unsigned)((t & 1) * P8_STAGE);
        const unsigned ra0 = fa_row + st + sw0, ra1 = fa_row + st + sw1, rb0 = fb_row + st + sw0, rb1 = fb_row + st + sw1;
        // phase 1: quadrant (A0, B0); prefetch B1 of tile t+1
        issue_b(1, P8_B1, t + 1, [&](unsigned v0, unsigned v1, u4 rs, unsigned l0, unsigned l1) {
            p8_load_a_b(fa, fb0, ra0, ra1, rb0, rb1, v0, v1, rs, l0, l1);
        });
        p8_mfma<0, 0>(acc, fa, fb0);
        // phase 2: quadrant (A0, B1); prefetch A1 of tile t+1
        issue_a(1, P8_A1, t + 1, k1, [&](unsigned v0, unsigned v1, u4 rs, unsigned l0, unsigned l1) {
            p8_load_b1(fb1, rb0, rb1, v0, v1, rs, l0, l1);
        });
        p8_mfma<0, 1>(acc, fa, fb1);
        // phase 3: quadrant (A1, B1); prefetch A0 of tile t+2
        issue_a(0, P8_A0, t + 2, k2, [&](unsigned v0, unsigned v1, u4 rs, unsigned l0, unsigned l1) {
            p8_load_a1(fa, ra0, ra1, v0, v1, rs, l0, l1);
        });
        p8_mfma<1, 1>(acc, fa, fb1);
        // phase 4: quadrant (A1, B0); prefetch B0 of tile t+2
        issue_b(0, P8_B0, t + 2, [&](unsigned v0, unsigned v1, u4 rs, unsigned l0, unsigned l1) {
            p8_load_none(v0, v1, rs, l0, l1);
        });
        p8_mfma<1, 0>(acc, fa, fb0);
        k1 = k2;
        k2.next(p);
    }
    if (wr == 0) __builtin_amdgcn_s_barrier();          // waves 0-3 catch up
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // the past-the-end zero fills must not land on the epilogue's staging
    __syncthreads();
    epilogue<T, 256, 256, 2, 4, DG>(p, acc, lds, mtile % NSH, n0, [&](int row) { const int m = m0 + row; return m < p.M ? (long)m : -1l; });
}

// the geometry this kernel takes: bf16, Cin a multiple of the 64-deep K tile, at most 32 taps, operands below 2 GiB
bool p8_supports(const Params &p) {
    return p.Cin % 64 == 0 && p.KH * p.KW <= 32 && !p.in_scale && !p.x2 &&
           (size_t)p.B * p.H * p.W * p.Cin * 2 < 0x7fff0000ull && (size_t)p.Cout * p.Kpad * 2 < 0x7fff0000ull;
}

int launch_p8(const Params &p, hipStream_t s) {
    const dim3 grid((p.M + 255) / 256, (p.Cout + 255) / 256), block(512);
    const bool taps = !(p.KH == 1 && p.KW == 1 && p.pad == 0);
    if (p.mask) {
        if (taps) hipLaunchKernelGGL((conv_p8_kernel<true, true>), grid, block, 0, s, p);
        else hipLaunchKernelGGL((conv_p8_kernel<true, false>), grid, block, 0, s, p);
    } else {
        if (taps) hipLaunchKernelGGL((conv_p8_kernel<false, true>), grid, block, 0, s, p);
        else hipLaunchKernelGGL((conv_p8_kernel<false, false>), grid, block, 0, s, p);
    }
    return check_launch("conv_p8_kernel");
}

}}  // namespace mhe::conv

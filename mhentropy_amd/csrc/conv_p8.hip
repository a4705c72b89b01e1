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
//    0-31 / 32-63 of every wave column) into a two-stage ring; its 64 MFMAs per wave run as two PHASES of 32:
//    phase 1 reads the fragments of A0, B0, B1 (16 ds_read_b128) and computes the quadrants (A0,B0) (A0,B1); phase 2 reads A1 (8)
//    and computes (A1,B1) (A1,B0).  Phase 1 also issues the DMA of A1 of the NEXT K tile, phase 2 that of A0, B0, B1 of the
//    tile after it; every phase retires, with a COUNTED s_waitcnt vmcnt(8), what was issued two phases earlier - one whole K tile
//    (64 KiB per CU) is in flight across the barriers at all times, vmcnt never drains to 0 inside the loop;
//  * ONE s_barrier per phase (measured: a barrier costs ~100 ns = the time of 12 MFMAs, and a first version with two per 16
//    MFMAs spent as long in barriers as in the matrix pipe), and the two wave rows run it at different points of their phase:
//    waves 0-3 do  load, MFMA, barrier;  waves 4-7 do  load, barrier, MFMA  - so between two barriers every SIMD has one wave
//    loading and then computing and its partner computing (the previous phase) and then loading.
//
// Hazards, per half-tile buffer (g = global phase number, barrier g ends phase g's load part for every wave):
//   RAW  every wave waits (vmcnt) for its share of the DMA inside the load part of phase g, i.e. before barrier g; the
//        fragments are read in phase g+1, after barrier g.
//   WAR  the load part ends with lgkmcnt(0) BEFORE barrier r, so a buffer last read in phase r is free after barrier r and is
//        re-staged in phase r+1 at the earliest (A1: read 2t+2, DMA 2t+3; A0/B0/B1: read 2t+1, DMA 2t+2).
#include "conv_shared.h"

namespace mhe { namespace conv {

typedef unsigned u4 __attribute__((ext_vector_type(4)));
typedef __attribute__((ext_vector_type(8))) __bf16 bf8;

constexpr unsigned P8_OOB = 0x80000000u;      // an offset no descriptor of < 2 GiB admits
constexpr int P8_STAGE = 65536;      // per stage: A0 at 0, B0 at 16384, B1 at 32768, A1 at 49152 (the literals in the asm strings)

// ---- the load part of a phase as ONE asm statement (hipcc neither counts nor reorders what is inside): fragment reads,
// DMA pieces of the half-tiles to prefetch (M0 = the DMA's LDS base, written and used inside), the counted wait that
// retires what was issued two phases ago, and the wait for the fragments.
// ABL (tuning builds only, 0 in production): 1 = no DMA inside the loop, 2 = no fragment reads, 4 = no MFMA, 8 = no epilogue
#define P8_PIECE(V, RS, OFF) "s_add_u32 m0, %[lb], " #OFF "\n\ts_nop 0\n\tbuffer_load_dwordx4 %[" #V "], %[" #RS "], 0 offen lds\n\t"
#define P8_WAIT "s_waitcnt vmcnt(8) lgkmcnt(0)"
#define P8_RD_1                                                                                       \
    "ds_read_b128 %[a00], %[ra0]\n\tds_read_b128 %[c00], %[rb0] offset:16384\n\t"                      \
    "ds_read_b128 %[a10], %[ra0] offset:2048\n\tds_read_b128 %[c10], %[rb0] offset:18432\n\t"          \
    "ds_read_b128 %[a20], %[ra0] offset:4096\n\tds_read_b128 %[a30], %[ra0] offset:6144\n\t"           \
    "ds_read_b128 %[a01], %[ra1]\n\tds_read_b128 %[c01], %[rb1] offset:16384\n\t"                      \
    "ds_read_b128 %[a11], %[ra1] offset:2048\n\tds_read_b128 %[c11], %[rb1] offset:18432\n\t"          \
    "ds_read_b128 %[a21], %[ra1] offset:4096\n\tds_read_b128 %[a31], %[ra1] offset:6144\n\t"           \
    "ds_read_b128 %[d00], %[rb0] offset:32768\n\tds_read_b128 %[d10], %[rb0] offset:34816\n\t"         \
    "ds_read_b128 %[d01], %[rb1] offset:32768\n\tds_read_b128 %[d11], %[rb1] offset:34816\n\t"
#define P8_RD_2                                                                                       \
    "ds_read_b128 %[a00], %[ra0] offset:49152\n\tds_read_b128 %[a10], %[ra0] offset:51200\n\t"         \
    "ds_read_b128 %[a20], %[ra0] offset:53248\n\tds_read_b128 %[a30], %[ra0] offset:55296\n\t"         \
    "ds_read_b128 %[a01], %[ra1] offset:49152\n\tds_read_b128 %[a11], %[ra1] offset:51200\n\t"         \
    "ds_read_b128 %[a21], %[ra1] offset:53248\n\tds_read_b128 %[a31], %[ra1] offset:55296\n\t"
// 256 x 128 form (NBH = 1): the wave columns are 32 channels wide, B0 holds all 128 channel rows, there is no B1: phase 1 reads A0 + B0
// (12 fragments) for 16 MFMAs, phase 2 reads A1 (8) for 16; six 1 KiB DMA pieces per wave and K tile instead of eight -> vmcnt(6)
#define P8_WAIT6 "s_waitcnt vmcnt(6) lgkmcnt(0)"
#define P8_RD_1H                                                                                      \
    "ds_read_b128 %[a00], %[ra0]\n\tds_read_b128 %[c00], %[rb0] offset:16384\n\t"                      \
    "ds_read_b128 %[a10], %[ra0] offset:2048\n\tds_read_b128 %[c10], %[rb0] offset:18432\n\t"          \
    "ds_read_b128 %[a20], %[ra0] offset:4096\n\tds_read_b128 %[a30], %[ra0] offset:6144\n\t"           \
    "ds_read_b128 %[a01], %[ra1]\n\tds_read_b128 %[c01], %[rb1] offset:16384\n\t"                      \
    "ds_read_b128 %[a11], %[ra1] offset:2048\n\tds_read_b128 %[c11], %[rb1] offset:18432\n\t"          \
    "ds_read_b128 %[a21], %[ra1] offset:4096\n\tds_read_b128 %[a31], %[ra1] offset:6144\n\t"
#define P8_DMA_2H "s_mov_b32 %[keep], m0\n\t" P8_PIECE(v0, rs, 0) P8_PIECE(v1, rs, 1024) P8_PIECE(v2, rw, 16384) P8_PIECE(v3, rw, 17408) \
                  "s_mov_b32 m0, %[keep]\n\t"
#define P8_OUT_BH [c00] "=&v"(fb0[0][0]), [c01] "=&v"(fb0[0][1]), [c10] "=&v"(fb0[1][0]), [c11] "=&v"(fb0[1][1])
#define P8_DMA_1 "s_mov_b32 %[keep], m0\n\t" P8_PIECE(v0, rs, 49152) P8_PIECE(v1, rs, 50176) "s_mov_b32 m0, %[keep]\n\t"
#define P8_DMA_2 "s_mov_b32 %[keep], m0\n\t" P8_PIECE(v0, rs, 0) P8_PIECE(v1, rs, 1024) P8_PIECE(v2, rw, 16384) P8_PIECE(v3, rw, 17408) \
                 P8_PIECE(v4, rw, 32768) P8_PIECE(v5, rw, 33792) "s_mov_b32 m0, %[keep]\n\t"
#define P8_OUT_A [a00] "=&v"(fa[0][0]), [a01] "=&v"(fa[0][1]), [a10] "=&v"(fa[1][0]), [a11] "=&v"(fa[1][1]), \
                 [a20] "=&v"(fa[2][0]), [a21] "=&v"(fa[2][1]), [a30] "=&v"(fa[3][0]), [a31] "=&v"(fa[3][1])
#define P8_OUT_B [c00] "=&v"(fb0[0][0]), [c01] "=&v"(fb0[0][1]), [c10] "=&v"(fb0[1][0]), [c11] "=&v"(fb0[1][1]), \
                 [d00] "=&v"(fb1[0][0]), [d01] "=&v"(fb1[0][1]), [d10] "=&v"(fb1[1][0]), [d11] "=&v"(fb1[1][1])

// phase 1: fragments of A half 0 (4 pixel tiles x 2 k-steps) and of both B halves (2 x 2 each); DMA of one half-tile
template <int ABL>
__device__ __forceinline__ void p8_load_1(u4 (&fa)[4][2], u4 (&fb0)[2][2], u4 (&fb1)[2][2], unsigned ra0, unsigned ra1, unsigned rb0,
                                          unsigned rb1, unsigned v0, unsigned v1, u4 rs, unsigned lb) {
    unsigned keep;
#define P8_INS [ra0] "v"(ra0), [ra1] "v"(ra1), [rb0] "v"(rb0), [rb1] "v"(rb1), [v0] "v"(v0), [v1] "v"(v1), [rs] "s"(rs), [lb] "s"(lb)
    if constexpr ((ABL & 3) == 0) asm volatile(P8_RD_1 P8_DMA_1 P8_WAIT : P8_OUT_A, P8_OUT_B, [keep] "=&s"(keep) : P8_INS : "memory");
    else if constexpr ((ABL & 3) == 1) asm volatile(P8_RD_1 P8_WAIT : P8_OUT_A, P8_OUT_B, [keep] "=&s"(keep) : P8_INS : "memory");
    else if constexpr ((ABL & 3) == 2) asm volatile("s_nop 4\n\t" P8_DMA_1 P8_WAIT : P8_OUT_A, P8_OUT_B, [keep] "=&s"(keep) : P8_INS : "memory");
    else asm volatile(P8_WAIT : P8_OUT_A, P8_OUT_B, [keep] "=&s"(keep) : P8_INS : "memory");
#undef P8_INS
}
// phase 2: fragments of A half 1; DMA of three half-tiles (A0 from x; B0, B1 from w)
template <int ABL>
__device__ __forceinline__ void p8_load_2(u4 (&fa)[4][2], unsigned ra0, unsigned ra1, unsigned v0, unsigned v1, unsigned v2, unsigned v3,
                                          unsigned v4, unsigned v5, u4 rs, u4 rw, unsigned lb) {
    unsigned keep;
#define P8_INS [ra0] "v"(ra0), [ra1] "v"(ra1), [v0] "v"(v0), [v1] "v"(v1), [v2] "v"(v2), [v3] "v"(v3), [v4] "v"(v4), [v5] "v"(v5), \
               [rs] "s"(rs), [rw] "s"(rw), [lb] "s"(lb)
    if constexpr ((ABL & 3) == 0) asm volatile(P8_RD_2 P8_DMA_2 P8_WAIT : P8_OUT_A, [keep] "=&s"(keep) : P8_INS : "memory");
    else if constexpr ((ABL & 3) == 1) asm volatile(P8_RD_2 P8_WAIT : P8_OUT_A, [keep] "=&s"(keep) : P8_INS : "memory");
    else if constexpr ((ABL & 3) == 2) asm volatile("s_nop 4\n\t" P8_DMA_2 P8_WAIT : P8_OUT_A, [keep] "=&s"(keep) : P8_INS : "memory");
    else asm volatile(P8_WAIT : P8_OUT_A, [keep] "=&s"(keep) : P8_INS : "memory");
#undef P8_INS
}
// the 256 x 128 form's load parts
__device__ __forceinline__ void p8h_load_1(u4 (&fa)[4][2], u4 (&fb0)[2][2], unsigned ra0, unsigned ra1, unsigned rb0, unsigned rb1, unsigned v0, unsigned v1,
                                           u4 rs, unsigned lb) {
    unsigned keep;
    asm volatile(P8_RD_1H P8_DMA_1 P8_WAIT6 : P8_OUT_A, P8_OUT_BH, [keep] "=&s"(keep)
                 : [ra0] "v"(ra0), [ra1] "v"(ra1), [rb0] "v"(rb0), [rb1] "v"(rb1), [v0] "v"(v0), [v1] "v"(v1), [rs] "s"(rs), [lb] "s"(lb) : "memory");
}
__device__ __forceinline__ void p8h_load_2(u4 (&fa)[4][2], unsigned ra0, unsigned ra1, unsigned v0, unsigned v1, unsigned v2, unsigned v3, u4 rs, u4 rw,
                                           unsigned lb) {
    unsigned keep;
    asm volatile(P8_RD_2 P8_DMA_2H P8_WAIT6 : P8_OUT_A, [keep] "=&s"(keep)
                 : [ra0] "v"(ra0), [ra1] "v"(ra1), [v0] "v"(v0), [v1] "v"(v1), [v2] "v"(v2), [v3] "v"(v3), [rs] "s"(rs), [rw] "s"(rw), [lb] "s"(lb) : "memory");
}
__device__ __forceinline__ void p8h_issue_2(unsigned v0, unsigned v1, unsigned v2, unsigned v3, u4 rs, u4 rw, unsigned lb) {
    unsigned keep;
    asm volatile("s_nop 4\n\t" P8_DMA_2H "s_nop 0" : [keep] "=&s"(keep)
                 : [v0] "v"(v0), [v1] "v"(v1), [v2] "v"(v2), [v3] "v"(v3), [rs] "s"(rs), [rw] "s"(rw), [lb] "s"(lb) : "memory");
}
// 16 MFMAs of one phase of the 256 x 128 form: pixel tiles 4*HA + {0..3} against this wave's two channel tiles, both k-steps
template <int HA>
__device__ __forceinline__ void p8h_mfma(v4f (&acc)[2][8], const u4 (&fa)[4][2], const u4 (&fb0)[2][2], int wr) {
    __builtin_amdgcn_sched_barrier(0);
    if (wr) __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
                acc[nt][4 * HA + mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8, fb0[nt][kk]), __builtin_bit_cast(bf8, fa[mt][kk]),
                                                                               acc[nt][4 * HA + mt], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    if (!wr) __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
}

// prologue pieces: issue only
__device__ __forceinline__ void p8_issue_1(unsigned v0, unsigned v1, u4 rs, unsigned lb) {
    unsigned keep;
    asm volatile("s_nop 4\n\t" P8_DMA_1 "s_nop 0" : [keep] "=&s"(keep) : [v0] "v"(v0), [v1] "v"(v1), [rs] "s"(rs), [lb] "s"(lb) : "memory");
}
__device__ __forceinline__ void p8_issue_2(unsigned v0, unsigned v1, unsigned v2, unsigned v3, unsigned v4, unsigned v5, u4 rs, u4 rw,
                                           unsigned lb) {
    unsigned keep;
    asm volatile("s_nop 4\n\t" P8_DMA_2 "s_nop 0" : [keep] "=&s"(keep)
                 : [v0] "v"(v0), [v1] "v"(v1), [v2] "v"(v2), [v3] "v"(v3), [v4] "v"(v4), [v5] "v"(v5), [rs] "s"(rs), [rw] "s"(rw), [lb] "s"(lb)
                 : "memory");
}

// 32 MFMAs of one phase: pixel tiles 4*HA + {0..3} against channel tiles {0,1} (fbx) and {2,3} (fby), both 32-deep k-steps;
// LATE = the barrier of the phase comes after the MFMAs (waves 0-3) instead of before them (waves 4-7)
template <int HA, int ABL>
__device__ __forceinline__ void p8_mfma(v4f (&acc)[4][8], const u4 (&fa)[4][2], const u4 (&fb0)[2][2], const u4 (&fb1)[2][2], int wr) {
    __builtin_amdgcn_sched_barrier(0);
    if (wr) __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
    if constexpr (ABL & 4) {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) { asm volatile("" :: "v"(fb0[nt][kk])); asm volatile("" :: "v"(fb1[nt][kk])); }
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) asm volatile("" :: "v"(fa[mt][kk]));
        }
    } else {
#pragma unroll
        for (int hb = 0; hb < 2; ++hb)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt) {
                        const int hbe = HA ? 1 - hb : hb;       // phase 2 starts with the B half whose fragments were read last
                        const u4 b = hbe ? fb1[nt][kk] : fb0[nt][kk];
                        acc[2 * hbe + nt][4 * HA + mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                            __builtin_bit_cast(bf8, b), __builtin_bit_cast(bf8, fa[mt][kk]), acc[2 * hbe + nt][4 * HA + mt], 0, 0, 0);
                    }
    }
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    if (!wr) __builtin_amdgcn_s_barrier();
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

// NBH = B half-tiles per K tile: 2 = the 256 x 256 tile; 1 = a 256 x 128 tile (round 3: the 3x3 layers with 128 output channels and the
// 512-channel ones of layer4, whose 16k pixels make only 128 tiles of 256 x 256 - see the macros above)
template <bool DG, bool TAPS, int ABL = 0, int NBH = 2>
__global__ __launch_bounds__(512) void conv_p8_kernel(const Params p) {
    using T = u16;
    __shared__ uint4 lds[2 * P8_STAGE / 16];          // 128 KiB: the two-stage ring, then the epilogue's staging

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = lane >> 4, l15 = lane & 15;
    const int wr = wave >> 2, wc = wave & 3;
    int mtile, ntile;
    tile_of_block(mtile, ntile);
    const int m0 = mtile * 256, n0 = ntile * (128 * NBH);
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
            int hi0, wi0, b;
            if (!TAPS && p.stride == 1) {     // 1x1, stride 1, no padding: input pixel = output pixel, no index arithmetic
                b = 0; hi0 = 0; wi0 = mm;
            } else {
                const int wo_ = mm % p.Wo, t2 = mm / p.Wo, ho = t2 % p.Ho;
                b = t2 / p.Ho;
                hi0 = ho * p.stride - p.pad; wi0 = wo_ * p.stride - p.pad;
            }
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
            const int n = NBH == 2 ? n0 + (rho >> 5) * 64 + h * 32 + (rho & 31) : n0 + rho;      // (NBH = 1: B0 row = channel, h = 1 unused)
            wo[h][i] = n < p.Cout ? (unsigned)n * (unsigned)p.Kpad * 2u + lc * 16u : P8_OOB;
        }
    }
    // buffer descriptors (base, num_records = the tensor's bytes, raw dword access)
    const size_t xbytes = (size_t)p.B * p.H * p.W * p.Cin * 2, wbytes = (size_t)p.Cout * p.Kpad * 2;
    const u4 rsx = {(unsigned)(size_t)p.x, (unsigned)((size_t)p.x >> 32) & 0xffffu, (unsigned)xbytes, 0x00020000u};
    const u4 rsw = {(unsigned)(size_t)p.w, (unsigned)((size_t)p.w >> 32) & 0xffffu, (unsigned)wbytes, 0x00020000u};
    const unsigned ldst = lds_base + (unsigned)(2 * wave) * 1024u;        // this wave's first piece within a half-tile buffer

    const int nk = p.Kpad / 64;
    // A half h of K tile u (position kp): the two DMA offsets of this lane
    auto a_offs = [&](int h, int u, const P8Kpos &kp, unsigned &v0, unsigned &v1) {
        const unsigned so = u < nk ? kp.aoff(p) : P8_OOB;
        v0 = xo[h][0] + so; v1 = xo[h][1] + so;
        if constexpr (TAPS) {
            v0 = (msk[h][0] >> kp.tap) & 1u ? v0 : P8_OOB;
            v1 = (msk[h][1] >> kp.tap) & 1u ? v1 : P8_OOB;
        }
    };
    auto b_offs = [&](int h, int u, unsigned &v0, unsigned &v1) {
        const unsigned so = u < nk ? (unsigned)u * 128u : P8_OOB;
        v0 = wo[h][0] + so; v1 = wo[h][1] + so;
    };
    auto stage_of = [&](int u) { return ldst + (unsigned)((u & 1) * P8_STAGE); };
    auto issue_x = [&](int u, const P8Kpos &kp) {          // A0, B0 (, B1) of K tile u
        unsigned v0, v1, v2, v3, v4, v5;
        a_offs(0, u, kp, v0, v1); b_offs(0, u, v2, v3);
        if constexpr (NBH == 2) { b_offs(1, u, v4, v5); p8_issue_2(v0, v1, v2, v3, v4, v5, rsx, rsw, stage_of(u)); }
        else p8h_issue_2(v0, v1, v2, v3, rsx, rsw, stage_of(u));
    };

    // ---- fragment read addresses: tile row l15 of a 16-row tile, logical chunk 4kk+q -> physical (4kk+q) ^ (l15>>1)
    const unsigned sw0 = (unsigned)((q ^ (l15 >> 1)) * 16), sw1 = (unsigned)(((4 + q) ^ (l15 >> 1)) * 16);
    const unsigned fa_row = lds_base + (unsigned)(wr * 64 + l15) * 128u, fb_row = lds_base + (unsigned)(wc * 32 + l15) * 128u;

    v4f acc[2 * NBH][8];
#pragma unroll
    for (int a = 0; a < 2 * NBH; ++a)
#pragma unroll
        for (int b = 0; b < 8; ++b) acc[a][b] = v4f{0.f, 0.f, 0.f, 0.f};
    u4 fa[4][2], fb0[2][2], fb1[2][2];

    // ---- prologue: K tile 0 whole, then A0 / B0 / B1 of K tile 1 (issue order = the order the loop continues in)
    P8Kpos k1{0, 0, 0, 0};            // position of K tile t+1 (starts at tile 0 here, advanced below)
    issue_x(0, k1);
    {
        unsigned v0, v1;
        a_offs(1, 0, k1, v0, v1);
        p8_issue_1(v0, v1, rsx, stage_of(0));
    }
    k1.next(p);
    issue_x(1, k1);
    P8Kpos k2 = k1;                   // position of K tile t+2
    k2.next(p);
    if constexpr (NBH == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");          // 4 + 2 + 4 pieces issued: K tile 0's A0 / B0 have landed
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);

    for (int t = 0; t < nk; ++t) {
        const unsigned st = (unsigned)((t & 1) * P8_STAGE);
        const unsigned ra0 = fa_row + st + sw0, ra1 = fa_row + st + sw1, rb0 = fb_row + st + sw0, rb1 = fb_row + st + sw1;
        if constexpr (NBH == 2) {
            {   // phase 1: quadrants (A0, B0), (A0, B1); prefetch A1 of tile t+1
                unsigned v0, v1;
                a_offs(1, t + 1, k1, v0, v1);
                p8_load_1<ABL>(fa, fb0, fb1, ra0, ra1, rb0, rb1, v0, v1, rsx, stage_of(t + 1));
                p8_mfma<0, ABL>(acc, fa, fb0, fb1, wr);
            }
            {   // phase 2: quadrants (A1, B1), (A1, B0); prefetch A0, B0, B1 of tile t+2
                unsigned v0, v1, v2, v3, v4, v5;
                a_offs(0, t + 2, k2, v0, v1); b_offs(0, t + 2, v2, v3); b_offs(1, t + 2, v4, v5);
                p8_load_2<ABL>(fa, ra0, ra1, v0, v1, v2, v3, v4, v5, rsx, rsw, stage_of(t + 2));
                p8_mfma<1, ABL>(acc, fa, fb0, fb1, wr);
            }
        } else {
            {   // phase 1: (A0, B0); prefetch A1 of tile t+1
                unsigned v0, v1;
                a_offs(1, t + 1, k1, v0, v1);
                p8h_load_1(fa, fb0, ra0, ra1, rb0, rb1, v0, v1, rsx, stage_of(t + 1));
                p8h_mfma<0>(acc, fa, fb0, wr);
            }
            {   // phase 2: (A1, B0); prefetch A0, B0 of tile t+2
                unsigned v0, v1, v2, v3;
                a_offs(0, t + 2, k2, v0, v1); b_offs(0, t + 2, v2, v3);
                p8h_load_2(fa, ra0, ra1, v0, v1, v2, v3, rsx, rsw, stage_of(t + 2));
                p8h_mfma<1>(acc, fa, fb0, wr);
            }
        }
        k1 = k2;
        k2.next(p);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // the past-the-end zero fills must not land on the epilogue's staging
    __syncthreads();
    if constexpr (ABL & 8) {
#pragma unroll
        for (int a = 0; a < 2 * NBH; ++a)
#pragma unroll
            for (int b = 0; b < 8; ++b) asm volatile("" :: "v"(acc[a][b]));
        return;
    }
    epilogue<T, 256, 128 * NBH, 2, 4, DG>(p, acc, lds, mtile % NSH, n0, [&](int row) { return out_pixel(p, m0 + row); });
}

// the geometry this kernel takes: bf16, Cin a multiple of the 64-deep K tile, at most 32 taps, operands below 2 GiB
bool p8_supports(const Params &p) {
    return p.Cin % 64 == 0 && p.KH * p.KW <= 32 && !p.in_scale && !p.x2 &&
           (size_t)p.B * p.H * p.W * p.Cin * 2 < 0x7fff0000ull && (size_t)p.Cout * p.Kpad * 2 < 0x7fff0000ull;
}

// the 256 x 128 form (variant 13)
int launch_p8h(const Params &p, hipStream_t s) {
    const dim3 grid((p.M + 255) / 256, (p.Cout + 127) / 128), block(512);
    const bool taps = !(p.KH == 1 && p.KW == 1 && p.pad == 0);
    if (p.mask) {
        if (taps) hipLaunchKernelGGL((conv_p8_kernel<true, true, 0, 1>), grid, block, 0, s, p);
        else hipLaunchKernelGGL((conv_p8_kernel<true, false, 0, 1>), grid, block, 0, s, p);
    } else {
        if (taps) hipLaunchKernelGGL((conv_p8_kernel<false, true, 0, 1>), grid, block, 0, s, p);
        else hipLaunchKernelGGL((conv_p8_kernel<false, false, 0, 1>), grid, block, 0, s, p);
    }
    return check_launch("conv_p8_kernel<256x128>");
}

int launch_p8(const Params &p, hipStream_t s) {
    const dim3 grid((p.M + 255) / 256, (p.Cout + 255) / 256), block(512);
    const bool taps = !(p.KH == 1 && p.KW == 1 && p.pad == 0);
#ifdef MHE_P8_ABLATIONS
    if (p.force > 7) {       // tuning builds: variant 7 + 16 * ABL
        switch (p.force >> 4) {
#define P8_ABL_CASE(A) case A: if (taps) hipLaunchKernelGGL((conv_p8_kernel<false, true, A>), grid, block, 0, s, p); \
                               else hipLaunchKernelGGL((conv_p8_kernel<false, false, A>), grid, block, 0, s, p); break;
            P8_ABL_CASE(1) P8_ABL_CASE(2) P8_ABL_CASE(3) P8_ABL_CASE(4) P8_ABL_CASE(5) P8_ABL_CASE(6) P8_ABL_CASE(7) P8_ABL_CASE(8) P8_ABL_CASE(12)
            default: set_error("conv_p8: no ablation build %d", p.force >> 4); return MHE_ERR_ARG;
        }
        return check_launch("conv_p8_kernel");
    }
#endif
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

// 3x3 / stride-1 / pad-1 bf16 convolution with the input tile resident in LDS (variant 14): conv2 of the bottlenecks of layer2 / layer3
// (32 x 32 and 16 x 16 images, 128 / 256 channels) and their data gradients; reference: torchvision Bottleneck.conv2 via hand/network.py:54-61,110.
//
// Why its own kernel.  The im2col kernels (conv.hip, conv_p8.hip) bring every input element to LDS once per TAP - nine times - and can only
// take the producer's BatchNorm + ReLU as a pass of its own over the tensor (bn_act_kernel: 17 launches, 2 GB and 0.47 ms of the forward
// step) or nine times over in their operand load.  Here a workgroup keeps the (rows + 2) x (W + 2) halo of its 256 output pixels in LDS, 64
// channels at a time; a tap is an address offset into that image, and every element is loaded, normalised and stored to LDS ONCE per workgroup:
//   multiply waves 0-7  : 64 pixels x 64 output channels each (2 x 2 v_mfma_f32_32x32x16_bf16 tiles); a STEP = one tap of one 64-channel
//                         chunk = 16 MFMAs per wave.  They also stream the weights: fragment-major 16 KiB stages (mhe_conv3x3_halo_pack_bf16)
//                         by LDS-DMA into a 4-stage ring, three steps ahead, behind a counted vmcnt (they issue no other VMEM in the loop);
//   transfer waves 8-11 : the NEXT chunk's halo from global memory through registers - all of a chunk's pieces in flight from the first step
//                         of the chunk before, written from its fourth step on (BatchNorm + ReLU of the producer applied on the way, zeros at
//                         the image border) into the other halo buffer; optionally the normalised tensor out to a_out once (the train
//                         step's weight-gradient operand).
// Workgroups are persistent (one per CU, XCD-aware tile order): the next tile's first chunk is staged while this tile's last chunk multiplies.
// One barrier per step.  32 x 32 MFMAs because their operand lanes of one LDS lane group all read the same 16-byte slot of 16 different
// pixels - 16 different image columns mod 16, on one row or two: with the slot XOR-swizzled by the halo COLUMN (pitch W + 2 is even, so the
// column's parity is the pixel's) the shifted (tap) reads are conflict-free for any shift (keyed by the pixel index the two-row fragments
// of the 16-wide maps were 2-way on 2 of 16 banks: 34 % of the LDS cycles).
// Epilogue: the shared one (conv_shared.h), staged through the weight ring; statistics / gate / BatchNorm-reverse sums as everywhere.
#include "conv_shared.h"

#ifndef MHE_HALO_ABL
#define MHE_HALO_ABL 0
#endif

namespace mhe { namespace conv {

namespace {

typedef unsigned u4 __attribute__((ext_vector_type(4)));
typedef __attribute__((ext_vector_type(8))) __bf16 bf8;
typedef float v16f __attribute__((ext_vector_type(16)));

constexpr int HBM = 256, HBN = 128;
constexpr int H_STAGE = 16384, H_RING = 4;            // one (chunk, tap): 64 x 128 bf16; stages in the ring
constexpr int H_MAXC = 512;
constexpr unsigned H_OOB = 0x80000000u;

template <int W> struct Geo {
    static constexpr int R = HBM / W;                  // output rows per tile
    static constexpr int PW = W + 2, HP = (R + 2) * PW; // halo pixels
    static constexpr int HALO = HP * 128;               // bytes per 64-channel chunk
    static constexpr int NP = HP * 8;                   // 16-byte pieces per chunk
    static constexpr int NJ = (NP + 255) / 256;         // pieces per transfer lane
};

}  // namespace

template <int W, bool DG>
__global__ __launch_bounds__(768) void conv_halo_kernel(const Params p) {
    using T = u16;
    using G = Geo<W>;
    constexpr int R = G::R, PW = G::PW, HALO = G::HALO, NP = G::NP, NJ = G::NJ;
    static_assert(NJ <= 12, "six load steps of two pieces");
    __shared__ uint4 lds[(H_RING * H_STAGE + 2 * HALO) / 16];
    __shared__ float aff[3 * H_MAXC];                  // scale | shift of the producer's BatchNorm, or k2 | k1 | k0 of the BatchNorm reverse on the load
    unsigned char *const ring = reinterpret_cast<unsigned char *>(lds), *const halo0 = ring + H_RING * H_STAGE;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int C = p.Cin, nC = C / 64, Tn = nC * 9;
    const int gm = p.M / HBM, gn = p.Cout / HBN, ntiles = gm * gn;
    const int TPI = p.H / R;                           // tiles per image
    auto tile_at = [&](int L, int &mt, int &nt) __attribute__((always_inline)) {
        if (gn > 1 && (gm & 7) == 0) {
            const int slot = L >> 3;
            nt = slot % gn;
            mt = (slot / gn) * 8 + (L & 7);
        } else { mt = L % gm; nt = L / gm; }
    };
    const bool bn = p.in_scale != nullptr;
    const bool rev = DG && p.rev_coef != nullptr;      // operand = k2 x + k1 x2 + k0 (bn_bwd_apply_kernel's arithmetic, trunk_bwd.hip)
    if (rev) for (int i = tid; i < C; i += 768) { aff[i] = p.rev_coef[i]; aff[H_MAXC + i] = p.rev_coef[C + i]; aff[2 * H_MAXC + i] = p.rev_coef[2 * C + i]; }
    constexpr int abl = MHE_HALO_ABL; // tuning builds (tools/halo_abl.sh): 1 no MFMA, 2 no fragment reads, 4 no weight DMA, 8 no halo staging, 16 no output walk
    if (bn) for (int i = tid; i < C; i += 768) { aff[i] = p.in_scale[i]; aff[H_MAXC + i] = p.in_shift[i]; }
    const int nbar_epi = (MHE_HALO_ABL & 16) ? 3 : 3 + (!DG && p.stats ? 2 : 0) + (DG && BN_EPILOGUE && p.bn_y[0] ? 2 : 0);
    __syncthreads();

    if (wave < 8) {
        // ------------------------------------------------------------------ multiply role
        const int l31 = lane & 31, kg = lane >> 5;
        const int wm = wave >> 1, wn = wave & 1;
        int hpb[2], xcb[2];                            // halo pixel index / halo column of this lane's pixel of row tile mt (tap 0, 0)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const int pm = wm * 64 + mt * 32 + l31;
            hpb[mt] = (pm / W + 1) * PW + (pm % W) + 1;
            xcb[mt] = (pm % W) + 1;
        }
        const size_t wbytes = (size_t)gn * Tn * H_STAGE;
        const u4 rsw = {(unsigned)(size_t)p.w, (unsigned)((size_t)p.w >> 32) & 0xffffu, (unsigned)wbytes, 0x00020000u};
        const unsigned ring_lds = (unsigned)(size_t)((__attribute__((address_space(3))) unsigned char *)ring);
        int gc = 0;                                    // chunks done by this workgroup: parity = halo buffer
        for (int L = blockIdx.x; L < ntiles; L += (int)gridDim.x) {
            int mtile, ntile;
            tile_at(L, mtile, ntile);
            auto issue = [&](int tau) __attribute__((always_inline)) {
                const unsigned off = tau < Tn ? (unsigned)((ntile * Tn + tau) * H_STAGE + wave * 2048 + lane * 16) : H_OOB;
                const unsigned off1 = tau < Tn ? off + 1024u : H_OOB;
                const unsigned lb = ring_lds + (unsigned)((tau & 3) * H_STAGE + wave * 2048);
                unsigned keep;
                asm volatile("s_mov_b32 %[keep], m0\n\ts_mov_b32 m0, %[lb]\n\ts_nop 0\n\tbuffer_load_dwordx4 %[v0], %[rs], 0 offen lds\n\t"
                             "s_add_u32 m0, %[lb], 1024\n\ts_nop 0\n\tbuffer_load_dwordx4 %[v1], %[rs], 0 offen lds\n\ts_mov_b32 m0, %[keep]"
                             : [keep] "=&s"(keep) : [v0] "v"(off), [v1] "v"(off1), [rs] "s"(rsw), [lb] "s"(lb) : "memory");
            };
            v16f acc[2][2];
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;
            if constexpr (!(abl & 4)) { issue(0); issue(1); issue(2); }
            __builtin_amdgcn_s_barrier();                                     // (pairs with the transfer waves' first barrier of the tile)
            __builtin_amdgcn_sched_barrier(0);
            // A step: this wave's share of stage tau has landed (issued three steps ago; tau + 1, tau + 2 stay in flight); barrier (everyone's
            // share, and at a chunk's first step its halo); DMA of stage tau + 3 into the buffer read during step tau - 1; then the 16 fragment
            // reads and 16 MFMAs, k-step by k-step, interleaved by the compiler.  Measured alternatives, all slower (EXPERIMENTS.md): all 16
            // reads up front behind counted waits (+10 %: the eight waves' reads arrive at the LDS together and nothing multiplies meanwhile);
            // conv_p8.hip's staggered phases (waves 0-3 load, MFMA, barrier; waves 4-7 load, barrier, MFMA: +5 % - reading before the barrier
            // shortens the DMA's lead from three steps to two); s_setprio around the MFMAs (equal).
            for (int c = 0; c < nC; ++c) {
                const unsigned char *hb = halo0 + ((gc + c) & 1) * HALO;
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) {
                    const int tau = c * 9 + tap;
                    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (!(abl & 4)) issue(tau + 3);
                    if constexpr (abl & 2) continue;
                    const unsigned char *ws = ring + ((c + tap) & 3) * H_STAGE + (wn * 8) * 1024 + lane * 16;
                    const int toff = (tap / 3 - 1) * PW + (tap % 3 - 1);
                    const unsigned char *ha[2];
                    int fx[2];
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt) {
                        int hp0 = hpb[mt];
                        asm volatile("" : "+v"(hp0));          // (keeps hipcc from hoisting the 18 tap addresses out of the chunk loop - and spilling them)
                        const int hp = hp0 + toff;
                        ha[mt] = hb + hp * 128;
                        fx[mt] = (kg ^ (((xcb[mt] + (tap % 3 - 1)) >> 1) & 7)) << 4;
                    }
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) {
                        uint4 fb[2], fa[2];
#pragma unroll
                        for (int nt = 0; nt < 2; ++nt) fb[nt] = *reinterpret_cast<const uint4 *>(ws + (nt * 4 + ks) * 1024);
#pragma unroll
                        for (int mt = 0; mt < 2; ++mt) fa[mt] = *reinterpret_cast<const uint4 *>(ha[mt] + (fx[mt] ^ (ks << 5)));
#pragma unroll
                        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                            for (int mt = 0; mt < 2; ++mt)
                                if constexpr (abl & 1) acc[nt][mt][0] += __uint_as_float((fb[nt].x ^ fa[mt].x) + (fb[nt].w ^ fa[mt].w));
                                else acc[nt][mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8, fb[nt]), __builtin_bit_cast(bf8, fa[mt]),
                                                                                           acc[nt][mt], 0, 0, 0);
                    }
                }
            }
            gc += nC;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                  // the past-the-end zero fills must not land on the staged outputs
            __builtin_amdgcn_s_barrier();                                     // every wave is done with the ring: it becomes the staging buffer
            __builtin_amdgcn_sched_barrier(0);
            // D of a 32 x 32 tile: lane (l31, kg) holds channels 8 b + 4 kg + r (register 4 b + r) of pixel l31
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int b = 0; b < 4; ++b) {
                        const int row = wm * 64 + mt * 32 + l31, boff = (wn * 64 + nt * 32 + 8 * b + 4 * kg) * 2;
                        const int chunk = (boff >> 4) ^ (row & 15);
                        uint2 o;
                        o.x = (unsigned)f32_to_bf16(acc[nt][mt][4 * b]) | ((unsigned)f32_to_bf16(acc[nt][mt][4 * b + 1]) << 16);
                        o.y = (unsigned)f32_to_bf16(acc[nt][mt][4 * b + 2]) | ((unsigned)f32_to_bf16(acc[nt][mt][4 * b + 3]) << 16);
                        *reinterpret_cast<uint2 *>(ring + ((size_t)row * 16 + chunk) * 16 + (boff & 15)) = o;
                    }
            __syncthreads();
            const int m0 = mtile * HBM;
            if constexpr (!(abl & 16)) epilogue_store<T, HBM, HBN, 512, DG, 2, 1>(p, ring, mtile % NSH, ntile * HBN, [&](int row) { return (long)(m0 + row); });
            __syncthreads();                                                  // the fold's reads of the ring are done: the next tile's weights may land
        }
    } else {
        // ------------------------------------------------------------------ transfer role
        const int tr = tid - 512;
        const T *xg = reinterpret_cast<const T *>(p.x);
        // piece j of this lane: halo pixel / slot -> global element offset (or -1: border zeros; -2: no such piece), LDS byte offset
        // (rb = (image * H + first halo row), r1 = that halo row within the image: both may be -1 at the top of an image)
        auto locate = [&](int j, int rb, int r1, int cc, long &goff, int &lo, bool &inner) __attribute__((always_inline)) {
            int trv = tr;
            asm volatile("" : "+v"(trv));              // (recomputed where it is used: hoisted out of the tile loop these 11 x 4 values spill)
            const int e = trv + 256 * j;
            if (e >= NP) { goff = -2; lo = 0; inner = false; return; }
            const int hp = e >> 3, slot = e & 7, hr = hp / PW, hc = hp - hr * PW;
            const int col = hc - 1;
            lo = hp * 128 + ((slot ^ ((hc >> 1) & 7)) << 4);
            const bool ok = (unsigned)(r1 + hr) < (unsigned)p.H && (unsigned)col < (unsigned)W;
            goff = ok ? ((long)(rb + hr) * W + col) * C + cc * 64 + slot * 8 : -1l;
            inner = ok && hr >= 1 && hr <= R;
        };
        // a chunk's pieces are all loaded at step 0 of the chunk before (NJ x 16 B per lane in flight) and written, two per step, from step 3 on:
        // three steps cover the load latency (with two pieces loaded per step and written two steps later the transfer waves set the pace:
        // 85 us against 51 without any multiply work - tools/halo_abl.sh)
        uint4 regs[NJ], regs2[DG ? NJ : 1];
        const T *x2g = reinterpret_cast<const T *>(p.x2);
        const int slot8 = (tr & 7) * 8;                // every piece of this lane is the same 16-byte slot of its pixel (256 % 8 == 0)
        float sc[8], sh[8], k0[DG ? 8 : 1];
        auto affine_of = [&](int cc) __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < 8; ++i) { sc[i] = aff[cc * 64 + slot8 + i]; sh[i] = aff[H_MAXC + cc * 64 + slot8 + i]; }
            if constexpr (DG) {
#pragma unroll
                for (int i = 0; i < 8; ++i) k0[i] = aff[2 * H_MAXC + cc * 64 + slot8 + i];
            }
        };
        auto load_all = [&](int rb, int r1, int cc) __attribute__((always_inline)) {
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                long goff; int lo; bool inner;
                locate(j, rb, r1, cc, goff, lo, inner);
                regs[j] = goff >= 0 ? *reinterpret_cast<const uint4 *>(xg + goff) : make_uint4(0u, 0u, 0u, 0u);
                if constexpr (DG) { if (rev) regs2[j] = goff >= 0 ? *reinterpret_cast<const uint4 *>(x2g + goff) : make_uint4(0u, 0u, 0u, 0u); }
            }
        };
        auto store2 = [&](int s, int rb, int r1, int cc, unsigned char *hb, T *ag) __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int j = 2 * s + i;
                if (j >= NJ) continue;
                long goff; int lo; bool inner;
                locate(j, rb, r1, cc, goff, lo, inner);
                if (goff == -2) continue;
                uint4 v = regs[j];
                if (bn && goff >= 0) {
                    float f[8];
                    Chunk<T>::unpack(v, f);
#pragma unroll
                    for (int k = 0; k < 8; ++k) { f[k] = fmaf(f[k], sc[k], sh[k]); if (p.relu_in) f[k] = fmaxf(f[k], 0.f); }
                    v = Chunk<T>::pack(f);
                }
                if constexpr (DG) {
                    if (rev && goff >= 0) {
                        float f[8], y[8];
                        Chunk<T>::unpack(v, f);
                        Chunk<T>::unpack(regs2[j], y);
#pragma unroll
                        for (int k = 0; k < 8; ++k) f[k] = fmaf(sc[k], f[k], fmaf(sh[k], y[k], k0[k]));
                        v = Chunk<T>::pack(f);
                    }
                }
                *reinterpret_cast<uint4 *>(hb + lo) = v;
                if (ag && inner) *reinterpret_cast<uint4 *>(ag + goff) = v;
            }
        };
        int gc = 0;
        {   // the first tile's first chunk: the one exposed staging of the workgroup
            int mt0, nt0;
            tile_at(blockIdx.x, mt0, nt0);
            T *ag = p.a_out && nt0 == 0 ? reinterpret_cast<T *>(p.a_out) : nullptr;
            const int r1 = (mt0 % TPI) * R - 1, rb = (mt0 / TPI) * p.H + r1;
            if (bn || rev) affine_of(0);
            load_all(rb, r1, 0);
#pragma unroll
            for (int s = 0; s < 6; ++s) store2(s, rb, r1, 0, halo0, ag);
        }
        for (int L = blockIdx.x; L < ntiles; L += (int)gridDim.x) {
            int mtile, ntile;
            tile_at(L, mtile, ntile);
            for (int c = 0; c < nC; ++c) {
                // the chunk staged during this one: the next of this tile, or the first of the next tile
                int mt2 = mtile, nt2 = ntile, c2 = c + 1;
                bool have = true;
                if (c2 == nC) {
                    c2 = 0;
                    have = L + (int)gridDim.x < ntiles;
                    if (have) tile_at(L + (int)gridDim.x, mt2, nt2);
                }
                unsigned char *hb = halo0 + ((gc + c + 1) & 1) * HALO;
                T *ag = p.a_out && nt2 == 0 ? reinterpret_cast<T *>(p.a_out) : nullptr;
                const int r1 = (mt2 % TPI) * R - 1, rb = (mt2 / TPI) * p.H + r1;
                if (c == 0) {                                                 // the tile's first barrier
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int s = 0; s < 9; ++s) {
                    if (have && !(abl & 8)) {
                        if (s == 0) { load_all(rb, r1, c2); if (bn || rev) affine_of(c2); }
                        if (s >= 3) store2(s - 3, rb, r1, c2, hb, ag);
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // (step 8: the next chunk is written - the barrier publishes it)
                    __builtin_amdgcn_s_barrier();
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            gc += nC;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            // (the output walk by THIS role from registers while the next tile multiplies - the walk is 13 us of write-rate-bound time in which
            // no CU multiplies - was slower: 111 against 101 us; these waves then arrive late at the next tile's first barriers.  EXPERIMENTS.md)
            for (int i = 0; i < nbar_epi; ++i) __builtin_amdgcn_s_barrier();
        }
    }
}

bool halo_supports(const Params &p) {
    return p.KH == 3 && p.KW == 3 && p.stride == 1 && p.pad == 1 && (p.W == 32 || p.W == 16) && p.H % (HBM / p.W) == 0 && p.Cin % 64 == 0 &&
           p.Cin >= 64 && p.Cin <= H_MAXC && p.Cout % HBN == 0 && p.M % HBM == 0 && (!p.x2 || (p.mask && p.rev_coef && !p.in_scale)) && !p.os2 && !p.res_s2 && !p.y32 && !p.xcat &&
           !(p.mask && p.stats) && (size_t)(p.Cout / HBN) * (p.Cin / 64) * 9 * H_STAGE < 0x7fff0000ull;
}

int launch_halo(const Params &p, hipStream_t s) {
    const int ntiles = (p.M / HBM) * (p.Cout / HBN);
    const dim3 grid((unsigned)(ntiles < 256 ? ntiles : 256)), block(768);
    if (p.W == 32) {
        if (p.mask) hipLaunchKernelGGL((conv_halo_kernel<32, true>), grid, block, 0, s, p);
        else hipLaunchKernelGGL((conv_halo_kernel<32, false>), grid, block, 0, s, p);
    } else {
        if (p.mask) hipLaunchKernelGGL((conv_halo_kernel<16, true>), grid, block, 0, s, p);
        else hipLaunchKernelGGL((conv_halo_kernel<16, false>), grid, block, 0, s, p);
    }
    return check_launch("conv_halo_kernel");
}

// standard pack [Cout][Kpad] (K ordered kh, kw, cin) -> fragment-major stages [Cout / 128][Cin / 64][9][16 fragments][64 lanes][8]:
// fragment f = 4 (channel tile of 32) + (16-deep k step), lane = (channel row, k half)
__global__ void halo_pack_kernel(const u16 *w, u16 *out, int Cout, int Cin, int Kpad) {
    const long n = (long)(Cout / HBN) * (Cin / 64) * 9 * 16 * 64;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int lane = (int)(i & 63), f = (int)((i >> 6) & 15);
    const long st = i >> 10;
    const int tap = (int)(st % 9), c = (int)((st / 9) % (Cin / 64)), ntile = (int)(st / 9 / (Cin / 64));
    const int cout = ntile * HBN + (f >> 2) * 32 + (lane & 31), k = tap * Cin + c * 64 + (f & 3) * 16 + (lane >> 5) * 8;
    *reinterpret_cast<uint4 *>(out + i * 8) = *reinterpret_cast<const uint4 *>(w + (size_t)cout * Kpad + k);
}

}}  // namespace mhe::conv

using namespace mhe;

extern "C" int mhe_conv3x3_halo_supported(int B, int H, int W, int Cin, int Cout) {
    conv::Params p{};
    p.B = B; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout; p.KH = p.KW = 3; p.stride = 1; p.pad = 1; p.Ho = H; p.Wo = W;
    const long long M = (long long)B * H * W;
    if (B <= 0 || H <= 0 || W <= 0 || M >= (1ll << 31)) return 0;
    p.M = (int)M;
    return conv::halo_supports(p) ? 1 : 0;
}

extern "C" int mhe_conv3x3_halo_pack_bf16(const void *w, void *w_halo, int Cout, int Cin, void *stream) {
    MHE_REQUIRE(w && w_halo, "mhe_conv3x3_halo_pack_bf16: null pointer");
    MHE_REQUIRE(Cout > 0 && Cout % 128 == 0 && Cin > 0 && Cin % 64 == 0, "mhe_conv3x3_halo_pack_bf16: Cout %% 128 and Cin %% 64 must be 0");
    const long n = (long)(Cout / 128) * (Cin / 64) * 9 * 16 * 64;
    hipLaunchKernelGGL(conv::halo_pack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const u16 *)w, (u16 *)w_halo, Cout,
                       Cin, 9 * Cin);
    return check_launch("halo_pack_kernel");
}

extern "C" int mhe_conv3x3_halo_nhwc(int B, int H, int W, int Cin, int Cout, const void *x, const void *w_halo, void *y, const float *in_scale,
                                     const float *in_shift, int relu_in, void *a_out, mhe_stat_t *stats, const void *residual, const void *mask,
                                     const void *bn_y0, const float *bn_mean_invstd0, mhe_stat_t *bn_stats0, void *stream) {
    MHE_REQUIRE(x && w_halo && y, "mhe_conv3x3_halo_nhwc: null pointer");
    MHE_REQUIRE((in_scale == nullptr) == (in_shift == nullptr), "mhe_conv3x3_halo_nhwc: in_scale/in_shift must come together");
    MHE_REQUIRE(!a_out || in_scale, "mhe_conv3x3_halo_nhwc: a_out is the normalised operand: it needs in_scale / in_shift");
    MHE_REQUIRE(!(mask && stats) && !(residual && !mask), "mhe_conv3x3_halo_nhwc: statistics in the forward form, residual in the data-gradient form only");
    MHE_REQUIRE(!bn_y0 || (mask && bn_mean_invstd0 && bn_stats0), "mhe_conv3x3_halo_nhwc: bn_y needs the gate, its mean_invstd and stats");
    conv::Params p{};
    p.x = x; p.w = w_halo; p.y = y; p.in_scale = in_scale; p.in_shift = in_shift; p.relu_in = relu_in; p.a_out = a_out; p.stats = stats;
    p.residual = residual; p.mask = mask; p.bn_y[0] = bn_y0; p.bn_mi[0] = bn_mean_invstd0; p.bn_stats[0] = bn_stats0;
    p.B = B; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout; p.KH = p.KW = 3; p.stride = 1; p.pad = 1; p.Ho = H; p.Wo = W;
    const long long M = (long long)B * H * W;
    MHE_REQUIRE(B > 0 && H > 0 && W > 0 && M < (1ll << 31), "mhe_conv3x3_halo_nhwc: bad geometry");
    p.M = (int)M; p.Kpad = 9 * Cin;
    p.force = -1;
    MHE_REQUIRE(conv::halo_supports(p), "mhe_conv3x3_halo_nhwc: geometry not taken (3x3 stride 1 pad 1, W 32 / 16, Cin %% 64, Cin <= 512, Cout %% 128)");
    return conv::launch_halo(p, (hipStream_t)stream);
}

extern "C" int mhe_conv3x3_halo_dgrad_bn_nhwc(int B, int H, int W, int Cin, int Cout, const void *g, const void *y_raw, const float *coef, const void *w_halo,
                                              void *gx, void *gy_out, const void *residual, const void *mask, const void *bn_y0,
                                              const float *bn_mean_invstd0, mhe_stat_t *bn_stats0, void *stream) {
    MHE_REQUIRE(g && y_raw && coef && w_halo && gx && mask, "mhe_conv3x3_halo_dgrad_bn_nhwc: null pointer");
    MHE_REQUIRE(!bn_y0 || (bn_mean_invstd0 && bn_stats0), "mhe_conv3x3_halo_dgrad_bn_nhwc: bn_y needs its mean_invstd and stats");
    conv::Params p{};
    p.x = g; p.x2 = y_raw; p.rev_coef = coef; p.w = w_halo; p.y = gx; p.a_out = gy_out;
    p.residual = residual; p.mask = mask; p.bn_y[0] = bn_y0; p.bn_mi[0] = bn_mean_invstd0; p.bn_stats[0] = bn_stats0;
    p.B = B; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout; p.KH = p.KW = 3; p.stride = 1; p.pad = 1; p.Ho = H; p.Wo = W;
    const long long M = (long long)B * H * W;
    MHE_REQUIRE(B > 0 && H > 0 && W > 0 && M < (1ll << 31), "mhe_conv3x3_halo_dgrad_bn_nhwc: bad geometry");
    p.M = (int)M; p.Kpad = 9 * Cin; p.force = -1;
    MHE_REQUIRE(conv::halo_supports(p), "mhe_conv3x3_halo_dgrad_bn_nhwc: geometry not taken (see mhe_conv3x3_halo_nhwc)");
    return conv::launch_halo(p, (hipStream_t)stream);
}

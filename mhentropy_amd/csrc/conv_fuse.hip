// Bottleneck tail with conv3 RE-EVALUATED in place of being read back (variant 12, forward-only path of layer1 / layer2 of ResNet-50;
// reference hand/network.py:54-61,110 = torchvision Bottleneck.forward: conv3 -> bn3 -> `out += identity` -> relu, then the next block's conv1).
//
// Why.  At 64 / 128 bottleneck channels the block output is 4x wider than anything else in the block, and train-mode BatchNorm makes the
// library write conv3's raw output y3 (to take its batch statistics) and read it back in the tail: 2 x 537 MB per block at layer1 of
// config C2, the largest single share of the forward's HBM traffic.  Here y3 never exists in memory:
//   (1) a statistics-only launch of the streaming 1x1 kernel (conv_stream.hip with y = NULL) reads the 64 / 128-channel operand and keeps
//       only sum / sum of squares of the bf16-rounded products -> bn3's scale / shift;
//   (2) this kernel evaluates T = W3 relu(bn2(y2)) again, 64 channels at a time, rounds it to bf16 exactly as the stored y3 was,
//       forms a = relu(bn3(T) + identity), writes a once (next block's identity / shortcut input) and multiplies it straight into the
//       next block's conv1:  y1 = W1 a  (+ batch statistics of y1 as stored).
// Per 128-pixel tile: read 16 / 32 KiB (y2) + 64 / 128 KiB (identity), write 64 / 128 KiB (a) + 16 / 32 KiB (y1) - against
// conv3 (write y3) + MODE 2 tail (read y3 + identity, write a + y1) one read and one write of a block-wide tensor less.
// The launch is HBM-bound by a wide margin (8.4 MFLOP per 160 KiB: the two products take ~10 % of a tile's memory time), so the
// structure optimises for bytes in flight, not for MFMA rate:
//   multiply waves 0-3 : a wave owns 32 pixels.  Slot s: G1(s): T_s = A2 W3[s]^T (A2 fragments live in registers for the whole tile)
//                        -> bf16 -> Tbuf[s & 1];  G2(s - 2): acc2 += Abuf[s & 1] W1[:, s - 2]^T;
//   transfer waves 4-7 : slot s: X(s - 1): Tbuf[(s - 1) & 1] (+ bn3) + identity (registers, loaded two ticks ahead) -> relu -> Abuf[(s - 1) & 1]
//                        and out to a;  the weight chunks of the coming slots global -> registers -> LDS rings;  the next tile's y2 tile.
// One barrier per slot, NT + 2 slots per tile (NT = C / 64 channel chunks); the tile's 128 x N2 outputs are staged through Tbuf and
// stored with 16-byte lanes by all eight waves, batch statistics accumulated per thread over the workgroup's whole life (one fold at the end).
// Arithmetic order = conv1x1_stream_kernel's for T and conv_kernel MODE 2's for y1, so the results equal the unfused path's bit for bit.
#include "conv_shared.h"

namespace mhe { namespace conv {

namespace {
constexpr int FPIX = 128;                 // pixels per tile
constexpr int FT = FPIX * 8;              // uint4 per 128 x 64-channel LDS tile (16 KiB)
// weight chunks only pass through registers (global -> LDS): as HIP's uint4 struct those copies become memcpy calls on an alloca that
// SROA leaves in scratch (or "promotes" to 32 KiB of LDS); a native vector type keeps them in VGPRs
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
// the transfer role's global traffic goes through buffer instructions: resource (SGPRs) + per-thread byte offset (ONE VGPR per tensor
// shape, fixed for the kernel's life) + uniform byte offset (an SGPR computed on the scalar unit per access).  Written as 64-bit pointer
// arithmetic hipcc formed (base + thread offset) in a VGPR pair and hoisted one pair per static access site out of the tile loop:
// ~100 VGPRs of addresses, spilled.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc_of(const void *p, size_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ uint4 bld(__amdgpu_buffer_rsrc_t r, unsigned toff, unsigned uoff) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)toff, (int)uoff, 0);
    return make_uint4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ u32x4 bldv(__amdgpu_buffer_rsrc_t r, unsigned toff, unsigned uoff) {
    return __builtin_amdgcn_raw_buffer_load_b128(r, (int)toff, (int)uoff, 0);
}
__device__ __forceinline__ void bst(__amdgpu_buffer_rsrc_t r, unsigned toff, unsigned uoff, uint4 v) {
    const u32x4 t = {v.x, v.y, v.z, v.w};
    __builtin_amdgcn_raw_buffer_store_b128(t, r, (int)toff, (int)uoff, 0);
}
}

template <int CB, int N2>
__global__ __launch_bounds__(512) void bottleneck_tail_kernel(const Params p) {
    using T = u16;
    constexpr int C = 4 * CB, NT = C / 64, KT1 = CB / 64;
    constexpr int R3S = KT1 * 64 * 8, R1S = N2 * 8;                  // uint4 per W3 / W1 ring stage
    constexpr int CPR2 = N2 * 2 / 16, RPP = 256 / CPR2, NJ2 = FPIX / RPP;        // output rows: 16-byte chunks per row, rows per pass (transfer waves), passes
    static_assert(NT % 2 == 0 && KT1 >= 1 && KT1 <= 2 && (N2 == 64 || N2 == 128), "geometry");
    static_assert(FPIX * CPR2 <= 2 * FT, "the output tile is staged over Tbuf");
    // Tbuf[2] | Abuf[2] (the tile's normalised y2, KT1 x 16 KiB, lies over Abuf before slot 1) | W3 ring[2] | W1 ring[2]
    __shared__ uint4 lds[4 * FT + 2 * R3S + 2 * R1S];
    __shared__ float aff[4 * C + 2 * CB];                            // bn3 scale | shift, identity scale | shift (1 | 0 without), bn2 scale | shift
    uint4 *Tbuf = lds, *Abuf = lds + 2 * FT, *R3 = lds + 4 * FT, *R1 = R3 + 2 * R3S;
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, q = lane >> 4, l15 = lane & 15;
    const bool mult = wave < 4;
    const int t2 = tid & 255, s8 = t2 & 7, rbase = t2 >> 3;
    const int ntiles = p.M / FPIX;
    const bool idaff = p.x2_scale != nullptr;
    for (int i = tid; i < C; i += 512) {
        aff[i] = p.mid_scale[i]; aff[C + i] = p.mid_shift[i];
        aff[2 * C + i] = idaff ? p.x2_scale[i] : 1.f; aff[3 * C + i] = idaff ? p.x2_shift[i] : 0.f;
    }
    for (int i = tid; i < CB; i += 512) { aff[4 * C + i] = p.in_scale[i]; aff[4 * C + CB + i] = p.in_shift[i]; }
    __syncthreads();                                           // the tables are read by the transfer role before its first top barrier
    const __amdgpu_buffer_rsrc_t y2g = rsrc_of(p.x, (size_t)p.M * CB * 2), idg = rsrc_of(p.x2, (size_t)p.M * C * 2);
    const __amdgpu_buffer_rsrc_t w3g = rsrc_of(p.w3, (size_t)C * CB * 2), w1g = rsrc_of(p.w, (size_t)N2 * C * 2);
    const __amdgpu_buffer_rsrc_t ag = rsrc_of(p.a_out, (size_t)p.M * C * 2), yg = rsrc_of(p.y, (size_t)p.M * N2 * 2);
    const __amdgpu_buffer_rsrc_t abg = rsrc_of(p.a_bits, (size_t)p.M * (C / 8));
    // output epilogue (the four transfer waves; the multiply waves go on to the next tile's top barrier): staged 128 x N2 tile -> 16-byte
    // stores + per-thread partial statistics of the stored values
    const int cc = t2 % CPR2, r0 = t2 / CPR2;
    const unsigned toffC = (unsigned)(rbase * C + s8 * 8) * 2u, toffB = (unsigned)(rbase * CB + s8 * 8) * 2u, toffY = (unsigned)(r0 * N2 + cc * 8) * 2u;
    const bool st_on = p.stats != nullptr;
    float ss1[8], ss2[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) ss1[i] = ss2[i] = 0.f;
    auto store_outputs = [&](int m0) __attribute__((always_inline)) {
        const unsigned char *ot = reinterpret_cast<const unsigned char *>(Tbuf);
        uint4 raw[NJ2];
#pragma unroll
        for (int j = 0; j < NJ2; ++j) {
            const int row = r0 + RPP * j;
            raw[j] = *reinterpret_cast<const uint4 *>(ot + ((size_t)row * CPR2 + (cc ^ (row & (CPR2 - 1)))) * 16);
        }
#pragma unroll
        for (int j = 0; j < NJ2; ++j) {
            if (st_on) {
                float f[8];
                Chunk<T>::unpack(raw[j], f);
#pragma unroll
                for (int i = 0; i < 8; ++i) { ss1[i] += f[i]; ss2[i] = fmaf(f[i], f[i], ss2[i]); }
            }
            bst(yg, toffY, (unsigned)(m0 + RPP * j) * (N2 * 2), raw[j]);
        }
    };

    if (mult) {
        for (int L = blockIdx.x; L < ntiles; L += (int)gridDim.x) {
            __syncthreads();                                   // top: y2 tile staged (over Abuf), W3[0] in ring stage 0, tables written
            uint4 fa[2][KT1 * 2];                              // this wave's 32 pixels x CB channels of relu(bn2(y2)): B operands of every G1
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int k = 0; k < KT1 * 2; ++k) fa[m][k] = Abuf[(k >> 1) * FT + swz(wave * 32 + m * 16 + l15, (k & 1) * 4 + q)];
            v4f acc2[N2 / 16][2];
#pragma unroll
            for (int a = 0; a < N2 / 16; ++a) { acc2[a][0] = v4f{0.f, 0.f, 0.f, 0.f}; acc2[a][1] = v4f{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
            for (int s = 0; s < NT + 2; ++s) {
                if (s > 0) __syncthreads();
                if (s < NT) {                                  // G1(s): 64 channels of T for this wave's 32 pixels
                    const uint4 *W3s = R3 + (s & 1) * R3S;
                    v4f acc1[4][2];
#pragma unroll
                    for (int a = 0; a < 4; ++a) { acc1[a][0] = v4f{0.f, 0.f, 0.f, 0.f}; acc1[a][1] = v4f{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
                    for (int k = 0; k < KT1 * 2; ++k) {
                        uint4 fb[4];
#pragma unroll
                        for (int nt = 0; nt < 4; ++nt) fb[nt] = W3s[(k >> 1) * 512 + swz(nt * 16 + l15, (k & 1) * 4 + q)];
#pragma unroll
                        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                            for (int m = 0; m < 2; ++m)
                                acc1[nt][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                                    __builtin_bit_cast(__attribute__((ext_vector_type(8))) __bf16, fb[nt]),
                                    __builtin_bit_cast(__attribute__((ext_vector_type(8))) __bf16, fa[m][k]), acc1[nt][m], 0, 0, 0);
                    }
                    // D layout: lane (l15, q) holds channels 16 nt + 4q .. + 3 of pixel 16 m + l15 -> the rounded value conv3 would have stored
                    unsigned char *tb = reinterpret_cast<unsigned char *>(Tbuf + (s & 1) * FT);
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                        for (int m = 0; m < 2; ++m) {
                            const int row = wave * 32 + m * 16 + l15, chunk = nt * 2 + (q >> 1);
                            const v4f v = acc1[nt][m];
                            uint2 o;
                            o.x = (unsigned)f32_to_bf16(v[0]) | ((unsigned)f32_to_bf16(v[1]) << 16);
                            o.y = (unsigned)f32_to_bf16(v[2]) | ((unsigned)f32_to_bf16(v[3]) << 16);
                            *reinterpret_cast<uint2 *>(tb + (size_t)swz(row, chunk) * 16 + (q & 1) * 8) = o;
                        }
                }
                if (s >= 2) {                                  // G2(s - 2): the next conv1 over the 64 channels of a evaluated in slot s - 1
                    const int t = s - 2;
                    const uint4 *As = Abuf + (t & 1) * FT, *W1s = R1 + (t & 1) * R1S;
#pragma unroll
                    for (int kk = 0; kk < 2; ++kk) {
                        uint4 fa2[2], fb2[N2 / 16];
#pragma unroll
                        for (int m = 0; m < 2; ++m) fa2[m] = As[swz(wave * 32 + m * 16 + l15, kk * 4 + q)];
#pragma unroll
                        for (int nt = 0; nt < N2 / 16; ++nt) fb2[nt] = W1s[swz(nt * 16 + l15, kk * 4 + q)];
#pragma unroll
                        for (int nt = 0; nt < N2 / 16; ++nt)
#pragma unroll
                            for (int m = 0; m < 2; ++m)
                                acc2[nt][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                                    __builtin_bit_cast(__attribute__((ext_vector_type(8))) __bf16, fb2[nt]),
                                    __builtin_bit_cast(__attribute__((ext_vector_type(8))) __bf16, fa2[m]), acc2[nt][m], 0, 0, 0);
                    }
                }
            }
            __syncthreads();                                   // every LDS read of the tile is done: Tbuf becomes the output staging buffer
            unsigned char *ot = reinterpret_cast<unsigned char *>(Tbuf);
#pragma unroll
            for (int nt = 0; nt < N2 / 16; ++nt)
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    const int row = wave * 32 + m * 16 + l15, boff = (nt * 16 + 4 * q) * 2;
                    const int chunk = (boff >> 4) ^ (row & (CPR2 - 1));
                    const v4f v = acc2[nt][m];
                    uint2 o;
                    o.x = (unsigned)f32_to_bf16(v[0]) | ((unsigned)f32_to_bf16(v[1]) << 16);
                    o.y = (unsigned)f32_to_bf16(v[2]) | ((unsigned)f32_to_bf16(v[3]) << 16);
                    *reinterpret_cast<uint2 *>(ot + ((size_t)row * CPR2 + chunk) * 16 + (boff & 15)) = o;
                }
            __syncthreads();                                   // staged: the transfer waves store it while this role waits at the next top barrier
        }
    } else {
        // ---- transfer role.  Thread: 16-byte chunk s8 of a 64-channel row piece, rows rbase + 32 j.
        struct IdSet { uint4 v[4]; };
        IdSet id0, id1;                                        // identity chunks of ticks t (set t & 1), loaded two ticks ahead
        uint4 a2r[KT1 * 4];
        u32x4 w3r[KT1 * 2], w1r[N2 / 32];
        auto load_id = [&](int m0, int t, IdSet &st) __attribute__((always_inline)) {
#pragma unroll
            for (int j = 0; j < 4; ++j) st.v[j] = bld(idg, toffC, (unsigned)(m0 + 32 * j) * (C * 2) + t * 128);
        };
        auto load_a2 = [&](int m0) __attribute__((always_inline)) {
#pragma unroll
            for (int kt = 0; kt < KT1; ++kt)
#pragma unroll
                for (int j = 0; j < 4; ++j) a2r[kt * 4 + j] = bld(y2g, toffB, (unsigned)(m0 + 32 * j) * (CB * 2) + kt * 128);
        };
        auto load_w3 = [&](int t) __attribute__((always_inline)) {             // rows 64 t .. 64 t + 63 of W3 [C][CB]
#pragma unroll
            for (int kt = 0; kt < KT1; ++kt)
#pragma unroll
                for (int j = 0; j < 2; ++j) w3r[kt * 2 + j] = bldv(w3g, toffB, (unsigned)(t * 64 + 32 * j) * (CB * 2) + kt * 128);
        };
        auto store_w3 = [&](int t) __attribute__((always_inline)) {
            uint4 *dst = R3 + (t & 1) * R3S;
#pragma unroll
            for (int kt = 0; kt < KT1; ++kt)
#pragma unroll
                for (int j = 0; j < 2; ++j) *reinterpret_cast<u32x4 *>(dst + kt * 512 + swz(rbase + 32 * j, s8)) = w3r[kt * 2 + j];
        };
        auto load_w1 = [&](int t) __attribute__((always_inline)) {             // columns 64 t .. + 63 of W1 [N2][C]
#pragma unroll
            for (int j = 0; j < N2 / 32; ++j) w1r[j] = bldv(w1g, toffC, (unsigned)(32 * j) * (C * 2) + t * 128);
        };
        auto store_w1 = [&](int t) __attribute__((always_inline)) {
            uint4 *dst = R1 + (t & 1) * R1S;
#pragma unroll
            for (int j = 0; j < N2 / 32; ++j) *reinterpret_cast<u32x4 *>(dst + swz(rbase + 32 * j, s8)) = w1r[j];
        };
        // X(t): a = relu(bn3(T) + identity) for the 64 channels of tick t, in conv_shared.h's in_transform order
        auto xform = [&](int m0, int t, const IdSet &st) __attribute__((always_inline)) {
            const uint4 *Ts = Tbuf + (t & 1) * FT;
            uint4 *As = Abuf + (t & 1) * FT;
            const int kc = t * 64 + s8 * 8;
            float sc[8], sh[8], s2[8], h2[8];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const float4 a0 = *reinterpret_cast<const float4 *>(aff + kc + 4 * h), a1 = *reinterpret_cast<const float4 *>(aff + C + kc + 4 * h);
                const float4 b0 = *reinterpret_cast<const float4 *>(aff + 2 * C + kc + 4 * h), b1 = *reinterpret_cast<const float4 *>(aff + 3 * C + kc + 4 * h);
                sc[4 * h] = a0.x; sc[4 * h + 1] = a0.y; sc[4 * h + 2] = a0.z; sc[4 * h + 3] = a0.w;
                sh[4 * h] = a1.x; sh[4 * h + 1] = a1.y; sh[4 * h + 2] = a1.z; sh[4 * h + 3] = a1.w;
                s2[4 * h] = b0.x; s2[4 * h + 1] = b0.y; s2[4 * h + 2] = b0.z; s2[4 * h + 3] = b0.w;
                h2[4 * h] = b1.x; h2[4 * h + 1] = b1.y; h2[4 * h + 2] = b1.z; h2[4 * h + 3] = b1.w;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int row = rbase + 32 * j;
                float v[8], w[8];
                Chunk<T>::unpack(Ts[swz(row, s8)], v);
                Chunk<T>::unpack(st.v[j], w);
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    v[i] = fmaf(v[i], sc[i], sh[i]);
                    if (idaff) w[i] = fmaf(w[i], s2[i], h2[i]);
                    v[i] = fmaxf(v[i] + w[i], 0.f);
                }
                const uint4 o = Chunk<T>::pack(v);
                As[swz(row, s8)] = o;
                bst(ag, toffC, (unsigned)(m0 + 32 * j) * (C * 2) + t * 128, o);
                // the gate of the reverse pass, one bit per stored value (bf16: sign clear and not zero)
                const unsigned wds[4] = {o.x, o.y, o.z, o.w};
                unsigned gb = 0;
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    gb |= ((unsigned)((wds[i] & 0x8000u) == 0 && (wds[i] & 0x7fffu) != 0) << (2 * i)) |
                          ((unsigned)((wds[i] & 0x80000000u) == 0 && (wds[i] & 0x7fff0000u) != 0) << (2 * i + 1));
                if (p.a_bits) __builtin_amdgcn_raw_buffer_store_b8((unsigned char)gb, abg, (int)((rbase * (C / 8)) + s8), (int)((unsigned)(m0 + 32 * j) * (C / 8) + t * 8), 0);
            }
        };
        auto stage_a2 = [&]() __attribute__((always_inline)) {                 // relu(bn2(y2)) tile -> LDS (over Abuf)
#pragma unroll
            for (int kt = 0; kt < KT1; ++kt) {
                float sc[8], sh[8];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const float4 a0 = *reinterpret_cast<const float4 *>(aff + 4 * C + kt * 64 + s8 * 8 + 4 * h);
                    const float4 a1 = *reinterpret_cast<const float4 *>(aff + 4 * C + CB + kt * 64 + s8 * 8 + 4 * h);
                    sc[4 * h] = a0.x; sc[4 * h + 1] = a0.y; sc[4 * h + 2] = a0.z; sc[4 * h + 3] = a0.w;
                    sh[4 * h] = a1.x; sh[4 * h + 1] = a1.y; sh[4 * h + 2] = a1.z; sh[4 * h + 3] = a1.w;
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float v[8];
                    Chunk<T>::unpack(a2r[kt * 4 + j], v);
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[i] = fmaxf(fmaf(v[i], sc[i], sh[i]), 0.f);
                    Abuf[kt * FT + swz(rbase + 32 * j, s8)] = Chunk<T>::pack(v);
                }
            }
        };
        // first tile: everything it needs before its first slot
        {
            const int m0 = (int)blockIdx.x * FPIX;
            load_a2(m0); load_w3(0);
            load_id(m0, 0, id0); load_id(m0, 1, id1);
            load_w1(0);
        }
        for (int L = blockIdx.x; L < ntiles; L += (int)gridDim.x) {
            const int m0 = L * FPIX;
            const int Ln = L + (int)gridDim.x;
            const bool more = Ln < ntiles;
            const int m0n = more ? Ln * FPIX : m0;             // (clamped: the loads of a tile that does not exist are harmless re-reads)
            stage_a2();
            store_w3(0);
            load_w3(1);
            __syncthreads();                                   // top
#pragma unroll
            for (int s = 0; s < NT + 2; ++s) {
                if (s > 0) __syncthreads();
                if (s >= 1 && s <= NT) {
                    const int t = s - 1;
                    if (t & 1) { xform(m0, t, id1); if (t + 2 < NT) load_id(m0, t + 2, id1); else load_id(m0n, t + 2 - NT, id1); }
                    else       { xform(m0, t, id0); if (t + 2 < NT) load_id(m0, t + 2, id0); else load_id(m0n, t + 2 - NT, id0); }
                    store_w1(t);
                    load_w1((t + 1) % NT);
                }
                if (s + 1 < NT) {                              // W3[s + 1] for the next slot's G1; then fetch the one after (wrapping to the next tile's first)
                    store_w3(s + 1);
                    load_w3((s + 2) % NT);
                }
                if (s == 1) load_a2(m0n);                      // the next tile's y2 rows: in flight for the rest of this tile
            }
            __syncthreads();
            __syncthreads();
            store_outputs(m0);
        }
    }
    // ---- batch statistics of y1 as stored: fold the 256 transfer threads' partial sums once per workgroup
    if (st_on) {
        float *red = reinterpret_cast<float *>(lds);
        __syncthreads();
        if (!mult) {
#pragma unroll
            for (int i = 0; i < 8; ++i) { red[t2 * 16 + i] = ss1[i]; red[t2 * 16 + 8 + i] = ss2[i]; }
        }
        __syncthreads();
        if (tid < N2) {
            const int ch = tid >> 3, e = tid & 7;
            float a = 0.f, b = 0.f;
            for (int k = 0; k < RPP; ++k) { a += red[(ch + CPR2 * k) * 16 + e]; b += red[(ch + CPR2 * k) * 16 + 8 + e]; }
            fx::add(p.stats, (int)blockIdx.x % NSH, 0, N2, tid, a);
            fx::add(p.stats, (int)blockIdx.x % NSH, 1, N2, tid, b);
        }
    }
}

bool fuse_supports(const Params &p, int cb) {
    return p.x && p.x2 && p.w && p.w3 && p.y && p.a_out && p.in_scale && p.in_shift && p.mid_scale && p.mid_shift &&
           (cb == 64 || cb == 128) && p.Cin == 4 * cb && (p.Cout == 64 || p.Cout == 128) && p.Kpad == p.Cin &&
           p.KH == 1 && p.KW == 1 && p.stride == 1 && p.pad == 0 && p.M % FPIX == 0 && p.M > 0;
}

int launch_fuse(const Params &p, int cb, hipStream_t s) {
    const int ntiles = p.M / FPIX;
    const dim3 grid((unsigned)(ntiles < 256 ? ntiles : 256)), block(512);
    if (cb == 64 && p.Cout == 64) hipLaunchKernelGGL((bottleneck_tail_kernel<64, 64>), grid, block, 0, s, p);
    else if (cb == 64 && p.Cout == 128) hipLaunchKernelGGL((bottleneck_tail_kernel<64, 128>), grid, block, 0, s, p);
    else if (cb == 128 && p.Cout == 64) hipLaunchKernelGGL((bottleneck_tail_kernel<128, 64>), grid, block, 0, s, p);
    else hipLaunchKernelGGL((bottleneck_tail_kernel<128, 128>), grid, block, 0, s, p);
    return check_launch("bottleneck_tail_kernel");
}

}}  // namespace mhe::conv

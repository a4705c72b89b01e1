// Bottleneck tail with conv3 RE-EVALUATED at 256 bottleneck channels (round 4): layer3 of ResNet-50 - conv3 256 -> 1024, the block tail
// relu(bn3(.) + identity), and the next block's conv1 1024 -> 256 in one kernel; reference hand/network.py:54-61,110 (torchvision
// Bottleneck.forward: conv3 -> bn3 -> `out += identity` -> relu, then the next block's conv1).
//
// Why.  Layer3's block chain wrote conv3's raw output y3 (134 MB at config C2, the widest tensor of the block) only to take its batch
// statistics and to read it back once in the tail: conv_wide_kernel 68 us + conv_tail_kernel 102 us per block for 604 MB of traffic.
// With bn3's statistics from a statistics-only launch of conv_wide_kernel (its stores dropped: the same bf16-rounded products, the same
// sums) the tail can evaluate conv3 again on its way: 335 MB per block (y2 and the identity in, the block output and y1 out).
//
// csrc/conv_fuse.hip does this for 64 / 128 bottleneck channels; its structure does not stretch to 256 (192 KiB of LDS, 72 fragment
// registers in its second product): the same two roles on other shapes -
//   * the block width is walked in 32-channel SLOTS (32 per tile): per slot a W3 stage (32 rows x 256 K = 16 KiB) and a W1 stage
//     (256 rows x 32 K = 16 KiB), both PRE-PACKED on the host as the bank-swizzled LDS image of the stage (mhe_bottleneck_tail256_pack):
//     the transfer waves copy them global -> registers -> LDS linearly, two slots ahead;
//   * multiply waves 0-3 (32 pixels each; the tile's y2 fragments, 64 VGPRs, live in registers for the whole tile):
//       slot s:  G1(s): T = W3[s] y2^T (32 x 32 per wave, K = 256) -> bf16 (what conv3 would have stored) -> ring buffer B[s % 4];
//                G2(s - 2): y1^T += W1[s - 2] a^T (256 x 32 per wave, K = 32), a from B[(s - 2) % 4];
//   * transfer waves 4-7: X(s - 1): B[(s - 1) % 4] <- relu(bn3(T) + identity) IN PLACE (identity chunks four slots ahead in registers) and
//     out to the block output; the weight stages; the next tile's y2 rows (in flight for the whole tile, normalised into their own LDS
//     image a quarter into the tile);
//   * one barrier per slot; slot groups 0 and 8 (where a product or the transform has nothing to do) are written out, groups 1-7 loop
//     with every register-set index a compile-time constant and no branch around a global-memory operation;
//   * the 128 x 256 outputs are staged over the weight rings and stored by the transfer waves with the batch statistics of y1 as stored.
// Arithmetic order = conv_wide_kernel's for T (K ascending, 32 per MFMA) and conv_tail_kernel's for y1: the block output equals the
// unfused path's bit for bit.
#include "conv_shared.h"

#ifndef MHE_T256_ABL
#define MHE_T256_ABL 0        // tuning builds (tools/tail256_abl.sh): 1 no MFMA, 2 no transform, 4 no weight stages, 8 no identity loads, 16 no a_out stores, 32 no output stores
#endif

namespace mhe { namespace conv {

namespace {
constexpr int QABL = MHE_T256_ABL;
constexpr int QPIX = 128, QCB = 256, QC = 1024, QN2 = 256, QSW = 32, QNT = QC / QSW;      // 32 slots per tile
constexpr int QG = (QNT + 2 + 3) / 4;                                                      // 9 slot groups of 4 (36 slots; the last two idle)
constexpr int Q_WS = 1024, Q_BB = QPIX * 4, Q_Y2 = 4 * QPIX * 8;                          // uint4 per weight stage / T-A buffer / y2 image
typedef unsigned int q32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ int qswz64(int row, int slot) { return row * 4 + (slot ^ ((0 - (row >> 2)) & 3)); }
__device__ __forceinline__ __amdgpu_buffer_rsrc_t qrsrc(const void *p, size_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ uint4 qld(__amdgpu_buffer_rsrc_t r, unsigned toff, unsigned uoff) {
    const q32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)toff, (int)uoff, 0);
    return make_uint4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ q32x4 qldv(__amdgpu_buffer_rsrc_t r, unsigned toff, unsigned uoff) {
    return __builtin_amdgcn_raw_buffer_load_b128(r, (int)toff, (int)uoff, 0);
}
__device__ __forceinline__ void qst(__amdgpu_buffer_rsrc_t r, unsigned toff, unsigned uoff, uint4 v) {
    const q32x4 t = {v.x, v.y, v.z, v.w};
    __builtin_amdgcn_raw_buffer_store_b128(t, r, (int)toff, (int)uoff, 0);
}
constexpr unsigned Q_OOB = 0x80000000u;
}  // namespace

template <bool IDAFF>
__global__ __launch_bounds__(512) void bottleneck_tail256_kernel(const Params p) {
    using T = u16;
    // W3 ring[2] | W1 ring[2] (the tile's staged outputs lie over both) | B[4] | Y2   = 160 KiB: bn3's tables are read from global memory
    // slot by slot (a ring of THREE buffers would do, but its index is not a compile-time constant of the four-slot loop body: hipcc then
    // keeps an address register per access and buffer - 30 spilled VGPRs)
    __shared__ uint4 lds[4 * Q_WS + 4 * Q_BB + Q_Y2];
    uint4 *const R3 = lds, *const R1 = lds + 2 * Q_WS, *const BB = lds + 4 * Q_WS, *const Y2 = BB + 4 * Q_BB;
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, q = lane >> 4, l15 = lane & 15;
    const bool mult = wave < 4;
    const int t2 = tid & 255;
    const int ntiles = p.M / QPIX, G = (int)gridDim.x;
    const bool st_on = p.stats != nullptr;
    float ss1[8], ss2[8];                                              // (transfer role: per-thread partial statistics of the stored y1)
#pragma unroll
    for (int i = 0; i < 8; ++i) ss1[i] = ss2[i] = 0.f;

    if (mult) {
        // ------------------------------------------------------------------ multiply role
        // LDS addresses as a few per-lane bases + compile-time offsets (hoisted per-access address arithmetic cost 32 VGPRs of spills):
        // 128-byte-row images (W3 stages, Y2): chunk (k & 1) * 4 + q of row r sits at r * 8 + (chunk ^ ((r >> 1) & 7)), and every row this lane
        // reads is l15 + a multiple of 16: the XOR term is (l15 >> 1) & 7;  64-byte-row images (W1 stages, the ring): f = -(l15 >> 2) & 3
        const int f = (0 - (l15 >> 2)) & 3, x7 = (l15 >> 1) & 7;
        const int e0 = l15 * 8 + (q ^ x7), e1 = l15 * 8 + ((q ^ x7) ^ 4);     // 128-byte rows: even / odd k-step
        const int c0 = l15 * 4 + (q ^ f);                                    // 64-byte rows: the K = 32 step
        const int d0 = l15 * 4 + ((q >> 1) ^ f), d1 = l15 * 4 + (((q >> 1) ^ f) ^ 2);      // 64-byte rows: T chunks of channel tile 0 / 1
        const int rowA = wave * 32 + l15;                                // + 16 m
        // B ring: G1(s) writes B[s & 3], X(s - 1) works in place on B[(s - 1) & 3], G2(s - 2) reads B[(s - 2) & 3]
        for (int L = blockIdx.x; L < ntiles; L += G) {
            __syncthreads();                                             // top: Y2 holds this tile's relu(bn2(y2)), W3[0] sits in R3[0]
            uint4 fa[2][8];                                              // this wave's 32 pixels x 256 channels: B operands of every G1
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int k = 0; k < 8; ++k) fa[m][k] = Y2[(k >> 1) * (QPIX * 8) + (wave * 32 + 16 * m) * 8 + ((k & 1) ? e1 : e0)];
            v4f acc2[16][2];
#pragma unroll
            for (int a = 0; a < 16; ++a) { acc2[a][0] = v4f{0.f, 0.f, 0.f, 0.f}; acc2[a][1] = v4f{0.f, 0.f, 0.f, 0.f}; }
            // (tried: the slot as 16 steps of 4 MFMAs with the next step's two weight fragments read while a step multiplies - hipcc emits each
            // group's reads right before its own MFMAs, so every group exposes an LDS latency; pinned with sched_barriers the pipelined form
            // spilled 75 VGPRs next to the 64 fragment + 128 accumulator registers: 145-150 us against 120-136)
            auto slot = [&](int s, bool g1, bool g2) __attribute__((always_inline)) {
                if (s > 0) __syncthreads();
                if (g1) {                                                // G1(s): 32 channels of T for this wave's 32 pixels
                    const uint4 *W3s = R3 + (s & 1) * Q_WS;
                    v4f acc1[2][2];
#pragma unroll
                    for (int a = 0; a < 2; ++a) { acc1[a][0] = v4f{0.f, 0.f, 0.f, 0.f}; acc1[a][1] = v4f{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        uint4 fb[2];
#pragma unroll
                        for (int nt = 0; nt < 2; ++nt) fb[nt] = W3s[(k >> 1) * 256 + nt * 128 + ((k & 1) ? e1 : e0)];
#pragma unroll
                        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                            for (int m = 0; m < 2; ++m)
                                if constexpr (!(QABL & 1)) acc1[nt][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                                    __builtin_bit_cast(__attribute__((ext_vector_type(8))) __bf16, fb[nt]),
                                    __builtin_bit_cast(__attribute__((ext_vector_type(8))) __bf16, fa[m][k]), acc1[nt][m], 0, 0, 0);
                    }
                    // D layout: lane (l15, q) holds channels 16 nt + 4q .. + 3 of pixel 16 m + l15 -> the rounded value conv3 would have stored
                    unsigned char *tb = reinterpret_cast<unsigned char *>(BB + (s & 3) * Q_BB);
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                        for (int m = 0; m < 2; ++m) {
                            const v4f v = acc1[nt][m];
                            uint2 o;
                            o.x = (unsigned)f32_to_bf16(v[0]) | ((unsigned)f32_to_bf16(v[1]) << 16);
                            o.y = (unsigned)f32_to_bf16(v[2]) | ((unsigned)f32_to_bf16(v[3]) << 16);
                            *reinterpret_cast<uint2 *>(tb + (size_t)((wave * 32 + 16 * m) * 4 + (nt ? d1 : d0)) * 16 + (q & 1) * 8) = o;
                        }
                }
                __builtin_amdgcn_sched_barrier(0);
                if (g2) {                                                // G2(s - 2): the next conv1 over the 32 channels of a evaluated in slot s - 1
                    const uint4 *As = BB + ((s + 2) & 3) * Q_BB, *W1s = R1 + (s & 1) * Q_WS;
                    uint4 fa2[2];
#pragma unroll
                    for (int m = 0; m < 2; ++m) fa2[m] = As[(wave * 32 + 16 * m) * 4 + c0];
#pragma unroll
                    for (int ng = 0; ng < 4; ++ng) {
                        uint4 fb2[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) fb2[j] = W1s[(ng * 4 + j) * 64 + c0];
#pragma unroll
                        for (int j = 0; j < 4; ++j)
#pragma unroll
                            for (int m = 0; m < 2; ++m)
                                if constexpr (!(QABL & 1)) acc2[ng * 4 + j][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                                    __builtin_bit_cast(__attribute__((ext_vector_type(8))) __bf16, fb2[j]),
                                    __builtin_bit_cast(__attribute__((ext_vector_type(8))) __bf16, fa2[m]), acc2[ng * 4 + j][m], 0, 0, 0);
                    }
                }
            };
            slot(0, true, false); slot(1, true, false); slot(2, true, true); slot(3, true, true);
            for (int g = 1; g < QG - 1; ++g) { slot(4 * g, true, true); slot(4 * g + 1, true, true); slot(4 * g + 2, true, true); slot(4 * g + 3, true, true); }
            slot(4 * (QG - 1), false, true); slot(4 * (QG - 1) + 1, false, true); slot(4 * (QG - 1) + 2, false, false); slot(4 * (QG - 1) + 3, false, false);
            __syncthreads();                                             // B1: every LDS read of the tile is done: the weight rings become the staging buffer
            unsigned char *ot = reinterpret_cast<unsigned char *>(R3);
            int rowE = rowA;
            asm volatile("" : "+v"(rowE));                               // (the 32 staging addresses are formed here, not hoisted out of the tile loop: 32 VGPRs)
#pragma unroll
            for (int nt = 0; nt < 16; ++nt)
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    const int row = rowE + 16 * m, boff = (nt * 16 + 4 * q) * 2;
                    const int chunk = (boff >> 4) ^ (row & 31);
                    const v4f v = acc2[nt][m];
                    uint2 o;
                    o.x = (unsigned)f32_to_bf16(v[0]) | ((unsigned)f32_to_bf16(v[1]) << 16);
                    o.y = (unsigned)f32_to_bf16(v[2]) | ((unsigned)f32_to_bf16(v[3]) << 16);
                    *reinterpret_cast<uint2 *>(ot + ((size_t)row * 32 + chunk) * 16 + (boff & 15)) = o;
                }
            __syncthreads();                                             // B2: staged: the transfer waves store it
            __syncthreads();                                             // B3: stored: the rings are free for the next tile's first stages
        }
    } else {
        // ------------------------------------------------------------------ transfer roles: waves 4-5 the activations (X), waves 6-7 the weights (W)
        // (one role doing both in turn - the first version - made a slot the SUM of the two: 120-136 us per launch, tools/tail256_abl.sh;
        // and vector-memory loads return in issue order, so a wave that mixes two-slot-ahead weight loads with four-slot-ahead identity
        // loads gives every load the shorter lead)
        const bool xrole = wave < 6;
        const int u = tid & 127;                                         // thread of its sub-role
        const int cc = t2 & 31, r0 = t2 >> 5;                            // outputs (all four waves): chunk cc of rows r0 + 8 j
        const __amdgpu_buffer_rsrc_t yg = qrsrc(p.y, (size_t)p.M * QN2 * 2);
        const unsigned toffY = (unsigned)(r0 * QN2 + cc * 8) * 2u;
        auto store_outputs = [&](int m0) __attribute__((always_inline)) {
            const unsigned char *ot = reinterpret_cast<const unsigned char *>(R3);
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
                    if (st_on) {
                        float fv[8];
                        Chunk<T>::unpack(raw[jj], fv);
#pragma unroll
                        for (int i = 0; i < 8; ++i) { ss1[i] += fv[i]; ss2[i] = fmaf(fv[i], fv[i], ss2[i]); }
                    }
                    qst(yg, toffY, (unsigned)(m0 + 8 * (j0 + jj)) * (QN2 * 2), raw[jj]);
                }
            }
        };
        if (xrole) {
            // ---- X: identity chunks (four slots ahead), bn3 / identity tables (two ahead), the in-place tail on the ring, the block output,
            // and the next tile's y2 (a K tile per slot group)
            const int s8 = u & 7, r16 = u >> 3;                          // y2 rows: 16-byte chunk s8 of rows r16 + 16 j (per 64-channel K tile)
            const int s4 = u & 3, r32 = u >> 2;                          // 64-byte-row images: chunk s4 of rows r32 + 32 j
            const __amdgpu_buffer_rsrc_t y2g = qrsrc(p.x, (size_t)p.M * QCB * 2), idg = qrsrc(p.x2, (size_t)p.M * QC * 2);
            const __amdgpu_buffer_rsrc_t ag = qrsrc(p.a_out, (size_t)p.M * QC * 2);
            const __amdgpu_buffer_rsrc_t s3g = qrsrc(p.mid_scale, (size_t)QC * 4), h3g = qrsrc(p.mid_shift, (size_t)QC * 4);
            const __amdgpu_buffer_rsrc_t s2g = qrsrc(IDAFF ? p.x2_scale : p.mid_scale, (size_t)QC * 4), h2g = qrsrc(IDAFF ? p.x2_shift : p.mid_shift, (size_t)QC * 4);
            const unsigned toffB = (unsigned)(r16 * QCB + s8 * 8) * 2u, toffC = (unsigned)(r32 * QC + s4 * 8) * 2u;
            struct IdSet { uint4 v[4]; };
            IdSet id0, id1, id2, id3;                                    // identity chunks of slots t (set t & 3)
            struct Aff { q32x4 s0, s1, h0, h1, i0, i1, j0, j1; };
            Aff A0, A1;                                                  // tables of slots t (set t & 1)
            uint4 a2r[8];                                                // one 64-channel K tile of the next tile's raw y2 rows
            auto load_id = [&](int m0, int t, IdSet &st) __attribute__((always_inline)) {
#pragma unroll
                for (int j = 0; j < 4; ++j) st.v[j] = qld(idg, toffC, (unsigned)(m0 + 32 * j) * (QC * 2) + (unsigned)t * 64u);
            };
            auto load_aff = [&](int t, Aff &A) __attribute__((always_inline)) {
                const unsigned o2 = (unsigned)((t % QNT) * QSW + s4 * 8) * 4u;
                A.s0 = qldv(s3g, o2, 0); A.s1 = qldv(s3g, o2 + 16u, 0); A.h0 = qldv(h3g, o2, 0); A.h1 = qldv(h3g, o2 + 16u, 0);
                if constexpr (IDAFF) { A.i0 = qldv(s2g, o2, 0); A.i1 = qldv(s2g, o2 + 16u, 0); A.j0 = qldv(h2g, o2, 0); A.j1 = qldv(h2g, o2 + 16u, 0); }
            };
            auto load_a2 = [&](int m0, int kt) __attribute__((always_inline)) {
#pragma unroll
                for (int j = 0; j < 8; ++j) a2r[j] = qld(y2g, toffB, (unsigned)(m0 + 16 * j) * (QCB * 2) + (unsigned)kt * 128u);
            };
            auto stage_a2 = [&](int kt) __attribute__((always_inline)) {   // one K tile of relu(bn2(y2)) -> its LDS image
                float sc[8], sh[8];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const float4 a0 = *reinterpret_cast<const float4 *>(p.in_scale + kt * 64 + s8 * 8 + 4 * h);
                    const float4 a1 = *reinterpret_cast<const float4 *>(p.in_shift + kt * 64 + s8 * 8 + 4 * h);
                    sc[4 * h] = a0.x; sc[4 * h + 1] = a0.y; sc[4 * h + 2] = a0.z; sc[4 * h + 3] = a0.w;
                    sh[4 * h] = a1.x; sh[4 * h + 1] = a1.y; sh[4 * h + 2] = a1.z; sh[4 * h + 3] = a1.w;
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    float v[8];
                    Chunk<T>::unpack(a2r[j], v);
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[i] = fmaxf(fmaf(v[i], sc[i], sh[i]), 0.f);
                    Y2[kt * (QPIX * 8) + swz(r16 + 16 * j, s8)] = Chunk<T>::pack(v);
                }
            };
            // X(t): a = relu(bn3(T) + identity) for the 32 channels of slot t, in place on the ring buffer (conv_shared.h's in_transform order)
            auto xform = [&](int m0, int t, int b, const IdSet &st, const Aff &A, bool valid) __attribute__((always_inline)) {
                uint4 *Bt = BB + b * Q_BB;
                float sc[8], sh[8];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    sc[i] = __uint_as_float(A.s0[i]); sc[4 + i] = __uint_as_float(A.s1[i]);
                    sh[i] = __uint_as_float(A.h0[i]); sh[4 + i] = __uint_as_float(A.h1[i]);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int row = r32 + 32 * j;
                    float v[8], w[8];
                    Chunk<T>::unpack(Bt[qswz64(row, s4)], v);
                    Chunk<T>::unpack(st.v[j], w);
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        v[i] = fmaf(v[i], sc[i], sh[i]);
                        if constexpr (IDAFF) {
                            const float s2 = __uint_as_float(i < 4 ? A.i0[i & 3] : A.i1[i & 3]), h2 = __uint_as_float(i < 4 ? A.j0[i & 3] : A.j1[i & 3]);
                            w[i] = fmaf(w[i], s2, h2);
                        }
                        v[i] = fmaxf(v[i] + w[i], 0.f);
                    }
                    const uint4 o = Chunk<T>::pack(v);
                    Bt[qswz64(row, s4)] = o;
                    qst(ag, valid ? toffC : Q_OOB, (unsigned)(m0 + 32 * j) * (QC * 2) + (unsigned)t * 64u, o);
                }
            };
            {   // first tile: everything it needs before its first slot
                const int m0 = (int)blockIdx.x * QPIX;
                load_id(m0, 0, id0); load_id(m0, 1, id1); load_id(m0, 2, id2); load_id(m0, 3, id3);
                for (int kt = 0; kt < 4; ++kt) { load_a2(m0, kt); stage_a2(kt); }      // (the exposed loads of the workgroup)
            }
            for (int L = blockIdx.x; L < ntiles; L += G) {
                const int m0 = L * QPIX;
                const int Ln = L + G;
                const int m0n = Ln < ntiles ? Ln * QPIX : m0;            // (clamped: the loads of a tile that does not exist are harmless re-reads)
                load_aff(0, A0); load_aff(1, A1);
                load_a2(m0n, 0);                                         // the next tile's y2, a K tile per slot group: staged one group after its load
                __syncthreads();                                         // top
                // slot s: X(s - 1) with identity set (s - 1) & 3 and table set (s - 1) & 1, then their reloads for slots t + 2 / t + 4 (of this
                // tile or the next one) - the shorter lead first
                auto slot = [&](int s, IdSet &ids, Aff &A, bool xvalid) __attribute__((always_inline)) {
                    if (s > 0) __syncthreads();
                    const int t = s - 1;
                    if constexpr (!(QABL & 2)) xform(m0, t < 0 ? 0 : (t < QNT ? t : QNT - 1), (s + 3) & 3, ids, A, xvalid && !(QABL & 16));
                    if (xvalid) {                                        // (compile-time: the written-out groups; always true in the loop)
                        load_aff(t + 2, A);                              // (wraps to the next tile's first slots: the same tables)
                        const int tn = t + 4;
                        if constexpr (!(QABL & 8)) load_id(tn < QNT ? m0 : m0n, tn < QNT ? tn : tn - QNT, ids);
                    }
                };
                slot(0, id3, A1, false); slot(1, id0, A0, true); slot(2, id1, A1, true); slot(3, id2, A0, true);
                for (int g1 = 1; g1 < QG - 1; ++g1) {
                    const int sb = 4 * g1;
                    slot(sb, id3, A1, true);
                    // the next tile's y2 (every multiply wave took this tile's fragments at the top): group g stages K tile min(g, 3) and loads
                    // min(g + 1, 3) - the same two operations in every group (the last three repeat K tile 3: no branch around a load)
                    const int g = g1 - 1;
                    stage_a2(g < 3 ? g : 3);
                    load_a2(m0n, g + 1 < 3 ? g + 1 : 3);
                    slot(sb + 1, id0, A0, true); slot(sb + 2, id1, A1, true); slot(sb + 3, id2, A0, true);
                }
                slot(4 * (QG - 1), id3, A1, true); slot(4 * (QG - 1) + 1, id0, A0, false); slot(4 * (QG - 1) + 2, id1, A1, false);
                slot(4 * (QG - 1) + 3, id2, A0, false);
                __syncthreads();                                         // B1
                __syncthreads();                                         // B2: the outputs are staged
                if constexpr (!(QABL & 32)) store_outputs(m0);
                __syncthreads();                                         // B3
            }
        } else {
            // ---- W: the weight stages, global -> registers (two slots ahead) -> LDS; the host packed them as the stages' LDS images, so
            // both copies are linear
            const __amdgpu_buffer_rsrc_t w3g = qrsrc(p.w3, (size_t)QNT * Q_WS * 16), w1g = qrsrc(p.w, (size_t)QNT * Q_WS * 16);
            const unsigned toffW = (unsigned)u * 16u;
            struct WSet { q32x4 v[8]; };
            WSet w3a, w3b, w1a, w1b;
            auto load_w = [&](__amdgpu_buffer_rsrc_t g, int t, WSet &ws) __attribute__((always_inline)) {
#pragma unroll
                for (int i = 0; i < 8; ++i) ws.v[i] = qldv(g, toffW, (unsigned)(t * Q_WS + i * 128) * 16u);
            };
            auto store_w = [&](uint4 *dst, const WSet &ws) __attribute__((always_inline)) {
#pragma unroll
                for (int i = 0; i < 8; ++i) *reinterpret_cast<q32x4 *>(dst + i * 128 + u) = ws.v[i];
            };
            for (int L = blockIdx.x; L < ntiles; L += G) {
                const int m0 = L * QPIX;
                // pre-top: the first weight stages (W3[0] straight into its ring stage; W3[1], W3[2], W1[0] into their register sets)
                load_w(w3g, 0, w3b); load_w(w3g, 1, w3a); load_w(w1g, 0, w1a); load_w(w1g, 0, w1b);
                store_w(R3, w3b);
                load_w(w3g, 2, w3b);
                __syncthreads();                                         // top
                // slot s: W3[s + 1] -> R3[(s + 1) & 1] from set (s + 1) & 1, reloaded with W3[s + 3]; W1[s - 1] -> R1[(s - 1) & 1] from set
                // (s - 1) & 1, reloaded with W1[s + 1]
                auto slot = [&](int s, WSet &w3s, WSet &w1s) __attribute__((always_inline)) {
                    if (s > 0) __syncthreads();
                    if constexpr (!(QABL & 4)) {
                        store_w(R3 + ((s + 1) & 1) * Q_WS, w3s);
                        store_w(R1 + ((s + 1) & 1) * Q_WS, w1s);       // (s - 1) & 1 == (s + 1) & 1
                        load_w(w3g, (s + 3) % QNT, w3s);
                        load_w(w1g, (s + 1) % QNT, w1s);
                    }
                };
                slot(0, w3a, w1b); slot(1, w3b, w1a); slot(2, w3a, w1b); slot(3, w3b, w1a);
                for (int g1 = 1; g1 < QG - 1; ++g1) { const int sb = 4 * g1; slot(sb, w3a, w1b); slot(sb + 1, w3b, w1a); slot(sb + 2, w3a, w1b); slot(sb + 3, w3b, w1a); }
                slot(4 * (QG - 1), w3a, w1b); slot(4 * (QG - 1) + 1, w3b, w1a); slot(4 * (QG - 1) + 2, w3a, w1b); slot(4 * (QG - 1) + 3, w3b, w1a);
                __syncthreads();                                         // B1
                __syncthreads();                                         // B2: the outputs are staged
                if constexpr (!(QABL & 32)) store_outputs(m0);
                __syncthreads();                                         // B3
            }
        }
    }
    // ---- batch statistics of y1 as stored: fold the 256 transfer threads' partial sums once per workgroup (8 threads share a column chunk)
    if (st_on) {
        float *red = reinterpret_cast<float *>(Y2);
        __syncthreads();
        if (!mult) {
#pragma unroll
            for (int i = 0; i < 8; ++i) { red[t2 * 16 + i] = ss1[i]; red[t2 * 16 + 8 + i] = ss2[i]; }
        }
        __syncthreads();
        if (tid < QN2) {
            const int ch = tid >> 3, e = tid & 7;
            float a = 0.f, b = 0.f;
            for (int k = 0; k < 8; ++k) { a += red[(ch + 32 * k) * 16 + e]; b += red[(ch + 32 * k) * 16 + 8 + e]; }
            fx::add(p.stats, (int)blockIdx.x % NSH, 0, QN2, tid, a);
            fx::add(p.stats, (int)blockIdx.x % NSH, 1, QN2, tid, b);
        }
    }
}

bool fuse256_supports(const Params &p) {
    return p.x && p.x2 && p.w && p.w3 && p.y && p.a_out && p.in_scale && p.in_shift && p.mid_scale && p.mid_shift && !p.a_bits &&
           p.Cin == QC && p.Cout == QN2 && p.KH == 1 && p.KW == 1 && p.stride == 1 && p.pad == 0 && p.M % QPIX == 0 && p.M > 0 &&
           (size_t)p.M * QC * 2 < 0x7fff0000ull;
}

int launch_fuse256(const Params &p, hipStream_t s) {
    const int ntiles = p.M / QPIX;
    const dim3 grid((unsigned)(ntiles < 256 ? ntiles : 256)), block(512);
    if (p.x2_scale) hipLaunchKernelGGL(bottleneck_tail256_kernel<true>, grid, block, 0, s, p);
    else hipLaunchKernelGGL(bottleneck_tail256_kernel<false>, grid, block, 0, s, p);
    return check_launch("bottleneck_tail256_kernel");
}

}}  // namespace mhe::conv

using namespace mhe;

extern "C" int mhe_bottleneck_tail256_supported(const mhe_conv_desc *d) {
    if (!d || d->dtype != MHE_BF16 || d->Cin != 1024 || d->Cout != 256 || d->KH != 1 || d->KW != 1 || d->stride != 1 || d->pad != 0) return 0;
    const long long M = (long long)d->B * d->H * d->W;
    return M > 0 && M % 128 == 0 && M * 2048 < 0x7fff0000ll;
}

extern "C" int mhe_bottleneck_tail256_nhwc(const mhe_conv_desc *d, const void *y2, const float *bn2_scale, const float *bn2_shift, const void *w3_stages,
                                           const float *bn3_scale, const float *bn3_shift, const void *identity, const float *id_scale,
                                           const float *id_shift, const void *w1_stages, void *a_out, void *y1, mhe_stat_t *stats, void *stream) {
    MHE_REQUIRE(d && y2 && bn2_scale && bn2_shift && w3_stages && bn3_scale && bn3_shift && identity && w1_stages && a_out && y1, "mhe_bottleneck_tail256_nhwc: null pointer");
    MHE_REQUIRE((id_scale == nullptr) == (id_shift == nullptr), "mhe_bottleneck_tail256_nhwc: id_scale/id_shift must come together");
    MHE_REQUIRE(mhe_bottleneck_tail256_supported(d), "mhe_bottleneck_tail256_nhwc: bf16, block width 1024 (bottleneck 256), 256 output channels, pixels %% 128 == 0");
    conv::Params p{};
    p.x = y2; p.in_scale = bn2_scale; p.in_shift = bn2_shift; p.w3 = w3_stages; p.mid_scale = bn3_scale; p.mid_shift = bn3_shift;
    p.x2 = identity; p.x2_scale = id_scale; p.x2_shift = id_shift; p.w = w1_stages; p.a_out = a_out; p.y = y1; p.stats = stats;
    p.B = d->B; p.H = d->H; p.W = d->W; p.Cin = d->Cin; p.Cout = d->Cout; p.KH = 1; p.KW = 1; p.stride = 1; p.pad = 0;
    p.Ho = d->H; p.Wo = d->W; p.Kpad = d->Cin; p.relu_in = 1; p.force = -1;
    p.M = (int)((long long)d->B * d->H * d->W);
    MHE_REQUIRE(conv::fuse256_supports(p), "mhe_bottleneck_tail256_nhwc: unsupported launch");
    return conv::launch_fuse256(p, (hipStream_t)stream);
}

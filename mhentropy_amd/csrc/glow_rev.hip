// Conditional Glow, REVERSE of the sampling direction: the data-gradient chain of all layers in ONE launch (round 5; train step of the Glow
// branch, bf16 mode, hidden 512, 64 hypotheses per image).  Forward: csrc/glow_fwd.hip, whose tape this kernel reads; reference call site
// hand/network.py:736-742 differentiated by autograd (hand/CrossModalHand.py:455-470); algorithm oracle/glow_ref.py (PARITY UNPINNED).
//
// Until this kernel the reverse pass was, per layer, ~25 launches (45 x 45 product, coupling reverse, final-layer reverse, per block: gate reverse,
// product, dropout + ReLU reverse, product, ReLU reverse; initial-layer reverse, per-image sums) with every [R, 512] gradient through HBM.
// Here a workgroup owns ONE IMAGE = its 64 hypothesis rows (r = n B + b) for the whole chain, exactly like flow_rev.hip's chain_kernel:
//   gy   = gv A^-1                              45 x 45 on the VALU from LDS copies (gv = dL / d layer output, lives in the flow variable's lanes)
//   coupling reverse where v lives: g_shift = -gy / scale, g_us = (-gy (v - shift) / scale^2 + a_q / scale) sig (1 - sig), identity columns pass
//   gh   = g_shift Ws'^T + g_us Wu'^T           K = 64 + 64 (the final layer in the flow variable's column order)
//   per block (last first):  g_t3 = gh s (s = sigmoid(gate[image]));  gate gradient = s (1 - s) sum_rows gh t3;  g_t2 = (g_t3 W1) [t2 != 0] / (1 - p);
//                            gh += (g_t2 W0) [relu(h) > 0]
//   gv'  = g_v(coupling) + gh Wx                K = 512 -> 64, one 16-column tile x two row tiles per wave
// eight waves, a wave = 64 rows x 64 hidden units of every 512-wide product (accumulators [unit tile 4][row tile 4] of v_mfma_f32_16x16x32_bf16): a
// column of a [64 x 512] gradient lies in ONE wave, so the sums over an image's rows (gate gradients, bias rows, the initial layer's context
// gradient) are in-lane sums + one DPP row reduction - plain stores, no atomics, no second pass.  Of the kept activations only t3 is read as
// values (the gate gradient needs it); the two ReLU gates of a block come as bits in the accumulator layout.
// Left behind for the weight gradients (grouped launches over the tape): g_t3, g_t2 (bf16 [L][2][R][512]), gh at the initial layer (bf16
// [L][R][512]), [g_shift | g_us] (bf16 [L][R][128], column order), gv (f32 [L][R][64], for dA^-1 / dc^-1), and the per-image rows.
#include "flow_frag.h"
#include "../../include/mhe.h"

namespace mhe { namespace glowr {

using namespace flowfrag;

constexpr int H = 512, ROWS = 64, KT = H / 64, NBLK = 2, YP = 68, MAXL = 16;

struct Args {
    const float *g_x, *g_logp;                            // [R][dim] dL / d sample; [B] dL / d log_p per image or NULL
    const float *v_e, *prmc_e;                            // tape: [L][R][64], [L][R][128]
    const u16 *t3_e;                                      // tape: [L][NBLK][R][512]
    const uint2 *bits_e;                                  // tape: [L][NBLK][2][B][512]
    const float *ctab;                                    // [B][cstride] (gate pre-activations at slots l * 3 + 1 + blk)
    const u16 *wsT, *wuT, *w1T, *w0T, *wxT;               // fragment-major bf16: [L][512][64] x 2 (row u, k = column), [L][NBLK][512][512] x 2 (W^T), [L][64][512]
    const float *ainv;                                    // [L][64][64]: A^-1 [i][j]
    float drop_scale, q_weight;
    float *gv_e;                                          // [L][R][64]
    u16 *gpc_e, *gt3_e, *gt2_e, *gh0_e;                   // [L][R][128], [L][NBLK][R][512] x 2, [L][R][512]
    float *gct, *bsum, *bfsum;                            // [B][cstride], [B][L * NBLK * 2 * 512], [B][L * 128]
    int R, B, dim, L, cstride;
};

__global__ __launch_bounds__(512) void chain_kernel(const Args a) {
    __shared__ uint4 act[KT * ROWS * 8];                  // 64 KiB: the operand image of the 512-wide products
    __shared__ uint4 gsb[2][ROWS * 8];                    // g_shift / g_us as [64 rows][64 columns] bf16 operand tiles
    __shared__ float gvb[ROWS * YP];                      // gv (f32) for the 45 x 45 product
    __shared__ float aI[64 * 64];                         // A^-1 of the layer in flight
    __shared__ float colred[2][2][64];                    // per-image sums of g_shift / g_us: [row half][which][column]
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, q = lane >> 4, l15 = lane & 15;
    const int B = a.B, R = a.R, dim = a.dim, b = blockIdx.x;
    const int xr = (l15 >> 1) & 7, r8 = lane >> 3, c8 = lane & 7;
    unsigned fa_off[2], ac_off[4], pc_off[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) fa_off[kk] = (unsigned)((l15 * 8 + ((kk * 4 + q) ^ xr)) * 16);
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) ac_off[nt] = (unsigned)((l15 * 8 + ((nt * 2 + (q >> 1)) ^ xr)) * 16 + (q & 1) * 8);
#pragma unroll
    for (int ip = 0; ip < 2; ++ip) pc_off[ip] = (unsigned)((r8 * 8 + (c8 ^ ((ip * 4 + (r8 >> 1)) & 7))) * 16);
    unsigned char *const actb = reinterpret_cast<unsigned char *>(act);
    unsigned char *const tile = actb + wave * 8192;
    const int nt3 = wave >> 1, mt3 = 2 * (wave & 1), d0 = nt3 * 16 + 4 * q;
    const unsigned xw_off = (unsigned)((l15 * 8 + ((nt3 * 2 + (q >> 1)) ^ xr)) * 16 + (q & 1) * 8 + mt3 * 2048);
    const unsigned lane16 = (unsigned)lane * 16u;
    const unsigned co = (unsigned)(((r8 * B + b) * H + 64 * wave + c8 * 8) * 2);
    const unsigned cstep = (unsigned)(8 * B * H) * 2u;
    const size_t hbytes = (size_t)R * H * 2;
    int grow[2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) grow[mi] = ((mt3 + mi) * 16 + l15) * B + b;
    const float aq = a.g_logp ? a.g_logp[b] * a.q_weight : 0.f;
    const unsigned cq = (unsigned)(64 * wave + 4 * q) * 4u;
    const rsrc_t ctr = rsrc_of(a.ctab + (size_t)b * a.cstride, (size_t)a.cstride * 4);

    v4f gvr[2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int e = 0; e < 4; ++e) gvr[mi][e] = d0 + e < dim ? a.g_x[(size_t)grow[mi] * dim + d0 + e] : 0.f;

    auto emit_tile = [&](u16 *base, size_t index) __attribute__((always_inline)) {
        const rsrc_t hr = rsrc_of(base + index * (size_t)R * H, hbytes);
        wave_sync();
#pragma unroll
        for (int i = 0; i < 8; ++i) bst(hr, co, i * cstep, *reinterpret_cast<const uint4 *>(tile + pc_off[i & 1] + 1024 * i));
    };
    // sum over the image's 64 rows of the wave's [4 unit tiles][4 row tiles] x 4 values: in-lane over the row tiles, DPP over the 16 rows of a tile;
    // the lanes l15 == 0 store unit 64 wave + 16 nt + 4 q + e
    auto store_colsum = [&](float *dst, int nt, const float (&sv)[4]) __attribute__((always_inline)) {
        v4f o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = row16_sum(sv[e]);
        if (l15 == 0) *reinterpret_cast<v4f *>(dst + 64 * wave + 16 * nt + 4 * q) = o;
    };

    for (int l = 0; l < a.L; ++l) {
        const int first = 1 - (l & 1), slot0 = l * (1 + NBLK);
        // ---- (1) gv out (for dA^-1, dc^-1), gv and A^-1 into LDS
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
            *reinterpret_cast<v4f *>(a.gv_e + ((size_t)l * R + grow[mi]) * 64 + d0) = gvr[mi];
            *reinterpret_cast<v4f *>(gvb + ((mt3 + mi) * 16 + l15) * YP + d0) = gvr[mi];
        }
        {
            const float4 *src = reinterpret_cast<const float4 *>(a.ainv + (size_t)l * 4096);
            reinterpret_cast<float4 *>(aI)[tid] = src[tid];
            reinterpret_cast<float4 *>(aI)[tid + 512] = src[tid + 512];
        }
        // tape of the coupling, requested before the barrier
        v4f vv[2], sh[2], us[2];
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
            vv[mi] = *reinterpret_cast<const v4f *>(a.v_e + ((size_t)l * R + grow[mi]) * 64 + d0);
            sh[mi] = *reinterpret_cast<const v4f *>(a.prmc_e + ((size_t)l * R + grow[mi]) * 128 + d0);
            us[mi] = *reinterpret_cast<const v4f *>(a.prmc_e + ((size_t)l * R + grow[mi]) * 128 + 64 + d0);
        }
        __syncthreads();
        // ---- (2) gy = gv A^-1
        v4f gy[2] = {v4f{0.f, 0.f, 0.f, 0.f}, v4f{0.f, 0.f, 0.f, 0.f}};
        {
            const float *y0 = gvb + (mt3 * 16 + l15) * YP, *y1 = y0 + 16 * YP;
            for (int k4 = 0; k4 < 12; ++k4) {
                const float4 ya = *reinterpret_cast<const float4 *>(y0 + 4 * k4), yc = *reinterpret_cast<const float4 *>(y1 + 4 * k4);
                const float yav[4] = {ya.x, ya.y, ya.z, ya.w}, ycv[4] = {yc.x, yc.y, yc.z, yc.w};
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    const float4 ar = *reinterpret_cast<const float4 *>(aI + (4 * k4 + kk) * 64 + d0);
                    gy[0][0] = fmaf(ar.x, yav[kk], gy[0][0]); gy[0][1] = fmaf(ar.y, yav[kk], gy[0][1]); gy[0][2] = fmaf(ar.z, yav[kk], gy[0][2]); gy[0][3] = fmaf(ar.w, yav[kk], gy[0][3]);
                    gy[1][0] = fmaf(ar.x, ycv[kk], gy[1][0]); gy[1][1] = fmaf(ar.y, ycv[kk], gy[1][1]); gy[1][2] = fmaf(ar.z, ycv[kk], gy[1][2]); gy[1][3] = fmaf(ar.w, ycv[kk], gy[1][3]);
                }
            }
        }
        // ---- (3) the coupling's reverse where v lives; g_shift / g_us as bf16 operand tiles + out; their per-image sums
        v4f gvc[2];
        float cs[2][4];
#pragma unroll
        for (int e = 0; e < 4; ++e) cs[0][e] = cs[1][e] = 0.f;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
            float gs[4], gu[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int c = d0 + e;
                gs[e] = gu[e] = 0.f;
                gvc[mi][e] = c < dim ? gy[mi][e] : 0.f;
                if (c < dim && c >= first && ((c - first) & 1) == 0) {
                    const float sig = 1.f / (1.f + expf(-(us[mi][e] + 2.f))), scale = sig + 1e-3f;
                    const float gq = gy[mi][e] / scale;
                    gvc[mi][e] = gq;
                    gs[e] = -gq;
                    gu[e] = (-gq * (vv[mi][e] - sh[mi][e]) / scale + aq / scale) * sig * (1.f - sig);
                }
            }
            uint2 o, p;
            o.x = (unsigned)f32_to_bf16(gs[0]) | ((unsigned)f32_to_bf16(gs[1]) << 16); o.y = (unsigned)f32_to_bf16(gs[2]) | ((unsigned)f32_to_bf16(gs[3]) << 16);
            p.x = (unsigned)f32_to_bf16(gu[0]) | ((unsigned)f32_to_bf16(gu[1]) << 16); p.y = (unsigned)f32_to_bf16(gu[2]) | ((unsigned)f32_to_bf16(gu[3]) << 16);
            *reinterpret_cast<uint2 *>(reinterpret_cast<unsigned char *>(gsb[0]) + xw_off + 2048 * mi) = o;
            *reinterpret_cast<uint2 *>(reinterpret_cast<unsigned char *>(gsb[1]) + xw_off + 2048 * mi) = p;
            u16 *gp = a.gpc_e + ((size_t)l * R + grow[mi]) * 128 + d0;
            *reinterpret_cast<uint2 *>(gp) = o;
            *reinterpret_cast<uint2 *>(gp + 64) = p;
            // (the bias gradient sums the values as stored: what the weight gradient multiplies)
            cs[0][0] += bf16_to_f32((u16)(o.x & 0xffffu)); cs[0][1] += bf16_to_f32((u16)(o.x >> 16)); cs[0][2] += bf16_to_f32((u16)(o.y & 0xffffu)); cs[0][3] += bf16_to_f32((u16)(o.y >> 16));
            cs[1][0] += bf16_to_f32((u16)(p.x & 0xffffu)); cs[1][1] += bf16_to_f32((u16)(p.x >> 16)); cs[1][2] += bf16_to_f32((u16)(p.y & 0xffffu)); cs[1][3] += bf16_to_f32((u16)(p.y >> 16));
        }
#pragma unroll
        for (int w = 0; w < 2; ++w) {
            v4f o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = row16_sum(cs[w][e]);
            if (l15 == 0) *reinterpret_cast<v4f *>(&colred[wave & 1][w][d0]) = o;      // rows 32 (wave & 1) .. + 31
        }
        __syncthreads();                                  // operand tiles and the column partial sums are complete
        if (tid < 128) a.bfsum[(size_t)b * (a.L * 128) + l * 128 + tid] = colred[0][tid >> 6][tid & 63] + colred[1][tid >> 6][tid & 63];
        // ---- (4) gh = g_shift Ws'^T + g_us Wu'^T (K = 64 + 64)
        v4f acc[4][4], gh[4][4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) gh[nt][mt] = v4f{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int w = 0; w < 2; ++w) {
            const rsrc_t wr = rsrc_of((w ? a.wuT : a.wsT) + (size_t)l * H * 64, (size_t)H * 64 * 2);
            const unsigned char *ob = reinterpret_cast<const unsigned char *>(gsb[w]);
            uint4 wf[2][4];
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) wf[kk][nt] = frag(wr, lane16, 4 * wave + nt, 2, kk);
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                uint4 fa[4];
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) fa[mt] = *reinterpret_cast<const uint4 *>(ob + fa_off[kk] + 2048 * mt);
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt) gh[nt][mt] = mfma(wf[kk][nt], fa[mt], gh[nt][mt]);
            }
        }
        auto product = [&](const u16 *wbase) __attribute__((always_inline)) {
            const rsrc_t w1 = rsrc_of(wbase, (size_t)H * H * 2);
            // the residual gradient gh (64 VGPRs) stays live across this product: the weight fragments are fetched ONE 32-deep k step ahead in
            // three rotating sets of 4 (48 VGPRs, two steps in flight) instead of flow_fwd.hip's two sets of 8 (64 VGPRs)
            uint4 fS[3][4];
            auto fetch_w = [&](uint4 (&f)[4], int ks) __attribute__((always_inline)) {
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) f[nt] = frag(w1, lane16, 4 * wave + nt, 16, ks);
            };
            fetch_w(fS[0], 0);
            fetch_w(fS[1], 1);
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) acc[nt][mt] = v4f{0.f, 0.f, 0.f, 0.f};
            __builtin_amdgcn_sched_barrier(0);
            __syncthreads();                              // the operand image is complete
#pragma unroll
            for (int ks = 0; ks < 2 * KT; ++ks) {
                if (ks + 2 < 2 * KT) fetch_w(fS[(ks + 2) % 3], ks + 2);
                __builtin_amdgcn_sched_barrier(0);
                uint4 fa[4];
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) fa[mt] = *reinterpret_cast<const uint4 *>(actb + fa_off[ks & 1] + (8192 * (ks >> 1) + 2048 * mt));
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt) acc[nt][mt] = mfma(fS[ks % 3][nt], fa[mt], acc[nt][mt]);
                __builtin_amdgcn_sched_barrier(0);
            }
            __syncthreads();                              // every wave has read the image
        };
        // ---- (5) the residual blocks, last first
#pragma unroll 1
        for (int blk = NBLK - 1; blk >= 0; --blk) {
            const size_t lb = (size_t)l * NBLK + blk;
            // t3 of this block: whole rows into the wave's k-tile, then read back in the accumulator layout
            {
                const rsrc_t tr = rsrc_of(a.t3_e + lb * (size_t)R * H, hbytes);
#pragma unroll
                for (int i = 0; i < 8; ++i) *reinterpret_cast<uint4 *>(tile + pc_off[i & 1] + 1024 * i) = bld(tr, co, i * cstep);
                wave_sync();
            }
            float4 sg[4];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                sg[nt] = __builtin_bit_cast(float4, bld(ctr, cq, (unsigned)((slot0 + 1 + blk) * H + nt * 16) * 4u));
                sg[nt].x = 1.f / (1.f + expf(-sg[nt].x)); sg[nt].y = 1.f / (1.f + expf(-sg[nt].y));
                sg[nt].z = 1.f / (1.f + expf(-sg[nt].z)); sg[nt].w = 1.f / (1.f + expf(-sg[nt].w));
            }
            float *const gct_row = a.gct + (size_t)b * a.cstride + (size_t)(slot0 + 1 + blk) * H;
            float *const bs_row = a.bsum + (size_t)b * (a.L * NBLK * 2 * H) + 2 * lb * H;
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const float sv[4] = {sg[nt].x, sg[nt].y, sg[nt].z, sg[nt].w};
                float sgate[4] = {0.f, 0.f, 0.f, 0.f}, sbias[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    const uint2 tq = *reinterpret_cast<const uint2 *>(tile + ac_off[nt] + 2048 * mt);
                    const float t3[4] = {__uint_as_float(tq.x << 16), __uint_as_float(tq.x & 0xffff0000u), __uint_as_float(tq.y << 16), __uint_as_float(tq.y & 0xffff0000u)};
                    float g3[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float g = gh[nt][mt][e];
                        sgate[e] = fmaf(g, t3[e], sgate[e]);
                        sbias[e] += g;
                        g3[e] = g * sv[e];
                    }
                    uint2 o;
                    o.x = (unsigned)f32_to_bf16(g3[0]) | ((unsigned)f32_to_bf16(g3[1]) << 16);
                    o.y = (unsigned)f32_to_bf16(g3[2]) | ((unsigned)f32_to_bf16(g3[3]) << 16);
                    *reinterpret_cast<uint2 *>(tile + ac_off[nt] + 2048 * mt) = o;      // (same lane reads and writes this 8-byte slot)
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) { sgate[e] *= sv[e] * (1.f - sv[e]); sbias[e] *= sv[e]; }
                store_colsum(gct_row, nt, sgate);
                store_colsum(bs_row + H, nt, sbias);
            }
            emit_tile(a.gt3_e, lb);
            product(a.w1T + lb * H * H);
            // g_t2 = (g_t3 W1) [t2 != 0] / (1 - p)
            const uint2 bt_2 = a.bits_e[((lb * 2 + 1) * B + b) * 512 + wave * 64 + lane];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                float sbias[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    const int idx = (nt * 4 + mt) * 4;
                    const unsigned nib = ((idx < 32 ? bt_2.x >> idx : bt_2.y >> (idx - 32))) & 15u;
                    float g2[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) g2[e] = (nib >> e) & 1u ? acc[nt][mt][e] * a.drop_scale : 0.f;
                    uint2 o;
                    o.x = (unsigned)f32_to_bf16(g2[0]) | ((unsigned)f32_to_bf16(g2[1]) << 16);
                    o.y = (unsigned)f32_to_bf16(g2[2]) | ((unsigned)f32_to_bf16(g2[3]) << 16);
                    *reinterpret_cast<uint2 *>(tile + ac_off[nt] + 2048 * mt) = o;
                    sbias[0] += __uint_as_float(o.x << 16); sbias[1] += __uint_as_float(o.x & 0xffff0000u);
                    sbias[2] += __uint_as_float(o.y << 16); sbias[3] += __uint_as_float(o.y & 0xffff0000u);
                }
                store_colsum(bs_row, nt, sbias);
            }
            emit_tile(a.gt2_e, lb);
            product(a.w0T + lb * H * H);
            // gh += (g_t2 W0) [relu(h) > 0]
            const uint2 bt_h = a.bits_e[((lb * 2 + 0) * B + b) * 512 + wave * 64 + lane];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    const int idx = (nt * 4 + mt) * 4;
                    const unsigned nib = ((idx < 32 ? bt_h.x >> idx : bt_h.y >> (idx - 32))) & 15u;
#pragma unroll
                    for (int e = 0; e < 4; ++e) gh[nt][mt][e] += (nib >> e) & 1u ? acc[nt][mt][e] : 0.f;
                }
        }
        // ---- (6) gh at the initial layer: its per-image sums (the context term's gradient), bf16 -> image + out
        {
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                float sv[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    const v4f g = gh[nt][mt];
                    uint2 o;
                    o.x = (unsigned)f32_to_bf16(g[0]) | ((unsigned)f32_to_bf16(g[1]) << 16);
                    o.y = (unsigned)f32_to_bf16(g[2]) | ((unsigned)f32_to_bf16(g[3]) << 16);
                    *reinterpret_cast<uint2 *>(tile + ac_off[nt] + 2048 * mt) = o;
                    sv[0] += g[0]; sv[1] += g[1]; sv[2] += g[2]; sv[3] += g[3];
                }
                store_colsum(a.gct + (size_t)b * a.cstride + (size_t)slot0 * H, nt, sv);
            }
            emit_tile(a.gh0_e, (size_t)l);
        }
        // ---- (7) gv' = g_v(coupling) + gh Wx (K = 512 -> 64)
        {
            const rsrc_t wx = rsrc_of(a.wxT + (size_t)l * 64 * H, (size_t)64 * H * 2);
            uint4 w2f[16];
#pragma unroll
            for (int ks = 0; ks < 16; ++ks) w2f[ks] = frag(wx, lane16, nt3, 16, ks);
            v4f o[2] = {v4f{0.f, 0.f, 0.f, 0.f}, v4f{0.f, 0.f, 0.f, 0.f}};
            __builtin_amdgcn_sched_barrier(0);
            __syncthreads();                              // bf16(gh) complete in the image
#pragma unroll
            for (int ks = 0; ks < 16; ++ks)
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
                    o[mi] = mfma(w2f[ks], *reinterpret_cast<const uint4 *>(actb + mt3 * 2048 + fa_off[ks & 1] + (8192 * (ks >> 1) + 2048 * mi)), o[mi]);
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int e = 0; e < 4; ++e) gvr[mi][e] = d0 + e < dim ? gvc[mi][e] + o[mi][e] : 0.f;
        }
        __syncthreads();                                  // the image, gvb, A^-1 and the operand tiles may be overwritten
    }
}

}}  // namespace mhe::glowr

using namespace mhe;

extern "C" int mhe_glow_reverse_chain_supported(int R, int B, int dim, int hidden, int layers, int blocks) {
    return R > 0 && B > 0 && R == 64 * B && dim > 1 && dim <= 48 && hidden == 512 && layers > 0 && layers <= glowr::MAXL && blocks == glowr::NBLK;
}

extern "C" int mhe_glow_reverse_chain_bf16(const float *g_x, const float *g_log_p, float q_weight, const float *v_e, const float *prmc_e, const void *t3_e,
                                           const void *bits_e, const float *ctab, int ctab_stride, const void *wsT, const void *wuT, const void *w1T,
                                           const void *w0T, const void *wxT, const float *ainv, float p_drop, float *gv_e, void *gpc_e, void *gt3_e,
                                           void *gt2_e, void *gh0_e, float *gct, float *bsum, float *bfsum, int R, int B, int dim, int hidden, int layers,
                                           int blocks, void *stream) {
    MHE_REQUIRE(g_x && v_e && prmc_e && t3_e && bits_e && ctab && wsT && wuT && w1T && w0T && wxT && ainv && gv_e && gpc_e && gt3_e && gt2_e && gh0_e && gct &&
                    bsum && bfsum, "mhe_glow_reverse_chain_bf16: null pointer");
    MHE_REQUIRE(mhe_glow_reverse_chain_supported(R, B, dim, hidden, layers, blocks),
                "mhe_glow_reverse_chain_bf16: needs hidden 512, 2 blocks per layer, 64 hypotheses per image, dim <= 48 (R=%d B=%d dim=%d)", R, B, dim);
    MHE_REQUIRE(ctab_stride % 4 == 0 && ctab_stride >= layers * (1 + blocks) * hidden, "mhe_glow_reverse_chain_bf16: bad context-table stride");
    MHE_REQUIRE(p_drop >= 0.f && p_drop < 1.f && (long)R * hidden < (1L << 30), "mhe_glow_reverse_chain_bf16: bad arguments");
    glowr::Args a;
    a.g_x = g_x; a.g_logp = g_log_p; a.q_weight = q_weight; a.v_e = v_e; a.prmc_e = prmc_e; a.t3_e = (const u16 *)t3_e; a.bits_e = (const uint2 *)bits_e;
    a.ctab = ctab; a.cstride = ctab_stride; a.wsT = (const u16 *)wsT; a.wuT = (const u16 *)wuT; a.w1T = (const u16 *)w1T; a.w0T = (const u16 *)w0T;
    a.wxT = (const u16 *)wxT; a.ainv = ainv; a.drop_scale = 1.f / (1.f - p_drop); a.gv_e = gv_e; a.gpc_e = (u16 *)gpc_e; a.gt3_e = (u16 *)gt3_e;
    a.gt2_e = (u16 *)gt2_e; a.gh0_e = (u16 *)gh0_e; a.gct = gct; a.bsum = bsum; a.bfsum = bfsum; a.R = R; a.B = B; a.dim = dim; a.L = layers;
    hipLaunchKernelGGL(glowr::chain_kernel, dim3(B), dim3(512), 0, (hipStream_t)stream, a);
    return check_launch("glowr::chain_kernel");
}

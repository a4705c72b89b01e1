// RealNVP reverse pass, data-gradient chain of ALL couplings in one launch (train step, bf16 mode, hidden = 512, forward activations
// kept by mhe_flow_couplings_bf16_emit; reference hand/flows.py:97-122,210-217 differentiated by autograd, hand/CrossModalHand.py:455-470).
//
// Per coupling (last -> first) the chain was 13 launches - mask_pad, couple_bwd, then per net: GO W2 (bf16 GEMM), leaky-ReLU reverse + per-image
// sums, G2 W1, leaky-ReLU reverse + sums, G1 W0, and couple_accum - 156 launches and 2.3 ms per step at config C2, every intermediate a round
// trip through HBM.  Here a workgroup owns ONE IMAGE = its 64 hypothesis rows for the whole chain (N = 64: the rows r = n B + b of image b),
// keeps the flow variable, its gradient and the 64 x 512 activation gradients on chip, and leaves behind only what the weight gradients need
// (GO, G2, G1, the masked inputs: read by the grouped launches of mhe_conv_wgrad_batched_nhwc), the conditioning table's gradient (per-image
// sums = sums over the workgroup's own rows: plain stores, no atomics) and the l2 bias gradients.
//
//   eight waves, a wave = 64 rows x 64 units of every 512-wide product (accumulators [unit tile 4][row tile 4] of v_mfma_f32_16x16x32_bf16):
//   phase 1  P2 = GO W2        K = 64    operand GO from LDS, the W2^T fragments of the wave's units fetched one net ahead
//            G2 = P2 * lrelu'(H2) -> bf16 -> LDS k-tile `wave` (+ global G2, + column sums over the 64 rows -> d cond[b][layer 1])
//   phase 2  GH1 = G2 W1       K = 512   operand G2 from LDS (all eight k-tiles), W1^T fragments from global (L2), one k-tile ahead
//            G1 = GH1 * lrelu'(H1) -> bf16 -> LDS k-tile `wave` (+ global G1, + sums -> d cond[b][layer 0])
//   phase 3  GX += G1 W0       K = 512   a wave = one 16-dim tile x two row tiles, both nets accumulated in registers
//   then the coupling's element-wise reverse (tanh / exp, hand/flows.py:213-216) on the 64 x 45 flow variable held in LDS.
// Every global access is whole 128-byte rows or 1 KiB runs: the outgoing G2 / G1 move as [8 rows][128 B] pieces out of the wave's own k-tile;
// of the kept activations only the SIGNS are read (64 bits per lane and epilogue, laid down by the forward kernel in the accumulator layout); the weight operands are FRAGMENT-MAJOR copies of the train step's
// bf16 packs (w2F from W2^T [512][64], w1F from W1^T [512][512], w0F from W0^T [64][512]; see frag()), one pitch apart from net to net.
// The kernel is bound by the L2 -> register weight stream (640 KiB per net and workgroup, as the forward kernel's), not by its 0.5 TFLOP.
#include "flow_frag.h"
#include "../../include/mhe.h"

namespace mhe { namespace flowrev {

using namespace flowfrag;

constexpr int H = 512, ROWS = 64, XP = 64, XG = 68, KT = H / 64;
struct Args {
    const float *x_out, *g_x, *g_logp, *mask, *oe;       // [R][dim], [R][dim], [B] | NULL, [ncoup][dim], [nets][R][64]
    const uint2 *hbits;                                   // signs of the kept H1 / H2: [nets][B][layer 2][wave 8][lane 64] x 64 bits (flow_fwd.hip)
    const u16 *w2F, *w1F, *w0F;                           // net 0's fragment-major operands; net k lies k * w_stride elements further
    long w_stride;
    u16 *GOb, *G2b, *G1b, *XPb;                           // [nets][R][64], [nets][R][512] x 2, [ncoup][R][64]
    float *Gc, *db2, *z0;                                 // [B][cstride], per-image l2 bias-gradient rows [B][2 ncoup][64] (written), [R][dim]
    int R, B, dim, ncoup, cstride;
    float q_weight;
};

__global__ __launch_bounds__(512) void chain_kernel(const Args a) {
    __shared__ uint4 act[KT * ROWS * 8];                  // 64 KiB: G2, then G1, as eight [64 rows][64 units] k-tiles (16-byte chunks swizzled)
    __shared__ uint4 gos[2][ROWS * 8];                    // GOs / GOt of the coupling as [64 rows][64 dims] operand tiles
    __shared__ float xc[ROWS * XP], gc[ROWS * XP], gp[ROWS * XP];     // flow variable, its gradient, g_part
    __shared__ float gx[ROWS * XG];                       // GXs + GXt
    __shared__ float red[2][8][XP];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, q = lane >> 4, l15 = lane & 15;
    const int b = blockIdx.x, B = a.B, R = a.R, dim = a.dim;
    const int d = tid & 63, rg = tid >> 6;                // element-wise role: dim d of rows rg, rg + 8, ...
    // (this thread's element of row rg + 8 k of the LDS arrays: one pointer each + a constant per k)
    float *const xcp = xc + rg * XP + d, *const gcp = gc + rg * XP + d, *const gpp = gp + rg * XP + d, *const gxp = gx + rg * XG + d;
    for (int k = 0; k < 8; ++k) {
        const size_t r = (size_t)(rg + 8 * k) * B + b;
        xcp[k * 8 * XP] = d < dim ? a.x_out[r * dim + d] : 0.f;
        gcp[k * 8 * XP] = d < dim ? a.g_x[r * dim + d] : 0.f;
    }
    const float aq = a.g_logp ? a.g_logp[b] * a.q_weight : 0.f;
    // element offset of the 16-byte row pieces a lane moves between a [R][512] tensor and the wave's k-tile: piece i = row 8 i + lane / 8,
    // chunk lane % 8 - a wave-instruction covers eight whole 128-byte rows (32-bit on purpose: 64-bit row addresses were hoisted and spilled)
    const unsigned co = (unsigned)(((lane >> 3) * B + b) * H + 64 * wave + (lane & 7) * 8) * 2u;     // (bytes)
    const unsigned cstep = (unsigned)(8 * B * H) * 2u, lane16 = (unsigned)lane * 16u;
    const size_t hbytes = (size_t)R * H * 2;
    // LDS byte offsets of the swizzled tiles, written so that everything but one lane-dependent term is an instruction's immediate (as
    // swz(row, slot) per use hipcc kept some sixty distinct addresses alive across the kernel and spilled them):
    //   fa_off[kk]   the MFMA operand piece (row l15, k step kk) of row tile 0        + 2048 per row tile, + 8192 per k-tile
    //   ac_off[nt]   the accumulator's 8 bytes (row l15, unit tile nt) of row tile 0   + 2048 per row tile
    //   pc_off[i&1]  the 16-byte piece (row lane / 8, chunk lane % 8) a lane moves     + 1024 i
    //   el_off[k&1]  the element-wise stage's bf16 element (row rg, dim d)             + 1024 k
    const int xr = (l15 >> 1) & 7, r8 = lane >> 3, c8 = lane & 7;
    unsigned fa_off[2], ac_off[4], pc_off[2], el_off[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) fa_off[kk] = (unsigned)((l15 * 8 + ((kk * 4 + q) ^ xr)) * 16);
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) ac_off[nt] = (unsigned)((l15 * 8 + ((nt * 2 + (q >> 1)) ^ xr)) * 16 + (q & 1) * 8);
#pragma unroll
    for (int ip = 0; ip < 2; ++ip) pc_off[ip] = (unsigned)((r8 * 8 + (c8 ^ ((ip * 4 + (r8 >> 1)) & 7))) * 16);
#pragma unroll
    for (int kp = 0; kp < 2; ++kp) el_off[kp] = (unsigned)((rg * 8 + ((d >> 3) ^ (rg >> 1) ^ (4 * kp))) * 16 + (d & 7) * 2);
    unsigned char *const actb = reinterpret_cast<unsigned char *>(act), *const gosb = reinterpret_cast<unsigned char *>(gos);
    unsigned char *const tile = actb + wave * 8192;       // this wave's k-tile
    const int nt3 = wave >> 1, mt3 = 2 * (wave & 1);      // phase 3: this wave's dim tile and pair of row tiles of GX
    v4f a3[2] = {v4f{0.f, 0.f, 0.f, 0.f}, v4f{0.f, 0.f, 0.f, 0.f}};
    // fetched a net ahead, under phase 3: phase 1's W2^T fragments and the kept H2 rows of its epilogue
    uint4 w2f[2][4];
    uint2 hc2;
    // the signs of the kept activations under this lane's accumulators: bit (unit tile nt * 4 + row tile mt) * 4 + e, written by the forward
    // kernel in this very layout - 8 bytes per lane and epilogue where the bf16 rows were 32 VGPRs and 16 KiB of HBM per wave
    auto fetch_bits = [&](int net, int layer) __attribute__((always_inline)) {
        return a.hbits[(((size_t)net * B + b) * 2 + layer) * 512 + wave * 64 + lane];
    };
    auto fetch_net = [&](int net) __attribute__((always_inline)) {
        const rsrc_t w2 = rsrc_of(a.w2F + (size_t)net * a.w_stride, (size_t)H * 64 * 2);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) w2f[kk][nt] = frag(w2, lane16, 4 * wave + nt, 2, kk);
        hc2 = fetch_bits(net, 1);
    };
    // the coupling's kept s / t pre-activations of this thread's eight elements, requested at the end of the coupling before: read where
    // they are used - inside the branch on the mask - they were eight dependent HBM round trips per coupling
    float os[8], ot[8];
    // (element (row rg + 8 k, dim d) of a [R][64] tensor: e0 + k * estep, 32-bit - as size_t expressions the sixteen addresses were spilled)
    const unsigned e0 = (unsigned)((rg * B + b) * 64 + d), estep = (unsigned)(8 * B * 64);
    const size_t ebytes = (size_t)R * 64 * 4;
    auto fetch_oe = [&](int ci) __attribute__((always_inline)) {
        const rsrc_t ps = rsrc_of(a.oe + (size_t)(2 * ci) * R * 64, 2 * ebytes);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            os[k] = d < dim ? __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ps, (int)(e0 * 4u), (int)(k * estep * 4u), 0)) : 0.f;
            ot[k] = d < dim ? __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ps, (int)(e0 * 4u), (int)(k * estep * 4u + (unsigned)ebytes), 0)) : 0.f;
        }
    };
    fetch_net(2 * (a.ncoup - 1));
    fetch_oe(a.ncoup - 1);
    __syncthreads();

    for (int ci = a.ncoup - 1; ci >= 0; --ci) {
        // ---- the coupling's element-wise reverse: x_in, GOs, GOt, g_part; bf16 operand tiles; what the weight gradients read
        const float m = d < dim ? a.mask[ci * dim + d] : 1.f;
        float cs = 0.f, ct = 0.f;
        const rsrc_t pG = rsrc_of(a.GOb + (size_t)(2 * ci) * R * 64, ebytes), pXP = rsrc_of(a.XPb + (size_t)ci * R * 64, ebytes / 2);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            float vs = 0.f, vt = 0.f, xpv = 0.f;
            if (d < dim) {
                const float xo = xcp[k * 8 * XP], go = gcp[k * 8 * XP];
                float xi = xo, gpv = go;
                if (m == 0.f) {
                    const float s = tanhf(os[k]), t = ot[k];
                    const float es = expf(s);
                    xi = (xo - t) / es;
                    vs = (go * xi * es - aq) * (1.f - s * s);
                    vt = go;
                    gpv = go * es;
                }
                xcp[k * 8 * XP] = xi;
                gpp[k * 8 * XP] = gpv;
                xpv = xo * m;
            }
            const u16 bs = f32_to_bf16(vs), bt = f32_to_bf16(vt);
            *reinterpret_cast<u16 *>(gosb + el_off[k & 1] + 1024 * k) = bs;
            *reinterpret_cast<u16 *>(gosb + 8192 + el_off[k & 1] + 1024 * k) = bt;
            __builtin_amdgcn_raw_buffer_store_b16(bs, pG, (int)(e0 * 2u), (int)(k * estep * 2u), 0);
            __builtin_amdgcn_raw_buffer_store_b16(bt, pG, (int)(e0 * 2u), (int)(k * estep * 2u + (unsigned)(ebytes / 2)), 0);
            __builtin_amdgcn_raw_buffer_store_b16(f32_to_bf16(xpv), pXP, (int)(e0 * 2u), (int)(k * estep * 2u), 0);
            cs += vs; ct += vt;
        }
        red[0][rg][d] = cs; red[1][rg][d] = ct;
        __syncthreads();                                  // operand tiles written; red filled
        if (tid < 128) {
            const int w2 = tid >> 6;
            float v = 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) v += red[w2][k][d];
            // this image's share of net (2 ci + w2)'s l2 bias gradient: a plain store - the sum over images is a fixed-order column sum
            // afterwards (f32 atomics from the B workgroups up to round 3: order-dependent)
            a.db2[((size_t)b * (2 * a.ncoup) + 2 * ci + w2) * 64 + d] = v;
        }
#pragma unroll 1
        for (int n = 0; n < 2; ++n) {
            const int net = 2 * ci + n, slot = net * 2;
            const rsrc_t w1 = rsrc_of(a.w1F + (size_t)net * a.w_stride, (size_t)H * H * 2), w0 = rsrc_of(a.w0F + (size_t)net * a.w_stride, (size_t)64 * H * 2);
            const rsrc_t G2o = rsrc_of(a.G2b + (size_t)net * R * H, hbytes), G1o = rsrc_of(a.G1b + (size_t)net * R * H, hbytes);
            v4f acc[4][4];                                // [unit tile of the wave's 64][row tile]
            uint2 hc1;
            // leaky-ReLU reverse of the accumulators against the kept activation's sign, bf16 result into the k-tile, column sums over the 64
            // rows -> conditioning gradient, the finished tile out to Go
            auto epilogue = [&](const uint2 hc, rsrc_t Go, int cslot) __attribute__((always_inline)) {
                float csum[4][4];
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                    for (int e = 0; e < 4; ++e) csum[nt][e] = 0.f;
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt) {
                        uint2 *pc = reinterpret_cast<uint2 *>(tile + ac_off[nt] + 2048 * mt);
                        const unsigned sg = ((nt * 4 + mt) < 8 ? hc.x : hc.y) >> (((nt * 4 + mt) & 7) * 4);
                        v4f g = acc[nt][mt];
                        g[0] = (sg & 1u) ? g[0] : 0.01f * g[0]; g[1] = (sg & 2u) ? g[1] : 0.01f * g[1];
                        g[2] = (sg & 4u) ? g[2] : 0.01f * g[2]; g[3] = (sg & 8u) ? g[3] : 0.01f * g[3];
                        csum[nt][0] += g[0]; csum[nt][1] += g[1]; csum[nt][2] += g[2]; csum[nt][3] += g[3];
                        uint2 o;
                        o.x = (unsigned)f32_to_bf16(g[0]) | ((unsigned)f32_to_bf16(g[1]) << 16);
                        o.y = (unsigned)f32_to_bf16(g[2]) | ((unsigned)f32_to_bf16(g[3]) << 16);
                        *pc = o;
                    }
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    float4 v;
                    v.x = row16_sum(csum[nt][0]); v.y = row16_sum(csum[nt][1]); v.z = row16_sum(csum[nt][2]); v.w = row16_sum(csum[nt][3]);
                    if (l15 == 0) *reinterpret_cast<float4 *>(a.Gc + (size_t)b * a.cstride + cslot * H + 64 * wave + nt * 16 + 4 * q) = v;
                }
                wave_sync();
#pragma unroll
                for (int i = 0; i < 8; ++i) bst(Go, co, i * cstep, *reinterpret_cast<const uint4 *>(tile + pc_off[i & 1] + 1024 * i));
            };
            // ================= phase 1: P2 = GO W2 (K = 64)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) acc[nt][mt] = v4f{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                uint4 fa[4];
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) fa[mt] = *reinterpret_cast<const uint4 *>(gosb + n * 8192 + fa_off[kk] + 2048 * mt);
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt) acc[nt][mt] = mfma(w2f[kk][nt], fa[mt], acc[nt][mt]);
            }
            // the signs of H1 for phase 2's epilogue: requested here, under phase 1's epilogue, so that the load is older than every W1
            // fragment of phase 2 (loads return in order: asked for later it would hold k-tiles up behind an HBM round trip)
            __builtin_amdgcn_sched_barrier(0);
            hc1 = fetch_bits(net, 0);
            if (n == 1) __syncthreads();                  // (B0) every wave is through net 0's phase 3: act is free (net 0: the barriers above)
            epilogue(hc2, G2o, slot + 1);
            __syncthreads();                              // (B1) G2 complete in act
            // ================= phase 2: GH1 = G2 W1 (K = 512).  The W1^T fragments come straight from global (L2) into two register sets,
            // k-tile kt + 1's in flight under k-tile kt's 32 MFMAs; the scheduling fences pin that order (left alone, hipcc sank every load
            // to just above its first use - eight exposed L2 round trips per k-tile)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) acc[nt][mt] = v4f{0.f, 0.f, 0.f, 0.f};
            uint4 fA[2][4], fB[2][4];
            auto fetch_w = [&](uint4 (&f)[2][4], int kt) __attribute__((always_inline)) {
#pragma unroll
                for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) f[kk][nt] = frag(w1, lane16, 4 * wave + nt, 16, 2 * kt + kk);
            };
            auto ktile = [&](const uint4 (&f)[2][4], int kt) __attribute__((always_inline)) {
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    uint4 fa[4];
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt) fa[mt] = *reinterpret_cast<const uint4 *>(actb + fa_off[kk] + (8192 * kt + 2048 * mt));
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                        for (int mt = 0; mt < 4; ++mt) acc[nt][mt] = mfma(f[kk][nt], fa[mt], acc[nt][mt]);
                }
            };
            fetch_w(fA, 0);
#pragma unroll
            for (int p = 0; p < KT / 2; ++p) {
                fetch_w(fB, 2 * p + 1);
                __builtin_amdgcn_sched_barrier(0);
                ktile(fA, 2 * p);
                __builtin_amdgcn_sched_barrier(0);
                if (p + 1 < KT / 2) fetch_w(fA, 2 * p + 2);
                __builtin_amdgcn_sched_barrier(0);
                ktile(fB, 2 * p + 1);
                __builtin_amdgcn_sched_barrier(0);
            }
            __syncthreads();                              // (B2) every wave has read G2: act is free for G1
            // the W0^T fragments of phase 3 (this wave's dim tile, all 16 k steps), in flight under the epilogue
            uint4 w0f[16];
#pragma unroll
            for (int ks = 0; ks < 16; ++ks) w0f[ks] = frag(w0, lane16, nt3, 16, ks);
            __builtin_amdgcn_sched_barrier(0);
            epilogue(hc1, G1o, slot);
            __syncthreads();                              // (B3) G1 complete in act
            // ================= phase 3: GX += G1 W0 (K = 512): a wave = one dim tile x two row tiles, accumulated over both nets in registers
            // (as a split over the waves' own k-tiles the 64 ds_add_f32 per wave and net that merged the partial tiles took 0.9 ms per step)
            // (always: the next net in the chain's order - the t net of this coupling, or the s net of the one before it; fetched only on one
            // of the two rounds the registers would count as live through the whole round, phase 2 included)
            fetch_net(n == 0 ? net + 1 : (ci > 0 ? net - 3 : 0));
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ks = 0; ks < 16; ++ks) {
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
                    a3[mi] = mfma(w0f[ks], *reinterpret_cast<const uint4 *>(actb + mt3 * 2048 + fa_off[ks & 1] + (8192 * (ks >> 1) + 2048 * mi)), a3[mi]);
            }
        }
        fetch_oe(ci > 0 ? ci - 1 : 0);                    // (under the barrier and the g_in pass below)
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
            *reinterpret_cast<v4f *>(&gx[((mt3 + mi) * 16 + l15) * XG + nt3 * 16 + 4 * q]) = a3[mi];
            a3[mi] = v4f{0.f, 0.f, 0.f, 0.f};
        }
        __syncthreads();                                  // (B4) GXs + GXt in; every wave is through phase 3
        // ---- g_in = g_part + m (GXs + GXt)
#pragma unroll
        for (int k = 0; k < 8; ++k) gcp[k * 8 * XP] = d < dim ? gpp[k * 8 * XP] + m * gxp[k * 8 * XG] : 0.f;
        __syncthreads();
    }
    if (a.z0)
        for (int k = 0; k < 8; ++k)
            if (d < dim) a.z0[((size_t)(rg + 8 * k) * B + b) * dim + d] = xcp[k * 8 * XP];
}

// the conditioning gradient as the two bf16 operands its consumers read: [R][C] rows (the weight gradient of the conditioning layer) and the
// transpose [C][R] (split-K operand of g_feat = Gc Wc).  64 x 64 tiles through LDS.
__global__ __launch_bounds__(256) void pack_transpose_kernel(const float *__restrict__ src, long src_stride, u16 *__restrict__ dst,
                                                             u16 *__restrict__ dstT, int R, int C) {
    __shared__ float tile[64][65];
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64, tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int i = ty; i < 64; i += 4) {
        const int r = r0 + i, c = c0 + tx;
        const float v = r < R && c < C ? src[(size_t)r * src_stride + c] : 0.f;
        tile[i][tx] = v;
        if (dst && r < R && c < C) dst[(size_t)r * C + c] = f32_to_bf16(v);
    }
    __syncthreads();
    for (int i = ty; i < 64; i += 4) {
        const int c = c0 + i, r = r0 + tx;
        if (r < R && c < C) dstT[(size_t)c * R + r] = f32_to_bf16(tile[tx][i]);
    }
}

}}  // namespace mhe::flowrev

using namespace mhe;

extern "C" int mhe_flow_reverse_chain_supported(int R, int B, int dim, int hidden, int ncoup) {
    return R > 0 && B > 0 && R == 64 * B && dim > 0 && dim <= 64 && hidden == 512 && ncoup > 0;
}

extern "C" int mhe_flow_reverse_chain_bf16(const float *x_out, const float *g_x, const float *g_logp, float q_weight, const float *mask,
                                           const float *o_pre, const void *sign_bits, const void *w2F, const void *w1F,
                                           const void *w0F, long w_net_stride, void *GO_bf16, void *G2_bf16, void *G1_bf16, void *XP_bf16,
                                           float *Gc, int cond_stride, float *db2_rows, float *z0, int R, int B, int dim,
                                           int hidden, int ncoup, void *stream) {
    MHE_REQUIRE(x_out && g_x && mask && o_pre && sign_bits && w2F && w1F && w0F && GO_bf16 && G2_bf16 && G1_bf16 && XP_bf16 && Gc && db2_rows,
                "mhe_flow_reverse_chain_bf16: null pointer");
    MHE_REQUIRE(mhe_flow_reverse_chain_supported(R, B, dim, hidden, ncoup), "mhe_flow_reverse_chain_bf16: needs hidden 512 and 64 hypotheses per image (R=%d B=%d)", R, B);
    MHE_REQUIRE((long)R * hidden < (1L << 31), "mhe_flow_reverse_chain_bf16: R x hidden beyond the 32-bit row offsets");
    MHE_REQUIRE(cond_stride % 4 == 0 && cond_stride >= 4 * ncoup * hidden && w_net_stride > 0, "mhe_flow_reverse_chain_bf16: bad strides");
    flowrev::Args a;
    a.x_out = x_out; a.g_x = g_x; a.g_logp = g_logp; a.mask = mask; a.oe = o_pre;
    a.hbits = (const uint2 *)sign_bits; a.w2F = (const u16 *)w2F; a.w1F = (const u16 *)w1F; a.w0F = (const u16 *)w0F;
    a.w_stride = w_net_stride;
    a.GOb = (u16 *)GO_bf16; a.G2b = (u16 *)G2_bf16; a.G1b = (u16 *)G1_bf16; a.XPb = (u16 *)XP_bf16;
    a.Gc = Gc; a.db2 = db2_rows; a.z0 = z0;
    a.R = R; a.B = B; a.dim = dim; a.ncoup = ncoup; a.cstride = cond_stride; a.q_weight = q_weight;
    hipLaunchKernelGGL(flowrev::chain_kernel, dim3(B), dim3(512), 0, (hipStream_t)stream, a);
    return check_launch("flowrev::chain_kernel");
}

extern "C" int mhe_pack_transpose_bf16(const float *src, long src_stride, void *dst_bf16, void *dstT_bf16, int R, int C, void *stream) {
    MHE_REQUIRE(src && dstT_bf16 && R > 0 && C > 0 && src_stride >= C, "mhe_pack_transpose_bf16: bad arguments");
    hipLaunchKernelGGL(flowrev::pack_transpose_kernel, dim3((C + 63) / 64, (R + 63) / 64), dim3(256), 0, (hipStream_t)stream, src, src_stride,
                       (u16 *)dst_bf16, (u16 *)dstT_bf16, R, C);
    return check_launch("flowrev::pack_transpose_kernel");
}

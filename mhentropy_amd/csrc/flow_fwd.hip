// bf16 RealNVP coupling stack at hidden = 512, third generation: the skeleton of the one-launch reverse chain (flow_rev.hip) run forwards.
// Reference hand/flows.py:105-122 (coupling nets), :210-226 (forward_p / inverse), :195-208 (log_prob); rounding points those of
// flow_bf16.hip / flow_ns.hip (bf16 x bf16 products with f32 accumulation; the flow variable, s, t, exp and the log-determinant in f32):
// oracle/flows_ref.py:forward_p_logdet_bf16.
//
// A workgroup owns 64 hypothesis rows of ONE image (rows (64 c + i) B + b) for the whole stack; eight waves, a wave = 64 rows x 64 hidden
// units of the 512-wide layers (accumulators [unit tile 4][row tile 4] of v_mfma_f32_16x16x32_bf16):
//   layer 0  H1 = lrelu(Xm W0^T + cond0)   K = 64    operand Xm (masked flow variable, bf16) from LDS, W0 fragments fetched a net ahead
//   layer 1  H2 = lrelu(H1 W1^T + cond1)   K = 512   operand H1 from LDS (eight k-tiles), W1 fragments from global (L2) one k-tile ahead
//   layer 2  O  = H2 W2^T + b2             K = 512   a wave = one 16-dim tile x two row tiles; the SAME wave gets the same elements of the
//            s net and of the t net, so the flow variable lives in those lanes' registers (8 floats per lane) and the affine update, the
//            log-determinant and the next coupling's masked operand are computed where the products land: no owner waves, no partial sums.
// Second generation (flow_ns.hip): weights by LDS-DMA into private rings, 32x32x16 MFMA, layer 2 split over K with partial sums through LDS
// and six owner waves: 326 us at config C2 (14 us per net); this kernel's weight stream is the reverse chain's (fragment-major copies, a
// fragment = one 1 KiB run, register ping-pong pinned by sched_barrier).
// EMIT (train step): H1, H2 (bf16 [net][R][512], after the leaky-ReLU) and the s / t pre-activations (f32 [net][R][64], bias included)
// are also written out, H1 / H2 as whole 128-byte rows from the wave's k-tile.
#include "flow_frag.h"
#include "../../include/mhe.h"

namespace mhe { namespace flowfwd {

using namespace flowfrag;

constexpr int H = 512, ROWS = 64, KT = H / 64, MAXC = 32;

struct Args {
    const float *in, *cond, *bias2, *mask;               // [R][dim], [B][cstride], [nets][64], [ncoup][dim]
    float *out, *sum_s, *logp;                            // [R][dim], [R] | NULL, [R] | NULL
    const u16 *w0F, *w1F, *w2F;                           // net 0's fragment-major operands W0 [512][64], W1 [512][512], W2 [64][512]
    long w_stride;                                        // net k lies k * w_stride elements further
    u16 *h1e, *h2e;                                       // EMIT: [nets][R][512]
    float *oe;                                            // EMIT: [nets][R][64]
    uint2 *hbits;                                         // EMIT (optional): signs of H1 / H2 in the accumulator layout, [nets][R / 64][layer 2][wave 8][lane 64]
                                                          //   bit (unit tile nt * 4 + row tile mt) * 4 + e - what the reverse chain masks with (flow_rev.hip)
    int R, B, dim, ncoup, cstride, inverse;
};

__device__ __forceinline__ float leaky(float v) { return fmaxf(v, 0.01f * v); }

template <bool EMIT, int SETS>
__global__ __launch_bounds__(512) void couplings_frag_kernel(const Args a) {
    __shared__ uint4 act[KT * ROWS * 8];                  // 64 KiB: H1, then H2, as eight [64 rows][64 units] k-tiles (16-byte chunks swizzled)
    __shared__ uint4 xb[ROWS * 8];                        // the masked flow variable as a [64 rows][64 dims] bf16 operand tile
    __shared__ float mk[MAXC * 64];                       // masks, padded to 64 dims with 1 (= pass through)
    __shared__ float red[2][4][ROWS];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, q = lane >> 4, l15 = lane & 15;
    const int B = a.B, R = a.R, dim = a.dim;
    const int b = blockIdx.x % B, chunk = blockIdx.x / B;
    // LDS byte offsets (flow_rev.hip): one lane-dependent base per access shape, the rest immediates
    const int xr = (l15 >> 1) & 7, r8 = lane >> 3, c8 = lane & 7;
    unsigned fa_off[2], ac_off[4], pc_off[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) fa_off[kk] = (unsigned)((l15 * 8 + ((kk * 4 + q) ^ xr)) * 16);
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) ac_off[nt] = (unsigned)((l15 * 8 + ((nt * 2 + (q >> 1)) ^ xr)) * 16 + (q & 1) * 8);
#pragma unroll
    for (int ip = 0; ip < 2; ++ip) pc_off[ip] = (unsigned)((r8 * 8 + (c8 ^ ((ip * 4 + (r8 >> 1)) & 7))) * 16);
    unsigned char *const actb = reinterpret_cast<unsigned char *>(act), *const xbb = reinterpret_cast<unsigned char *>(xb);
    unsigned char *const tile = actb + wave * 8192;       // this wave's k-tile
    // layer 2 / flow variable role: dims nt3 * 16 + 4 q + e, rows (mt3 + mi) * 16 + l15
    const int nt3 = wave >> 1, mt3 = 2 * (wave & 1), d0 = nt3 * 16 + 4 * q;
    const unsigned xw_off = (unsigned)((l15 * 8 + ((nt3 * 2 + (q >> 1)) ^ xr)) * 16 + (q & 1) * 8 + mt3 * 2048);
    const unsigned lane16 = (unsigned)lane * 16u;
    const unsigned co = (unsigned)((((64 * chunk + r8) * B + b) * H + 64 * wave + c8 * 8) * 2);       // EMIT: 16-byte row pieces (bytes)
    const unsigned cstep = (unsigned)(8 * B * H) * 2u;
    const size_t hbytes = (size_t)R * H * 2;
    // hypotheses per image need not be a multiple of 64 in the forward-only form (round 5: the metrics pass draws N = 200): the last chunk's
    // surplus rows are computed on zeros and never stored (csrc/glow_fwd.hip does the same)
    const int Nh = R / B;
    int grow[2];                                          // global rows of this lane's two flow-variable rows
    bool ok[2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
        const int n = 64 * chunk + (mt3 + mi) * 16 + l15;
        ok[mi] = n < Nh;
        grow[mi] = ok[mi] ? n * B + b : b;
    }

    for (int i = tid; i < a.ncoup * 64; i += 512) mk[i] = (i & 63) < dim ? a.mask[(i >> 6) * dim + (i & 63)] : 1.f;
    v4f x[2];
    float sum_s[2] = {0.f, 0.f}, sq_in[2] = {0.f, 0.f};
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            x[mi][e] = (ok[mi] && d0 + e < dim) ? a.in[(size_t)grow[mi] * dim + d0 + e] : 0.f;
            sq_in[mi] = fmaf(x[mi][e], x[mi][e], sq_in[mi]);
        }
    __syncthreads();                                      // masks staged
    auto coupling_at = [&](int step) { return a.inverse ? a.ncoup - 1 - step : step; };
    // the masked flow variable of coupling ci as the bf16 operand tile of layer 0
    auto publish_x = [&](int ci) __attribute__((always_inline)) {
        const float4 m = *reinterpret_cast<const float4 *>(mk + ci * 64 + d0);
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
            uint2 o;
            o.x = (unsigned)f32_to_bf16(x[mi][0] * m.x) | ((unsigned)f32_to_bf16(x[mi][1] * m.y) << 16);
            o.y = (unsigned)f32_to_bf16(x[mi][2] * m.z) | ((unsigned)f32_to_bf16(x[mi][3] * m.w) << 16);
            *reinterpret_cast<uint2 *>(xbb + xw_off + 2048 * mi) = o;
        }
    };
    publish_x(coupling_at(0));
    // fetched a net ahead, under layer 2: layer 0's fragments, its conditioning row and the l2 bias
    uint4 w0f[2][4];
    float4 c0[4], bz;
    const unsigned cq = (unsigned)(64 * wave + 4 * q) * 4u;
    const rsrc_t condr = rsrc_of(a.cond + (size_t)b * a.cstride, (size_t)a.cstride * 4);
    auto fetch_net = [&](int net) __attribute__((always_inline)) {
        const rsrc_t w0 = rsrc_of(a.w0F + (size_t)net * a.w_stride, (size_t)H * 64 * 2);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) w0f[kk][nt] = frag(w0, lane16, 4 * wave + nt, 2, kk);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) c0[nt] = __builtin_bit_cast(float4, bld(condr, cq, (unsigned)((2 * net) * H + nt * 16) * 4u));
        bz = *reinterpret_cast<const float4 *>(a.bias2 + (size_t)net * 64 + d0);
    };
    fetch_net(2 * coupling_at(0));
    __syncthreads();                                      // operand tile of the first coupling written

    v4f sv[2];                                            // s of the coupling in flight
    for (int step = 0; step < a.ncoup; ++step) {
        const int ci = coupling_at(step);
#pragma unroll 1
        for (int n = 0; n < 2; ++n) {
            const int net = 2 * ci + n;
            const rsrc_t w1 = rsrc_of(a.w1F + (size_t)net * a.w_stride, (size_t)H * H * 2), w2 = rsrc_of(a.w2F + (size_t)net * a.w_stride, (size_t)64 * H * 2);
            v4f acc[4][4];                                // [unit tile of the wave's 64][row tile]
            // + conditioning row, leaky-ReLU, bf16 -> the wave's k-tile (accumulator layout: 8 bytes per (unit tile, row tile)); EMIT: the
            // finished tile out as whole 128-byte rows
            auto finish = [&](const float4 (&c)[4], u16 *he, int layer) __attribute__((always_inline)) {
                uint2 sg = make_uint2(0u, 0u);
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt) {
                        const v4f g = acc[nt][mt];
                        uint2 o;
                        o.x = (unsigned)f32_to_bf16(leaky(g[0] + c[nt].x)) | ((unsigned)f32_to_bf16(leaky(g[1] + c[nt].y)) << 16);
                        o.y = (unsigned)f32_to_bf16(leaky(g[2] + c[nt].z)) | ((unsigned)f32_to_bf16(leaky(g[3] + c[nt].w)) << 16);
                        *reinterpret_cast<uint2 *>(tile + ac_off[nt] + 2048 * mt) = o;
                        if constexpr (EMIT) {             // sign of the STORED value (bf16: sign bit clear and not zero)
                            const unsigned b4 = (unsigned)((o.x & 0x8000u) == 0 && (o.x & 0x7fffu) != 0) | ((unsigned)((o.x & 0x80000000u) == 0 && (o.x & 0x7fff0000u) != 0) << 1) |
                                                ((unsigned)((o.y & 0x8000u) == 0 && (o.y & 0x7fffu) != 0) << 2) | ((unsigned)((o.y & 0x80000000u) == 0 && (o.y & 0x7fff0000u) != 0) << 3);
                            if (nt * 4 + mt < 8) sg.x |= b4 << ((nt * 4 + mt) * 4);
                            else sg.y |= b4 << (((nt * 4 + mt) & 7) * 4);
                        }
                    }
                if constexpr (EMIT) {
                    if (a.hbits) a.hbits[(((size_t)net * gridDim.x + blockIdx.x) * 2 + layer) * 512 + wave * 64 + lane] = sg;
                    const rsrc_t hr = rsrc_of(he + (size_t)net * R * H, hbytes);
                    wave_sync();                          // the tile is this wave's own: wave-local ordering suffices
#pragma unroll
                    for (int i = 0; i < 8; ++i) bst(hr, co, i * cstep, *reinterpret_cast<const uint4 *>(tile + pc_off[i & 1] + 1024 * i));
                }
            };
            // ================= layer 0 (K = 64)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) acc[nt][mt] = v4f{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                uint4 fa[4];
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) fa[mt] = *reinterpret_cast<const uint4 *>(xbb + fa_off[kk] + 2048 * mt);
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt) acc[nt][mt] = mfma(w0f[kk][nt], fa[mt], acc[nt][mt]);
            }
            // layer 1's conditioning row and its first W1 k-tile: requested here, under layer 0's epilogue
            float4 c1[4];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) c1[nt] = __builtin_bit_cast(float4, bld(condr, cq, (unsigned)((2 * net + 1) * H + nt * 16) * 4u));
            uint4 fS[SETS][2][4];                         // SETS - 1 k-tiles of W1 fragments in flight beside the one being multiplied
            auto fetch_w = [&](uint4 (&f)[2][4], int kt) __attribute__((always_inline)) {
#pragma unroll
                for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) f[kk][nt] = frag(w1, lane16, 4 * wave + nt, 16, 2 * kt + kk);
            };
#pragma unroll
            for (int k0 = 0; k0 < SETS - 1; ++k0) fetch_w(fS[k0], k0);
            __builtin_amdgcn_sched_barrier(0);
            __syncthreads();                              // (B0) every wave is through the layer 2 before: act is free
            finish(c0, a.h1e, 0);
            __syncthreads();                              // (B1) H1 complete in act
            // ================= layer 1 (K = 512): W1 fragments into two register sets, k-tile kt + 1's in flight under k-tile kt's 32 MFMAs
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) acc[nt][mt] = v4f{0.f, 0.f, 0.f, 0.f};
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
#pragma unroll
            for (int kt = 0; kt < KT; ++kt) {
                if (kt + SETS - 1 < KT) fetch_w(fS[(kt + SETS - 1) % SETS], kt + SETS - 1);
                __builtin_amdgcn_sched_barrier(0);
                ktile(fS[kt % SETS], kt);
                __builtin_amdgcn_sched_barrier(0);
            }
            __syncthreads();                              // (B2) every wave has read H1: act is free for H2
            // layer 2's fragments (this wave's dim tile, all 16 k steps), in flight under the epilogue
            uint4 w2f[16];
#pragma unroll
            for (int ks = 0; ks < 16; ++ks) w2f[ks] = frag(w2, lane16, nt3, 16, ks);
            const float4 bias = bz;
            {   // ... and the next net's layer-0 operands (the next net in the stack's order; always fetched: see flow_rev.hip)
                const int nstep = step + 1 < a.ncoup ? step + 1 : step;
                fetch_net(n == 0 ? net + 1 : 2 * coupling_at(nstep));
            }
            __builtin_amdgcn_sched_barrier(0);
            finish(c1, a.h2e, 1);
            __syncthreads();                              // (B3) H2 complete in act
            // ================= layer 2 (K = 512) + the coupling
            v4f o[2] = {v4f{0.f, 0.f, 0.f, 0.f}, v4f{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
            for (int ks = 0; ks < 16; ++ks)
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
                    o[mi] = mfma(w2f[ks], *reinterpret_cast<const uint4 *>(actb + mt3 * 2048 + fa_off[ks & 1] + (8192 * (ks >> 1) + 2048 * mi)), o[mi]);
            const float4 m = *reinterpret_cast<const float4 *>(mk + ci * 64 + d0);
            const float mv[4] = {m.x, m.y, m.z, m.w}, bv[4] = {bias.x, bias.y, bias.z, bias.w};
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
                for (int e = 0; e < 4; ++e) o[mi][e] += bv[e];
                if constexpr (EMIT) *reinterpret_cast<v4f *>(a.oe + ((size_t)net * R + grow[mi]) * 64 + d0) = o[mi];
                if (n == 0) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) sv[mi][e] = mv[e] != 0.f ? 0.f : tanhf(o[mi][e]);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (mv[e] == 0.f) {
                            if (!a.inverse) x[mi][e] = x[mi][e] * expf(sv[mi][e]) + o[mi][e];       // flows.py:216
                            else            x[mi][e] = (x[mi][e] - o[mi][e]) * expf(-sv[mi][e]);    // flows.py:225
                            sum_s[mi] += sv[mi][e];
                        }
                }
            }
            if (n == 1 && step + 1 < a.ncoup) publish_x(coupling_at(step + 1));      // (xb: every wave read it before (B1) of this net)
        }
        // (the next coupling's layer 0 reads xb after ... the barrier below; act is rewritten only after (B0))
        __syncthreads();                                  // (B4) next coupling's operand tile published
    }

    // ---- outputs: the flow variable by the lanes that hold it, the per-row scalars through a small LDS table
    float sq[2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
        float so = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (ok[mi] && d0 + e < dim) a.out[(size_t)grow[mi] * dim + d0 + e] = x[mi][e];
            so = fmaf(x[mi][e], x[mi][e], so);
        }
        sq[mi] = a.inverse ? so : sq_in[mi];
        // over the four dim groups q of the wave's 16-dim tile
        sq[mi] += __shfl_xor(sq[mi], 16, 64); sq[mi] += __shfl_xor(sq[mi], 32, 64);
        sum_s[mi] += __shfl_xor(sum_s[mi], 16, 64); sum_s[mi] += __shfl_xor(sum_s[mi], 32, 64);
        if (q == 0) { red[0][nt3][(mt3 + mi) * 16 + l15] = sq[mi]; red[1][nt3][(mt3 + mi) * 16 + l15] = sum_s[mi]; }
    }
    __syncthreads();
    if (tid < ROWS && 64 * chunk + tid < Nh) {
        const int r = (64 * chunk + tid) * B + b;
        const float s2 = red[0][0][tid] + red[0][1][tid] + red[0][2][tid] + red[0][3][tid];
        const float ss = red[1][0][tid] + red[1][1][tid] + red[1][2][tid] + red[1][3][tid];
        if (a.sum_s) a.sum_s[r] = ss;
        if (a.logp) a.logp[r] = (-0.5f * s2 - 0.5f * (float)dim * 1.8378770664093453f) - ss;
    }
}

}}  // namespace mhe::flowfwd

using namespace mhe;

extern "C" int mhe_flow_couplings_frag_supported(int R, int B, int dim, int hidden, int ncoup) {
    return R > 0 && B > 0 && R % B == 0 && dim > 0 && dim <= 64 && hidden == 512 && ncoup > 0 && ncoup <= flowfwd::MAXC;
}

extern "C" int mhe_flow_couplings_frag_bf16(const float *in, float *out, const float *cond, int cond_stride, const void *w0F, const void *w1F,
                                            const void *w2F, long w_net_stride, const float *bias2, const float *mask, float *sum_s,
                                            float *log_prob, void *h1, void *h2, float *o, void *sign_bits, int R, int B, int dim, int hidden,
                                            int ncoup, int direction, void *stream) {
    MHE_REQUIRE(in && out && cond && w0F && w1F && w2F && bias2 && mask, "mhe_flow_couplings_frag_bf16: null pointer");
    MHE_REQUIRE(mhe_flow_couplings_frag_supported(R, B, dim, hidden, ncoup),
                "mhe_flow_couplings_frag_bf16: needs hidden 512, R a multiple of B, at most %d couplings (R=%d B=%d)", flowfwd::MAXC, R, B);
    MHE_REQUIRE(direction == MHE_FLOW_FORWARD || direction == MHE_FLOW_INVERSE, "mhe_flow_couplings_frag_bf16: direction=%d", direction);
    MHE_REQUIRE(cond_stride % 4 == 0 && cond_stride >= 4 * ncoup * hidden && w_net_stride > 0, "mhe_flow_couplings_frag_bf16: bad strides");
    MHE_REQUIRE((long)R * hidden < (1L << 31), "mhe_flow_couplings_frag_bf16: R x hidden beyond the 32-bit row offsets");
    const bool emit = h1 || h2 || o;
    MHE_REQUIRE(!emit || (h1 && h2 && o), "mhe_flow_couplings_frag_bf16: h1, h2 and o come together");
    MHE_REQUIRE(!sign_bits || emit, "mhe_flow_couplings_frag_bf16: sign_bits come with h1, h2 and o");
    MHE_REQUIRE(!emit || R % (64 * B) == 0, "mhe_flow_couplings_frag_bf16: the train step's form (h1, h2, o) needs a multiple of 64 hypotheses per image");
    flowfwd::Args a;
    a.in = in; a.cond = cond; a.bias2 = bias2; a.mask = mask; a.out = out; a.sum_s = sum_s; a.logp = log_prob;
    a.w0F = (const u16 *)w0F; a.w1F = (const u16 *)w1F; a.w2F = (const u16 *)w2F; a.w_stride = w_net_stride;
    a.h1e = (u16 *)h1; a.h2e = (u16 *)h2; a.oe = o; a.hbits = (uint2 *)sign_bits;
    a.R = R; a.B = B; a.dim = dim; a.ncoup = ncoup; a.cstride = cond_stride; a.inverse = direction == MHE_FLOW_INVERSE;
    // W1 register sets of the forward-only form: 3 = two k-tiles in flight (4 VGPRs spilled).  Measured at C2: 313.7 us (2) against 312.1 us (3) -
    // the layer-1 stream is not latency-bound: 64 KiB per CU and k-tile at 75-90 GB/s per CU IS the L2 -> CU rate of this chip
    // (MI355X_MICROARCH.md "Indexed rows": 66-73 GB/s per CU for rows served by the XCD's L2); the default stays 2
    static const int sets = getenv("MHE_FLOW_W1_SETS") ? atoi(getenv("MHE_FLOW_W1_SETS")) : 2;
    const dim3 grid((unsigned)(((R / B + 63) / 64) * B));
    if (emit) hipLaunchKernelGGL((flowfwd::couplings_frag_kernel<true, 2>), grid, dim3(512), 0, (hipStream_t)stream, a);
    else if (sets == 3) hipLaunchKernelGGL((flowfwd::couplings_frag_kernel<false, 3>), grid, dim3(512), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL((flowfwd::couplings_frag_kernel<false, 2>), grid, dim3(512), 0, (hipStream_t)stream, a);
    return check_launch("flowfwd::couplings_frag_kernel");
}

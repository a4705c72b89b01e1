// Conditional Glow, sampling direction (noise -> sample, with log q), ALL layers in ONE launch at hidden = 512 (round 5).
//
// What it replaces: glow.py's layer-by-layer chain - per layer `linear` -> add_image_rows -> 2 x (relu_copy, product, dropout, product,
// glu_residual) -> `linear` -> coupling -> `linear`: ~60 launches, every [R, 512] activation through HBM (1.4 ms at config C2 where the
// RealNVP stack takes 0.31).  The reference builds the flow with `ConditionalGlow(45, 512, 4, 2, context_features=512,
// dropout_probability=0.2)` (hand/network.py:342-344) and samples it at :736-742; the class is the unpinned third-party nkolot/nflows
// (hand/environment.yml:284), absent from the reference tree: the algorithm is the published nflows one restated in oracle/glow_ref.py
// (PARITY UNPINNED).  Per layer l = L-1 .. 0, on the flow variable v (45 dims carried as 64):
//     h   = Wx v + ctab[image][slot]                                   ResidualNet.initial_layer (context columns + bias hoisted per image)
//     2 x { t = relu(h); t2 = dropout(relu(W0 t + b0)); h += (W1 t2 + b1) sigmoid(gate[image]) }     ResidualBlock with GLU context gate
//     prm = Wf h + bf;  scale = sigmoid(us + 2) + 1e-3;  y_t = (v_t - shift) / scale on the transform columns;  log q += sum log scale
//     v   = A^-1 y + c^-1                                               inverse of ActNorm + LU (csrc/glow_affine.hip)
//
// Skeleton: flow_fwd.hip's (the RealNVP stack).  A workgroup owns 64 hypothesis rows of ONE image for the whole flow; eight waves, a wave =
// 64 rows x 64 hidden units of every 512-wide layer (accumulators [unit tile 4][row tile 4] of v_mfma_f32_16x16x32_bf16):
//   * the RESIDUAL STREAM h stays in the wave's registers in f32 (64 VGPRs) for the whole layer - only its bf16 operand copies go through LDS;
//   * activations travel as ONE 64 KiB LDS image of eight [64 rows][64 units] k-tiles, each written by the wave that produced it;
//   * weights are bf16 FRAGMENT-MAJOR copies (a fragment = one 1 KiB run) streamed L2 -> registers one k-tile ahead, pinned by sched_barrier;
//   * the final layer is evaluated as TWO 64-row products whose output rows are laid out in the flow variable's own column order (shift
//     and unconstrained scale of column c land in the lane that holds v[c]), so the coupling needs no shuffles;
//   * the 45 x 45 inverse affine map runs on the VALU from an f32 LDS copy of y and (A^-1)^T: 384 FMAs per lane and layer;
//   * dropout masks are the bits ops.dropout_ draws (one bit per element of a [R, 512] tensor, bit k of byte i = element 8 i + k), drawn for
//     all blocks by one launch (mhe_dropout_bits) and read here - the same bits the reverse pass applies to the gradient.
// EMIT (train step): the tape the reverse pass reads - per layer v (f32 [R][64]), prm ([shift | us], f32 [R][64]), y (f32 [R][64]), the final
// layer's operand bf16(h) and per block relu(h), t2 (after dropout), t3 = W1 t2 + b1 as bf16 [R][512] - written as whole 128-byte rows from
// the wave's k-tile.  With EMIT the gate multiplies the bf16-ROUNDED t3, i.e. the forward value is the one the tape describes.
// Rows: hypothesis n of image b is row n * row_n + b * row_b (sample-major: row_n = B, row_b = 1; nflows' batch-major: row_n = 1, row_b = N);
// N need not be a multiple of 64: the last chunk's surplus rows are computed on zeros and never stored.
#include "flow_frag.h"
#include "../../include/mhe.h"

namespace mhe { namespace glowf {

using namespace flowfrag;

constexpr int H = 512, ROWS = 64, KT = H / 64, MAXL = 16, NBLK = 2, YP = 68;

struct Args {
    const float *in, *ctab;                               // [R][dim] base noise; [B][cstride] context-only terms, slot s at + s * 512
    const u16 *wxF, *w0F, *w1F, *wsF, *wuF;               // fragment-major bf16: [L][512][64], [L][NBLK][512][512] x 2, [L][64][512] x 2
    const float *b0, *b1, *bs, *bu;                       // [L][NBLK][512] x 2, [L][64] x 2 (final-layer biases in flow-variable column order)
    const float *ainvT, *cinv, *const_parts;              // [L][64][64] ((A^-1)^T: [k][d]), [L][64], [L]
    const unsigned char *drop;                            // [L][NBLK][R * 64] mask bytes (row r: 64 bytes) or NULL (eval mode)
    float drop_scale;
    float *out, *logp;                                    // [R][dim], [R]
    float *v_e, *y_e, *prm_e;                             // EMIT: [L][R][64]
    u16 *tb_e, *t2_e, *t3_e, *hf_e;                       // EMIT: [L][NBLK][R][512] x 3, [L][R][512]
    // EMIT, for the one-launch reverse chain (csrc/glow_rev.hip; all three or none): the coupling parameters in the flow variable's column order
    // ([shift' (64) | us' (64)] f32 [L][R][128]), the layer input as the bf16 operand it was multiplied as ([L][R][64]), and the gates of the two
    // ReLUs of every block as bits in the accumulator layout ([L][NBLK][2: relu(h) > 0 | t2 != 0][R / 64][8 waves][64 lanes] x 64 bits, bit
    // (unit tile nt * 4 + row tile mt) * 4 + e - the layout flow_fwd.hip / flow_rev.hip use)
    float *prmc_e;
    u16 *vb_e;
    uint2 *bits_e;
    int R, B, N, dim, L, cstride, row_n, row_b;
};

template <bool EMIT>
__global__ __launch_bounds__(512) void layers_kernel(const Args a) {
    __shared__ uint4 act[KT * ROWS * 8];                  // 64 KiB: the operand image of the 512-wide products
    __shared__ uint4 xb[ROWS * 8];                        // the flow variable as a [64 rows][64 dims] bf16 operand tile
    __shared__ float yb[ROWS * YP];                       // y (f32) for the inverse affine map, rows padded to 68 floats
    __shared__ float aT[64 * 64];                         // (A^-1)^T of the layer in flight
    __shared__ unsigned long long mk[ROWS * 8];           // one block's dropout mask: [64 rows][64 bytes]
    __shared__ float red[2][4][ROWS];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, q = lane >> 4, l15 = lane & 15;
    const int B = a.B, R = a.R, dim = a.dim;
    const int b = blockIdx.x % B, chunk = blockIdx.x / B;
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
    // flow-variable role: dims nt3 * 16 + 4 q + e, rows (mt3 + mi) * 16 + l15
    const int nt3 = wave >> 1, mt3 = 2 * (wave & 1), d0 = nt3 * 16 + 4 * q;
    const unsigned xw_off = (unsigned)((l15 * 8 + ((nt3 * 2 + (q >> 1)) ^ xr)) * 16 + (q & 1) * 8 + mt3 * 2048);
    const unsigned lane16 = (unsigned)lane * 16u;
    // EMIT row pieces: lane (r8, c8) moves the 16-byte piece c8 of rows r8 + 8 i of the wave's k-tile
    const long rn = a.row_n, rb = a.row_b;
    const unsigned co = (unsigned)((((long)(64 * chunk + r8) * rn + (long)b * rb) * H + 64 * wave + c8 * 8) * 2);
    const unsigned cstep = (unsigned)(8 * rn * H) * 2u;
    const size_t hbytes = (size_t)R * H * 2;
    int grow[2];
    bool ok[2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
        const int n = 64 * chunk + (mt3 + mi) * 16 + l15;
        ok[mi] = n < a.N;
        grow[mi] = ok[mi] ? (int)((long)n * rn + (long)b * rb) : 0;
    }
    v4f x[2];
    float ld[2] = {0.f, 0.f}, sq_in[2] = {0.f, 0.f};
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            x[mi][e] = (ok[mi] && d0 + e < dim) ? a.in[(size_t)grow[mi] * dim + d0 + e] : 0.f;
            sq_in[mi] = fmaf(x[mi][e], x[mi][e], sq_in[mi]);
        }
    auto publish_x = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
            uint2 o;
            o.x = (unsigned)f32_to_bf16(x[mi][0]) | ((unsigned)f32_to_bf16(x[mi][1]) << 16);
            o.y = (unsigned)f32_to_bf16(x[mi][2]) | ((unsigned)f32_to_bf16(x[mi][3]) << 16);
            *reinterpret_cast<uint2 *>(xbb + xw_off + 2048 * mi) = o;
        }
    };
    publish_x();
    const unsigned cq = (unsigned)(64 * wave + 4 * q) * 4u;
    const rsrc_t ctr = rsrc_of(a.ctab + (size_t)b * a.cstride, (size_t)a.cstride * 4);
    // the wave's k-tile out as whole 128-byte rows of a [R][512] bf16 tensor (rows past R - surplus hypotheses - fall outside the resource)
    auto emit_tile = [&](u16 *base, size_t index) __attribute__((always_inline)) {
        const rsrc_t hr = rsrc_of(base + index * (size_t)R * H, hbytes);
        wave_sync();                                      // the tile is this wave's own: wave-local ordering suffices
#pragma unroll
        for (int i = 0; i < 8; ++i)
            if (64 * chunk + r8 + 8 * i < a.N) bst(hr, co, i * cstep, *reinterpret_cast<const uint4 *>(tile + pc_off[i & 1] + 1024 * i));
    };
    __syncthreads();                                      // operand tile of the first layer written

    for (int step = 0; step < a.L; ++step) {
        const int l = a.L - 1 - step;
        const int first = 1 - (l & 1);                    // first transform column: the alternating mask starts with the odd columns
        const int T = first ? dim / 2 : (dim + 1) / 2;
        const int slot0 = l * (1 + NBLK);
        // ---- stage (A^-1)^T of this layer (read after the coupling, many barriers from here)
        {
            const float4 *src = reinterpret_cast<const float4 *>(a.ainvT + (size_t)l * 4096);
            reinterpret_cast<float4 *>(aT)[tid] = src[tid];
            reinterpret_cast<float4 *>(aT)[tid + 512] = src[tid + 512];
        }
        if constexpr (EMIT) {
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
                if (ok[mi]) {
                    *reinterpret_cast<v4f *>(a.v_e + ((size_t)l * R + grow[mi]) * 64 + d0) = x[mi];
                    if (a.vb_e) {
                        uint2 o;
                        o.x = (unsigned)f32_to_bf16(x[mi][0]) | ((unsigned)f32_to_bf16(x[mi][1]) << 16);
                        o.y = (unsigned)f32_to_bf16(x[mi][2]) | ((unsigned)f32_to_bf16(x[mi][3]) << 16);
                        *reinterpret_cast<uint2 *>(a.vb_e + ((size_t)l * R + grow[mi]) * 64 + d0) = o;
                    }
                }
        }
        v4f acc[4][4], h[4][4];
        // ================= initial layer (K = 64): h = Wx v + ctab[image][slot0]
        {
            const rsrc_t wx = rsrc_of(a.wxF + (size_t)l * H * 64, (size_t)H * 64 * 2);
            uint4 w0f[2][4];
            float4 c0[4];
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) w0f[kk][nt] = frag(wx, lane16, 4 * wave + nt, 2, kk);
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) c0[nt] = __builtin_bit_cast(float4, bld(ctr, cq, (unsigned)(slot0 * H + nt * 16) * 4u));
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
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    h[nt][mt][0] = acc[nt][mt][0] + c0[nt].x; h[nt][mt][1] = acc[nt][mt][1] + c0[nt].y;
                    h[nt][mt][2] = acc[nt][mt][2] + c0[nt].z; h[nt][mt][3] = acc[nt][mt][3] + c0[nt].w;
                }
        }
        // the K = 512 product of the wave's 64 units over the LDS image; fragments of k-tile kt + 1 in flight under k-tile kt's 32 MFMAs
        auto product = [&](const u16 *wbase) __attribute__((always_inline)) {
            const rsrc_t w1 = rsrc_of(wbase, (size_t)H * H * 2);
            uint4 fS[2][2][4];
            auto fetch_w = [&](uint4 (&f)[2][4], int kt) __attribute__((always_inline)) {
#pragma unroll
                for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) f[kk][nt] = frag(w1, lane16, 4 * wave + nt, 16, 2 * kt + kk);
            };
            fetch_w(fS[0], 0);
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) acc[nt][mt] = v4f{0.f, 0.f, 0.f, 0.f};
            __builtin_amdgcn_sched_barrier(0);
            __syncthreads();                              // the operand image is complete
#pragma unroll
            for (int kt = 0; kt < KT; ++kt) {
                if (kt + 1 < KT) fetch_w(fS[(kt + 1) & 1], kt + 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    uint4 fa[4];
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt) fa[mt] = *reinterpret_cast<const uint4 *>(actb + fa_off[kk] + (8192 * kt + 2048 * mt));
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                        for (int mt = 0; mt < 4; ++mt) acc[nt][mt] = mfma(fS[kt & 1][kk][nt], fa[mt], acc[nt][mt]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            __syncthreads();                              // every wave has read the image: it may be rewritten
        };
#pragma unroll 1
        for (int blk = 0; blk < NBLK; ++blk) {
            const size_t lb = (size_t)l * NBLK + blk;
            // ---- t = relu(h) -> the wave's k-tile; this block's dropout mask -> LDS
            if (a.drop) {
                const int row = tid >> 3, n = 64 * chunk + row;
                unsigned long long m = ~0ull;
                if (n < a.N) m = *reinterpret_cast<const unsigned long long *>(a.drop + (lb * R + ((size_t)n * rn + (size_t)b * rb)) * 64 + (tid & 7) * 8);
                mk[tid] = m;
            }
            uint2 sg = make_uint2(0u, 0u);                // EMIT: [stored value != 0] of this lane's 64 elements
            auto note = [&](uint2 o, int nt, int mt) __attribute__((always_inline)) {
                const unsigned b4 = (unsigned)((o.x & 0x7fffu) != 0) | ((unsigned)((o.x & 0x7fff0000u) != 0) << 1) |
                                    ((unsigned)((o.y & 0x7fffu) != 0) << 2) | ((unsigned)((o.y & 0x7fff0000u) != 0) << 3);
                if (nt * 4 + mt < 8) sg.x |= b4 << ((nt * 4 + mt) * 4);
                else sg.y |= b4 << (((nt * 4 + mt) & 7) * 4);
            };
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    const v4f g = h[nt][mt];
                    uint2 o;
                    o.x = (unsigned)f32_to_bf16(fmaxf(g[0], 0.f)) | ((unsigned)f32_to_bf16(fmaxf(g[1], 0.f)) << 16);
                    o.y = (unsigned)f32_to_bf16(fmaxf(g[2], 0.f)) | ((unsigned)f32_to_bf16(fmaxf(g[3], 0.f)) << 16);
                    *reinterpret_cast<uint2 *>(tile + ac_off[nt] + 2048 * mt) = o;
                    if constexpr (EMIT) note(o, nt, mt);
                }
            if constexpr (EMIT) {
                emit_tile(a.tb_e, lb);
                if (a.bits_e) a.bits_e[((lb * 2 + 0) * gridDim.x + blockIdx.x) * 512 + wave * 64 + lane] = sg;
                sg = make_uint2(0u, 0u);
            }
            product(a.w0F + lb * H * H);
            // ---- t2 = dropout(relu(acc + b0)) -> the wave's k-tile
            {
                const rsrc_t br = rsrc_of(a.b0 + lb * H, (size_t)H * 4);
                float4 bb[4];
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) bb[nt] = __builtin_bit_cast(float4, bld(br, cq, (unsigned)(nt * 16) * 4u));
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    // the mask bits of row 16 mt + l15, units 64 wave .. + 63: one 64-bit word; unit 16 nt + 4 q + e is bit 16 nt + 4 q + e
                    const unsigned long long mw = a.drop ? mk[(16 * mt + l15) * 8 + wave] : ~0ull;
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) {
                        const v4f g = acc[nt][mt];
                        const unsigned nib = (unsigned)(mw >> (16 * nt + 4 * q)) & 15u;
                        const float sc = a.drop ? a.drop_scale : 1.f;
                        const float v0 = (nib & 1u) ? fmaxf(g[0] + bb[nt].x, 0.f) * sc : 0.f, v1 = (nib & 2u) ? fmaxf(g[1] + bb[nt].y, 0.f) * sc : 0.f;
                        const float v2 = (nib & 4u) ? fmaxf(g[2] + bb[nt].z, 0.f) * sc : 0.f, v3 = (nib & 8u) ? fmaxf(g[3] + bb[nt].w, 0.f) * sc : 0.f;
                        uint2 o;
                        o.x = (unsigned)f32_to_bf16(v0) | ((unsigned)f32_to_bf16(v1) << 16);
                        o.y = (unsigned)f32_to_bf16(v2) | ((unsigned)f32_to_bf16(v3) << 16);
                        *reinterpret_cast<uint2 *>(tile + ac_off[nt] + 2048 * mt) = o;
                        if constexpr (EMIT) note(o, nt, mt);
                    }
                }
            }
            if constexpr (EMIT) {
                emit_tile(a.t2_e, lb);
                if (a.bits_e) a.bits_e[((lb * 2 + 1) * gridDim.x + blockIdx.x) * 512 + wave * 64 + lane] = sg;
            }
            product(a.w1F + lb * H * H);
            // ---- h += (acc + b1) sigmoid(gate[image])
            {
                const rsrc_t br = rsrc_of(a.b1 + lb * H, (size_t)H * 4);
                float4 bb[4], gt[4];
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    bb[nt] = __builtin_bit_cast(float4, bld(br, cq, (unsigned)(nt * 16) * 4u));
                    gt[nt] = __builtin_bit_cast(float4, bld(ctr, cq, (unsigned)((slot0 + 1 + blk) * H + nt * 16) * 4u));
                    gt[nt].x = 1.f / (1.f + __expf(-gt[nt].x)); gt[nt].y = 1.f / (1.f + __expf(-gt[nt].y));
                    gt[nt].z = 1.f / (1.f + __expf(-gt[nt].z)); gt[nt].w = 1.f / (1.f + __expf(-gt[nt].w));
                }
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt) {
                        v4f t3 = acc[nt][mt];
                        t3[0] += bb[nt].x; t3[1] += bb[nt].y; t3[2] += bb[nt].z; t3[3] += bb[nt].w;
                        if constexpr (EMIT) {             // the tape keeps t3 as bf16: the forward multiplies the same rounded value
                            uint2 o;
                            o.x = (unsigned)f32_to_bf16(t3[0]) | ((unsigned)f32_to_bf16(t3[1]) << 16);
                            o.y = (unsigned)f32_to_bf16(t3[2]) | ((unsigned)f32_to_bf16(t3[3]) << 16);
                            *reinterpret_cast<uint2 *>(tile + ac_off[nt] + 2048 * mt) = o;
                            t3[0] = __uint_as_float(o.x << 16); t3[1] = __uint_as_float(o.x & 0xffff0000u);
                            t3[2] = __uint_as_float(o.y << 16); t3[3] = __uint_as_float(o.y & 0xffff0000u);
                        }
                        h[nt][mt][0] = fmaf(t3[0], gt[nt].x, h[nt][mt][0]); h[nt][mt][1] = fmaf(t3[1], gt[nt].y, h[nt][mt][1]);
                        h[nt][mt][2] = fmaf(t3[2], gt[nt].z, h[nt][mt][2]); h[nt][mt][3] = fmaf(t3[3], gt[nt].w, h[nt][mt][3]);
                    }
                if constexpr (EMIT) emit_tile(a.t3_e, lb);
            }
        }
        // ================= final layer: prm = Wf h + bf as two 64-row products in the flow variable's column order
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const v4f g = h[nt][mt];
                uint2 o;
                o.x = (unsigned)f32_to_bf16(g[0]) | ((unsigned)f32_to_bf16(g[1]) << 16);
                o.y = (unsigned)f32_to_bf16(g[2]) | ((unsigned)f32_to_bf16(g[3]) << 16);
                *reinterpret_cast<uint2 *>(tile + ac_off[nt] + 2048 * mt) = o;
            }
        if constexpr (EMIT) emit_tile(a.hf_e, (size_t)l);
        v4f os[2] = {v4f{0.f, 0.f, 0.f, 0.f}, v4f{0.f, 0.f, 0.f, 0.f}}, ou[2] = {v4f{0.f, 0.f, 0.f, 0.f}, v4f{0.f, 0.f, 0.f, 0.f}};
        {
            const rsrc_t ws = rsrc_of(a.wsF + (size_t)l * 64 * H, (size_t)64 * H * 2), wu = rsrc_of(a.wuF + (size_t)l * 64 * H, (size_t)64 * H * 2);
            uint4 w2f[16];
#pragma unroll
            for (int ks = 0; ks < 16; ++ks) w2f[ks] = frag(ws, lane16, nt3, 16, ks);
            const float4 bsv = *reinterpret_cast<const float4 *>(a.bs + (size_t)l * 64 + d0), buv = *reinterpret_cast<const float4 *>(a.bu + (size_t)l * 64 + d0);
            __builtin_amdgcn_sched_barrier(0);
            __syncthreads();                              // bf16(h) complete in the image
#pragma unroll
            for (int ks = 0; ks < 16; ++ks)
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
                    os[mi] = mfma(w2f[ks], *reinterpret_cast<const uint4 *>(actb + mt3 * 2048 + fa_off[ks & 1] + (8192 * (ks >> 1) + 2048 * mi)), os[mi]);
#pragma unroll
            for (int ks = 0; ks < 16; ++ks) w2f[ks] = frag(wu, lane16, nt3, 16, ks);
#pragma unroll
            for (int ks = 0; ks < 16; ++ks)
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
                    ou[mi] = mfma(w2f[ks], *reinterpret_cast<const uint4 *>(actb + mt3 * 2048 + fa_off[ks & 1] + (8192 * (ks >> 1) + 2048 * mi)), ou[mi]);
            const float bsa[4] = {bsv.x, bsv.y, bsv.z, bsv.w}, bua[4] = {buv.x, buv.y, buv.z, buv.w};
            // ---- the coupling (inverse direction) where the products land, y -> LDS (f32) for the affine map
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) {
                v4f y = x[mi];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int c = d0 + e, j = (c - first) >> 1;
                    if (c < dim && c >= first && ((c - first) & 1) == 0) {
                        const float shift = os[mi][e] + bsa[e], us = ou[mi][e] + bua[e];
                        const float scale = 1.f / (1.f + expf(-(us + 2.f))) + 1e-3f;
                        y[e] = (x[mi][e] - shift) / scale;
                        ld[mi] -= logf(scale);
                        if constexpr (EMIT) {
                            if (ok[mi]) {
                                float *pr = a.prm_e + ((size_t)l * R + grow[mi]) * 64;
                                pr[j] = shift; pr[T + j] = us;
                            }
                        }
                    }
                }
                if constexpr (EMIT) {
                    if (ok[mi]) {
                        *reinterpret_cast<v4f *>(a.y_e + ((size_t)l * R + grow[mi]) * 64 + d0) = y;
                        if (nt3 == 3 && q == 3) {         // prm columns 2 T .. 63 are zero (the reverse pass reads whole rows)
                            float *pr = a.prm_e + ((size_t)l * R + grow[mi]) * 64;
                            for (int c = 2 * T; c < 64; ++c) pr[c] = 0.f;
                        }
                    }
                }
                *reinterpret_cast<v4f *>(yb + ((mt3 + mi) * 16 + l15) * YP + d0) = y;
                if constexpr (EMIT) {
                    if (a.prmc_e && ok[mi]) {
                        float *pc = a.prmc_e + ((size_t)l * R + grow[mi]) * 128 + d0;
                        *reinterpret_cast<v4f *>(pc) = v4f{os[mi][0] + bsa[0], os[mi][1] + bsa[1], os[mi][2] + bsa[2], os[mi][3] + bsa[3]};
                        *reinterpret_cast<v4f *>(pc + 64) = v4f{ou[mi][0] + bua[0], ou[mi][1] + bua[1], ou[mi][2] + bua[2], ou[mi][3] + bua[3]};
                    }
                }
            }
        }
        __syncthreads();                                  // y complete; every wave is through the final layer's reads of the image
        // ================= v = A^-1 y + c^-1 (45 x 45, f32): outputs d0 .. d0 + 3 of this lane's two rows
        {
            const float4 cv = *reinterpret_cast<const float4 *>(a.cinv + (size_t)l * 64 + d0);
            v4f v0 = {cv.x, cv.y, cv.z, cv.w}, v1 = v0;
            const float *y0 = yb + (mt3 * 16 + l15) * YP, *y1 = y0 + 16 * YP;
            for (int k4 = 0; k4 < 12; ++k4) {             // k = 0 .. 47 (columns past dim are zero on both sides)
                const float4 ya = *reinterpret_cast<const float4 *>(y0 + 4 * k4), yc = *reinterpret_cast<const float4 *>(y1 + 4 * k4);
                const float yav[4] = {ya.x, ya.y, ya.z, ya.w}, ycv[4] = {yc.x, yc.y, yc.z, yc.w};
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    const float4 ar = *reinterpret_cast<const float4 *>(aT + (4 * k4 + kk) * 64 + d0);
                    v0[0] = fmaf(ar.x, yav[kk], v0[0]); v0[1] = fmaf(ar.y, yav[kk], v0[1]); v0[2] = fmaf(ar.z, yav[kk], v0[2]); v0[3] = fmaf(ar.w, yav[kk], v0[3]);
                    v1[0] = fmaf(ar.x, ycv[kk], v1[0]); v1[1] = fmaf(ar.y, ycv[kk], v1[1]); v1[2] = fmaf(ar.z, ycv[kk], v1[2]); v1[3] = fmaf(ar.w, ycv[kk], v1[3]);
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) { x[0][e] = d0 + e < dim ? v0[e] : 0.f; x[1][e] = d0 + e < dim ? v1[e] : 0.f; }
        }
        publish_x();
        __syncthreads();                                  // next layer's operand tile published; y / (A^-1)^T may be overwritten
    }

    // ---- outputs: the sample by the lanes that hold it, log q through a small LDS table
    float cst = 0.f;
    for (int i = 0; i < a.L; ++i) cst += a.const_parts[i];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
        if (ok[mi])
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (d0 + e < dim) a.out[(size_t)grow[mi] * dim + d0 + e] = x[mi][e];
        float s2 = sq_in[mi], sl = ld[mi];
        s2 += __shfl_xor(s2, 16, 64); s2 += __shfl_xor(s2, 32, 64);
        sl += __shfl_xor(sl, 16, 64); sl += __shfl_xor(sl, 32, 64);
        if (q == 0) { red[0][nt3][(mt3 + mi) * 16 + l15] = s2; red[1][nt3][(mt3 + mi) * 16 + l15] = sl; }
    }
    __syncthreads();
    if (tid < ROWS && 64 * chunk + tid < a.N) {
        const long r = (long)(64 * chunk + tid) * rn + (long)b * rb;
        const float s2 = red[0][0][tid] + red[0][1][tid] + red[0][2][tid] + red[0][3][tid];
        const float sl = red[1][0][tid] + red[1][1][tid] + red[1][2][tid] + red[1][3][tid];
        // log q(sample) = log N(z0) - log|det d sample / d z0| = log N(z0) + sum log scale + sum_l (sum log_scale_l + sum log diag U_l)
        a.logp[r] = (-0.5f * s2 - 0.5f * (float)dim * 1.8378770664093453f) - sl + cst;
    }
}

}}  // namespace mhe::glowf

using namespace mhe;

extern "C" int mhe_glow_layers_supported(int N, int B, int dim, int hidden, int layers, int blocks) {
    return N > 0 && B > 0 && dim > 1 && dim <= 48 && hidden == 512 && layers > 0 && layers <= glowf::MAXL && blocks == glowf::NBLK;
}

extern "C" int mhe_glow_layers_bf16(const float *noise, const float *ctab, int ctab_stride, const void *wxF, const void *w0F, const void *w1F, const void *wsF,
                                    const void *wuF, const float *b0, const float *b1, const float *bs, const float *bu, const float *ainvT,
                                    const float *cinv, const float *const_parts, const unsigned char *drop_bits, float p_drop, float *out, float *log_q,
                                    float *v_e, float *y_e, float *prm_e, void *tb_e, void *t2_e, void *t3_e, void *hf_e, float *prmc_e, void *vb_e,
                                    void *bits_e, int N, int B, int dim, int hidden, int layers, int blocks, long row_n, long row_b, void *stream) {
    MHE_REQUIRE(noise && ctab && wxF && w0F && w1F && wsF && wuF && b0 && b1 && bs && bu && ainvT && cinv && const_parts && out && log_q,
                "mhe_glow_layers_bf16: null pointer");
    MHE_REQUIRE(mhe_glow_layers_supported(N, B, dim, hidden, layers, blocks),
                "mhe_glow_layers_bf16: needs hidden 512, 2 blocks per layer, at most %d layers, dim <= 48 (N=%d B=%d dim=%d)", glowf::MAXL, N, B, dim);
    MHE_REQUIRE((row_n == B && row_b == 1) || (row_n == 1 && row_b == N), "mhe_glow_layers_bf16: rows are sample-major (B, 1) or batch-major (1, N)");
    MHE_REQUIRE(ctab_stride % 4 == 0 && ctab_stride >= layers * (1 + blocks) * hidden, "mhe_glow_layers_bf16: bad context-table stride");
    MHE_REQUIRE(p_drop >= 0.f && p_drop < 1.f, "mhe_glow_layers_bf16: p_drop=%f", (double)p_drop);
    const long R = (long)N * B;
    MHE_REQUIRE(R * hidden < (1L << 30), "mhe_glow_layers_bf16: R x hidden beyond the 32-bit byte offsets of the tape rows");
    const bool emit = v_e || y_e || prm_e || tb_e || t2_e || t3_e || hf_e;
    MHE_REQUIRE(!emit || (v_e && y_e && prm_e && tb_e && t2_e && t3_e && hf_e), "mhe_glow_layers_bf16: the tape tensors come together");
    MHE_REQUIRE((!prmc_e && !vb_e && !bits_e) || (emit && prmc_e && vb_e && bits_e && N % 64 == 0),
                "mhe_glow_layers_bf16: prmc_e, vb_e and bits_e come together, with the tape, for N %% 64 == 0");
    glowf::Args a;
    a.in = noise; a.ctab = ctab; a.wxF = (const u16 *)wxF; a.w0F = (const u16 *)w0F; a.w1F = (const u16 *)w1F; a.wsF = (const u16 *)wsF; a.wuF = (const u16 *)wuF;
    a.b0 = b0; a.b1 = b1; a.bs = bs; a.bu = bu; a.ainvT = ainvT; a.cinv = cinv; a.const_parts = const_parts;
    a.drop = drop_bits; a.drop_scale = 1.f / (1.f - p_drop); a.out = out; a.logp = log_q;
    a.v_e = v_e; a.y_e = y_e; a.prm_e = prm_e; a.tb_e = (u16 *)tb_e; a.t2_e = (u16 *)t2_e; a.t3_e = (u16 *)t3_e; a.hf_e = (u16 *)hf_e;
    a.prmc_e = prmc_e; a.vb_e = (u16 *)vb_e; a.bits_e = (uint2 *)bits_e;
    a.R = (int)R; a.B = B; a.N = N; a.dim = dim; a.L = layers; a.cstride = ctab_stride; a.row_n = (int)row_n; a.row_b = (int)row_b;
    const dim3 grid((unsigned)(((N + 63) / 64) * B));
    if (emit) hipLaunchKernelGGL(glowf::layers_kernel<true>, grid, dim3(512), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(glowf::layers_kernel<false>, grid, dim3(512), 0, (hipStream_t)stream, a);
    return check_launch("glowf::layers_kernel");
}

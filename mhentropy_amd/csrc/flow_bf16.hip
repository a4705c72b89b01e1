// bf16-operand variant of the fused RealNVP coupling stack (see flow.hip for the f32 one and for the
// formulation).  Differences that matter on gfx950:
//   * v_mfma_f32_32x32x16_bf16: a wavefront carries 32 hypothesis rows; weights (A operand, 1 KiB per
//     32x16 block) are read from LDS once per 32-cycle MFMA = 128 B/clk per CU, half the LDS peak
//     (a 16-row-per-wave layout would sit exactly at the LDS ceiling);
//   * hidden activations stay in registers: the f32 accumulator of layer L (column = row of the batch
//     on lane&31, row = (r&3)+8(r>>2)+4(lane>>5)) is rounded pairwise to bf16 and is then, without
//     moving a lane, the B operand of layer L+1 (k = (j&3)+8(j>>2)+4h inside a 16-deep block; the host
//     packs the weights with the same k permutation);
//   * the weight stream (bf16, 39 stages of 16 KiB per network at h=512) goes HBM/L2 -> LDS by
//     global_load_lds_dwordx4 (the packed stream is lane-linear, exactly the DMA's destination order)
//     into a 5-deep ring, 4 stages in flight behind a counted vmcnt, one raw s_barrier per stage.
// Arithmetic: bf16 x bf16 products, f32 accumulation; the flow variable, s, t, exp and log-det are f32.
#include "common.h"
#include <cstring>

namespace mhe { namespace flowns {      // flow_ns.hip: hidden = 512
size_t packed_bytes_per_net();
void pack_net_host(const float *W0, const float *W1, const float *W2, int dim, unsigned short *out);
int launch(const float *in, float *out, const float *cond, const void *wstream, const float *bias2, const float *mask,
           float *sum_s, float *log_prob, int R, int B, int dim, int ncoup, int inv, hipStream_t s,
           unsigned short *h1e, unsigned short *h2e, float *oe);
}}

namespace mhe { namespace flowb {

constexpr int STAGE_BLOCKS = 16;                 // 1 KiB blocks per stage
constexpr int STAGE_BYTES = STAGE_BLOCKS * 1024;
constexpr int RING = 5, DEPTH = 4;               // LDS ring slots, stages in flight
constexpr int KB0 = 3;                           // 45 -> 48 input dims = 3 k-blocks of 16
constexpr int DT = 2;                            // output dims padded to 2 tiles of 32

// Stream of one network (NT = hidden/32 tiles of 32 units).  The layer-1 accumulators of ALL NT tiles
// stay resident (NT*16 accumulator registers); layer-0 output is produced 4 tiles at a time and consumed
// immediately as the B operand of layer 1:
//   per chunk of 4 layer-0 tiles: 1 stage  = 12 blocks (tile-major, 3 k-blocks each) + 4 pad
//                                 8*NT/16 stages of layer 1, blocks ordered (k-block of the chunk, output tile)
//   then NT/4 stages of layer 2, blocks ordered (hidden tile, half, output tile)
// At NT = 16 (h = 512) the resident accumulators are limited to RT = 8 tiles (128 registers) and the
// network is evaluated in NT/RT = 2 passes over the layer-0 chunks (layer 0 recomputed: +8 % MFMAs,
// +10 % stream) - 16 resident tiles fill the whole accumulator file and spill.
static inline int net_stages(int hidden) {
    const int nt = hidden / 32, rt = nt > 8 ? 8 : nt;
    return (nt / rt) * ((nt / 4) * (1 + 8 * rt / 16) + rt / 4);
}

typedef __attribute__((ext_vector_type(8))) __bf16 bf8;

__device__ __forceinline__ unsigned pk(float a, float b) {
    return (unsigned)f32_to_bf16(a) | ((unsigned)f32_to_bf16(b) << 16);
}
// 8 accumulator registers (r0..r0+7 of a 32x32 tile) -> one B fragment
__device__ __forceinline__ uint4 pack8(const v16f &a, int r0) {
    return make_uint4(pk(a[r0], a[r0 + 1]), pk(a[r0 + 2], a[r0 + 3]), pk(a[r0 + 4], a[r0 + 5]), pk(a[r0 + 6], a[r0 + 7]));
}
__device__ __forceinline__ float leaky(float v) { return fmaxf(v, 0.01f * v); }

// LDS fragment reads as inline asm: hipcc treats every ds_read after a global_load_lds as possibly
// aliasing the DMA and inserts s_waitcnt vmcnt(0) in front of it, which drains the whole weight pipeline
// once per stage (cdna_hip_programming.md section 5.7 item 1).  The reads' completion is therefore counted
// here: every asm block ends with lgkmcnt(0) itself and has early-clobber outputs.  (Leaving the reads
// in flight behind "=v" outputs and waiting later was tried and is WRONG at this register pressure: the
// compiler copies the not-yet-landed registers.)  Only lgkmcnt(0) is used because scalar loads share the
// counter and return out of order.
typedef unsigned u4 __attribute__((ext_vector_type(4)));
#define LDS_READ8_(W, ADDR, TAIL, CONSTR)                                                                   \
    asm volatile("ds_read_b128 %0, %8\n\tds_read_b128 %1, %8 offset:1024\n\tds_read_b128 %2, %8 offset:2048\n\t"   \
                 "ds_read_b128 %3, %8 offset:3072\n\tds_read_b128 %4, %8 offset:4096\n\tds_read_b128 %5, %8 offset:5120\n\t" \
                 "ds_read_b128 %6, %8 offset:6144\n\tds_read_b128 %7, %8 offset:7168" TAIL                    \
                 : CONSTR(W[0]), CONSTR(W[1]), CONSTR(W[2]), CONSTR(W[3]), CONSTR(W[4]), CONSTR(W[5]),       \
                   CONSTR(W[6]), CONSTR(W[7])                                                               \
                 : "v"(ADDR)                                                                                \
                 : "memory")
#define C_EARLY(x) "=&v"(x)
#define LDS_READ8_WAIT(W, ADDR) LDS_READ8_(W, ADDR, "\n\ts_waitcnt lgkmcnt(0)", C_EARLY)

#define MFMA32(w, b, acc)                                                                                 \
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8, (w)), __builtin_bit_cast(bf8, (b)), (acc), 0, 0, 0)

// UNI: all 32 rows of a wavefront belong to one image (N % 32 == 0), so the conditioning vector is
// wave-uniform and is fetched with SCALAR loads (lgkmcnt) - ordinary vector loads would make the
// compiler wait vmcnt(0) and drain the in-flight weight DMAs at every use.
template <int NT, bool UNI>     // NT = hidden / 32
__global__ __launch_bounds__(256) void couplings_bf16_kernel(
    const float *__restrict__ in, float *__restrict__ out, const float *__restrict__ cond,
    const unsigned char *__restrict__ wstream, const float *__restrict__ bias2, const float *__restrict__ mask,
    float *__restrict__ sum_s_o, float *__restrict__ logp_o, int R, int B, int dim, int ncoup, int inverse) {
    constexpr int H = NT * 32, CH = NT / 4;
    constexpr int RT = NT > 8 ? 8 : NT, NP = NT / RT;      // resident layer-1 tiles, passes
    constexpr int L1S = 8 * RT / 16;                // layer-1 stages per chunk of 4 layer-0 tiles
    constexpr int L2S = RT / 4;
    constexpr int NET_STAGES = NP * (CH * (1 + L1S) + L2S);
    extern __shared__ __attribute__((aligned(16))) unsigned char ring[];       // RING * STAGE_BYTES, then s_keep
    float *s_lds = reinterpret_cast<float *>(ring + RING * STAGE_BYTES);        // [4 waves][32 values][64 lanes]
    constexpr int COND_OFF = RING * STAGE_BYTES + 4 * 32 * 64 * 4;              // [4 waves][2 buffers][2 layers][H] f32 (UNI)
    constexpr int COND_BYTES = 2 * H * 4;
    constexpr int MASK_OFF = COND_OFF + 4 * 2 * COND_BYTES;

    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int h = lane >> 5, m = lane & 31;
    const int N = R / B;
    const int g = (blockIdx.x * 4 + wave) * 32 + m;          // image-major tiling
    const bool valid = g < R;
    const int gc = valid ? g : R - 1;
    const int b = gc / N, r = (gc % N) * B + b;

    // D-layout dim of (tile T, reg i) for this lane
    auto dim_of = [&](int T, int i) { return 32 * T + (i & 3) + 8 * (i >> 2) + 4 * h; };
    const int b_uni = __builtin_amdgcn_readfirstlane(b);
    // accumulator tile <- 32 conditioning values (unit u of the tile sits in reg i of half h: u = (i&3)+8(i>>2)+4h).
    // UNI: the wave's conditioning row was DMA'd into its private LDS region (see cond_issue below) and is
    // read with asm ds_reads (compiler-visible LDS reads would be fenced with vmcnt(0), like the fragments).
    auto init_tile = [&](v16f &a, const float *p_lane, unsigned lds_addr) {
        if constexpr (UNI) {
            u4 c0, c1, c2, c3;
            asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:32\n\tds_read_b128 %2, %4 offset:64\n\t"
                         "ds_read_b128 %3, %4 offset:96\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(c0), "=&v"(c1), "=&v"(c2), "=&v"(c3) : "v"(lds_addr) : "memory");
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                a[k] = __uint_as_float(c0[k]); a[4 + k] = __uint_as_float(c1[k]);
                a[8 + k] = __uint_as_float(c2[k]); a[12 + k] = __uint_as_float(c3[k]);
            }
        } else {
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const float4 c4 = *reinterpret_cast<const float4 *>(p_lane + 8 * gq + 4 * h);
                a[4 * gq] = c4.x; a[4 * gq + 1] = c4.y; a[4 * gq + 2] = c4.z; a[4 * gq + 3] = c4.w;
            }
        }
    };
    v16f x[DT];
#pragma unroll
    for (int T = 0; T < DT; ++T)
#pragma unroll
        for (int i = 0; i < 16; ++i) {      // unconditional (clamped) loads so that they are issued back to back
            const int d = dim_of(T, i);
            const float v = in[(size_t)r * dim + (d < dim ? d : dim - 1)];
            x[T][i] = d < dim ? v : 0.f;
        }
    float sq_in = 0.f, sum_s = 0.f;
    // pass-through masks of all couplings as 64-bit words (bit d = mask[ci][d] != 0, padded dims pass through),
    // built once with ballots BEFORE the DMA pipeline starts: RealNVP masks are 0/1 (hand/flows.py:153-155)
    unsigned long long *mask_w = reinterpret_cast<unsigned long long *>(ring + MASK_OFF);
    for (int ci = 0; ci < ncoup; ++ci) {
        const float mv = lane < dim ? mask[ci * dim + lane] : 1.f;
        const unsigned long long bits = __ballot(mv != 0.f);
        if (tid == 0) mask_w[ci] = bits;
    }
    __syncthreads();
#pragma unroll
    for (int T = 0; T < DT; ++T)
#pragma unroll
        for (int i = 0; i < 16; ++i) sq_in = fmaf(x[T][i], x[T][i], sq_in);

    // ---- weight-stream ring (all state block-uniform)
    const int nnets = 2 * ncoup;
    const int total = nnets * NET_STAGES;
    const size_t net_bytes = (size_t)NET_STAGES * STAGE_BYTES;
    auto stage_src = [&](int gs) {       // global stage index -> source address of this wave's first block
        const int seq = gs / NET_STAGES, st = gs - seq * NET_STAGES, step = seq >> 1;
        const int net = 2 * (inverse ? ncoup - 1 - step : step) + (seq & 1);
        return wstream + (size_t)net * net_bytes + (size_t)st * STAGE_BYTES;
    };
    const unsigned lds_base = (unsigned)(size_t)((__attribute__((address_space(3))) unsigned char *)ring);
    // UNI: DMA the conditioning row (2 layers x H floats, contiguous) of network `seq` of this wave's image into
    // the wave's private double buffer, one whole network ahead of its use; vmcnt is in-order, so every later
    // counted weight-stage wait also covers it
    auto cond_issue = [&](int seq) {
        if constexpr (UNI) {
            if (seq < nnets) {
                const int step = seq >> 1, net = 2 * (inverse ? ncoup - 1 - step : step) + (seq & 1);
                const unsigned char *src = reinterpret_cast<const unsigned char *>(cond + ((size_t)(b_uni * nnets + net) * 2) * H) + lane * 16;
                unsigned char *dst = ring + COND_OFF + (wave * 2 + (seq & 1)) * COND_BYTES;
#pragma unroll
                for (int j = 0; j < COND_BYTES / 1024; ++j)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + j * 1024),
                                                     (__attribute__((address_space(3))) void *)(dst + j * 1024), 16, 0, 0);
            }
        }
    };
    cond_issue(0);
    auto issue = [&](int gsi) {          // each wave DMA's 4 of the stage's 16 blocks; branch-free: past the end
        const int gcl = gsi < total ? gsi : total - 1;      // of the stream the last stage is re-fetched into a free slot
        const unsigned char *src = stage_src(gcl) + (size_t)(wave * 4) * 1024 + lane * 16;
        unsigned char *dst = ring + (gsi % RING) * STAGE_BYTES + (wave * 4) * 1024;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + j * 1024),
                                             (__attribute__((address_space(3))) void *)(dst + j * 1024), 16, 0, 0);
    };
    int gs = 0;                          // next stage to consume
#pragma unroll
    for (int s0 = 0; s0 < DEPTH; ++s0) issue(s0);
    // wait until all but the 4*(DEPTH-1) youngest DMAs of this wave have landed (= stage gs is in LDS; ordinary
    // loads in between only make the wait stricter), barrier so every wave's share is visible, refill the ring.
    // Slot (gs+DEPTH)%RING last held stage gs+DEPTH-RING <= gs-2, which every wave finished before this barrier.
    const unsigned lds_lane = lds_base + lane * 16;
    auto stage_begin = [&]() -> unsigned {       // returns the LDS byte address of this lane's 16 bytes of block 0
        asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        issue(gs + DEPTH);
        const unsigned a = lds_lane + (gs % RING) * STAGE_BYTES;
        ++gs;
        return a;
    };
    static_assert(DEPTH == 4, "the counted wait above is 4 DMAs x (DEPTH-1) stages");

    for (int step = 0; step < ncoup; ++step) {
        const int ci = inverse ? ncoup - 1 - step : step;
        uint4 xin[KB0];
        unsigned mbits = 0;              // bit (16T+i): this lane's dim is a pass-through (mask == 1) dim
        {
            const unsigned long long mw = mask_w[ci];
            v16f xm[DT];
#pragma unroll
            for (int T = 0; T < DT; ++T)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const bool keep = (mw >> dim_of(T, i)) & 1ull;
                    mbits |= keep ? (1u << (16 * T + i)) : 0u;
                    xm[T][i] = keep ? x[T][i] : 0.f;
                }
            xin[0] = pack8(xm[0], 0); xin[1] = pack8(xm[0], 8); xin[2] = pack8(xm[1], 0);
        }
#pragma unroll 1
        for (int netk = 0; netk < 2; ++netk) {
            const int net = 2 * ci + netk;
            const float *cb = cond + ((size_t)(b * nnets + net) * 2) * H;
            const int seq = 2 * step + netk;
            cond_issue(seq + 1);
            const unsigned cl = lds_base + COND_OFF + (wave * 2 + (seq & 1)) * COND_BYTES + 16 * h;   // + 4*(unit of the tile's reg 0)
            v16f o[DT];
#pragma unroll
            for (int T = 0; T < DT; ++T)
#pragma unroll
                for (int i = 0; i < 16; ++i)        // bias2 is padded to 64 per net: no bounds test, loads issue back to back
                    o[T][i] = bias2[net * 64 + dim_of(T, i)];
#pragma unroll 1
            for (int pass = 0; pass < NP; ++pass) {
                // ---- layer-1 accumulators of RT resident tiles, initialised with c1(feat) + biases
                v16f H2[RT];
#pragma unroll
                for (int Tl = 0; Tl < RT; ++Tl) {
                    if ((Tl & 1) == 0) __builtin_amdgcn_sched_barrier(0);     // at most 2 tiles of loads in flight
                    init_tile(H2[Tl], cb + H + 32 * (pass * RT + Tl), cl + 4 * (H + 32 * (pass * RT + Tl)));
                }
#pragma unroll 1
                for (int c = 0; c < CH; ++c) {
                    // layer 0 for hidden tiles 4c..4c+3 -> leaky -> 8 bf16 B fragments
                    uint4 H1c[8];
                    {
                        const unsigned la = stage_begin();
                        u4 wa[8], wb[4];                 // 12 blocks used
                        LDS_READ8_WAIT(wa, la);
                        asm volatile("ds_read_b128 %0, %4 offset:8192\n\tds_read_b128 %1, %4 offset:9216\n\tds_read_b128 %2, %4 offset:10240\n\t"
                                     "ds_read_b128 %3, %4 offset:11264\n\ts_waitcnt lgkmcnt(0)"
                                     : "=&v"(wb[0]), "=&v"(wb[1]), "=&v"(wb[2]), "=&v"(wb[3]) : "v"(la) : "memory");
#pragma unroll
                        for (int tl = 0; tl < 4; ++tl) {
                            v16f acc;
                            init_tile(acc, cb + 32 * (4 * c + tl), cl + 4 * 32 * (4 * c + tl));
#pragma unroll
                            for (int kb = 0; kb < KB0; ++kb) {
                                const int blk = tl * KB0 + kb;
                                if (blk < 8) MFMA32(wa[blk], xin[kb], acc); else MFMA32(wb[blk - 8], xin[kb], acc);
                            }
#pragma unroll
                            for (int i = 0; i < 16; ++i) acc[i] = leaky(acc[i]);
                            H1c[2 * tl] = pack8(acc, 0);
                            H1c[2 * tl + 1] = pack8(acc, 8);
                        }
                    }
                    // layer 1: every resident accumulator tile += W1[tile][chunk] * H1c
#pragma unroll
                    for (int u = 0; u < L1S; ++u) {
                        const unsigned la = stage_begin();
                        u4 wa[8];
#pragma unroll
                        for (int hf = 0; hf < 2; ++hf) {
                            LDS_READ8_WAIT(wa, la + hf * 8192);
#pragma unroll
                            for (int j = 0; j < 8; ++j) {
                                const int f = u * STAGE_BLOCKS + hf * 8 + j;           // flat = kb_local * RT + tile_local
                                MFMA32(wa[j], H1c[f / RT], H2[f % RT]);
                            }
                        }
                    }
                }
                // ---- layer 2: leaky -> bf16 fragments of each resident hidden tile -> output accumulators
#pragma unroll
                for (int v = 0; v < L2S; ++v) {
                    const unsigned la = stage_begin();
                    u4 wa[8];
#pragma unroll
                    for (int tl = 0; tl < 4; ++tl) {
                        const int Tl = 4 * v + tl;
                        if ((tl & 1) == 0) LDS_READ8_WAIT(wa, la + (tl >> 1) * 8192);
#pragma unroll
                        for (int i = 0; i < 16; ++i) H2[Tl][i] = leaky(H2[Tl][i]);
                        const uint4 f0 = pack8(H2[Tl], 0), f1 = pack8(H2[Tl], 8);
#pragma unroll
                        for (int T = 0; T < DT; ++T) {      // block ((tl*2 + half)*2 + T)
                            MFMA32(wa[((tl & 1) * 2 + 0) * 2 + T], f0, o[T]);
                            MFMA32(wa[((tl & 1) * 2 + 1) * 2 + T], f1, o[T]);
                        }
                    }
                }
            }
            if (netk == 0) {
#pragma unroll
                for (int T = 0; T < DT; ++T)
#pragma unroll
                    for (int i = 0; i < 16; ++i)      // parked in LDS while the t network runs (register budget)
                        s_lds[(wave * 32 + 16 * T + i) * 64 + lane] = ((mbits >> (16 * T + i)) & 1u) ? 0.f : tanhf(o[T][i]);
            } else {
#pragma unroll
                for (int T = 0; T < DT; ++T)
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const bool keep = (mbits >> (16 * T + i)) & 1u;
                        const float s = s_lds[(wave * 32 + 16 * T + i) * 64 + lane], t = o[T][i];
                        if (!keep) {
                            if (!inverse) x[T][i] = x[T][i] * expf(s) + t;        // flows.py:216
                            else          x[T][i] = (x[T][i] - t) * expf(-s);     // flows.py:225
                            sum_s += s;
                        }
                    }
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // no DMA may still target LDS when the workgroup retires
    float sq_out = 0.f;
#pragma unroll
    for (int T = 0; T < DT; ++T)
#pragma unroll
        for (int i = 0; i < 16; ++i) sq_out = fmaf(x[T][i], x[T][i], sq_out);
    float base_sq = inverse ? sq_out : sq_in;
    base_sq += __shfl_xor(base_sq, 32, 64);
    sum_s += __shfl_xor(sum_s, 32, 64);
    if (valid) {
#pragma unroll
        for (int T = 0; T < DT; ++T)
#pragma unroll
            for (int i = 0; i < 16; ++i) { const int d = dim_of(T, i); if (d < dim) out[(size_t)r * dim + d] = x[T][i]; }
        if (h == 0) {
            if (sum_s_o) sum_s_o[r] = sum_s;
            if (logp_o) logp_o[r] = (-0.5f * base_sq - 0.5f * (float)dim * 1.8378770664093453f) - sum_s;
        }
    }
}

}}  // namespace mhe::flowb

using namespace mhe;

extern "C" size_t mhe_flow_packed_bytes_per_net_bf16(int dim, int hidden) {
    if (dim <= 0 || dim > 48 || hidden < 128 || hidden % 128) return 0;
    if (hidden == 512) return flowns::packed_bytes_per_net();
    return (size_t)flowb::net_stages(hidden) * flowb::STAGE_BYTES;
}

// 1 KiB block (To, kb) of W[rows][cols]: lane l = 32h + n, element j <-> W[32To+n][16kb + (j&3) + 8(j>>2) + 4h]
namespace mhe { namespace flowb {
void pack_block_bf16(const float *W, int rows, int cols, int To, int kb, unsigned short *dst) {
    for (int l = 0; l < 64; ++l) {
        const int h = l >> 5, n = l & 31;
        for (int j = 0; j < 8; ++j) {
            const int rr = 32 * To + n, cc = 16 * kb + (j & 3) + 8 * (j >> 2) + 4 * h;
            float v = (rr < rows && cc < cols) ? W[(size_t)rr * cols + cc] : 0.f;
            unsigned u;
            memcpy(&u, &v, 4);
            if ((u & 0x7fffffffu) > 0x7f800000u) u |= 0x00400000u;                 // keep NaN a NaN
            else u += 0x7fffu + ((u >> 16) & 1u);                                   // round to nearest even
            dst[l * 8 + j] = (unsigned short)(u >> 16);
        }
    }
}
}}
using mhe::flowb::pack_block_bf16;

extern "C" int mhe_flow_pack_net_bf16_host(const float *W0, const float *W1, const float *W2, int dim, int hidden,
                                           void *out_host) {
    const size_t total = mhe_flow_packed_bytes_per_net_bf16(dim, hidden);
    MHE_REQUIRE(total && W0 && W1 && W2 && out_host, "mhe_flow_pack_net_bf16_host: dim=%d (<=48) hidden=%d (multiple of 128)", dim, hidden);
    if (hidden == 512) {
        flowns::pack_net_host(W0, W1, W2, dim, reinterpret_cast<unsigned short *>(out_host));
        return MHE_OK;
    }
    const int NT = hidden / 32, CH = NT / 4, RT = NT > 8 ? 8 : NT, NP = NT / RT, L1S = 8 * RT / 16, L2S = RT / 4;
    unsigned short *out = reinterpret_cast<unsigned short *>(out_host);
    memset(out, 0, total);
    const size_t SH = flowb::STAGE_BYTES / 2;        // bf16 elements per stage
    unsigned short *p = out;
    for (int pass = 0; pass < NP; ++pass) {
        for (int c = 0; c < CH; ++c) {
            for (int tl = 0; tl < 4; ++tl)
                for (int kb = 0; kb < flowb::KB0; ++kb) pack_block_bf16(W0, hidden, dim, 4 * c + tl, kb, p + (tl * flowb::KB0 + kb) * 512);
            p += SH;
            for (int f = 0; f < 8 * RT; ++f)        // flat = kb_local * RT + tile_local ; k-block 8c + kb_local of W1's input
                pack_block_bf16(W1, hidden, hidden, pass * RT + f % RT, 8 * c + f / RT, p + (size_t)f * 512);
            p += (size_t)L1S * SH;
        }
        for (int v = 0; v < L2S; ++v, p += SH)
            for (int tl = 0; tl < 4; ++tl)
                for (int half = 0; half < 2; ++half)
                    for (int T = 0; T < flowb::DT; ++T)
                        pack_block_bf16(W2, dim, hidden, T, 2 * (pass * RT + 4 * v + tl) + half, p + ((tl * 2 + half) * 2 + T) * 512);
    }
    return MHE_OK;
}

extern "C" int mhe_flow_couplings_bf16_emit(const float *in, float *out, const float *cond, const void *wstream,
                                            const float *bias2, const float *mask, float *sum_s, float *log_prob, void *h1,
                                            void *h2, float *o, int R, int B, int dim, int hidden, int ncoup, int direction,
                                            void *stream) {
    MHE_REQUIRE(in && out && cond && wstream && bias2 && mask && h1 && h2 && o, "mhe_flow_couplings_bf16_emit: null pointer");
    MHE_REQUIRE(hidden == 512, "mhe_flow_couplings_bf16_emit: hidden=%d (only the 512-wide kernel writes its activations out)", hidden);
    MHE_REQUIRE(R > 0 && B > 0 && R % B == 0, "mhe_flow_couplings_bf16_emit: R=%d must be a positive multiple of B=%d", R, B);
    MHE_REQUIRE(dim > 0 && dim <= 48 && ncoup > 0 && ncoup <= 64, "mhe_flow_couplings_bf16_emit: dim=%d (1..48) ncoup=%d (1..64)", dim, ncoup);
    MHE_REQUIRE(direction == MHE_FLOW_FORWARD || direction == MHE_FLOW_INVERSE, "mhe_flow_couplings_bf16_emit: direction=%d", direction);
    return flowns::launch(in, out, cond, wstream, bias2, mask, sum_s, log_prob, R, B, dim, ncoup, direction == MHE_FLOW_INVERSE,
                          (hipStream_t)stream, reinterpret_cast<unsigned short *>(h1), reinterpret_cast<unsigned short *>(h2), o);
}

extern "C" int mhe_flow_couplings_bf16(const float *in, float *out, const float *cond, const void *wstream,
                                       const float *bias2, const float *mask, float *sum_s, float *log_prob, int R,
                                       int B, int dim, int hidden, int ncoup, int direction, void *stream) {
    MHE_REQUIRE(in && out && cond && wstream && bias2 && mask, "mhe_flow_couplings_bf16: null pointer");
    MHE_REQUIRE(R > 0 && B > 0 && R % B == 0, "mhe_flow_couplings_bf16: R=%d must be a positive multiple of B=%d", R, B);
    MHE_REQUIRE(dim > 0 && dim <= 48, "mhe_flow_couplings_bf16: dim=%d unsupported (1..48)", dim);
    MHE_REQUIRE(ncoup > 0, "mhe_flow_couplings_bf16: ncoup=%d", ncoup);
    MHE_REQUIRE(direction == MHE_FLOW_FORWARD || direction == MHE_FLOW_INVERSE, "mhe_flow_couplings_bf16: direction=%d", direction);
    const int inv = direction == MHE_FLOW_INVERSE;
    if (hidden == 512)
        return flowns::launch(in, out, cond, wstream, bias2, mask, sum_s, log_prob, R, B, dim, ncoup, inv, (hipStream_t)stream, nullptr, nullptr, nullptr);
    const dim3 grid((R + 127) / 128), block(256);
    const bool uni = ((R / B) % 32) == 0;          // every 32-row wavefront tile lies inside one image
    MHE_REQUIRE(ncoup <= 64, "mhe_flow_couplings_bf16: ncoup=%d > 64", ncoup);
    const size_t lds = (size_t)flowb::RING * flowb::STAGE_BYTES + 4 * 32 * 64 * sizeof(float) + 4 * 2 * 2 * (size_t)hidden * 4 +
                       64 * sizeof(unsigned long long);
    hipStream_t s = (hipStream_t)stream;
    const unsigned char *ws = reinterpret_cast<const unsigned char *>(wstream);
#define LAUNCH(NT_)                                                                                                   \
    do {                                                                                                              \
        static bool attr_set = false;                                                                                 \
        if (!attr_set) {                                                                                              \
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(flowb::couplings_bf16_kernel<NT_, true>),              \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                                \
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(flowb::couplings_bf16_kernel<NT_, false>),             \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                                \
            attr_set = true;                                                                                          \
        }                                                                                                             \
        if (uni)                                                                                                      \
            hipLaunchKernelGGL((flowb::couplings_bf16_kernel<NT_, true>), grid, block, lds, s, in, out, cond, ws, bias2,  \
                               mask, sum_s, log_prob, R, B, dim, ncoup, inv);                                         \
        else                                                                                                          \
            hipLaunchKernelGGL((flowb::couplings_bf16_kernel<NT_, false>), grid, block, lds, s, in, out, cond, ws, bias2, \
                               mask, sum_s, log_prob, R, B, dim, ncoup, inv);                                         \
    } while (0)
    switch (hidden) {
        case 128: LAUNCH(4); break;
        case 256: LAUNCH(8); break;
        default: MHE_REQUIRE(false, "mhe_flow_couplings_bf16: hidden=%d unsupported (128,256,512)", hidden);
    }
#undef LAUNCH
    return check_launch("flowb::couplings_bf16_kernel");
}

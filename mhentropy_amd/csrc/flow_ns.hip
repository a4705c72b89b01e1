// bf16 RealNVP coupling stack at hidden = 512, second generation: the hidden units of a net are SPLIT OVER THE
// EIGHT WAVES of a workgroup (64 units each) and a workgroup carries only 64 hypothesis rows, so that C2's 16,384
// rows make 256 workgroups = one per CU (the first-generation kernel in flow_bf16.hip keeps whole nets inside one
// wave: 32 rows per wave -> 512 wave tiles on 1,024 SIMDs, and every wave reads all weights from LDS).
//
//   * every weight block (1 KiB A fragment of v_mfma_f32_32x32x16_bf16) is needed by exactly ONE wave, which streams
//     its own blocks HBM/L2 -> LDS into a PRIVATE 4-slot ring with global_load_lds_dwordx4 behind a counted vmcnt:
//     the main loop has no workgroup barrier at all (first generation: one per 16 KiB stage, 39 per net);
//   * layer 0 -> layer 1: the 64 x 512 bf16 activations are exchanged through LDS as ready-made B fragments
//     (64 KiB); layer 1 -> layer 2 stays in registers (accumulator layout = B layout), layer 2 is split over K and
//     its 45 x 64 partial sums are combined by six "owner" waves that also hold the flow variable in f32,
//     apply the affine update and publish the masked bf16 fragments the next coupling's layer 0 reads;
//   * 6 workgroup barriers per net (h1 ready, h1 dead, 2 x partials written/consumed), all LDS traffic in inline
//     asm (compiler-visible ds_reads after an LDS-DMA are fenced with vmcnt(0), see flow_bf16.hip).
// Arithmetic and rounding points are those of flow_bf16.hip (bf16 x bf16 products, f32 accumulation; the flow variable,
// s, t, exp and the log-determinant in f32): oracle/flows_ref.py:forward_p_logdet_bf16.
// Reference: hand/flows.py:105-122 (coupling nets), :210-226 (forward_p / inverse), :195-208 (log_prob).
#include "common.h"
#include <cstring>

namespace mhe { namespace flowns {

constexpr int H = 512, WAVES = 8, ROWS = 64;
constexpr int P_L0 = 3, P_L1 = H / 16, P_L2 = 4, NET_PAIRS = P_L0 + P_L1 + P_L2;     // 39 pairs of blocks per wave per net
constexpr int PAIR_BYTES = 2048, STAGE_BYTES = WAVES * PAIR_BYTES;                     // one pair of every wave = 16 KiB
constexpr int RS = 4;                                                                 // private ring slots (pairs)
// LDS map (bytes)
constexpr int XB_OFF = 0;                                   // [32 k-blocks][2 row tiles][1 KiB]  h1 fragments; partial sums alias it
constexpr int RING_OFF = XB_OFF + (H / 16) * 2 * 1024;      // [8 waves][RS][2 KiB]
constexpr int XF_OFF = RING_OFF + WAVES * RS * PAIR_BYTES;  // [3 k-blocks][2 row tiles][1 KiB]  masked flow variable, bf16 B fragments
constexpr int CW_OFF = XF_OFF + 6 * 1024;                   // [8 waves][1536]: conditioning (row tile, layer) x 64 units f32 | 2 x 64 bias2
constexpr int CW_BYTES = 1536;
constexpr int MASK_OFF = CW_OFF + WAVES * CW_BYTES;         // 64 x u64
constexpr int RED_OFF = MASK_OFF + 512;                     // [2 quantities][3 k-blocks][2 row tiles][32 rows] f32
constexpr int LDS_BYTES = RED_OFF + 2 * 3 * 2 * 32 * 4;
static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");

typedef __attribute__((ext_vector_type(8))) __bf16 bf8;
typedef unsigned u4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned pk(float a, float b) { return (unsigned)f32_to_bf16(a) | ((unsigned)f32_to_bf16(b) << 16); }
__device__ __forceinline__ u4 pack8(const v16f &a, int r0) {
    u4 r;
    r[0] = pk(a[r0], a[r0 + 1]); r[1] = pk(a[r0 + 2], a[r0 + 3]); r[2] = pk(a[r0 + 4], a[r0 + 5]); r[3] = pk(a[r0 + 6], a[r0 + 7]);
    return r;
}
__device__ __forceinline__ float leaky(float v) { return fmaxf(v, 0.01f * v); }

#define NS_STR_(x) #x
#define NS_STR(x) NS_STR_(x)
// complete read groups (issue + lgkmcnt(0) inside one statement, early-clobber outputs: see flow_bf16.hip)
#define NS_R2(A, B, ADDR, O0, O1)                                                                                         \
    asm volatile("ds_read_b128 %0, %2 offset:" NS_STR(O0) "\n\tds_read_b128 %1, %2 offset:" NS_STR(O1) "\n\ts_waitcnt lgkmcnt(0)" \
                 : "=&v"(A), "=&v"(B) : "v"(ADDR) : "memory")
#define NS_R4(A, B, C_, D, ADDR, O0, O1, O2, O3)                                                                          \
    asm volatile("ds_read_b128 %0, %4 offset:" NS_STR(O0) "\n\tds_read_b128 %1, %4 offset:" NS_STR(O1) "\n\t"               \
                 "ds_read_b128 %2, %4 offset:" NS_STR(O2) "\n\tds_read_b128 %3, %4 offset:" NS_STR(O3) "\n\ts_waitcnt lgkmcnt(0)" \
                 : "=&v"(A), "=&v"(B), "=&v"(C_), "=&v"(D) : "v"(ADDR) : "memory")
// two ring slots (2 A blocks each) + 4 B fragments
#define NS_R8(A00, A01, A10, A11, B00, B01, B10, B11, RA0, RA1, XB)                                                        \
    asm volatile("ds_read_b128 %0, %8\n\tds_read_b128 %1, %8 offset:1024\n\tds_read_b128 %2, %9\n\tds_read_b128 %3, %9 offset:1024\n\t" \
                 "ds_read_b128 %4, %10\n\tds_read_b128 %5, %10 offset:1024\n\tds_read_b128 %6, %10 offset:2048\n\t"        \
                 "ds_read_b128 %7, %10 offset:3072\n\ts_waitcnt lgkmcnt(0)"                                               \
                 : "=&v"(A00), "=&v"(A01), "=&v"(A10), "=&v"(A11), "=&v"(B00), "=&v"(B01), "=&v"(B10), "=&v"(B11)          \
                 : "v"(RA0), "v"(RA1), "v"(XB) : "memory")
#define NS_W(ADDR, V, OFF) asm volatile("ds_write_b128 %0, %1 offset:" NS_STR(OFF) :: "v"(ADDR), "v"(V) : "memory")
#define NS_LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#define NS_BARRIER() do { NS_LGKM0(); __builtin_amdgcn_s_barrier(); } while (0)

#define MFMA32(w, b, acc)                                                                                 \
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8, (w)), __builtin_bit_cast(bf8, (b)), (acc), 0, 0, 0)

__device__ __forceinline__ void u4_to_acc(v16f &a, int r0, const u4 &c) {
#pragma unroll
    for (int k = 0; k < 4; ++k) a[r0 + k] = __uint_as_float(c[k]);
}

// UNI: every 32-row tile lies inside one image (N % 32 == 0): the conditioning rows are wave-uniform and travel by DMA
// EMIT (train step): the hidden activations of every net (bf16, [net][row][512], after the leaky-ReLU) and the s / t
// pre-activations (f32, [net][row][64]) are also written out - the reverse pass reads them instead of re-evaluating the nets
template <bool UNI, bool EMIT>
__global__ __launch_bounds__(512) void couplings_ns_kernel(
    const float *__restrict__ in, float *__restrict__ out, const float *__restrict__ cond,
    const unsigned char *__restrict__ wstream, const float *__restrict__ bias2, const float *__restrict__ mask,
    float *__restrict__ sum_s_o, float *__restrict__ logp_o, int R, int B, int dim, int ncoup, int inverse,
    unsigned short *__restrict__ h1e, unsigned short *__restrict__ h2e, float *__restrict__ oe) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, w = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int h = lane >> 5, m = lane & 31;
    const int N = R / B, nnets = 2 * ncoup;
    const unsigned lds_base = (unsigned)(size_t)((__attribute__((address_space(3))) unsigned char *)lds);
    const unsigned l16 = lds_base + lane * 16;

    // rows of the two 32-row tiles (image-major tiling of the sample-major rows r = n*B + b)
    int bimg[2], erow[2];
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        const int g = blockIdx.x * ROWS + a * 32 + m, gc = g < R ? g : R - 1;
        bimg[a] = gc / N;
        erow[a] = g < R ? (gc % N) * B + bimg[a] : -1;
    }
    // one hidden tile of this lane's row -> 4 x 8 bytes of the row-major bf16 activations (units 32T + 8g + 4h + {0..3})
    auto emit_tile = [&](unsigned short *base, int net, int a, int t, const u4 &f0, const u4 &f1) {
        if (erow[a] >= 0) {
            unsigned short *p = base + ((size_t)net * R + erow[a]) * H + 64 * w + 32 * t + 4 * h;
            typedef unsigned u2 __attribute__((ext_vector_type(2)));
            u2 d0, d1, d2, d3;
            d0[0] = f0[0]; d0[1] = f0[1]; d1[0] = f0[2]; d1[1] = f0[3]; d2[0] = f1[0]; d2[1] = f1[1]; d3[0] = f1[2]; d3[1] = f1[3];
            asm volatile("global_store_dwordx2 %0, %1, off\n\tglobal_store_dwordx2 %0, %2, off offset:16\n\t"
                         "global_store_dwordx2 %0, %3, off offset:32\n\tglobal_store_dwordx2 %0, %4, off offset:48"
                         :: "v"(p), "v"(d0), "v"(d1), "v"(d2), "v"(d3) : "memory");
        }
    };
    const int b_uni0 = __builtin_amdgcn_readfirstlane(bimg[0]), b_uni1 = __builtin_amdgcn_readfirstlane(bimg[1]);

    // ---- owner waves: wave o < 6 holds the flow variable of row tile o/3, dims 16*(o%3) + (j&3) + 8(j>>2) + 4h, j < 8
    const bool owner = w < 6;
    const int ort = w >= 3 ? 1 : 0, okb = w - 3 * ort;
    const int og = blockIdx.x * ROWS + ort * 32 + m;
    const bool ovalid = owner && og < R;
    const int ogc = og < R ? og : R - 1;
    const int orow = (ogc % N) * B + ogc / N;
    auto odim = [&](int j) { return 16 * okb + (j & 3) + 8 * (j >> 2) + 4 * h; };
    float x[8], sv[8];
    float sq_in = 0.f, sum_s = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int d = odim(j);
        const float v = in[(size_t)orow * dim + (d < dim ? d : dim - 1)];
        x[j] = (owner && d < dim) ? v : 0.f;
        sv[j] = 0.f;
        sq_in = fmaf(x[j], x[j], sq_in);
    }
    // pass-through masks as 64-bit words (bit d = mask[ci][d] != 0; padded dims pass through), before any DMA is in flight
    {
        unsigned long long *mask_w = reinterpret_cast<unsigned long long *>(lds + MASK_OFF);
        for (int ci = w; ci < ncoup; ci += WAVES) {
            const float mv = lane < dim ? mask[ci * dim + lane] : 1.f;
            const unsigned long long bits = __ballot(mv != 0.f);
            if (lane == 0) mask_w[ci] = bits;
        }
    }
    __syncthreads();
    auto mask_word = [&](int ci) -> unsigned long long {
        unsigned long long mw;
        const unsigned a = lds_base + MASK_OFF + ci * 8;
        asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(mw) : "v"(a) : "memory");
        return mw;
    };
    auto publish_x = [&](int ci) {                  // masked bf16 B fragment of the owner's 8 dims for coupling ci's nets
        const unsigned long long mw = mask_word(ci);
        v16f xm;
#pragma unroll
        for (int j = 0; j < 8; ++j) xm[j] = ((mw >> odim(j)) & 1ull) ? x[j] : 0.f;
        const u4 f = pack8(xm, 0);
        const unsigned a = l16 + XF_OFF + (okb * 2 + ort) * 1024;
        NS_W(a, f, 0);
    };
    const int ci0 = inverse ? ncoup - 1 : 0;
    if (owner) publish_x(ci0);

    // ---- private weight ring
    auto net_of = [&](int seq) { const int step = seq >> 1; return 2 * (inverse ? ncoup - 1 - step : step) + (seq & 1); };
    int iss_q = 0, iss_seq = 0, iss_p = 0;          // next pair to issue: global index, net sequence number, pair inside the net
    auto issue = [&]() {
        const int seqc = iss_seq < nnets ? iss_seq : nnets - 1;      // past the end: re-fetch into a free slot (branch-free)
        const unsigned char *src = wstream + ((size_t)net_of(seqc) * NET_PAIRS + iss_p) * STAGE_BYTES + w * PAIR_BYTES + lane * 16;
        unsigned char *dst = lds + RING_OFF + (w * RS + (iss_q & (RS - 1))) * PAIR_BYTES;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                         (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + 1024),
                                         (__attribute__((address_space(3))) void *)(dst + 1024), 16, 0, 0);
        ++iss_q;
        if (++iss_p == NET_PAIRS) { iss_p = 0; ++iss_seq; }
    };
    static_assert((RS & (RS - 1)) == 0, "ring slots: power of two");
    // conditioning of net `seq` for this wave's 64 units: lanes 16s..16s+15 carry segment s = 2*row tile + layer (256 B each);
    // bias2 of the net (64 floats) into one of two slots
    auto cond_issue = [&](int seq) {
        if (seq < nnets) {
            const int net = net_of(seq);
            unsigned char *dst = lds + CW_OFF + w * CW_BYTES;
            if constexpr (UNI) {
                const int seg = lane >> 4, bi = (seg >> 1) ? b_uni1 : b_uni0;
                const float *src = cond + ((size_t)(bi * nnets + net) * 2 + (seg & 1)) * H + 64 * w + (lane & 15) * 4;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                                 (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
            }
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(bias2 + net * 64 + lane),
                                             (__attribute__((address_space(3))) void *)(dst + 1024 + (seq & 1) * 256), 4, 0, 0);
        }
    };
    cond_issue(0);
#pragma unroll
    for (int i = 0; i < RS; ++i) issue();
    int q = 0;                                       // next pair to consume
    const unsigned ring_l = l16 + RING_OFF + w * RS * PAIR_BYTES;
    auto slot_addr = [&](int qq) { return ring_l + ((qq & (RS - 1)) * PAIR_BYTES); };
    NS_BARRIER();                                    // x fragments of the first coupling are published

    // accumulator tile <- conditioning of units 32t.. of layer `layer` for row tile a
    auto init_tile = [&](v16f &acc, int a, int layer, int t, int net) {
        if constexpr (UNI) {
            u4 c0, c1, c2, c3;
            const unsigned ca = lds_base + CW_OFF + w * CW_BYTES + (a * 2 + layer) * 256 + t * 128 + 16 * h;
            NS_R4(c0, c1, c2, c3, ca, 0, 32, 64, 96);
            u4_to_acc(acc, 0, c0); u4_to_acc(acc, 4, c1); u4_to_acc(acc, 8, c2); u4_to_acc(acc, 12, c3);
        } else {
            const float *p = cond + ((size_t)(bimg[a] * nnets + net) * 2 + layer) * H + 64 * w + 32 * t + 4 * h;
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const float4 c4 = *reinterpret_cast<const float4 *>(p + 8 * gq);
                acc[4 * gq] = c4.x; acc[4 * gq + 1] = c4.y; acc[4 * gq + 2] = c4.z; acc[4 * gq + 3] = c4.w;
            }
        }
    };

    for (int seq = 0; seq < nnets; ++seq) {
        const int net = net_of(seq), ci = net >> 1, netk = seq & 1;
        // ================= layer 0: h1[own 64 units][64 rows] = W0 . xm + c0
        v16f a1[2][2];                               // [unit tile][row tile]
        asm volatile("s_waitcnt vmcnt(2)" ::: "memory");            // 3 of the RS = 4 pairs in flight are needed (and, older, the conditioning)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int a = 0; a < 2; ++a) init_tile(a1[t][a], a, 0, t, net);
        {
            const unsigned xf = l16 + XF_OFF;
#pragma unroll
            for (int kb = 0; kb < P_L0; ++kb) {
                u4 A0, A1, B0, B1;
                const unsigned ra = slot_addr(q + kb);
                NS_R2(A0, A1, ra, 0, 1024);
                if (kb == 0) NS_R2(B0, B1, xf, 0, 1024);
                else if (kb == 1) NS_R2(B0, B1, xf, 2048, 3072);
                else NS_R2(B0, B1, xf, 4096, 5120);
                MFMA32(A0, B0, a1[0][0]); MFMA32(A0, B1, a1[0][1]);
                MFMA32(A1, B0, a1[1][0]); MFMA32(A1, B1, a1[1][1]);
            }
            q += P_L0;
            issue(); issue(); issue();
        }
        // leaky -> bf16 B fragments of k-blocks 4w + 2t + half -> exchange buffer
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int a = 0; a < 2; ++a) {
#pragma unroll
                for (int i = 0; i < 16; ++i) a1[t][a][i] = leaky(a1[t][a][i]);
                const u4 f0 = pack8(a1[t][a], 0), f1 = pack8(a1[t][a], 8);
                const unsigned xa = l16 + XB_OFF + ((4 * w + 2 * t) * 2 + a) * 1024;
                NS_W(xa, f0, 0);
                NS_W(xa, f1, 2048);
                if constexpr (EMIT) emit_tile(h1e, net, a, t, f0, f1);
            }
        // ================= layer 1: h2[own 64 units][64 rows] = W1 . h1 + c1
        v16f a2[2][2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int a = 0; a < 2; ++a) init_tile(a2[t][a], a, 1, t, net);
        cond_issue(seq + 1);                          // this wave has read both layers' conditioning of net `seq`
        NS_BARRIER();                                 // (1) h1 complete
#pragma unroll 1
        for (int k = 0; k < P_L1; k += 2) {
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");        // 2 of the 4 pairs in flight are needed
            u4 A00, A01, A10, A11, B00, B01, B10, B11;
            const unsigned ra0 = slot_addr(q), ra1 = slot_addr(q + 1), xb = l16 + XB_OFF + k * 2048;
            NS_R8(A00, A01, A10, A11, B00, B01, B10, B11, ra0, ra1, xb);
            q += 2;
            issue(); issue();
            MFMA32(A00, B00, a2[0][0]); MFMA32(A00, B01, a2[0][1]);
            MFMA32(A01, B00, a2[1][0]); MFMA32(A01, B01, a2[1][1]);
            MFMA32(A10, B10, a2[0][0]); MFMA32(A10, B11, a2[0][1]);
            MFMA32(A11, B10, a2[1][0]); MFMA32(A11, B11, a2[1][1]);
        }
        // ================= layer 2, split over K: partial o[dim tile][row tile] over this wave's 4 k-blocks
        u4 hf[2][2][2];                               // [unit tile][half][row tile]
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int a = 0; a < 2; ++a) {
#pragma unroll
                for (int i = 0; i < 16; ++i) a2[t][a][i] = leaky(a2[t][a][i]);
                hf[t][0][a] = pack8(a2[t][a], 0);
                hf[t][1][a] = pack8(a2[t][a], 8);
                if constexpr (EMIT) emit_tile(h2e, net, a, t, hf[t][0][a], hf[t][1][a]);
            }
        v16f o[2][2];                                 // [dim tile][row tile]
#pragma unroll
        for (int T = 0; T < 2; ++T)
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int i = 0; i < 16; ++i) o[T][a][i] = 0.f;
#pragma unroll
        for (int t = 0; t < 2; ++t) {                 // pairs kl = 2t (half 0), 2t+1 (half 1): blocks (T=0, kb=4w+kl), (T=1, kb=4w+kl)
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            u4 A00, A01, A10, A11;
            const unsigned ra0 = slot_addr(q), ra1 = slot_addr(q + 1);
            NS_R2(A00, A01, ra0, 0, 1024);
            NS_R2(A10, A11, ra1, 0, 1024);
            q += 2;
            issue(); issue();
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                MFMA32(A00, hf[t][0][a], o[0][a]); MFMA32(A01, hf[t][0][a], o[1][a]);
                MFMA32(A10, hf[t][1][a], o[0][a]); MFMA32(A11, hf[t][1][a], o[1][a]);
            }
        }
        NS_BARRIER();                                 // (2) every wave is through layer 1: the exchange buffer is free
        // ================= combine the 8 partial sums: row tile 0 by waves 0-2, row tile 1 by waves 3-5
        float acc8[8];
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            // partial of row tile a: P[kb 3][src wave 8][4-register group 2][lane][16 B]
#pragma unroll
            for (int kb = 0; kb < 3; ++kb) {
                const int T = kb >> 1, r0 = 8 * (kb & 1);
                u4 p0, p1;
#pragma unroll
                // "+ 0.f" is a real v_add: the inline-asm stores below must not read MFMA results directly (the wait states between an
                // MFMA and an LDS instruction reading its result are software-managed, and the compiler does not look inside asm)
                for (int k = 0; k < 4; ++k) { p0[k] = __float_as_uint(o[T][a][r0 + k] + 0.f); p1[k] = __float_as_uint(o[T][a][r0 + 4 + k] + 0.f); }
                const unsigned pa = l16 + XB_OFF + ((kb * 8 + w) * 2) * 1024;
                NS_W(pa, p0, 0);
                NS_W(pa, p1, 1024);
            }
            NS_BARRIER();                             // (3)/(5) partials of row tile a written
            if (owner && ort == a) {
#pragma unroll
                for (int j = 0; j < 8; ++j) acc8[j] = 0.f;
#pragma unroll
                for (int s2 = 0; s2 < 8; s2 += 2) {   // fixed order: waves 0..7
                    u4 p0, p1, p2, p3;
                    const unsigned pa = l16 + XB_OFF + ((okb * 8 + s2) * 2) * 1024;
                    NS_R4(p0, p1, p2, p3, pa, 0, 1024, 2048, 3072);
#pragma unroll
                    for (int k = 0; k < 4; ++k) { acc8[k] += __uint_as_float(p0[k]); acc8[4 + k] += __uint_as_float(p1[k]); }
#pragma unroll
                    for (int k = 0; k < 4; ++k) { acc8[k] += __uint_as_float(p2[k]); acc8[4 + k] += __uint_as_float(p3[k]); }
                }
            }
            if (a == 0) NS_BARRIER();                 // (4) partials of row tile 0 consumed
        }
        if (owner) {
            // + l2.bias (dims 16*okb + 4h + {0..3} and + 8)
            u4 b0, b1;
            const unsigned ba = lds_base + CW_OFF + w * CW_BYTES + 1024 + (seq & 1) * 256 + (16 * okb + 4 * h) * 4;
            NS_R2(b0, b1, ba, 0, 32);
            const unsigned long long mw = mask_word(ci);
            if constexpr (EMIT) {
                if (ovalid) {                           // pre-activation incl. bias, dims 16*okb + 4h + {0..3} and + 8
                    float *p = oe + ((size_t)net * R + orow) * 64 + 16 * okb + 4 * h;
                    v4f e0, e1;
#pragma unroll
                    for (int k = 0; k < 4; ++k) { e0[k] = acc8[k] + __uint_as_float(b0[k]); e1[k] = acc8[4 + k] + __uint_as_float(b1[k]); }
                    asm volatile("global_store_dwordx4 %0, %1, off\n\tglobal_store_dwordx4 %0, %2, off offset:32" :: "v"(p), "v"(e0), "v"(e1) : "memory");
                }
            }
            if (netk == 0) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const bool keep = (mw >> odim(j)) & 1ull;
                    const float bj = __uint_as_float(j < 4 ? b0[j & 3] : b1[j & 3]);
                    sv[j] = keep ? 0.f : tanhf(acc8[j] + bj);
                }
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const bool keep = (mw >> odim(j)) & 1ull;
                    const float tj = acc8[j] + __uint_as_float(j < 4 ? b0[j & 3] : b1[j & 3]);
                    if (!keep) {
                        if (!inverse) x[j] = x[j] * expf(sv[j]) + tj;          // flows.py:216
                        else          x[j] = (x[j] - tj) * expf(-sv[j]);       // flows.py:225
                        sum_s += sv[j];
                    }
                }
                if (seq + 1 < nnets) publish_x(net_of(seq + 1) >> 1);
            }
        }
        NS_BARRIER();                                 // (6) partials consumed, next coupling's x fragments published
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // no DMA may still target LDS when the workgroup retires

    // ---- outputs: rows by the owners, per-row scalars through a small LDS table
    float sq_out = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) sq_out = fmaf(x[j], x[j], sq_out);
    float base_sq = inverse ? sq_out : sq_in;
    base_sq += __shfl_xor(base_sq, 32, 64);
    sum_s += __shfl_xor(sum_s, 32, 64);
    if (ovalid) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { const int d = odim(j); if (d < dim) out[(size_t)orow * dim + d] = x[j]; }
    }
    float *red = reinterpret_cast<float *>(lds + RED_OFF);
    if (owner && h == 0) {
        red[(okb * 2 + ort) * 32 + m] = base_sq;
        red[192 + (okb * 2 + ort) * 32 + m] = sum_s;
    }
    __syncthreads();
    if (w < 2 && h == 0) {
        const int g = blockIdx.x * ROWS + w * 32 + m;
        if (g < R) {
            const int r = (g % N) * B + g / N;
            const float sq = red[(0 * 2 + w) * 32 + m] + red[(1 * 2 + w) * 32 + m] + red[(2 * 2 + w) * 32 + m];
            const float ss = red[192 + (0 * 2 + w) * 32 + m] + red[192 + (1 * 2 + w) * 32 + m] + red[192 + (2 * 2 + w) * 32 + m];
            if (sum_s_o) sum_s_o[r] = ss;
            if (logp_o) logp_o[r] = (-0.5f * sq - 0.5f * (float)dim * 1.8378770664093453f) - ss;
        }
    }
}

}}  // namespace mhe::flowns

namespace mhe { namespace flowb {
void pack_block_bf16(const float *W, int rows, int cols, int To, int kb, unsigned short *dst);
}}

namespace mhe { namespace flowns {

size_t packed_bytes_per_net() { return (size_t)NET_PAIRS * STAGE_BYTES; }

// stream of one net: [pair 39][wave 8][2 blocks of 1 KiB]
void pack_net_host(const float *W0, const float *W1, const float *W2, int dim, unsigned short *out) {
    memset(out, 0, packed_bytes_per_net());
    for (int p = 0; p < NET_PAIRS; ++p)
        for (int w = 0; w < WAVES; ++w) {
            unsigned short *d = out + ((size_t)p * WAVES + w) * (PAIR_BYTES / 2);
            for (int j = 0; j < 2; ++j, d += 512) {
                if (p < P_L0) flowb::pack_block_bf16(W0, H, dim, 2 * w + j, p, d);                        // (unit tile 2w+j, k-block p)
                else if (p < P_L0 + P_L1) flowb::pack_block_bf16(W1, H, H, 2 * w + j, p - P_L0, d);
                else flowb::pack_block_bf16(W2, dim, H, j, 4 * w + (p - P_L0 - P_L1), d);                 // (dim tile j, k-block 4w+kl)
            }
        }
}

int launch(const float *in, float *out, const float *cond, const void *wstream, const float *bias2, const float *mask,
           float *sum_s, float *log_prob, int R, int B, int dim, int ncoup, int inv, hipStream_t s,
           unsigned short *h1e, unsigned short *h2e, float *oe) {
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(couplings_ns_kernel<true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(couplings_ns_kernel<false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(couplings_ns_kernel<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(couplings_ns_kernel<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        attr_set = true;
    }
    const dim3 grid((R + ROWS - 1) / ROWS), block(512);
    const unsigned char *ws = reinterpret_cast<const unsigned char *>(wstream);
    const bool uni = ((R / B) % 32) == 0, emit = h1e != nullptr;
#define NS_LAUNCH(U, E)                                                                                                        \
    hipLaunchKernelGGL((couplings_ns_kernel<U, E>), grid, block, LDS_BYTES, s, in, out, cond, ws, bias2, mask, sum_s, log_prob, R, B, dim, \
                       ncoup, inv, h1e, h2e, oe)
    if (uni && emit) NS_LAUNCH(true, true);
    else if (uni) NS_LAUNCH(true, false);
    else if (emit) NS_LAUNCH(false, true);
    else NS_LAUNCH(false, false);
#undef NS_LAUNCH
    return check_launch("flowns::couplings_ns_kernel");
}

}}  // namespace mhe::flowns

// Weight gradients: dW[co][kh][kw][ci] += sum_{b,ho,wo} gy[b,ho,wo,co] * x[b, ho*s+kh-p, wo*s+kw-p, ci]
// for NHWC convolutions, and with H = W = KH = KW = 1 the weight gradient of a dense layer
// (dW[N][K] += gy[R][N]^T x[R][K]).  Stands where autograd's conv/addmm backward runs in the
// reference's train step (hand/CrossModalHand.py:455-470).
//
// GEMM view: M = Cout, N = KH*KW*Cin (tap-major, channel fastest = the forward's packed weight order),
// reduction over pixels.  Both operands are read as they lie in HBM ([pixel][channel], channel
// contiguous, 8/16-byte loads); a stage of BK = 16 pixels is staged in LDS as f32 [pixel][channel] rows,
// which IS the [k][m] / [k][n] image v_mfma_f32_16x16x4_f32 wants (lane = 16*k + m reads one dword,
// conflict-free with a 16-dword row pad).  f32 accumulate of f32 or bf16 operands; the pixel range is
// split over gridDim.z; the partial tiles go through workspace slabs + a fixed-order reducer (mhe_conv_wgrad_ws_nhwc: what the train step
// uses - bit-reproducible sums) or, without a workspace, are added to dW with f32 atomics (dW zeroed by the caller).
// Roofline: MFMA f32 (157 TFLOP/s); algorithmic bytes per launch = |x| + |gy| + 4|dW|.
#include "common.h"
#include <cstring>
#include "../../include/mhe.h"
#include <cstdlib>

namespace mhe { namespace wgrad {

struct Params {
    const void *x, *gy;
    float *dw;
    int B, H, W, Cin, Cout, KH, KW, stride, pad, Ho, Wo;
    int stride_w, pad_w;   // the width direction's own stride / left padding (= stride / pad except for mhe_conv_wgrad_rect_nhwc)
    int ldw;             // row pitch of dW (floats)
    int N;               // KH*KW*Cin
    long P;              // B*Ho*Wo
    int chunk;           // pixels per gridDim.z slice (multiple of BK)
    // partial-slab mode (ws != nullptr): slice z stores its tile into ws[z][Mp][Np] with plain coalesced stores and a reducer
    // pass adds the slabs to dW; otherwise the tiles are added to dW with f32 atomics (~1.3 TB/s chip-wide on gfx950: with
    // ~1000 slices of 64 KiB that was 10-35 % of a launch)
    float *ws;
    int Mp, Np;
    int direct;          // no workspace and ONE pixel slice: the tile is added to dW with plain read-modify-write stores (every element has exactly one
                         // writer - no atomics, no slab; the multi-problem launch's unsplit problems)
    // grouped launch (mhe_conv_wgrad_batched_nhwc, LDS-DMA kernel only): gridDim.z = nbatch * gz; problem b = blockIdx.z / gz reads
    // x + b * x_bs, gy + b * gy_bs (elements) and adds into dw + b * dw_bs (floats); slabs lie [b][slice][Mp][Np].  gz = 0: not grouped
    int gz;
    long x_bs, gy_bs, dw_bs;
};

constexpr int BK = 16;

template <typename T> struct Vec4;
template <> struct Vec4<float> {
    static __device__ __forceinline__ v4f load(const float *p) { return *reinterpret_cast<const v4f *>(p); }
};
template <> struct Vec4<u16> {
    static __device__ __forceinline__ v4f load(const u16 *p) {
        const uint2 r = *reinterpret_cast<const uint2 *>(p);
        v4f o;
        o[0] = __uint_as_float(r.x << 16); o[1] = __uint_as_float(r.x & 0xffff0000u);
        o[2] = __uint_as_float(r.y << 16); o[3] = __uint_as_float(r.y & 0xffff0000u);
        return o;
    }
};

// BM x BN output tile, 4 waves as 2 x 2, each wave (BM/2) x (BN/2) = TM x TN MFMA tiles of 16 x 16.
template <typename T, int BM, int BN>
__global__ __launch_bounds__(256) void wgrad_kernel(const Params p) {
    constexpr int LDA = BM + 16, LDB = BN + 16;
    constexpr int TM = BM / 32, TN = BN / 32;
    constexpr int A4 = BK * BM / 4 / 256, B4 = BK * BN / 4 / 256;       // vec4 loads per thread per stage
    static_assert(A4 >= 1 && B4 >= 1, "tile too small for 256 threads");
    __shared__ __attribute__((aligned(16))) float As[2][BK * LDA];
    __shared__ __attribute__((aligned(16))) float Bs[2][BK * LDB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const long k_begin = (long)blockIdx.z * p.chunk;
    const long k_end = k_begin + p.chunk < p.P ? k_begin + p.chunk : p.P;
    const T *x = reinterpret_cast<const T *>(p.x), *gy = reinterpret_cast<const T *>(p.gy);

    // per-thread fixed columns: A (gy) channel quad, B (x) column quad -> (tap, ci)
    int a_m[A4], a_k[A4], b_k[B4], b_ci[B4], b_dh[B4], b_dw[B4];
    bool a_ok[A4], b_ok[B4];
#pragma unroll
    for (int i = 0; i < A4; ++i) {
        const int f = tid + 256 * i;
        a_m[i] = (f % (BM / 4)) * 4; a_k[i] = f / (BM / 4);
        a_ok[i] = m0 + a_m[i] < p.Cout;
    }
#pragma unroll
    for (int i = 0; i < B4; ++i) {
        const int f = tid + 256 * i;
        const int n = n0 + (f % (BN / 4)) * 4;
        b_k[i] = f / (BN / 4);
        b_ok[i] = n < p.N;
        const int tap = b_ok[i] ? n / p.Cin : 0;
        b_ci[i] = b_ok[i] ? n % p.Cin : 0;
        b_dh[i] = tap / p.KW - p.pad; b_dw[i] = tap % p.KW - p.pad_w;
    }
    v4f ra[A4], rb[B4];
    auto fetch = [&](long k0) {
#pragma unroll
        for (int i = 0; i < A4; ++i) {
            const long pix = k0 + a_k[i];
            ra[i] = (a_ok[i] && pix < k_end) ? Vec4<T>::load(gy + pix * p.Cout + m0 + a_m[i]) : v4f{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int i = 0; i < B4; ++i) {
            const long pix = k0 + b_k[i];
            v4f v = {0.f, 0.f, 0.f, 0.f};
            if (b_ok[i] && pix < k_end) {
                const int wo = (int)(pix % p.Wo);
                const long t = pix / p.Wo;
                const int ho = (int)(t % p.Ho), b = (int)(t / p.Ho);
                const int hi = ho * p.stride + b_dh[i], wi = wo * p.stride_w + b_dw[i];
                if (hi >= 0 && hi < p.H && wi >= 0 && wi < p.W)
                    v = Vec4<T>::load(x + (((long)b * p.H + hi) * p.W + wi) * p.Cin + b_ci[i]);
            }
            rb[i] = v;
        }
    };
    auto stash = [&](int buf) {
#pragma unroll
        for (int i = 0; i < A4; ++i) *reinterpret_cast<v4f *>(&As[buf][a_k[i] * LDA + a_m[i]]) = ra[i];
#pragma unroll
        for (int i = 0; i < B4; ++i) *reinterpret_cast<v4f *>(&Bs[buf][b_k[i] * LDB + (tid + 256 * i) % (BN / 4) * 4]) = rb[i];
    };

    v4f acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = v4f{0.f, 0.f, 0.f, 0.f};

    const int fk = lane >> 4, fc = lane & 15;
    int buf = 0;
    if (k_begin < k_end) {
        fetch(k_begin);
        stash(0);
    }
    __syncthreads();
    for (long k0 = k_begin; k0 < k_end; k0 += BK) {
        const bool more = k0 + BK < k_end;
        if (more) fetch(k0 + BK);
        const float *a = As[buf] + wm * (BM / 2) + fc, *b = Bs[buf] + wn * (BN / 2) + fc;
#pragma unroll
        for (int kk = 0; kk < BK; kk += 4) {
            float fa[TM], fb[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[i] = a[(kk + fk) * LDA + 16 * i];
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[j] = b[(kk + fk) * LDB + 16 * j];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[i], fb[j], acc[i][j], 0, 0, 0);
        }
        if (more) stash(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }
    // D: column = lane & 15, rows 4*(lane >> 4) + r
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + wn * (BN / 2) + 16 * j + fc;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + wm * (BM / 2) + 16 * i + 4 * fk + r;
                if (p.ws) p.ws[((size_t)blockIdx.z * p.Mp + m) * p.Np + n] = acc[i][j][r];
                else if (m < p.Cout && n < p.N) atomicAdd(p.dw + (size_t)m * p.ldw + n, acc[i][j][r]);
            }
        }
}

// ---------------------------------------------------------------------------
// bf16 operands on v_mfma_f32_32x32x16_bf16.  The reduction index (pixels) is the ROW index of both operands
// as they lie in HBM, while the MFMA wants 8 consecutive k per lane: the stage is stored in LDS exactly as
// loaded ([pixel][channel] bf16 rows, 16-byte chunks) and read back with the gfx950 transposing LDS read
// ds_read_b64_tr_b16 (a 4-pixel x 16-channel block per 16-lane group, delivered channel-major), two reads per
// operand fragment.  Row pitch = tile bytes + 64: the four 64-byte row segments a 32-lane half touches land in
// four different 64-byte bank groups (pitch = 64 or 192 mod 256) -> conflict-free transposed reads.
// Measured on MI355X at ResNet-50's shapes (tools/wgrad_bench.py, B=256): 230-515 TFLOP/s; the f32-atomic epilogue is
// 10-35 % of a launch (main loop alone: 3x3 256ch@16x16 135 of 155 us); ~1024 workgroups with >= 512 pixels each is the
// best split (512 or 2048 workgroups: -15 %); a 4-deep register prefetch ring ran 20 % SLOWER (116 VGPRs: 4 -> 2
// waves/SIMD), so one stage of register prefetch under 4 co-resident workgroups per CU stays.
typedef short v4s __attribute__((ext_vector_type(4)));
typedef __attribute__((ext_vector_type(8))) __bf16 bf8;
typedef __attribute__((address_space(3))) v4s lds_v4s;

template <int BM, int BN, int WM, int WN>      // WM x WN waves, wave tile (BM/WM) x (BN/WN) of 32x32 MFMA tiles
__global__ __launch_bounds__(256) void wgrad_bf16_kernel(const Params p) {
    constexpr int BKB = 32;                                   // pixels per stage
    constexpr int LDA = BM * 2 + 64, LDB = BN * 2 + 64;       // row pitch in bytes
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr int CA = BM / 8, CB = BN / 8;                   // 16-byte chunks per row
    constexpr int A4 = BKB * CA / 256, B4 = BKB * CB / 256;
    static_assert(WM * WN == 4 && A4 >= 1 && B4 >= 1 && 256 % CA == 0 && 256 % CB == 0, "tile");
    static_assert((LDA % 256 == 64 || LDA % 256 == 192) && (LDB % 256 == 64 || LDB % 256 == 192), "pitch");
    __shared__ __attribute__((aligned(16))) char As[2][BKB * LDA];
    __shared__ __attribute__((aligned(16))) char Bs[2][BKB * LDB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const long k_begin = (long)blockIdx.z * p.chunk;
    const long k_end = k_begin + p.chunk < p.P ? k_begin + p.chunk : p.P;
    const u16 *x = reinterpret_cast<const u16 *>(p.x), *gy = reinterpret_cast<const u16 *>(p.gy);

    const int a_c = (tid % CA) * 8, a_r = tid / CA;            // this thread's chunk column / first row (rows step 256/CA)
    const int b_c = (tid % CB) * 8, b_r = tid / CB;
    const bool a_ok = m0 + a_c < p.Cout;
    const int nb = n0 + b_c;
    const bool b_ok = nb < p.N;
    const int tap = b_ok ? nb / p.Cin : 0, b_ci = b_ok ? nb % p.Cin : 0;
    const int b_dh = tap / p.KW - p.pad, b_dw = tap % p.KW - p.pad_w;
    uint4 ra[A4], rb[B4];
    auto fetch = [&](long k0) {
#pragma unroll
        for (int i = 0; i < A4; ++i) {
            const long pix = k0 + a_r + i * (256 / CA);
            ra[i] = (a_ok && pix < k_end) ? *reinterpret_cast<const uint4 *>(gy + pix * p.Cout + m0 + a_c) : make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < B4; ++i) {
            const long pix = k0 + b_r + i * (256 / CB);
            uint4 v = make_uint4(0, 0, 0, 0);
            if (b_ok && pix < k_end) {
                const int wo = (int)(pix % p.Wo);
                const long t = pix / p.Wo;
                const int ho = (int)(t % p.Ho), b = (int)(t / p.Ho);
                const int hi = ho * p.stride + b_dh, wi = wo * p.stride_w + b_dw;
                if (hi >= 0 && hi < p.H && wi >= 0 && wi < p.W)
                    v = *reinterpret_cast<const uint4 *>(x + (((long)b * p.H + hi) * p.W + wi) * p.Cin + b_ci);
            }
            rb[i] = v;
        }
    };
    auto stash = [&](int buf) {
#pragma unroll
        for (int i = 0; i < A4; ++i) *reinterpret_cast<uint4 *>(&As[buf][(a_r + i * (256 / CA)) * LDA + a_c * 2]) = ra[i];
#pragma unroll
        for (int i = 0; i < B4; ++i) *reinterpret_cast<uint4 *>(&Bs[buf][(b_r + i * (256 / CB)) * LDB + b_c * 2]) = rb[i];
    };
    v16f acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // transposed-read lane geometry: 16-lane group g = lane>>4 reads pixels 8*(g>>1) + {0..3 | 4..7}, channels 16*(g&1)..+15;
    // lane 4q+p of the group addresses row q, channel quad p
    const int g = lane >> 4, q = (lane >> 2) & 3, pq = lane & 3;
    const int fr = 8 * (g >> 1) + q, fcol = 16 * (g & 1) + 4 * pq;
    const int a_off = fr * LDA + (wm * (BM / WM) + fcol) * 2;
    const int b_off = fr * LDB + (wn * (BN / WN) + fcol) * 2;

    int buf = 0;
    if (k_begin < k_end) { fetch(k_begin); stash(0); }
    __syncthreads();
    for (long k0 = k_begin; k0 < k_end; k0 += BKB) {
        const bool more = k0 + BKB < k_end;
        if (more) fetch(k0 + BKB);
        const char *a = As[buf] + a_off, *b = Bs[buf] + b_off;
#pragma unroll
        for (int kk = 0; kk < BKB; kk += 16) {
            v8s fa[TM], fb[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s *)(a + kk * LDA + i * 64));
                const v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s *)(a + (kk + 4) * LDA + i * 64));
                fa[i] = v8s{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s *)(b + kk * LDB + j * 64));
                const v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s *)(b + (kk + 4) * LDB + j * 64));
                fb[j] = v8s{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8, fa[i]), __builtin_bit_cast(bf8, fb[j]), acc[i][j], 0, 0, 0);
        }
        if (more) stash(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }
    // D (32x32): column = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + wn * (BN / WN) + 32 * j + (lane & 31);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * (BM / WM) + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (p.ws) p.ws[((size_t)blockIdx.z * p.Mp + m) * p.Np + n] = acc[i][j][r];
                else if (m < p.Cout && n < p.N) atomicAdd(p.dw + (size_t)m * p.ldw + n, acc[i][j][r]);
            }
        }
}

// ---------------------------------------------------------------------------
// LDS-DMA form of the bf16 kernel.  The register-staged kernel above keeps ONE 32-pixel stage of global loads in flight per
// wave (PMC: 45 % of its wave time parked at s_waitcnt, MFMA utilisation 0.16): with three workgroups per CU that hides
// ~1/3 of a ~2,000-cycle load.  Here both operands go HBM/L2 -> LDS by buffer_load_dwordx4 ... lds into a 4-slot ring, three
// stages in flight behind a counted vmcnt, one raw s_barrier per stage, no VGPR staging and no ds_write pass:
//   * the stage image is the tensor's own [pixel][channel] rows WITHOUT the row pad (a DMA wave-instruction writes 1 KiB
//     lane-linearly, i.e. several whole rows); the bank-conflict remedy of the transposing reads moves into a 64-byte-segment
//     XOR applied to the per-lane SOURCE offset (256-byte rows: segment ^= row & 3; 128-byte rows: segment ^= (row >> 1) & 1),
//     so the four rows x 64 bytes a 32-lane half reads with ds_read_b64_tr_b16 land in four different bank groups;
//   * pixels past the slice end, channels / columns past the tensor and padding taps get an out-of-range offset: the DMA
//     writes zeros (tools/probes/lds_dma_oob.hip);
//   * the DMA and its wait are inline asm (hipcc would fence every LDS read behind a DMA it sees with vmcnt(0)); the
//     transposing reads and MFMAs stay compiler-scheduled;
//   * pixel -> (image, row, column) by multiply-high with host-made reciprocals instead of two integer divisions per load.
struct DmaParams {
    Params p;
    unsigned rcp_wo, rcp_ho;          // ceil(2^32 / Wo), ceil(2^32 / Ho): exact quotients for pixel counts < 2^32 / max(Wo, Ho)
    unsigned x_bytes, gy_bytes;
    int plain;                        // 1x1, stride 1, pad 0: the operand row is the pixel's own channel vector
    // XCD-aware order: the output tiles of ONE pixel slice read the same pixels (3x3: each of its column tiles another tap pair) - dispatched
    // in grid order they land on eight different XCDs and every L2 fetches the slice for itself (counters: 1,080 MB per launch for the
    // 268 MB of layer1's 3x3 layer).  xcd = 1: workgroup L (hardware order, XCD L % 8) takes tile (L / 8) % tiles of slice 8 (L / 8 / tiles)
    // + L % 8 - a slice's tiles run on one XCD, next to each other in time; nzt = slices that exist (gridDim.z is rounded up to 8)
    int xcd, nzt;
};
typedef unsigned u4v __attribute__((ext_vector_type(4)));

// <256, 256, 4, 2> (round 3): eight waves, a wave = 64 x 128 of the tile (TM = 2, TN = 4).  The 4-wave 128 x 128 tile reads (2 + 2) fragments
// per 4 MFMAs - every staged byte twice, 128 B/clk/CU of transposing reads at the full MFMA rate, which the LDS does not deliver next to
// the DMA writes (PMC: MFMA utilisation 0.20, 44 % issue-stalled); 64 x 128 per wave reads (2 + 4) per 8: 96 B/clk.  One workgroup per CU
// (128 KiB ring), two waves per SIMD as before.
// <256, 256, 4, 4> (round 5, the default for this tile): SIXTEEN waves of 64 x 64 (TM = TN = 2: 4 fragment reads per 4 MFMAs again), four per SIMD,
// 108 registers.  More LDS read traffic, and faster: train step 24.69 -> 24.35 ms (two A/B pairs), while FOUR waves of 128 x 128 (a third less
// read traffic, one wave per SIMD) lost 1.3 ms: what the loop lacks is something to issue while a wave sits at the stage's barrier or waits
// for its fragments, not LDS bandwidth (profiles/EXPERIMENTS.md "Round 5").  MHE_WGRAD_W16=0: the eight-wave form.
template <int BM, int BN, int WM, int WN> struct DmaTile {
    static constexpr int BKB = 32, NBUF = 4, DEPTH = 3, NW = WM * WN;
    static constexpr int PA = BM * 2, PB = BN * 2;                // row pitches in bytes (128, 256 or 512)
    static constexpr int STAGE = BKB * (PA + PB);
};

// the workgroup's work: output tile (bx, by) of pixel slice bzl of the problem dp; ring = NBUF * STAGE bytes of LDS, 1 KiB aligned
template <int BM, int BN, int WM, int WN>
__device__ __forceinline__ void wgrad_dma_body(const DmaParams &dp, const int bx, const int by, const int bzl, char *const ring) {
    const Params &p = dp.p;
    constexpr int BKB = 32, NBUF = 4, DEPTH = 3, NW = WM * WN;
    constexpr int PA = BM * 2, PB = BN * 2;                       // row pitches in bytes (128, 256 or 512)
    constexpr int STAGE = BKB * (PA + PB);
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr int RA = 1024 / PA, RB = 1024 / PB;                 // rows per DMA wave-instruction
    constexpr int IA = BKB / RA / NW, IB = BKB / RB / NW;         // DMA instructions per wave per stage
    static_assert((NW == 4 || NW == 8 || NW == 16) && (PA == 128 || PA == 256 || PA == 512) && (PB == 128 || PB == 256 || PB == 512) && IA >= 1 && IB >= 1, "tile");
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int m0 = by * BM, n0 = bx * BN;
    const int bz = p.gz ? bzl / p.gz : 0, zs = p.gz ? bzl - bz * p.gz : bzl;      // problem of a grouped launch, slice
    const long k_begin = (long)zs * p.chunk;
    const long k_end = k_begin + p.chunk < p.P ? k_begin + p.chunk : p.P;
    const unsigned OOB = 0x80000000u;
    // (512-byte rows sit a multiple of 256 bytes apart like 256-byte ones: same remedy, the XOR stays inside the row's 256-byte half)
    auto seg_swz = [](int pitch, int row) { return pitch >= 256 ? (row & 3) : ((row >> 1) & 1); };

    // ---- DMA role: per operand, instruction j of this wave fills rows R*(I*wave + j) .. +R-1 of the stage
    unsigned a_col[IA], b_col[IB];          // byte offset of this lane's 16 source bytes within a pixel's channel vector, or OOB
    int a_row[IA], b_row[IB], b_dh[IB], b_dw[IB];
#pragma unroll
    for (int j = 0; j < IA; ++j) {
        const int row = RA * (IA * wave + j) + lane / (PA / 16), pc = lane % (PA / 16);
        const int lc = (((pc >> 2) ^ seg_swz(PA, row)) << 2) | (pc & 3);
        a_row[j] = row;
        a_col[j] = m0 + lc * 8 < p.Cout ? (unsigned)(m0 + lc * 8) * 2u : OOB;
    }
#pragma unroll
    for (int j = 0; j < IB; ++j) {
        const int row = RB * (IB * wave + j) + lane / (PB / 16), pc = lane % (PB / 16);
        const int lc = (((pc >> 2) ^ seg_swz(PB, row)) << 2) | (pc & 3);
        const int n = n0 + lc * 8;
        b_row[j] = row;
        const int tap = n < p.N ? n / p.Cin : 0;
        b_col[j] = n < p.N ? (unsigned)(n - tap * p.Cin) * 2u : OOB;
        b_dh[j] = tap / p.KW - p.pad; b_dw[j] = tap % p.KW - p.pad_w;
    }
    const size_t gy_base = (size_t)p.gy + (size_t)bz * p.gy_bs * 2, x_base = (size_t)p.x + (size_t)bz * p.x_bs * 2;
    const u4v rs_g = {(unsigned)gy_base, (unsigned)(gy_base >> 32) & 0xffffu, dp.gy_bytes, 0x00020000u};
    const u4v rs_x = {(unsigned)x_base, (unsigned)(x_base >> 32) & 0xffffu, dp.x_bytes, 0x00020000u};
    const unsigned lds_base = (unsigned)(size_t)((__attribute__((address_space(3))) char *)ring);
    const long nst = (k_end - k_begin + BKB - 1) / BKB;

    auto dma = [&](unsigned voff, u4v rs, unsigned dst) {
        unsigned keep;
        asm volatile("s_nop 4\n\ts_mov_b32 %[keep], m0\n\ts_mov_b32 m0, %[dst]\n\ts_nop 0\n\tbuffer_load_dwordx4 %[v], %[rs], 0 offen lds\n\ts_mov_b32 m0, %[keep]"
                     : [keep] "=&s"(keep) : [v] "v"(voff), [rs] "s"(rs), [dst] "s"(dst) : "memory");
    };
    // stage st (may lie past the end: zero fill, keeps the counted wait uniform).  Branch-free: every offset is computed with
    // wrapping 32-bit arithmetic and replaced by the out-of-range value when the row is not to be loaded.
    const unsigned kend32 = (unsigned)k_end;
    auto issue = [&](long st) {
        const unsigned slot = lds_base + (unsigned)(st % NBUF) * STAGE;
        const unsigned k0 = (unsigned)(k_begin + st * BKB);
        const bool st_ok = st < nst;
#pragma unroll
        for (int j = 0; j < IA; ++j) {
            const unsigned pix = k0 + (unsigned)a_row[j];
            const bool ok = st_ok && pix < kend32 && a_col[j] != OOB;
            const unsigned v = pix * (unsigned)(p.Cout * 2) + a_col[j];
            dma(ok ? v : OOB, rs_g, slot + (unsigned)(IA * wave + j) * 1024u);
        }
#pragma unroll
        for (int j = 0; j < IB; ++j) {
            const unsigned pix = k0 + (unsigned)b_row[j];
            bool ok = st_ok && pix < kend32 && b_col[j] != OOB;
            unsigned v;
            if (dp.plain) v = pix * (unsigned)(p.Cin * 2) + b_col[j];
            else {
                const unsigned t = __umulhi(pix, dp.rcp_wo), wo = pix - t * (unsigned)p.Wo;
                const unsigned bi = __umulhi(t, dp.rcp_ho), ho = t - bi * (unsigned)p.Ho;
                const unsigned hi = ho * (unsigned)p.stride + (unsigned)b_dh[j], wi = wo * (unsigned)p.stride_w + (unsigned)b_dw[j];
                ok = ok && hi < (unsigned)p.H && wi < (unsigned)p.W;
                v = ((bi * (unsigned)p.H + hi) * (unsigned)p.W + wi) * (unsigned)(p.Cin * 2) + b_col[j];
            }
            dma(ok ? v : OOB, rs_x, slot + (unsigned)(BKB * PA) + (unsigned)(IB * wave + j) * 1024u);
        }
    };

    v16f acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // transposed-read lane geometry (as in wgrad_bf16_kernel): 16-lane group g reads pixels 8(g>>1) + q (+4), channels 16(g&1) + 4pq
    const int g = lane >> 4, q = (lane >> 2) & 3, pq = lane & 3;
    const int fr = 8 * (g >> 1) + q;                              // fr & 3 == q, (fr >> 1) & 1 == q >> 1 (the row adds 0/4/16 below)
    int a_off[TM], b_off[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int seg = (wm * (BM / WM) + 32 * i) / 32;
        a_off[i] = fr * PA + ((seg ^ seg_swz(PA, fr)) * 64) + (16 * (g & 1) + 4 * pq) * 2;
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int seg = (wn * (BN / WN) + 32 * j) / 32;
        b_off[j] = BKB * PA + fr * PB + ((seg ^ seg_swz(PB, fr)) * 64) + (16 * (g & 1) + 4 * pq) * 2;
    }

#pragma unroll
    for (int s0 = 0; s0 < DEPTH; ++s0) issue(s0);
    for (long st = 0; st < nst; ++st) {
        if constexpr (IA + IB == 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");        // (IA + IB) * (DEPTH - 1): stage st has landed
        else if constexpr (IA + IB == 3) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        issue(st + DEPTH);                                           // slot (st+3)%4 was last read in iteration st-1
        const char *base = ring + (st % NBUF) * STAGE;
#pragma unroll
        for (int kk = 0; kk < BKB; kk += 16) {
            v8s fa[TM], fb[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s *)(base + a_off[i] + kk * PA));
                const v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s *)(base + a_off[i] + (kk + 4) * PA));
                fa[i] = v8s{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s *)(base + b_off[j] + kk * PB));
                const v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s *)(base + b_off[j] + (kk + 4) * PB));
                fb[j] = v8s{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8, fa[i]), __builtin_bit_cast(bf8, fb[j]), acc[i][j], 0, 0, 0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the past-the-end zero fills land before the workgroup retires
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + wn * (BN / WN) + 32 * j + (lane & 31);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * (BM / WM) + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (p.ws) p.ws[((size_t)bzl * p.Mp + m) * p.Np + n] = acc[i][j][r];
                else if (m < p.Cout && n < p.N) {
                    float *dst = p.dw + (size_t)bz * p.dw_bs + (size_t)m * p.ldw + n;
                    if (p.direct) *dst += acc[i][j][r];
                    else atomicAdd(dst, acc[i][j][r]);
                }
            }
        }
}

template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(64 * WM * WN) void wgrad_dma_kernel(const DmaParams dp) {
    __shared__ __attribute__((aligned(1024))) char ring[DmaTile<BM, BN, WM, WN>::NBUF * DmaTile<BM, BN, WM, WN>::STAGE];
    int bx = blockIdx.x, by = blockIdx.y, bzl = blockIdx.z;
    if (dp.xcd) {
        const int S = gridDim.x * gridDim.y, L = bx + gridDim.x * (by + gridDim.y * bzl), j = L >> 3;
        const int sl = j / S, t = j - sl * S;
        bzl = sl * 8 + (L & 7); bx = t % (int)gridDim.x; by = t / (int)gridDim.x;
        if (bzl >= dp.nzt) return;
    }
    wgrad_dma_body<BM, BN, WM, WN>(dp, bx, by, bzl, ring);
}

// ---------------------------------------------------------------------------
// SEVERAL weight gradients of one tile shape in ONE launch (round 5).  A single layer's launch must cut its pixel range into as many slices as
// it takes to fill 256 CUs with ITS tiles - 28 to 64 slices for the 4 - 9 tiles of a layer3 layer - and every slice writes a full output tile
// of partial sums: slab bytes = workgroups x tile bytes, whatever the tile (PMC: 383 MB moved per launch for 208 MB of operands and gradient).
// The weight gradients of a gradient bucket do not depend on one another, so the train step queues them and launches them together: the
// chip is filled by the tiles of ALL queued layers, each layer's pixel range is cut only as far as a common slice length asks (layer4: not at
// all; layer3: 4 - 8 slices), and the slab traffic falls by that factor.  The problems ride in the kernel argument (up to 16 x 176 bytes; a
// captured HIP graph keeps them), workgroup L belongs to the problem whose range [first[i], first[i + 1]) holds it and takes that problem's
// tile / slice in the XCD-aware order of the single launch (every range is a multiple of 8 long and starts at a multiple of 8).
constexpr int MAXMULTI = 16;
struct MultiParams {
    DmaParams p[MAXMULTI];
    int first[MAXMULTI + 1];          // first workgroup of problem i (first[n] = grid size)
    int tx[MAXMULTI];                 // column tiles of problem i (tiles per slice = tx * ty; ty is not needed: the tile index is decoded with tx)
    int ts[MAXMULTI];                 // tiles per slice
    int n;
};

template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(64 * WM * WN) void wgrad_dma_multi_kernel(const MultiParams mp) {
    __shared__ __attribute__((aligned(1024))) char ring[DmaTile<BM, BN, WM, WN>::NBUF * DmaTile<BM, BN, WM, WN>::STAGE];
    const int L = blockIdx.x;
    int i = 0;
    while (i + 1 < mp.n && mp.first[i + 1] <= L) ++i;
    i = __builtin_amdgcn_readfirstlane(i);
    const int loc = L - mp.first[i], S = mp.ts[i];
    int bzl, t;
    if (mp.p[i].xcd) { const int j = loc >> 3, sl = j / S; t = j - sl * S; bzl = sl * 8 + (loc & 7); }
    else { bzl = loc / S; t = loc - bzl * S; }
    if (bzl >= mp.p[i].nzt) return;
    wgrad_dma_body<BM, BN, WM, WN>(mp.p[i], t % mp.tx[i], t / mp.tx[i], bzl, ring);
}

// the slab reducers of a multi-problem launch in one launch: dW[m][n .. n + 3] += sum_z ws[z][m][n .. n + 3] in slice order (fixed order: reproducible)
struct MultiReduce {
    const float *ws[MAXMULTI];
    float *dw[MAXMULTI];
    int gz[MAXMULTI], Mp[MAXMULTI], Np[MAXMULTI], M[MAXMULTI], N[MAXMULTI], ldw[MAXMULTI];
    int first[MAXMULTI + 1];          // first block of problem i
    int n;
};
__global__ __launch_bounds__(256) void slab_reduce_multi_kernel(const MultiReduce mr) {
    __shared__ v4f part[256];
    int i = 0;
    while (i + 1 < mr.n && mr.first[i + 1] <= (int)blockIdx.x) ++i;
    i = __builtin_amdgcn_readfirstlane(i);
    const int gz = mr.gz[i], Mp = mr.Mp[i], Np = mr.Np[i], n4 = mr.N[i] / 4;
    const int el = threadIdx.x & 63, zl = threadIdx.x >> 6;      // four slice lanes per float4, folded in lane order
    const long e = (long)(blockIdx.x - mr.first[i]) * 64 + el;
    const bool on = e < (long)mr.M[i] * n4;
    const int m = on ? (int)(e / n4) : 0, n = on ? (int)(e % n4) * 4 : 0;
    v4f a = {0.f, 0.f, 0.f, 0.f};
    const float *src = mr.ws[i] + (size_t)m * Np + n;
    if (on)
        for (int z = zl; z < gz; z += 4) {
            const v4f v = *reinterpret_cast<const v4f *>(src + (size_t)z * Mp * Np);
            a[0] += v[0]; a[1] += v[1]; a[2] += v[2]; a[3] += v[3];
        }
    part[threadIdx.x] = a;
    __syncthreads();
    if (zl || !on) return;
#pragma unroll
    for (int l = 1; l < 4; ++l) { const v4f v = part[l * 64 + el]; a[0] += v[0]; a[1] += v[1]; a[2] += v[2]; a[3] += v[3]; }
    float *dst = mr.dw[i] + (size_t)m * mr.ldw[i] + n;
    dst[0] += a[0]; dst[1] += a[1]; dst[2] += a[2]; dst[3] += a[3];
}

// dW[m][n..n+3] += sum_z ws[z][m][n..n+3]: the reducer of the partial-slab mode (one float4 per thread, slabs read coalesced).
// ZL slice lanes per float4 (blockDim = 64 * ZL): lane l sums slices l, l + ZL, ... in that order, the lanes are folded through LDS in
// lane order - a FIXED summation order for a given slice count (the train step's weight gradients are bit-reproducible), and the
// 100 - 500 slices of the small early-layer gradients (one or two output tiles, pixel range cut 512 ways) are walked 16 abreast
// instead of serially (those layers added their tiles with f32 atomics up to round 3: order-dependent sums).
template <int ZL>
__global__ __launch_bounds__(64 * ZL) void slab_reduce_kernel(const float *__restrict__ ws, float *__restrict__ dw, int gz, int Mp, int Np,
                                                              int M, int N, int ldw, long dw_bs) {
    __shared__ v4f part[ZL > 1 ? 64 * ZL : 1];
    ws += (size_t)blockIdx.y * gz * Mp * Np; dw += (size_t)blockIdx.y * dw_bs;          // (grouped launch: one problem per blockIdx.y)
    const int n4 = N / 4;
    const int el = threadIdx.x & 63, zl = threadIdx.x >> 6;
    const long i = (long)blockIdx.x * 64 + el;
    const bool on = i < (long)M * n4;
    const int m = on ? (int)(i / n4) : 0, n = on ? (int)(i % n4) * 4 : 0;
    v4f a = {0.f, 0.f, 0.f, 0.f};
    const float *src = ws + (size_t)m * Np + n;
    if (on) {
#pragma unroll 4
        for (int z = zl; z < gz; z += ZL) {
            const v4f v = *reinterpret_cast<const v4f *>(src + (size_t)z * Mp * Np);
            a[0] += v[0]; a[1] += v[1]; a[2] += v[2]; a[3] += v[3];
        }
    }
    if constexpr (ZL > 1) {
        part[threadIdx.x] = a;
        __syncthreads();
        if (zl) return;
#pragma unroll
        for (int l = 1; l < ZL; ++l) { const v4f v = part[l * 64 + el]; a[0] += v[0]; a[1] += v[1]; a[2] += v[2]; a[3] += v[3]; }
    }
    if (!on) return;
    float *dst = dw + (size_t)m * ldw + n;
    dst[0] += a[0]; dst[1] += a[1]; dst[2] += a[2]; dst[3] += a[3];
}

// out[c] += sum_r in[r][c]  (bias gradients).  A block owns a slab of rows and ALL of its 256 threads: thread t reads column
// t % CW of row-lane t / CW (CW = min(C, 256) columns per pass), the row-lanes are folded through LDS in lane order - for the 64-wide
// operands of the flow (C = 64) the one-thread-per-column form idled 3/4 of a block.  A FIXED summation order (f32 atomics from several
// row slabs into `out` up to round 3: order-dependent): either one block per column group adds its sum to `out`, or every row slab
// stores its sums as a row of `partial` and a second launch folds those rows.  out column c lands at out[(c / gw) * gs + c % gw]
// (gw = 0: out[c]).
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T *__restrict__ in, float *__restrict__ out, float *__restrict__ partial, long R, int C,
                                                     long rows_per_block, int gw, long gs, int) {
    __shared__ float part[256];
    const int CW = C < 256 ? C : 256, NRL = 256 / CW;           // C is a power of two below 256 or a multiple of 256 (checked by the launcher)
    const int cl = threadIdx.x % CW, rl = threadIdx.x / CW;
    const long r0 = (long)blockIdx.y * rows_per_block;
    const long r1 = r0 + rows_per_block < R ? r0 + rows_per_block : R;
    const int c = blockIdx.x * CW + cl;
    float acc = 0.f;
    if (rl < NRL && c < C) {
        long r = r0 + rl;
        for (; r + 3 * NRL < r1; r += 4 * NRL) {                 // four rows in flight
            float v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if constexpr (sizeof(T) == 4) v[u] = in[(r + u * NRL) * C + c];
                else v[u] = bf16_to_f32(in[(r + u * NRL) * C + c]);
            }
            acc += v[0]; acc += v[1]; acc += v[2]; acc += v[3];
        }
        for (; r < r1; r += NRL) {
            if constexpr (sizeof(T) == 4) acc += in[r * C + c];
            else acc += bf16_to_f32(in[r * C + c]);
        }
    }
    part[threadIdx.x] = acc;
    __syncthreads();
    if (rl == 0 && c < C) {
        for (int k = 1; k < NRL; ++k) acc += part[k * CW + cl];
        if (partial) partial[(size_t)blockIdx.y * C + c] = acc;
        else {
            float *dst = out + (gw ? (size_t)(c / gw) * gs + c % gw : (size_t)c);
            *dst += acc;                                         // the only block of this column
        }
    }
}

// dst[i] = (idx[i] < 0 ? 0 : src[idx[i]]) + (idx2 && idx2[i] >= 0 ? src[idx2[i]] : 0)  (f32 source; f32 or bf16 destination): every weight re-layout of the
// train step (forward packs, transposed / flipped dgrad operands, gradient un-packing) is one such gather
// over a table built once on the host.
template <typename TO>
__global__ __launch_bounds__(256) void gather_kernel(const float *__restrict__ src, const int *__restrict__ idx,
                                                     const int *__restrict__ idx2, TO *__restrict__ dst, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const int j = idx[i];
        float v = j < 0 ? 0.f : src[j];
        if (idx2) {
            const int j2 = idx2[i];
            if (j2 >= 0) v += src[j2];
        }
        if constexpr (sizeof(TO) == 4) dst[i] = v;
        else dst[i] = f32_to_bf16(v);
    }
}
// ... eight bf16 destinations per thread (one 16-byte store, two 16-byte index loads): the train step's bf16 operand arena is ~10^8 elements,
// and with 2-byte stores per lane the scalar form above ran at 2.5 TB/s of its index + destination bytes (0.44 ms per step)
__global__ __launch_bounds__(256) void gather8_bf16_kernel(const float *__restrict__ src, const int *__restrict__ idx, u16 *__restrict__ dst, size_t n8) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (size_t)gridDim.x * 256) {
        const int4 a = reinterpret_cast<const int4 *>(idx)[2 * i], b = reinterpret_cast<const int4 *>(idx)[2 * i + 1];
        const int j[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = j[k] < 0 ? 0.f : src[j[k]];
        uint4 o;
        o.x = (unsigned)f32_to_bf16(v[0]) | ((unsigned)f32_to_bf16(v[1]) << 16); o.y = (unsigned)f32_to_bf16(v[2]) | ((unsigned)f32_to_bf16(v[3]) << 16);
        o.z = (unsigned)f32_to_bf16(v[4]) | ((unsigned)f32_to_bf16(v[5]) << 16); o.w = (unsigned)f32_to_bf16(v[6]) | ((unsigned)f32_to_bf16(v[7]) << 16);
        reinterpret_cast<uint4 *>(dst)[i] = o;
    }
}
// ... the same eight destinations from ONE (base, stride) pair and a validity byte (round 5): the operand layouts are permutations of weight
// tensors with zero padding, so eight consecutive destination elements are almost always eight steps along one source dimension -
// src[base + k * stride] - and the 32 bytes of indices per group (472 MB per step for the 118 M bf16 elements, more than the 236 MB they
// produce) shrink to 9.  The host encodes an arena this way only if EVERY group of it is affine (mhentropy_amd/train.py: _affine8).
__global__ __launch_bounds__(256) void gather8_affine_bf16_kernel(const float *__restrict__ src, const int2 *__restrict__ bs, const unsigned char *__restrict__ mask,
                                                                  u16 *__restrict__ dst, size_t n8) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (size_t)gridDim.x * 256) {
        const int2 g = bs[i];
        const unsigned m = mask[i];
        float v[8];
        if (m == 0xffu && g.y == 1 && (g.x & 3) == 0) {
            const float4 a = *reinterpret_cast<const float4 *>(src + g.x), b = *reinterpret_cast<const float4 *>(src + g.x + 4);
            v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
        } else {
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = (m >> k) & 1u ? src[(long)g.x + (long)k * g.y] : 0.f;
        }
        uint4 o;
        o.x = (unsigned)f32_to_bf16(v[0]) | ((unsigned)f32_to_bf16(v[1]) << 16); o.y = (unsigned)f32_to_bf16(v[2]) | ((unsigned)f32_to_bf16(v[3]) << 16);
        o.z = (unsigned)f32_to_bf16(v[4]) | ((unsigned)f32_to_bf16(v[5]) << 16); o.w = (unsigned)f32_to_bf16(v[6]) | ((unsigned)f32_to_bf16(v[7]) << 16);
        reinterpret_cast<uint4 *>(dst)[i] = o;
    }
}
// ... and four f32 destinations per thread (the f32 arena: 11 M elements, a second index for a handful of them)
__global__ __launch_bounds__(256) void gather4_f32_kernel(const float *__restrict__ src, const int *__restrict__ idx, const int *__restrict__ idx2,
                                                          float *__restrict__ dst, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const int4 a = reinterpret_cast<const int4 *>(idx)[i];
        const int j[4] = {a.x, a.y, a.z, a.w};
        float v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = j[k] < 0 ? 0.f : src[j[k]];
        if (idx2) {
            const int4 b = reinterpret_cast<const int4 *>(idx2)[i];
            const int j2[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) if (j2[k] >= 0) v[k] += src[j2[k]];
        }
        reinterpret_cast<float4 *>(dst)[i] = make_float4(v[0], v[1], v[2], v[3]);
    }
}
}}  // namespace mhe::wgrad

using namespace mhe;

// geometry of a launch: tile, grid and pixel chunk (shared by the launcher and the workspace query)
struct WgradPlan { int BM, BN, gx, gy, gz; long chunk; bool bf16k, small, narrow, dma, big, xcd; };
static WgradPlan plan_wgrad(const mhe_conv_desc *d, int Ho_ = 0, int Wo_ = 0, int nbatch = 1) {
    WgradPlan w;
    const int Ho = Ho_ > 0 ? Ho_ : (d->H + 2 * d->pad - d->KH) / d->stride + 1, Wo = Wo_ > 0 ? Wo_ : (d->W + 2 * d->pad - d->KW) / d->stride + 1;
    const int N = d->KH * d->KW * d->Cin;
    const long P = (long)d->B * Ho * Wo;
    w.small = d->Cout <= 64;
    w.bf16k = d->dtype == MHE_BF16 && d->Cin % 8 == 0 && d->Cout % 8 == 0 && !getenv("MHE_WGRAD_F32MFMA");
    w.narrow = w.bf16k && N <= 64;          // 1x1 layers with 64 input channels: a 128-wide N tile would be half empty
    w.BM = w.small ? 64 : 128; w.BN = w.narrow ? 64 : 128;
    // LDS-DMA kernel: 32-bit byte offsets into x / gy, exact reciprocal quotients (pixel count < 2^32 / max(Wo, Ho))
    const size_t xb = (size_t)d->B * d->H * d->W * d->Cin * 2, gb = (size_t)P * d->Cout * 2;
    static const int dma_env = getenv("MHE_WGRAD_DMA") ? atoi(getenv("MHE_WGRAD_DMA")) : 1;
    w.dma = w.bf16k && dma_env && xb < 0x7fff0000ull && gb < 0x7fff0000ull && P < (long)(0xffffffffull / (Wo > Ho ? Wo : Ho)) && d->KH * d->KW <= 64;
    // 256 x 256 tile on eight waves (64 x 128 per wave) where both dimensions fill it: the layers from 256 output channels up
    static const int big_env = getenv("MHE_WGRAD_BIG") ? atoi(getenv("MHE_WGRAD_BIG")) : 1;
    // (measured, tools/wgrad_bench.py: -7 ... -26 % from 32k pixels up and on the 3x3 layers; the 16k-pixel 1x1 layers of layer4 and the
    // flow's 512 x 512 products, a handful of tiles with short pixel slices each, lose 2 - 14 % and stay on the 128 x 128 tile)
    w.big = w.dma && big_env && d->Cout % 256 == 0 && N >= 256 && (N % 256 == 0 || N >= 1024) && (P >= 32768 || N >= 2304 || nbatch > 1);
    if (w.big) w.BM = w.BN = 256;
    w.gx = (N + w.BN - 1) / w.BN; w.gy = (d->Cout + w.BM - 1) / w.BM;
    // split the pixel range: enough workgroups to fill 256 CUs a few times over; every split adds a full output tile of
    // partial sums - the bf16 kernel (4x faster mainloop) wants longer slices
    static const long target_wgs = getenv("MHE_WGRAD_WGS") ? atol(getenv("MHE_WGRAD_WGS")) : 512;      // 2 workgroups of the LDS-DMA kernel per CU: one resident wave of workgroups (measured 256-2048: 512 best)
    // (big tile: one workgroup per CU; a grouped launch - nbatch problems side by side - aims at three resident rounds of workgroups)
    long want = (nbatch > 1 ? (w.big ? 768 : 1536) : w.big ? 256 : w.bf16k ? target_wgs : 2048) / ((long)w.gx * w.gy * nbatch);
    if (want < 1) want = 1;
    // XCD-aware order (DmaParams::xcd): the launch rounds the slice count UP to a multiple of 8 with empty slices - aim at a multiple of 8 from
    // below, or the big tile's one-workgroup-per-CU launch grows a second round (28 slices x 9 tiles -> 32 x 9 = 288 workgroups: +65 %)
    // (the big tile keeps its slice count - and the grid order - where that is not a multiple of 8 already: 24 x 9 = 216 workgroups on 256 CUs
    // cost the 3x3 layers of 256 channels more than the shared L2 gave them)
    w.xcd = w.dma && (long)w.gx * w.gy > 1 && want >= 8 && (!w.big || want % 8 == 0);
    if (w.xcd) want = want / 8 * 8;
    long chunk = (P + want - 1) / want;
    static const long min_bf16 = getenv("MHE_WGRAD_MINCHUNK") ? atol(getenv("MHE_WGRAD_MINCHUNK")) : 512;
    const long min_chunk = w.bf16k ? min_bf16 : 64;
    if (chunk < min_chunk) chunk = min_chunk;
    w.chunk = (chunk + 31) / 32 * 32;
    w.gz = (int)((P + w.chunk - 1) / w.chunk);
    return w;
}

// Which launches go through partial slabs + the fixed-order reducer.  MHE_WGRAD_SLABS = "all" (default): every launch whose pixel range is
// split - the weight gradients do not depend on the order in which workgroups finish; "big": the round-3 policy (<= 64 slices and
// >= 128K weights; the rest adds its tiles with f32 atomics - order-dependent sums, kept for A/B timing).
static bool slabs_all() {
    static const bool all = !(getenv("MHE_WGRAD_SLABS") && !strcmp(getenv("MHE_WGRAD_SLABS"), "big"));
    return all;
}
static bool takes_slabs(const mhe_conv_desc *d, int gz) {
    if (gz <= 1) return false;
    if (slabs_all()) return gz <= 4096;
    return gz <= 64 && (size_t)d->Cout * d->KH * d->KW * d->Cin >= 131072;
}

// which instantiation a launch takes (bench.py names the kernel it times as rocprofv3 prints it): BM * 1000 + BN, + 1,000,000 for the
// LDS-DMA kernel, + 2,000,000 for the register-staged bf16 kernel (0 + ... : the f32 / generic kernel)
extern "C" int mhe_conv_wgrad_variant(const mhe_conv_desc *d, int Ho, int Wo, int nbatch) {
    if (!d || d->B <= 0 || d->H <= 0 || d->W <= 0 || d->Cin <= 0 || d->Cout <= 0 || d->KH <= 0 || d->KW <= 0 || d->stride <= 0 || nbatch < 1) return -1;
    const WgradPlan w = plan_wgrad(d, Ho, Wo, nbatch);
    return w.BM * 1000 + w.BN + (w.bf16k && w.dma ? 1000000 : w.bf16k ? 2000000 : 0);
}

extern "C" size_t mhe_conv_wgrad_workspace_floats(const mhe_conv_desc *d) {
    if (!d || d->B <= 0 || d->H <= 0 || d->W <= 0 || d->Cin <= 0 || d->Cout <= 0 || d->KH <= 0 || d->KW <= 0 || d->stride <= 0) return 0;
    const WgradPlan w = plan_wgrad(d);
    // measured in round 2 (tools/wgrad_bench.py, SLABS=0/1) with the serial reducer: slabs win 3-25 % from ~16 output tiles up (<= 64
    // slices each) and lost below that (many thin slices of a small dW) - the reducer now walks those 16 abreast
    return takes_slabs(d, w.gz) ? (size_t)w.gz * (w.gy * w.BM) * (size_t)(w.gx * w.BN) : 0;
}

static int wgrad_entry(const mhe_conv_desc *d, const void *x, const void *gy, float *dw, int ldw, float *ws, size_t ws_floats, void *stream,
                       int stride_w = 0, int pad_w = -1, int Ho_ = 0, int Wo_ = 0, int nbatch = 1, long x_bs = 0, long gy_bs = 0, long dw_bs = 0);

// nbatch independent weight gradients of one geometry in ONE launch (a grouped GEMM): problem b reads x + b * x_batch_stride and
// gy + b * gy_batch_stride (elements) and accumulates into dw + b * dw_batch_stride (floats).  The 24 coupling nets of the RealNVP reverse
// pass (hand/flows.py:105-122; three products per net over the same 16k hypothesis rows) ran as 72 launches of 4 - 16 output tiles each,
// 30 - 55 us apiece at 150 TF; grouped, each of the three shapes fills the chip.  bf16 operands (the LDS-DMA kernel) only.
extern "C" size_t mhe_conv_wgrad_batched_workspace_floats(const mhe_conv_desc *d, int nbatch) {
    if (!d || nbatch < 1 || d->B <= 0 || d->H <= 0 || d->W <= 0 || d->Cin <= 0 || d->Cout <= 0 || d->KH <= 0 || d->KW <= 0 || d->stride <= 0) return 0;
    const WgradPlan w = plan_wgrad(d, 0, 0, nbatch);
    return takes_slabs(d, w.gz) ? (size_t)nbatch * w.gz * (w.gy * w.BM) * (size_t)(w.gx * w.BN) : 0;
}

extern "C" int mhe_conv_wgrad_batched_nhwc(const mhe_conv_desc *d, int nbatch, const void *x, long x_batch_stride, const void *gy, long gy_batch_stride,
                                           float *dw, long dw_batch_stride, int ldw, float *workspace, size_t workspace_floats, void *stream) {
    MHE_REQUIRE(d && nbatch >= 1 && x_batch_stride >= 0 && gy_batch_stride >= 0 && dw_batch_stride >= 0, "mhe_conv_wgrad_batched_nhwc: bad arguments");
    MHE_REQUIRE(d->dtype == MHE_BF16 && d->Cin % 8 == 0 && d->Cout % 8 == 0, "mhe_conv_wgrad_batched_nhwc: bf16 operands with channel counts in multiples of 8");
    return wgrad_entry(d, x, gy, dw, ldw, workspace, workspace_floats, stream, 0, -1, 0, 0, nbatch, x_batch_stride, gy_batch_stride, dw_batch_stride);
}

extern "C" int mhe_conv_wgrad_nhwc(const mhe_conv_desc *d, const void *x, const void *gy, float *dw, int ldw, void *stream) {
    return wgrad_entry(d, x, gy, dw, ldw, nullptr, 0, stream);
}

extern "C" int mhe_conv_wgrad_ws_nhwc(const mhe_conv_desc *d, const void *x, const void *gy, float *dw, int ldw, float *workspace,
                                      size_t workspace_floats, void *stream) {
    return wgrad_entry(d, x, gy, dw, ldw, workspace, workspace_floats, stream);
}

// a convolution whose width direction has its own stride / left padding and whose output size is given (not derived): the stem's 7x7 /
// stride-2 weight gradient over PIXEL PAIRS - two neighbouring pixels x 4 padded channels = one 8-channel "pixel", kernel 7 x 4, stride
// (2, 1), padding (3, 2), output width W/2 exactly - which multiplies 224 columns instead of the 392 of 3 channels padded to 8
extern "C" size_t mhe_conv_wgrad_rect_workspace_floats(const mhe_conv_desc *d, int Ho, int Wo) {
    if (!d || d->B <= 0 || d->H <= 0 || d->W <= 0 || d->Cin <= 0 || d->Cout <= 0 || d->KH <= 0 || d->KW <= 0 || d->stride <= 0 || Ho <= 0 || Wo <= 0) return 0;
    const WgradPlan w = plan_wgrad(d, Ho, Wo);
    return takes_slabs(d, w.gz) ? (size_t)w.gz * (w.gy * w.BM) * (size_t)(w.gx * w.BN) : 0;
}

extern "C" int mhe_conv_wgrad_rect_nhwc(const mhe_conv_desc *d, int stride_w, int pad_w, int Ho, int Wo, const void *x, const void *gy, float *dw,
                                        int ldw, float *workspace, size_t workspace_floats, void *stream) {
    MHE_REQUIRE(stride_w > 0 && pad_w >= 0 && Ho > 0 && Wo > 0, "mhe_conv_wgrad_rect_nhwc: bad width geometry");
    MHE_REQUIRE(d && (Ho - 1) * d->stride - d->pad < d->H && (Wo - 1) * stride_w - pad_w < d->W, "mhe_conv_wgrad_rect_nhwc: output larger than the input allows");
    return wgrad_entry(d, x, gy, dw, ldw, workspace, workspace_floats, stream, stride_w, pad_w, Ho, Wo);
}

static int wgrad_entry(const mhe_conv_desc *d, const void *x, const void *gy, float *dw, int ldw, float *ws, size_t ws_floats, void *stream,
                       int stride_w, int pad_w, int Ho_, int Wo_, int nbatch, long x_bs, long gy_bs, long dw_bs) {
    MHE_REQUIRE(d && x && gy && dw, "mhe_conv_wgrad_nhwc: null pointer");
    MHE_REQUIRE(d->B > 0 && d->H > 0 && d->W > 0 && d->Cin > 0 && d->Cout > 0 && d->KH > 0 && d->KW > 0 && d->stride > 0 && d->pad >= 0,
                "mhe_conv_wgrad_nhwc: bad geometry");
    MHE_REQUIRE(d->Cin % 4 == 0 && d->Cout % 4 == 0, "mhe_conv_wgrad_nhwc: Cin=%d and Cout=%d must be multiples of 4", d->Cin, d->Cout);
    MHE_REQUIRE(d->dtype == MHE_F32 || d->dtype == MHE_BF16, "mhe_conv_wgrad_nhwc: dtype=%d", d->dtype);
    wgrad::Params p{};
    p.x = x; p.gy = gy; p.dw = dw;
    p.B = d->B; p.H = d->H; p.W = d->W; p.Cin = d->Cin; p.Cout = d->Cout; p.KH = d->KH; p.KW = d->KW;
    p.stride = d->stride; p.pad = d->pad;
    p.stride_w = stride_w > 0 ? stride_w : d->stride; p.pad_w = pad_w >= 0 ? pad_w : d->pad;
    p.Ho = Ho_ > 0 ? Ho_ : (d->H + 2 * d->pad - d->KH) / d->stride + 1;
    p.Wo = Wo_ > 0 ? Wo_ : (d->W + 2 * d->pad - d->KW) / d->stride + 1;
    MHE_REQUIRE(p.Ho > 0 && p.Wo > 0, "mhe_conv_wgrad_nhwc: empty output");
    p.N = d->KH * d->KW * d->Cin;
    p.ldw = ldw > 0 ? ldw : p.N;
    MHE_REQUIRE(p.ldw >= p.N, "mhe_conv_wgrad_nhwc: ldw=%d < KH*KW*Cin=%d", ldw, p.N);
    p.P = (long)d->B * p.Ho * p.Wo;
    const WgradPlan w = plan_wgrad(d, Ho_, Wo_, nbatch);
    const bool small = w.small, bf16k = w.bf16k, narrow = w.narrow;
    const int gx = w.gx, gyy = w.gy, gz = w.gz;
    p.chunk = (int)w.chunk;
    p.Mp = gyy * w.BM; p.Np = gx * w.BN;
    const size_t need = (size_t)nbatch * gz * p.Mp * p.Np;
    // a workspace was passed for a launch that takes slabs, but it is too small: fail (round 4 fell back to order-dependent f32 atomics without
    // a word - the bit-reproducibility the train step asserts would have been gone silently)
    MHE_REQUIRE(!(ws && takes_slabs(d, gz) && need > ws_floats), "mhe_conv_wgrad_ws_nhwc: workspace of %zu floats, this launch needs %zu (mhe_conv_wgrad_workspace_floats)",
                ws_floats, need);
    p.ws = (ws && takes_slabs(d, gz)) ? ws : nullptr;
    MHE_REQUIRE(nbatch == 1 || (w.dma && (long)gz * nbatch < 65536), "mhe_conv_wgrad_batched_nhwc: the grouped form runs on the LDS-DMA kernel (bf16, operands below 2 GiB)");
    if (nbatch > 1) { p.gz = gz; p.x_bs = x_bs; p.gy_bs = gy_bs; p.dw_bs = dw_bs; }
    dim3 grid(gx, gyy, gz * nbatch);
    const dim3 block(256);
    hipStream_t s = (hipStream_t)stream;
    const size_t xb = (size_t)d->B * d->H * d->W * d->Cin * 2, gb = (size_t)p.P * d->Cout * 2;
    const bool use_dma = w.dma;
    if (d->dtype == MHE_F32) {
        if (small) hipLaunchKernelGGL((wgrad::wgrad_kernel<float, 64, 128>), grid, block, 0, s, p);
        else hipLaunchKernelGGL((wgrad::wgrad_kernel<float, 128, 128>), grid, block, 0, s, p);
    } else if (bf16k && use_dma) {
        wgrad::DmaParams dp;
        dp.p = p;
        dp.rcp_wo = (unsigned)((0x100000000ull + p.Wo - 1) / p.Wo); dp.rcp_ho = (unsigned)((0x100000000ull + p.Ho - 1) / p.Ho);
        dp.x_bytes = (unsigned)xb; dp.gy_bytes = (unsigned)gb;
        dp.plain = d->KH == 1 && d->KW == 1 && d->stride == 1 && d->pad == 0 && p.stride_w == 1 && p.pad_w == 0;
        static const int xcd_env = getenv("MHE_WGRAD_XCD") ? atoi(getenv("MHE_WGRAD_XCD")) : 1;
        dp.nzt = gz * nbatch;
        dp.xcd = xcd_env && w.xcd && dp.nzt >= 8;
        if (dp.xcd) grid.z = (unsigned)((dp.nzt + 7) / 8 * 8);
        static const int w16 = getenv("MHE_WGRAD_W16") ? atoi(getenv("MHE_WGRAD_W16")) : 1;
        if (w.big && w16) hipLaunchKernelGGL((wgrad::wgrad_dma_kernel<256, 256, 4, 4>), grid, dim3(1024), 0, s, dp);
        else if (w.big) hipLaunchKernelGGL((wgrad::wgrad_dma_kernel<256, 256, 4, 2>), grid, dim3(512), 0, s, dp);
        else if (narrow) {
            if (small) hipLaunchKernelGGL((wgrad::wgrad_dma_kernel<64, 64, 2, 2>), grid, block, 0, s, dp);
            else hipLaunchKernelGGL((wgrad::wgrad_dma_kernel<128, 64, 4, 1>), grid, block, 0, s, dp);
        } else {
            if (small) hipLaunchKernelGGL((wgrad::wgrad_dma_kernel<64, 128, 1, 4>), grid, block, 0, s, dp);
            else hipLaunchKernelGGL((wgrad::wgrad_dma_kernel<128, 128, 2, 2>), grid, block, 0, s, dp);
        }
    } else if (narrow) {
        if (small) hipLaunchKernelGGL((wgrad::wgrad_bf16_kernel<64, 64, 2, 2>), grid, block, 0, s, p);
        else hipLaunchKernelGGL((wgrad::wgrad_bf16_kernel<128, 64, 4, 1>), grid, block, 0, s, p);
    } else if (bf16k) {
        if (small) hipLaunchKernelGGL((wgrad::wgrad_bf16_kernel<64, 128, 1, 4>), grid, block, 0, s, p);
        else hipLaunchKernelGGL((wgrad::wgrad_bf16_kernel<128, 128, 2, 2>), grid, block, 0, s, p);
    } else {
        if (small) hipLaunchKernelGGL((wgrad::wgrad_kernel<u16, 64, 128>), grid, block, 0, s, p);
        else hipLaunchKernelGGL((wgrad::wgrad_kernel<u16, 128, 128>), grid, block, 0, s, p);
    }
    if (int rc = check_launch("wgrad_kernel")) return rc;
    if (p.ws) {
        const long n = (long)d->Cout * (p.N / 4);
        const dim3 rg((unsigned)((n + 63) / 64), (unsigned)nbatch);
        if (gz > 64) hipLaunchKernelGGL(wgrad::slab_reduce_kernel<16>, rg, dim3(1024), 0, s, p.ws, dw, gz, p.Mp, p.Np, d->Cout, p.N, p.ldw, dw_bs);
        else if (gz > 8 || n < 65536) hipLaunchKernelGGL(wgrad::slab_reduce_kernel<4>, rg, dim3(256), 0, s, p.ws, dw, gz, p.Mp, p.Np, d->Cout, p.N, p.ldw, dw_bs);
        else hipLaunchKernelGGL(wgrad::slab_reduce_kernel<1>, rg, dim3(64), 0, s, p.ws, dw, gz, p.Mp, p.Np, d->Cout, p.N, p.ldw, dw_bs);
        return check_launch("slab_reduce_kernel");
    }
    return MHE_OK;
}

// ---- several weight gradients in one launch (see wgrad_dma_multi_kernel) ---------------------------------------------------------------
namespace {
struct MultiItemPlan { WgradPlan w; long P; int Ho, Wo, tiles; };
static long multi_target(bool big) {
    // measured at config C2 on one box (gpurun_out/r5, profiles/EXPERIMENTS.md "Round 5"): 512 / 1024 -> 25.85 ms per train step, 768 / 1024 -> 25.60,
    // 512 / 2048 -> 25.62, 768 / 2048 -> 25.30 (896 / 2048 -> 26.0: a fourth partial round of the one-workgroup-per-CU tile); one launch per layer: 26.0
    static const long tb = getenv("MHE_WGRAD_MULTI_WGS_BIG") ? atol(getenv("MHE_WGRAD_MULTI_WGS_BIG")) : 768;     // 256 x 256 tile: one workgroup per CU, three rounds
    static const long ts = getenv("MHE_WGRAD_MULTI_WGS") ? atol(getenv("MHE_WGRAD_MULTI_WGS")) : 2048;            // the 4-wave tiles: two per CU, four rounds
    return big ? tb : ts;
}
// common slice length of a batch: the shortest (multiple of 32, >= 512 pixels) at which the batch's workgroups - tiles x slices - fit the target
static long common_chunk(const MultiItemPlan *it, const int *idx, int n, long target) {
    long lo = 512, hi = 512;
    for (int k = 0; k < n; ++k) if (it[idx[k]].P > hi) hi = it[idx[k]].P;
    hi = (hi + 31) / 32 * 32;
    auto total = [&](long c) { long t = 0; for (int k = 0; k < n; ++k) t += (long)it[idx[k]].tiles * ((it[idx[k]].P + c - 1) / c); return t; };
    if (total(lo) <= target) return lo;
    while (lo + 32 < hi) {
        const long mid = (lo + hi) / 2 / 32 * 32;
        if (total(mid) <= target) hi = mid; else lo = mid;
    }
    return hi;
}
struct MultiLayout { int gz[wgrad::MAXMULTI]; long chunk[wgrad::MAXMULTI]; size_t ws_off[wgrad::MAXMULTI]; size_t ws_total; };
static void layout_batch(const mhe_wgrad_item *items, const MultiItemPlan *it, const int *idx, int n, bool big, MultiLayout &lo) {
    const long c = common_chunk(it, idx, n, multi_target(big));
    lo.ws_total = 0;
    for (int k = 0; k < n; ++k) {
        const MultiItemPlan &q = it[idx[k]];
        int gz = (int)((q.P + c - 1) / c);
        long chunk = ((q.P + gz - 1) / gz + 31) / 32 * 32;          // balanced slices of this problem
        gz = (int)((q.P + chunk - 1) / chunk);
        lo.gz[k] = gz; lo.chunk[k] = chunk; lo.ws_off[k] = lo.ws_total;
        if (gz > 1) lo.ws_total += (size_t)gz * (q.w.gy * q.w.BM) * (size_t)(q.w.gx * q.w.BN);
    }
}
static bool multi_plan(const mhe_wgrad_item *items, int n, MultiItemPlan *it) {
    for (int i = 0; i < n; ++i) {
        const mhe_conv_desc *d = &items[i].d;
        if (d->B <= 0 || d->H <= 0 || d->W <= 0 || d->Cin <= 0 || d->Cout <= 0 || d->KH <= 0 || d->KW <= 0 || d->stride <= 0 || d->pad < 0) return false;
        it[i].w = plan_wgrad(d);
        it[i].Ho = (d->H + 2 * d->pad - d->KH) / d->stride + 1; it[i].Wo = (d->W + 2 * d->pad - d->KW) / d->stride + 1;
        it[i].P = (long)d->B * it[i].Ho * it[i].Wo;
        it[i].tiles = it[i].w.gx * it[i].w.gy;
    }
    return true;
}
}  // namespace

extern "C" size_t mhe_conv_wgrad_multi_workspace_floats(const mhe_wgrad_item *items, int n) {
    if (!items || n <= 0 || n > 256) return 0;
    MultiItemPlan it[256];
    if (!multi_plan(items, n, it)) return 0;
    size_t need = 0;
    bool used[256] = {false};
    for (int i = 0; i < n; ++i) {
        if (used[i]) continue;
        if (!(it[i].w.bf16k && it[i].w.dma)) { used[i] = true; const size_t w = mhe_conv_wgrad_workspace_floats(&items[i].d); if (w > need) need = w; continue; }
        int idx[wgrad::MAXMULTI], m = 0;
        for (int j = i; j < n && m < wgrad::MAXMULTI; ++j)
            if (!used[j] && it[j].w.bf16k && it[j].w.dma && it[j].w.BM == it[i].w.BM && it[j].w.BN == it[i].w.BN) { idx[m++] = j; used[j] = true; }
        if (m == 1) { const size_t w = mhe_conv_wgrad_workspace_floats(&items[i].d); if (w > need) need = w; continue; }
        MultiLayout lo;
        layout_batch(items, it, idx, m, it[i].w.big, lo);
        if (lo.ws_total > need) need = lo.ws_total;
    }
    return need;
}

template <int BM, int BN, int WM, int WN>
static void launch_multi(const wgrad::MultiParams &mp, hipStream_t s) {
    hipLaunchKernelGGL((wgrad::wgrad_dma_multi_kernel<BM, BN, WM, WN>), dim3((unsigned)mp.first[mp.n]), dim3(64 * WM * WN), 0, s, mp);
}

extern "C" int mhe_conv_wgrad_multi_nhwc(const mhe_wgrad_item *items, int n, float *workspace, size_t workspace_floats, void *stream) {
    MHE_REQUIRE(items && n > 0 && n <= 256, "mhe_conv_wgrad_multi_nhwc: 1 .. 256 problems");
    MultiItemPlan it[256];
    MHE_REQUIRE(multi_plan(items, n, it), "mhe_conv_wgrad_multi_nhwc: bad geometry");
    for (int i = 0; i < n; ++i) MHE_REQUIRE(items[i].x && items[i].gy && items[i].dw, "mhe_conv_wgrad_multi_nhwc: null pointer (problem %d)", i);
    hipStream_t s = (hipStream_t)stream;
    bool used[256] = {false};
    for (int i = 0; i < n; ++i) {
        if (used[i]) continue;
        if (!(it[i].w.bf16k && it[i].w.dma)) {          // not a shape of the LDS-DMA kernel (f32 operands, odd channel counts): its own launch
            used[i] = true;
            if (int rc = wgrad_entry(&items[i].d, items[i].x, items[i].gy, items[i].dw, items[i].ldw, workspace, workspace_floats, stream)) return rc;
            continue;
        }
        int idx[wgrad::MAXMULTI], m = 0;
        for (int j = i; j < n && m < wgrad::MAXMULTI; ++j)
            if (!used[j] && it[j].w.bf16k && it[j].w.dma && it[j].w.BM == it[i].w.BM && it[j].w.BN == it[i].w.BN) { idx[m++] = j; used[j] = true; }
        if (m == 1) {            // alone in its tile class: the single launch's own plan (one resident round of workgroups, its reducer) is the better one
            if (int rc = wgrad_entry(&items[i].d, items[i].x, items[i].gy, items[i].dw, items[i].ldw, workspace, workspace_floats, stream)) return rc;
            continue;
        }
        MultiLayout lo;
        layout_batch(items, it, idx, m, it[i].w.big, lo);
        MHE_REQUIRE(lo.ws_total == 0 || (workspace && lo.ws_total <= workspace_floats), "mhe_conv_wgrad_multi_nhwc: workspace of %zu floats, this batch needs %zu",
                    workspace_floats, lo.ws_total);
        wgrad::MultiParams mp;
        wgrad::MultiReduce mr;
        memset(&mp, 0, sizeof(mp)); memset(&mr, 0, sizeof(mr));
        mp.n = m; mr.n = 0;
        int first = 0, rfirst = 0;
        for (int k = 0; k < m; ++k) {
            const mhe_wgrad_item &q = items[idx[k]];
            const MultiItemPlan &pl = it[idx[k]];
            const mhe_conv_desc *d = &q.d;
            wgrad::DmaParams &dp = mp.p[k];
            wgrad::Params &p = dp.p;
            p.x = q.x; p.gy = q.gy; p.dw = q.dw;
            p.B = d->B; p.H = d->H; p.W = d->W; p.Cin = d->Cin; p.Cout = d->Cout; p.KH = d->KH; p.KW = d->KW; p.stride = d->stride; p.pad = d->pad;
            p.stride_w = d->stride; p.pad_w = d->pad; p.Ho = pl.Ho; p.Wo = pl.Wo;
            p.N = d->KH * d->KW * d->Cin; p.ldw = q.ldw > 0 ? q.ldw : p.N;
            MHE_REQUIRE(p.ldw >= p.N, "mhe_conv_wgrad_multi_nhwc: ldw=%d < KH*KW*Cin=%d (problem %d)", q.ldw, p.N, idx[k]);
            p.P = pl.P; p.chunk = (int)lo.chunk[k];
            p.Mp = pl.w.gy * pl.w.BM; p.Np = pl.w.gx * pl.w.BN;
            const int gz = lo.gz[k];
            p.ws = gz > 1 ? workspace + lo.ws_off[k] : nullptr;
            p.direct = gz == 1;
            dp.rcp_wo = (unsigned)((0x100000000ull + p.Wo - 1) / p.Wo); dp.rcp_ho = (unsigned)((0x100000000ull + p.Ho - 1) / p.Ho);
            dp.x_bytes = (unsigned)((size_t)d->B * d->H * d->W * d->Cin * 2); dp.gy_bytes = (unsigned)((size_t)p.P * d->Cout * 2);
            dp.plain = d->KH == 1 && d->KW == 1 && d->stride == 1 && d->pad == 0;
            dp.nzt = gz;
            // a problem cut at least 8 ways takes the single launch's XCD-aware order (8 slices side by side, one per XCD, their tiles next to each
            // other in time: slot = 8 * (tile + tiles * (slice / 8)) + slice % 8); fewer slices: tile-major within the slice
            dp.xcd = gz >= 8;
            mp.tx[k] = pl.w.gx; mp.ts[k] = pl.tiles;
            mp.first[k] = first;
            const int wgs = dp.xcd ? 8 * pl.tiles * ((gz + 7) / 8) : (pl.tiles * gz + 7) / 8 * 8;
            first += wgs;
            if (gz > 1) {
                const int r = mr.n++;
                mr.ws[r] = p.ws; mr.dw[r] = q.dw; mr.gz[r] = gz; mr.Mp[r] = p.Mp; mr.Np[r] = p.Np; mr.M[r] = d->Cout; mr.N[r] = p.N; mr.ldw[r] = p.ldw;
                mr.first[r] = rfirst;
                rfirst += (int)(((long)d->Cout * (p.N / 4) + 63) / 64);
            }
        }
        mp.first[m] = first; mr.first[mr.n] = rfirst;
        const WgradPlan &w = it[i].w;
        static const int w16 = getenv("MHE_WGRAD_W16") ? atoi(getenv("MHE_WGRAD_W16")) : 1;
        if (w.big && w16) launch_multi<256, 256, 4, 4>(mp, s);
        else if (w.big) launch_multi<256, 256, 4, 2>(mp, s);
        else if (w.narrow) { if (w.small) launch_multi<64, 64, 2, 2>(mp, s); else launch_multi<128, 64, 4, 1>(mp, s); }
        else { if (w.small) launch_multi<64, 128, 1, 4>(mp, s); else launch_multi<128, 128, 2, 2>(mp, s); }
        if (int rc = check_launch("wgrad_dma_multi_kernel")) return rc;
        if (mr.n) {
            hipLaunchKernelGGL(wgrad::slab_reduce_multi_kernel, dim3((unsigned)rfirst), dim3(256), 0, s, mr);
            if (int rc = check_launch("slab_reduce_multi_kernel")) return rc;
        }
    }
    return MHE_OK;
}

// Fixed-order column sums (see colsum_kernel).  Without a workspace ONE block walks all rows of its column group (any `out`, up to 4,096
// rows).  With one, the rows are cut into up to 256 slabs (enough blocks to fill the chip: the [256][24,576] conditioning-gradient sums of
// config C2 ran on 192 blocks at 0.5 TB/s) whose sums go to the workspace with plain stores, and a second launch folds those rows.
extern "C" size_t mhe_colsum_workspace_floats(long R, int C) { return R > 16 && C > 0 ? (size_t)256 * (size_t)C : 0; }

template <typename T>
static void colsum_launch(const T *rows, float *out, float *partial, long R, int C, long rpb, int gw, long gs, hipStream_t s) {
    const int cw = C < 256 ? C : 256;
    const dim3 grid((C + cw - 1) / cw, (unsigned)((R + rpb - 1) / rpb));
    hipLaunchKernelGGL(wgrad::colsum_kernel<T>, grid, dim3(256), 0, s, rows, out, partial, R, C, rpb, gw, gs, 1);
}

extern "C" int mhe_colsum_ws_f32(const void *rows, float *out, long R, int C, int dtype, int out_group_width, long out_group_stride,
                                 float *workspace, size_t workspace_floats, void *stream) {
    MHE_REQUIRE(rows && out && R > 0 && C > 0 && (dtype == MHE_F32 || dtype == MHE_BF16), "mhe_colsum_f32: bad arguments");
    const int cw = C < 256 ? C : 256;
    MHE_REQUIRE(256 % cw == 0, "mhe_colsum_f32: C=%d must divide 256 or be at least 256", C);
    MHE_REQUIRE(out_group_width >= 0 && (out_group_width == 0 || (C % out_group_width == 0 && out_group_stride >= out_group_width)),
                "mhe_colsum_f32: bad output grouping (width %d, stride %ld)", out_group_width, out_group_stride);
    hipStream_t s = (hipStream_t)stream;
    const long ncg = (C + cw - 1) / cw;
    long nby = 1024 / ncg;
    if (nby > 256) nby = 256;
    if (nby > (R + 15) / 16) nby = (R + 15) / 16;
    if (nby < 1 || !workspace || workspace_floats < (size_t)nby * C) nby = 1;
    if (nby == 1) {
        MHE_REQUIRE(R <= 4096, "mhe_colsum_f32: %ld rows need a workspace of mhe_colsum_workspace_floats(R, C) floats (fixed summation order)", R);
        if (dtype == MHE_F32) colsum_launch((const float *)rows, out, nullptr, R, C, R, out_group_width, out_group_stride, s);
        else colsum_launch((const u16 *)rows, out, nullptr, R, C, R, out_group_width, out_group_stride, s);
        return check_launch("colsum_kernel");
    }
    const long rpb = (R + nby - 1) / nby, nb = (R + rpb - 1) / rpb;
    if (dtype == MHE_F32) colsum_launch((const float *)rows, nullptr, workspace, R, C, rpb, 0, 0, s);
    else colsum_launch((const u16 *)rows, nullptr, workspace, R, C, rpb, 0, 0, s);
    if (int rc = check_launch("colsum_kernel")) return rc;
    colsum_launch((const float *)workspace, out, nullptr, nb, C, nb, out_group_width, out_group_stride, s);
    return check_launch("colsum_kernel");
}

extern "C" int mhe_colsum_f32(const void *rows, float *out, long R, int C, int dtype, void *stream) {
    return mhe_colsum_ws_f32(rows, out, R, C, dtype, 0, 0, nullptr, 0, stream);
}

extern "C" int mhe_gather_affine8_bf16(const float *src, const int *base_stride, const unsigned char *mask, void *dst, size_t n, void *stream) {
    MHE_REQUIRE(src && base_stride && mask && dst && n > 0 && n % 8 == 0, "mhe_gather_affine8_bf16: bad arguments (n = %zu must be a multiple of 8)", n);
    MHE_REQUIRE(((size_t)base_stride & 7) == 0 && ((size_t)dst & 15) == 0, "mhe_gather_affine8_bf16: alignment");
    size_t b8 = (n / 8 + 255) / 256;
    if (b8 > 16384) b8 = 16384;
    hipLaunchKernelGGL(wgrad::gather8_affine_bf16_kernel, dim3((unsigned)b8), dim3(256), 0, (hipStream_t)stream, src,
                       reinterpret_cast<const int2 *>(base_stride), mask, (u16 *)dst, n / 8);
    return check_launch("gather8_affine_bf16_kernel");
}

extern "C" int mhe_gather_f32(const float *src, const int *idx, const int *idx2, void *dst, size_t n, int dst_dtype, void *stream) {
    MHE_REQUIRE(src && idx && dst && n > 0, "mhe_gather_f32: bad arguments");
    MHE_REQUIRE(dst_dtype == MHE_F32 || dst_dtype == MHE_BF16, "mhe_gather_f32: dst_dtype=%d", dst_dtype);
    size_t blocks = (n + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    if (dst_dtype == MHE_BF16 && !idx2 && n % 8 == 0 && ((size_t)idx & 15) == 0 && ((size_t)dst & 15) == 0) {
        size_t b8 = (n / 8 + 255) / 256;
        if (b8 > 16384) b8 = 16384;
        hipLaunchKernelGGL(wgrad::gather8_bf16_kernel, dim3((unsigned)b8), dim3(256), 0, (hipStream_t)stream, src, idx, (u16 *)dst, n / 8);
        return check_launch("gather8_bf16_kernel");
    }
    if (dst_dtype == MHE_F32 && n % 4 == 0 && ((size_t)idx & 15) == 0 && ((size_t)dst & 15) == 0 && (!idx2 || ((size_t)idx2 & 15) == 0)) {
        size_t b4 = (n / 4 + 255) / 256;
        if (b4 > 16384) b4 = 16384;
        hipLaunchKernelGGL(wgrad::gather4_f32_kernel, dim3((unsigned)b4), dim3(256), 0, (hipStream_t)stream, src, idx, idx2, (float *)dst, n / 4);
        return check_launch("gather4_f32_kernel");
    }
    if (dst_dtype == MHE_F32)
        hipLaunchKernelGGL(wgrad::gather_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, src, idx, idx2, (float *)dst, n);
    else
        hipLaunchKernelGGL(wgrad::gather_kernel<u16>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, src, idx, idx2, (u16 *)dst, n);
    return check_launch("gather_kernel");
}

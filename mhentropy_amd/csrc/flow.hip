// Conditional RealNVP coupling stack (hand/flows.py:75-122,210-227) as ONE launch:
// every wavefront carries 16 hypothesis rows through all couplings with the hidden
// activations resident in MFMA accumulator registers, never in LDS or HBM.
//
// Formulation (per network, transposed so that rows sit on MFMA columns):
//     H1^T[h][m] = W0[h][:] x_^T[:][m] + cond0[h]     (cond0 = c0(feat) + biases, per image)
//     H2^T       = W1 leaky(H1^T) + cond1
//     out^T      = W2 leaky(H2^T) + b2
// with v_mfma_f32_16x16x4_f32 (exact f32 fma chains).  The D layout of that
// instruction (column = lane&15, row = 4*(lane>>4)+reg) is exactly the B-operand
// layout of the next product (k = lane>>4 group), so an accumulator register is
// fed back as the next layer's B operand without moving a lane
// (cdna_hip_programming.md section 3, "An accumulator tile as the next MFMA's operand").
// Weights are pre-packed on the host in A-fragment order (one 1 KiB block per
// 16x16 weight tile, lane-linear) and streamed HBM/L2 -> LDS in 16 KiB stages
// shared by the 4 waves of a workgroup; LDS reads are conflict-free by construction.
//
// Algorithmic work: 2*(48*H + H*H + H*48) MAC per row per coupling; weights
// 2*NET_STAGES*16 KiB per coupling per workgroup from L2.
#include "common.h"

namespace mhe { namespace flow {

constexpr int DT = 3;           // 45 -> 48 = 3 tiles of 16
constexpr int STAGE_BLOCKS = 16;
constexpr int STAGE_FLOATS = STAGE_BLOCKS * 256;

template <int NT> struct Plan {
    static constexpr int CH = NT / 4;                                   // hidden chunks of 4 tiles
    static constexpr int L0S = (DT * NT + STAGE_BLOCKS - 1) / STAGE_BLOCKS;
    static constexpr int NET_STAGES = L0S + CH * (CH + 1);
};

static inline int net_stages(int nt) {
    const int ch = nt / 4, l0s = (DT * nt + STAGE_BLOCKS - 1) / STAGE_BLOCKS;
    return l0s + ch * (ch + 1);
}

__device__ __forceinline__ v4f leaky4(v4f v) {
    v4f r;
#pragma unroll
    for (int i = 0; i < 4; ++i) r[i] = fmaxf(v[i], 0.01f * v[i]);   // F.leaky_relu default slope (flows.py:117)
    return r;
}

#define MFMA4(w, bfrag, acc)                                                         \
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32((w), (bfrag), (acc), 0, 0, 0)

template <int NT>
__global__ __launch_bounds__(256) void couplings_kernel(
    const float *__restrict__ in, float *__restrict__ out, const float *__restrict__ cond,
    const float *__restrict__ wstream, const float *__restrict__ bias2, const float *__restrict__ mask,
    float *__restrict__ sum_s_o, float *__restrict__ logp_o, int R, int B, int dim, int ncoup, int inverse) {
    using P = Plan<NT>;
    constexpr int H = NT * 16;
    __shared__ __attribute__((aligned(16))) v4f lbuf[2][STAGE_BLOCKS * 64];

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int q = lane >> 4, m = lane & 15;
    const int N = R / B;
    // image-major tiling: the 16 rows of a wave belong to one image when N % 16 == 0
    const int g = (blockIdx.x * 4 + wave) * 16 + m;
    const bool valid = g < R;
    const int gc = valid ? g : R - 1;
    const int b = gc / N, r = (gc % N) * B + b;

    // flow variable, D layout: x[T][i] <-> dim 16T + 4q + i
    v4f x[DT];
#pragma unroll
    for (int T = 0; T < DT; ++T)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int d = 16 * T + 4 * q + i;
            x[T][i] = d < dim ? in[(size_t)r * dim + d] : 0.f;
        }
    float sq_in = 0.f;
#pragma unroll
    for (int T = 0; T < DT; ++T)
#pragma unroll
        for (int i = 0; i < 4; ++i) sq_in = fmaf(x[T][i], x[T][i], sq_in);
    float sum_s = 0.f;

    // ---- weight-stream pipeline state (block-uniform)
    const int nnets = 2 * ncoup;
    const size_t net_floats = (size_t)P::NET_STAGES * STAGE_FLOATS;
    int seq = 0, st = 0, cur = 0;          // net sequence position, stage within net, LDS buffer
    auto net_of_seq = [&](int s) { const int step = s >> 1; return 2 * (inverse ? ncoup - 1 - step : step) + (s & 1); };
    v4f pf[4];
    bool has_next = false;
    // prologue: stage 0 of the first net
    {
        const v4f *src = reinterpret_cast<const v4f *>(wstream + (size_t)net_of_seq(0) * net_floats);
#pragma unroll
        for (int j = 0; j < 4; ++j) lbuf[0][tid + 256 * j] = src[tid + 256 * j];
        __syncthreads();
    }
#define STAGE_BEGIN()                                                                                 \
    {                                                                                                 \
        int nseq = seq, nst = st + 1;                                                                 \
        if (nst == P::NET_STAGES) { nst = 0; ++nseq; }                                                \
        has_next = nseq < nnets;                                                                      \
        if (has_next) {                                                                               \
            const v4f *src = reinterpret_cast<const v4f *>(wstream + (size_t)net_of_seq(nseq) * net_floats + \
                                                           (size_t)nst * STAGE_FLOATS);              \
            _Pragma("unroll") for (int j = 0; j < 4; ++j) pf[j] = src[tid + 256 * j];                  \
        }                                                                                             \
        seq = nseq; st = nst;                                                                         \
    }
#define STAGE_END()                                                                                   \
    {                                                                                                 \
        if (has_next) {                                                                               \
            _Pragma("unroll") for (int j = 0; j < 4; ++j) lbuf[cur ^ 1][tid + 256 * j] = pf[j];        \
        }                                                                                             \
        __syncthreads();                                                                              \
        cur ^= 1;                                                                                     \
    }

    for (int step = 0; step < ncoup; ++step) {
        const int ci = inverse ? ncoup - 1 - step : step;
        v4f mk[DT], xin[DT], s_keep[DT];
#pragma unroll
        for (int T = 0; T < DT; ++T)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int d = 16 * T + 4 * q + i;
                mk[T][i] = d < dim ? mask[ci * dim + d] : 1.f;     // padded dims stay fixed at 0
                xin[T][i] = x[T][i] * mk[T][i];
            }
#pragma unroll 1
        for (int netk = 0; netk < 2; ++netk) {
            const int net = 2 * ci + netk;
            const float *cb = cond + ((size_t)(b * nnets + net) * 2) * H;
            // ---- layer 0: H1 = W0 x_ + (c0(feat) + biases)
            v4f H1[NT];
#pragma unroll
            for (int To = 0; To < NT; ++To) H1[To] = *reinterpret_cast<const v4f *>(cb + 16 * To + 4 * q);
#pragma unroll
            for (int s0 = 0; s0 < P::L0S; ++s0) {
                STAGE_BEGIN();
                const v4f *L = lbuf[cur];
#pragma unroll
                for (int g4 = 0; g4 < STAGE_BLOCKS / 4; ++g4) {
                    v4f w[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) w[j] = L[(g4 * 4 + j) * 64 + lane];
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            constexpr int dummy = 0; (void)dummy;
                            const int f = s0 * STAGE_BLOCKS + g4 * 4 + j;
                            if (f < DT * NT) MFMA4(w[j][i], xin[f / NT][i], H1[f % NT]);
                        }
                }
                STAGE_END();
            }
#pragma unroll
            for (int To = 0; To < NT; ++To) H1[To] = leaky4(H1[To]);
            // ---- layers 1 and 2 interleaved per chunk of 4 hidden tiles
            v4f o[DT];
#pragma unroll
            for (int T = 0; T < DT; ++T)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int d = 16 * T + 4 * q + i;
                    o[T][i] = d < dim ? bias2[net * dim + d] : 0.f;
                }
#pragma unroll 1
            for (int c = 0; c < P::CH; ++c) {
                v4f H2[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) H2[j] = *reinterpret_cast<const v4f *>(cb + H + 16 * (4 * c + j) + 4 * q);
#pragma unroll
                for (int u = 0; u < P::CH; ++u) {
                    STAGE_BEGIN();
                    const v4f *L = lbuf[cur];
#pragma unroll
                    for (int tk = 0; tk < 4; ++tk) {
                        v4f w[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) w[j] = L[(tk * 4 + j) * 64 + lane];
#pragma unroll
                        for (int i = 0; i < 4; ++i)
#pragma unroll
                            for (int j = 0; j < 4; ++j) MFMA4(w[j][i], H1[4 * u + tk][i], H2[j]);
                    }
                    STAGE_END();
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) H2[j] = leaky4(H2[j]);
                STAGE_BEGIN();
                {
                    const v4f *L = lbuf[cur];
#pragma unroll
                    for (int tk = 0; tk < 4; ++tk) {
                        v4f w[DT];
#pragma unroll
                        for (int T = 0; T < DT; ++T) w[T] = L[(T * 4 + tk) * 64 + lane];
#pragma unroll
                        for (int i = 0; i < 4; ++i)
#pragma unroll
                            for (int T = 0; T < DT; ++T) MFMA4(w[T][i], H2[tk][i], o[T]);
                    }
                }
                STAGE_END();
            }
            if (netk == 0) {
#pragma unroll
                for (int T = 0; T < DT; ++T)
#pragma unroll
                    for (int i = 0; i < 4; ++i) s_keep[T][i] = tanhf(o[T][i]) * (1.f - mk[T][i]);   // flows.py:120-121,214
            } else {
#pragma unroll
                for (int T = 0; T < DT; ++T)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float om = 1.f - mk[T][i];
                        const float s = s_keep[T][i], t = o[T][i] * om;
                        if (!inverse) x[T][i] = xin[T][i] + om * (x[T][i] * expf(s) + t);       // flows.py:216
                        else          x[T][i] = om * (x[T][i] - t) * expf(-s) + xin[T][i];      // flows.py:225
                        sum_s += s;
                    }
            }
        }
    }
#undef STAGE_BEGIN
#undef STAGE_END

    float sq_out = 0.f;
#pragma unroll
    for (int T = 0; T < DT; ++T)
#pragma unroll
        for (int i = 0; i < 4; ++i) sq_out = fmaf(x[T][i], x[T][i], sq_out);
    // reduce over the 4 lane groups that share a row
    float base_sq = inverse ? sq_out : sq_in;
    base_sq += __shfl_xor(base_sq, 16, 64); base_sq += __shfl_xor(base_sq, 32, 64);
    sum_s += __shfl_xor(sum_s, 16, 64);     sum_s += __shfl_xor(sum_s, 32, 64);
    if (valid) {
#pragma unroll
        for (int T = 0; T < DT; ++T)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int d = 16 * T + 4 * q + i;
                if (d < dim) out[(size_t)r * dim + d] = x[T][i];
            }
        if (q == 0) {
            if (sum_s_o) sum_s_o[r] = sum_s;
            // MultivariateNormal(0,I).log_prob(z) + log_det   (flows.py:157,320)
            if (logp_o) logp_o[r] = (-0.5f * base_sq - 0.5f * (float)dim * 1.8378770664093453f) - sum_s;
        }
    }
}

}}  // namespace mhe::flow

using namespace mhe;

extern "C" size_t mhe_flow_packed_floats_per_net(int dim, int hidden) {
    if (dim <= 0 || dim > 16 * flow::DT || hidden <= 0 || hidden % 64) return 0;
    return (size_t)flow::net_stages(hidden / 16) * flow::STAGE_FLOATS;
}

// block (To,Tk) of a weight matrix W[rows][cols] in A-fragment order:
// lane l = 16*q + n holds W[16To+n][16Tk+4q+i], i = 0..3
static void pack_block(const float *W, int rows, int cols, int To, int Tk, float *dst) {
    for (int l = 0; l < 64; ++l) {
        const int q = l >> 4, n = l & 15;
        for (int i = 0; i < 4; ++i) {
            const int rr = 16 * To + n, cc = 16 * Tk + 4 * q + i;
            dst[l * 4 + i] = (rr < rows && cc < cols) ? W[(size_t)rr * cols + cc] : 0.f;
        }
    }
}

extern "C" int mhe_flow_pack_net_host(const float *W0, const float *W1, const float *W2, int dim, int hidden,
                                      float *out) {
    const size_t total = mhe_flow_packed_floats_per_net(dim, hidden);
    MHE_REQUIRE(total && W0 && W1 && W2 && out, "mhe_flow_pack_net_host: dim=%d (<=48) hidden=%d (multiple of 64)", dim, hidden);
    const int NT = hidden / 16, CH = NT / 4, DT = flow::DT;
    const int L0S = (DT * NT + flow::STAGE_BLOCKS - 1) / flow::STAGE_BLOCKS;
    for (size_t i = 0; i < total; ++i) out[i] = 0.f;
    // layer 0: flat block list f = Tk*NT + To
    for (int f = 0; f < DT * NT; ++f) pack_block(W0, hidden, dim, f % NT, f / NT, out + (size_t)f * 256);
    float *p = out + (size_t)L0S * flow::STAGE_FLOATS;
    for (int c = 0; c < CH; ++c) {
        for (int u = 0; u < CH; ++u, p += flow::STAGE_FLOATS)
            for (int tk = 0; tk < 4; ++tk)
                for (int j = 0; j < 4; ++j) pack_block(W1, hidden, hidden, 4 * c + j, 4 * u + tk, p + (tk * 4 + j) * 256);
        for (int T = 0; T < DT; ++T)
            for (int tk = 0; tk < 4; ++tk) pack_block(W2, dim, hidden, T, 4 * c + tk, p + (T * 4 + tk) * 256);
        p += flow::STAGE_FLOATS;
    }
    return MHE_OK;
}

extern "C" int mhe_flow_couplings_f32(const float *in, float *out, const float *cond, const float *wstream,
                                      const float *bias2, const float *mask, float *sum_s, float *log_prob, int R,
                                      int B, int dim, int hidden, int ncoup, int direction, void *stream) {
    MHE_REQUIRE(in && out && cond && wstream && bias2 && mask, "mhe_flow_couplings_f32: null pointer");
    MHE_REQUIRE(R > 0 && B > 0 && R % B == 0, "mhe_flow_couplings_f32: R=%d must be a positive multiple of B=%d", R, B);
    MHE_REQUIRE(dim > 0 && dim <= 16 * flow::DT, "mhe_flow_couplings_f32: dim=%d unsupported (1..48)", dim);
    MHE_REQUIRE(ncoup > 0, "mhe_flow_couplings_f32: ncoup=%d", ncoup);
    MHE_REQUIRE(direction == MHE_FLOW_FORWARD || direction == MHE_FLOW_INVERSE, "mhe_flow_couplings_f32: direction=%d", direction);
    const dim3 grid((R + 63) / 64), block(256);
    const int inv = direction == MHE_FLOW_INVERSE;
    hipStream_t s = (hipStream_t)stream;
#define LAUNCH(NT_)                                                                                                  \
    hipLaunchKernelGGL(flow::couplings_kernel<NT_>, grid, block, 0, s, in, out, cond, wstream, bias2, mask, sum_s,    \
                       log_prob, R, B, dim, ncoup, inv)
    switch (hidden) {
        case 64: LAUNCH(4); break;
        case 128: LAUNCH(8); break;
        case 256: LAUNCH(16); break;
        case 512: LAUNCH(32); break;
        default: MHE_REQUIRE(false, "mhe_flow_couplings_f32: hidden=%d unsupported (64,128,256,512)", hidden);
    }
#undef LAUNCH
    return check_launch("flow::couplings_kernel");
}

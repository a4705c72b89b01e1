// Base noise of the flow drawn on the device INSIDE the step: z0 ~ N(0, I) for the R = N*B hypothesis rows
// (reference hand/flows.py:339 `prior.sample((N*B,))`, MultivariateNormal(0, I_45); hand/network.py:733-735).
// Counter-based: Philox4x32-10 (Salmon et al., SC'11) keyed by a seed, four uniforms per call -> two Box-Muller pairs.  The
// generator state lives in DEVICE memory and the launch that consumed it advances it (last workgroup to finish), so a HIP
// graph that contains this launch draws fresh noise on every replay with no host involvement.  GPU draws can never equal the
// reference's CPU generator stream - parity runs pass `noise=`; this is the product's default draw.
#include "common.h"

namespace mhe { namespace rng {

__device__ __forceinline__ void philox_round(unsigned &c0, unsigned &c1, unsigned &c2, unsigned &c3, unsigned k0, unsigned k1) {
    const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0, p1 = (unsigned long long)0xCD9E8D57u * c2;
    const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0, n1 = (unsigned)p1, n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1, n3 = (unsigned)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
}

__device__ __forceinline__ void philox4x32_10(unsigned long long ctr, unsigned long long key, unsigned out[4]) {
    unsigned c0 = (unsigned)ctr, c1 = (unsigned)(ctr >> 32), c2 = 0u, c3 = 0u, k0 = (unsigned)key, k1 = (unsigned)(key >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        philox_round(c0, c1, c2, c3, k0, k1);
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// state: [0] seed (key), [1] next counter, [2] workgroups finished in the running launch
__global__ __launch_bounds__(256) void randn_kernel(float *__restrict__ out, long n, unsigned long long *__restrict__ state, float scale) {
    const unsigned long long key = state[0], base = state[1];
    const long n4 = (n + 3) / 4;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        unsigned r[4];
        philox4x32_10(base + (unsigned long long)i, key, r);
        float v[4];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            // u1 in (0, 1], u2 in [0, 1): 24 mantissa bits each
            const float u1 = ((float)(r[2 * h] >> 8) + 1.0f) * (1.0f / 16777216.0f), u2 = (float)(r[2 * h + 1] >> 8) * (1.0f / 16777216.0f);
            // hardware transcendentals (v_log_f32, v_sqrt_f32, v_sin_f32 / v_cos_f32 take revolutions): ~1e-6 of the value - noise, not parity
            // data; the library forms (logf, sincospif with full range reduction) made this launch 61 us for 2.3 M values
            const float rad = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u1));
            const float sn = __builtin_amdgcn_sinf(u2), cs = __builtin_amdgcn_cosf(u2);
            v[2 * h] = rad * cs * scale; v[2 * h + 1] = rad * sn * scale;
        }
        if (4 * i + 3 < n) *reinterpret_cast<float4 *>(out + 4 * i) = make_float4(v[0], v[1], v[2], v[3]);
        else for (int e = 0; e < 4 && 4 * i + e < n; ++e) out[4 * i + e] = v[e];
    }
    // every workgroup has read `base` before it gets here; the last one to arrive moves the counter past this launch
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        const unsigned long long done = atomicAdd(&state[2], 1ull);
        if (done == (unsigned long long)gridDim.x - 1ull) {
            state[2] = 0ull;
            state[1] = base + (unsigned long long)n4;
            __threadfence();
        }
    }
}

}}  // namespace mhe::rng

extern "C" int mhe_randn_f32(float *out, long n, unsigned long long *state, float scale, void *stream) {
    using namespace mhe;
    MHE_REQUIRE(out && state && n > 0, "mhe_randn_f32: bad arguments");
    MHE_REQUIRE((reinterpret_cast<uintptr_t>(out) & 15) == 0, "mhe_randn_f32: out must be 16-byte aligned");
    const long n4 = (n + 3) / 4;
    long blocks = (n4 + 255) / 256;
    // one workgroup per CU: every workgroup ends with an atomic on ONE counter word, and same-address atomics retire one per ~28 ns - with 2,048
    // workgroups that tail WAS the launch (61 us for 2.3 M values; 20 us at the train step's 0.74 M)
    if (blocks > 256) blocks = 256;
    hipLaunchKernelGGL(rng::randn_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, out, n, state, scale);
    return check_launch("randn_kernel");
}

// ---- dropout of the ConditionalGlow's residual blocks in train mode (reference hand/network.py:343-344: `dropout_probability=0.2`, and the
// comment at :781 "Since uses Dropout"; nflows ResidualBlock: activation -> linear -> activation -> DROPOUT -> linear).  x <- x * keep / (1 - p)
// in place with keep ~ Bernoulli(1 - p) drawn from the same device-resident Philox state as the base noise (so a captured graph draws a fresh
// mask on every replay); one mask BIT per element goes to `bits` for the reverse pass, which applies the same launch with draw = 0 to the
// gradient.  Thread = 8 elements = one byte of bits = one Philox call (eight 16-bit uniforms; P(drop) = round(p * 65536) / 65536).
namespace mhe { namespace rng {
template <typename T>
__global__ __launch_bounds__(256) void dropout_kernel(T *__restrict__ x, unsigned char *__restrict__ bits, long n8, unsigned long long *__restrict__ state,
                                                      unsigned thr, float scale, int draw) {
    const unsigned long long key = draw ? state[0] : 0ull, base = draw ? state[1] : 0ull;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long)gridDim.x * 256) {
        unsigned m;
        if (draw) {
            unsigned r[4];
            philox4x32_10(base + (unsigned long long)i, key, r);
            m = 0;
#pragma unroll
            for (int k = 0; k < 8; ++k) m |= (unsigned)(((r[k >> 1] >> (16 * (k & 1))) & 0xffffu) >= thr) << k;
            if (bits) bits[i] = (unsigned char)m;
        } else m = bits[i];
        float v[8];
        load4<T>(x + 8 * i, v); load4<T>(x + 8 * i + 4, v + 4);
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = (m >> k) & 1u ? v[k] * scale : 0.f;
        store4<T>(x + 8 * i, v); store4<T>(x + 8 * i + 4, v + 4);
    }
    if (!draw) return;
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        const unsigned long long done = atomicAdd(&state[2], 1ull);
        if (done == (unsigned long long)gridDim.x - 1ull) {
            state[2] = 0ull;
            state[1] = base + (unsigned long long)n8;
            __threadfence();
        }
    }
}
}}  // namespace mhe::rng

extern "C" int mhe_dropout(void *x, int dtype, unsigned char *bits, long n, float p_drop, unsigned long long *state, int draw, void *stream) {
    using namespace mhe;
    MHE_REQUIRE(x && n > 0 && n % 8 == 0 && (dtype == MHE_F32 || dtype == MHE_BF16), "mhe_dropout: n=%ld must be a multiple of 8, dtype f32 / bf16", n);
    MHE_REQUIRE(p_drop >= 0.f && p_drop < 1.f, "mhe_dropout: p=%f", (double)p_drop);
    MHE_REQUIRE(draw ? state != nullptr : bits != nullptr, "mhe_dropout: drawing needs the generator state, applying needs the mask bits");
    MHE_REQUIRE((reinterpret_cast<uintptr_t>(x) & 15) == 0, "mhe_dropout: x must be 16-byte aligned");
    const long n8 = n / 8;
    long blocks = (n8 + 255) / 256;
    if (blocks > 512) blocks = 512;
    const unsigned thr = (unsigned)(p_drop * 65536.f + 0.5f);
    const float scale = 1.f / (1.f - p_drop);
    if (dtype == MHE_F32)
        hipLaunchKernelGGL(rng::dropout_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (float *)x, bits, n8, state, thr, scale, draw);
    else
        hipLaunchKernelGGL(rng::dropout_kernel<u16>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (u16 *)x, bits, n8, state, thr, scale, draw);
    return check_launch("dropout_kernel");
}

// ... the mask bits alone (no tensor touched): all dropouts of the fused Glow kernel (csrc/glow_fwd.hip) are drawn by ONE launch; byte i holds
// the keep bits of elements 8 i .. 8 i + 7, the format mhe_dropout writes and applies
namespace mhe { namespace rng {
__global__ __launch_bounds__(256) void dropout_bits_kernel(unsigned char *__restrict__ bits, long n8, unsigned long long *__restrict__ state, unsigned thr) {
    const unsigned long long key = state[0], base = state[1];
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long)gridDim.x * 256) {
        unsigned r[4];
        philox4x32_10(base + (unsigned long long)i, key, r);
        unsigned m = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) m |= (unsigned)(((r[k >> 1] >> (16 * (k & 1))) & 0xffffu) >= thr) << k;
        bits[i] = (unsigned char)m;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        const unsigned long long done = atomicAdd(&state[2], 1ull);
        if (done == (unsigned long long)gridDim.x - 1ull) {
            state[2] = 0ull;
            state[1] = base + (unsigned long long)n8;
            __threadfence();
        }
    }
}
}}  // namespace mhe::rng

extern "C" int mhe_dropout_bits(unsigned char *bits, long n, float p_drop, unsigned long long *state, void *stream) {
    using namespace mhe;
    MHE_REQUIRE(bits && state && n > 0 && n % 8 == 0 && p_drop >= 0.f && p_drop < 1.f, "mhe_dropout_bits: bad arguments (n=%ld, p=%f)", n, (double)p_drop);
    const long n8 = n / 8;
    long blocks = (n8 + 255) / 256;
    if (blocks > 256) blocks = 256;            // the counter word's atomic tail: see mhe_randn_f32 (58 us at 2,048 workgroups for the Glow train step's 4.2 M bytes of bits)
    hipLaunchKernelGGL(rng::dropout_bits_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, bits, n8, state, (unsigned)(p_drop * 65536.f + 0.5f));
    return check_launch("dropout_bits_kernel");
}

// ---- BasicEnc's stochastic head (reference hand/network.py:121-138): sd = exp(l2 / 2) | sigmoid(l2), z = mn + sd * eps.
// Dead for MHEnt (it keeps only mn, :779,862) - built so that the exported class returns the reference's (z, mn, sd).
namespace mhe { namespace rng {
__global__ __launch_bounds__(256) void reparam_kernel(const float *__restrict__ mn, const float *__restrict__ l2, const float *__restrict__ eps,
                                                      float *__restrict__ sd, float *__restrict__ z, long n, int sigmoid_act, int deterministic) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float s = sigmoid_act ? 1.f / (1.f + expf(-l2[i])) : expf(0.5f * l2[i]);
        sd[i] = s;
        z[i] = deterministic ? mn[i] : fmaf(s, eps[i], mn[i]);
    }
}
}}  // namespace mhe::rng

extern "C" int mhe_reparam_f32(const float *mn, const float *l2, const float *eps, float *sd, float *z, long n, int sigmoid_act,
                               int deterministic, void *stream) {
    using namespace mhe;
    MHE_REQUIRE(mn && l2 && sd && z && n > 0 && (deterministic || eps), "mhe_reparam_f32: bad arguments");
    long blocks = (n + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(rng::reparam_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, mn, l2, eps, sd, z, n, sigmoid_act, deterministic);
    return check_launch("reparam_kernel");
}

// Shared pieces of the fragment-streaming flow kernels (flow_rev.hip: reverse chain; flow_fwd.hip: coupling stack): 16x16x32 bf16 MFMA,
// DPP row sums, buffer-instruction access, fragment-major weight operands.
#pragma once
#include "common.h"

namespace mhe { namespace flowfrag {

typedef __attribute__((ext_vector_type(8))) __bf16 bf8;

__device__ __forceinline__ v4f mfma(const uint4 &a, const uint4 &b, v4f c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8, a), __builtin_bit_cast(bf8, b), c, 0, 0, 0);
}
// sum over the 16 lanes of a DPP row (the lanes that share q): quad butterflies, then the row rotated by 4 and by 8 - four v_add_f32_dpp,
// where __shfl_xor lowered to four dependent ds_bpermute round trips per value (64 per epilogue; 0.5 ms of the first version's 2.4)
template <int CTRL> __device__ __forceinline__ float dpp_add(float v) {
    return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float row16_sum(float v) {
    v = dpp_add<0xB1>(v);                                 // quad_perm [1,0,3,2]
    v = dpp_add<0x4E>(v);                                 // quad_perm [2,3,0,1]
    v = dpp_add<0x124>(v);                                // row_ror:4
    return dpp_add<0x128>(v);                             // row_ror:8
}
// Global traffic goes through buffer instructions: one lane-dependent VGPR byte offset per access shape, everything else (net, row piece,
// fragment) a scalar offset - as flat addresses hipcc built a 64-bit VGPR pair per access, hoisted them out of the coupling loop and spilled
// over three hundred registers.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ rsrc_t rsrc_of(const void *p, size_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, (int)(bytes > 0xffffffffu ? 0xffffffffu : bytes), 0x00020000);
}
__device__ __forceinline__ uint4 bld(rsrc_t r, unsigned voff, unsigned soff) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0);
    return make_uint4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ void bst(rsrc_t r, unsigned voff, unsigned soff, const uint4 &v) {
    const u32x4 t = {v.x, v.y, v.z, v.w};
    __builtin_amdgcn_raw_buffer_store_b128(t, r, (int)voff, (int)soff, 0);
}
// fragment-major operand: the 16-byte piece lane (q, l15) feeds to the MFMA of (row tile t, k step ks) lies at ((t * KS + ks) * 64 + lane) * 8,
// so one wave-instruction reads 1 KiB in a row (from plain [row][k] storage the 64 lanes of a fragment load are 64 separate 16-byte requests
// to 16 rows: the weight stream of the first version of flow_rev.hip ran at 30 GB/s per CU)
__device__ __forceinline__ uint4 frag(rsrc_t r, unsigned lane16, int t, int KS, int ks) { return bld(r, lane16, (unsigned)(t * KS + ks) * 1024u); }

}}  // namespace mhe::flowfrag

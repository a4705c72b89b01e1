// joint_pass: everything up to the 21 normalised joints of one hypothesis on one wavefront.
// Shared by the loss/sampling kernels (mano.hip) and the reverse-mode kernel (mano_bwd.hip).
#pragma once
#include "common.h"
#include "mano_layout.h"

namespace mhe { namespace mano {

// MANO joint j -> index into the level-ordered transform list is already folded:
// G[j] below IS the global transform of MANO joint j (manolayer.py:228 reorder).
// final joint k (RHD order) <- pre-reorder index (16 chain joints then 5 tips):
// pre[ JOINT_REORDER[ FREIHAND2RHD[k] ] ]   (manolayer.py:260, utils.py:15, ManoLayer.py:54-56)
static __device__ constexpr int kJointReorder[21] = {0, 13, 14, 15, 16, 1, 2, 3, 17, 4, 5, 6, 18, 10, 11, 12, 19, 7, 8, 9, 20};
static __device__ constexpr int kFreihand2Rhd[21] = {0, 4, 3, 2, 1, 8, 7, 6, 5, 12, 11, 10, 9, 16, 15, 14, 13, 20, 19, 18, 17};
constexpr int kCenterPre = 4;     // kJointReorder[center_idx = 9]   (manolayer.py:262-266)
constexpr int kRootIdx = 12;      // hand/network.py:477
constexpr int kNormIdx = 11;      // hand/network.py:478

// per-wave LDS scratch (floats)
constexpr int S_POSE = 0;      // [48]
constexpr int S_ROT = 48;      // [16][9]
constexpr int S_JR = 192;      // [16][3]  rest joints
constexpr int S_G = 240;       // [16][12] global transform: R(9), t(3)
constexpr int S_GR = 432;      // [16][12] R(9), t - R*j_rest
constexpr int S_TIPV = 624;    // [5][3]   posed tip vertices (rest frame)
constexpr int S_PRE = 640;     // [21][3]  chain joints + skinned tips
constexpr int S_J21 = 704;     // [21][3]  final joints, mm, centred
constexpr int SCRATCH = 768;

__device__ __forceinline__ float bcast(float v, int srclane) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), srclane));
}

struct RowOut {
    float xyz;      // lane < 63: normalised joint coordinate (joint lane/3, comp lane%3)
    float bone;     // |J11 - J12| in mm
    float center_c; // lane%3 component of the centring joint (metres, pre-centre)
    float root_c;   // lane%3 component of final joint 12 (mm, centred)
};

// Everything up to the 21 normalised joints for one hypothesis.  `th45` lane<45,
// `det` lane<16 = [th3(3) bt(10) logs t(2)] (hand/network.py:370-372 order).
// tb = joint section of the table blob in LDS; sc = this wave's scratch.
__device__ __forceinline__ RowOut joint_pass(const float *tb, float *sc, int lane, float th45, float det) {
    // -- PCA coefficients -> axis-angle (manolayer.py:131-143)
    {
        float acc = 0.f;
        const int lc = lane < 45 ? lane : 44;
#pragma unroll
        for (int k = 0; k < 45; ++k) acc = fmaf(bcast(th45, k), tb[COMPS + k * 45 + lc], acc);
        if (lane < 45) sc[S_POSE + 3 + lane] = tb[MEAN + lc] + acc;
        if (lane < 3) sc[S_POSE + lane] = det;
    }
    wave_sync();
    // -- Rodrigues through a unit quaternion, 16 joints on 16 lanes
    //    (rodrigues_layer.py:43-54, :15-40)
    if (lane < 16) {
        const float ax = sc[S_POSE + 3 * lane], ay = sc[S_POSE + 3 * lane + 1], az = sc[S_POSE + 3 * lane + 2];
        const float px = ax + 1e-8f, py = ay + 1e-8f, pz = az + 1e-8f;
        const float angle = sqrtf(px * px + py * py + pz * pz);
        const float nx = ax / angle, ny = ay / angle, nz = az / angle;
        const float half = angle * 0.5f;
        const float cs = cosf(half), sn = sinf(half);
        float w = cs, x = sn * nx, y = sn * ny, z = sn * nz;
        const float qn = sqrtf(w * w + x * x + y * y + z * z);
        w /= qn; x /= qn; y /= qn; z /= qn;
        const float w2 = w * w, x2 = x * x, y2 = y * y, z2 = z * z;
        const float wx = w * x, wy = w * y, wz = w * z, xy = x * y, xz = x * z, yz = y * z;
        float *r = sc + S_ROT + 9 * lane;
        r[0] = w2 + x2 - y2 - z2; r[1] = 2 * xy - 2 * wz;    r[2] = 2 * wy + 2 * xz;
        r[3] = 2 * wz + 2 * xy;   r[4] = w2 - x2 + y2 - z2;  r[5] = 2 * yz - 2 * wx;
        r[6] = 2 * xz - 2 * wy;   r[7] = 2 * wx + 2 * yz;    r[8] = w2 - x2 - y2 + z2;
    }
    // -- rest joints: J_regressor @ (template + shapedirs beta) is affine in beta
    //    (manolayer.py:181-184; SURVEY.md A2 iii)
    if (lane < 48) {
        float a = 0.f;
#pragma unroll
        for (int k = 0; k < 10; ++k) a = fmaf(tb[JSD + lane * 10 + k], bcast(det, 3 + k), a);
        sc[S_JR + lane] = tb[JT + lane] + a;
    }
    wave_sync();
    // -- kinematic chain root -> 3 levels, one finger per lane (manolayer.py:193-229)
    if (lane < 5) {
        float PR[9], Pt[3];
#pragma unroll
        for (int e = 0; e < 9; ++e) PR[e] = sc[S_ROT + e];
#pragma unroll
        for (int c = 0; c < 3; ++c) Pt[c] = sc[S_JR + c];
        if (lane == 0) {
#pragma unroll
            for (int e = 0; e < 9; ++e) { sc[S_G + e] = PR[e]; sc[S_GR + e] = PR[e]; }
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                sc[S_G + 9 + c] = Pt[c];
                sc[S_GR + 9 + c] = Pt[c] - (PR[3 * c] * Pt[0] + PR[3 * c + 1] * Pt[1] + PR[3 * c + 2] * Pt[2]);
            }
        }
        int parent = 0;
#pragma unroll
        for (int lvl = 0; lvl < 3; ++lvl) {
            const int j = 1 + 3 * lane + lvl;
            float Rj[9], rel[3], jr[3], CR[9], Ct[3];
#pragma unroll
            for (int e = 0; e < 9; ++e) Rj[e] = sc[S_ROT + 9 * j + e];
#pragma unroll
            for (int c = 0; c < 3; ++c) { jr[c] = sc[S_JR + 3 * j + c]; rel[c] = jr[c] - sc[S_JR + 3 * parent + c]; }
#pragma unroll
            for (int a = 0; a < 3; ++a) {
#pragma unroll
                for (int b = 0; b < 3; ++b)
                    CR[3 * a + b] = PR[3 * a] * Rj[b] + PR[3 * a + 1] * Rj[3 + b] + PR[3 * a + 2] * Rj[6 + b];
                Ct[a] = PR[3 * a] * rel[0] + PR[3 * a + 1] * rel[1] + PR[3 * a + 2] * rel[2] + Pt[a];
            }
#pragma unroll
            for (int e = 0; e < 9; ++e) { sc[S_G + 12 * j + e] = CR[e]; sc[S_GR + 12 * j + e] = CR[e]; PR[e] = CR[e]; }
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                sc[S_G + 12 * j + 9 + c] = Ct[c];
                // subtract the rest-pose joint (manolayer.py:231-234)
                sc[S_GR + 12 * j + 9 + c] = Ct[c] - (CR[3 * c] * jr[0] + CR[3 * c + 1] * jr[1] + CR[3 * c + 2] * jr[2]);
                Pt[c] = Ct[c];
            }
            parent = j;
        }
    }
    // -- pose-corrective blend of the 5 fingertip vertices (manolayer.py:187-188),
    //    135 terms split over 4 lane groups
    {
        const int q = lane >> 4, tc = lane & 15, tcc = tc < 15 ? tc : 14;
        const int k0 = q * 34, k1 = (k0 + 34 < 135) ? k0 + 34 : 135;
        float part = 0.f;
        for (int k = k0; k < k1; ++k) {
            const int e = k % 9;
            const float pm = sc[S_ROT + 9 + k] - ((e == 0 || e == 4 || e == 8) ? 1.f : 0.f);
            part = fmaf(tb[TIP_PD + tcc * 135 + k], pm, part);
        }
        part += __shfl_xor(part, 16, 64);
        part += __shfl_xor(part, 32, 64);
        float shaped = 0.f;
#pragma unroll
        for (int k = 0; k < 10; ++k) shaped = fmaf(tb[TIP_SD + tcc * 10 + k], bcast(det, 3 + k), shaped);
        if (lane < 15) sc[S_TIPV + lane] = (shaped + tb[TIP_T + lane]) + part;
    }
    wave_sync();
    // -- skin the tips: T = sum_j w_j Gr_j ; v' = T [v;1]   (manolayer.py:236-246)
    if (lane < 15) {
        const int tip = lane / 3, c = lane % 3;
        float T0 = 0.f, T1 = 0.f, T2 = 0.f, T3 = 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const float w = tb[TIP_W + tip * 16 + j];
            const float *g = sc + S_GR + 12 * j;
            T0 = fmaf(g[3 * c], w, T0); T1 = fmaf(g[3 * c + 1], w, T1); T2 = fmaf(g[3 * c + 2], w, T2);
            T3 = fmaf(g[9 + c], w, T3);
        }
        const float *v = sc + S_TIPV + 3 * tip;
        sc[S_PRE + 48 + lane] = T0 * v[0] + T1 * v[1] + T2 * v[2] + T3;
    }
    if (lane < 48) sc[S_PRE + lane] = sc[S_G + 12 * (lane / 3) + 9 + lane % 3];
    wave_sync();
    // -- reorder, centre on joint 9, metres -> mm (manolayer.py:260-273, ManoLayer.py:54-56)
    RowOut o;
    const int c3 = lane % 3;
    const int k21 = lane < 63 ? lane / 3 : 20;
    const int src = kJointReorder[kFreihand2Rhd[k21]];
    o.center_c = sc[S_PRE + 3 * kCenterPre + c3];
    const float J = (sc[S_PRE + 3 * src + c3] - o.center_c) * 1000.f;
    if (lane < 63) sc[S_J21 + lane] = J;
    wave_sync();
    // -- root-relative, bone-length normalised (hand/utils.py:46-66)
    o.root_c = sc[S_J21 + 3 * kRootIdx + c3];
    const float d0 = sc[S_J21 + 3 * kNormIdx] - sc[S_J21 + 3 * kRootIdx];
    const float d1 = sc[S_J21 + 3 * kNormIdx + 1] - sc[S_J21 + 3 * kRootIdx + 1];
    const float d2 = sc[S_J21 + 3 * kNormIdx + 2] - sc[S_J21 + 3 * kRootIdx + 2];
    o.bone = sqrtf(d0 * d0 + d1 * d1 + d2 * d2);
    o.xyz = (J - o.root_c) / o.bone;
    return o;
}

}}  // namespace mhe::mano

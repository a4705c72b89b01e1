// MANO forward kinematics, fingertip / full linear-blend skinning, projection and
// the per-hypothesis log-likelihood + priors, one 64-lane wavefront per hypothesis.
//
// Reference arithmetic being replaced (all fp32):
//   hand/manopth/manolayer.py:131-273, rodrigues_layer.py:15-54, tensutils.py:6-22,
//   hand/ManoLayer.py:45-60,150-165, hand/utils.py:46-66,
//   hand/network.py:155-165,233-258,455-558,612-667,703-717,787-788.
//
// HBM view (loss pass): reads 180 B (th45) per row + per-image det/crop_uv/vis,
// writes <= 0.6 KB per row; the 19.5 KB of joint-section tables live in LDS.
#include "mano_joint_pass.h"

namespace mhe { namespace mano {

__global__ __launch_bounds__(256) void mano_joints_kernel(
    const float *__restrict__ th45_g, const float *__restrict__ det_g, const float *__restrict__ crop_uv,
    const float *__restrict__ vis, const float *__restrict__ tables,
    float *__restrict__ z_o, float *__restrict__ xyz_o, float *__restrict__ uv_o, float *__restrict__ terms_o,
    float *__restrict__ logp_o, float *__restrict__ norms_o, float *__restrict__ jmm_o,
    int R, int B, float lap_b, float th45_alpha, int inv_norm, float image_size) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *tb = smem;
    for (int i = threadIdx.x; i < JOINT_FLOATS / 4; i += 256)
        reinterpret_cast<float4 *>(tb)[i] = reinterpret_cast<const float4 *>(tables)[i];
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float *sc = smem + JOINT_FLOATS + wave * SCRATCH;
    const float log2b = logf(2.f * lap_b);

    for (int r = blockIdx.x * 4 + wave; r < R; r += gridDim.x * 4) {
        const int b = r % B;
        const float th45 = lane < 45 ? th45_g[(size_t)r * 45 + lane] : 0.f;
        const float det = lane < 16 ? det_g[b * 16 + lane] : 0.f;
        const RowOut o = joint_pass(tb, sc, lane, th45, det);

        if (xyz_o && lane < 63) xyz_o[(size_t)r * 63 + lane] = o.xyz;
        if (jmm_o && lane < 63) jmm_o[(size_t)r * 63 + lane] = sc[S_J21 + lane];
        // -- orthographic projection (hand/network.py:497-514, ManoLayer.py:150-165)
        const float s_cam = expf(bcast(det, 13));
        const float t_cam = (lane & 1) ? bcast(det, 15) : bcast(det, 14);
        const int lu = lane < 42 ? lane : 41;
        float uv = s_cam * __shfl(o.xyz, 3 * (lu >> 1) + (lu & 1), 64) + t_cam;
        if (inv_norm) uv = (uv + 1.f) / 2.f * image_size;
        if (uv_o && lane < 42) uv_o[(size_t)r * 42 + lane] = uv;

        if (z_o) {      // z = [th3 th45 bt logs t]  (hand/network.py:703-717)
            const float a = __shfl(th45, lane >= 3 ? lane - 3 : 0, 64);
            const float d = __shfl(det, lane >= 45 ? lane - 45 : 0, 64);
            if (lane < 61) z_o[(size_t)r * 61 + lane] = lane < 3 ? det : (lane < 48 ? a : d);
        }
        // |theta|, |beta|  (hand/network.py:787-788)
        const float th_sq = wave_sum((lane < 45 ? th45 * th45 : 0.f) + (lane < 3 ? det * det : 0.f));
        const float bt_sq = wave_sum((lane >= 3 && lane < 13) ? det * det : 0.f);
        if (norms_o && lane == 0) { norms_o[(size_t)r * 2] = sqrtf(th_sq); norms_o[(size_t)r * 2 + 1] = sqrtf(bt_sq); }

        if (terms_o || logp_o) {
            // visibility-masked Laplace over the 42 projected coordinates (hand/network.py:255-257)
            float lt = 0.f;
            if (lane < 42) {
                const float y = crop_uv[b * 42 + lane];
                const float w = vis[b * 21 + (lane >> 1)];
                const float e = -(fmaxf(fabsf(y - uv) - 1e-4f, 0.f) + 1e-4f) / lap_b - log2b;
                lt = (w == 1.f) ? e : 0.f;
            }
            const float lp_uv = wave_sum(lt);
            // soft box on th45 (+-2) and beta (+-0.03), soft ball on th3 (pi)  (hand/network.py:155-163,429-435)
            float v45 = lane < 45 ? fmaxf(fabsf(th45) / 2.f - 1.f, 0.f) : 0.f;
            const float lp_45 = -wave_sum(th45_alpha * v45 * v45);
            float vbt = (lane >= 3 && lane < 13) ? fmaxf(fabsf(det) / 0.03f - 1.f, 0.f) : 0.f;
            const float lp_bt = -wave_sum(50.f * vbt * vbt);
            const float r3 = sqrtf(bcast(det, 0) * bcast(det, 0) + bcast(det, 1) * bcast(det, 1) + bcast(det, 2) * bcast(det, 2));
            const float v3 = fmaxf(r3 / 3.14159265358979323846f - 1.f, 0.f);
            const float lp_3 = -5.f * v3 * v3;
            if (lane == 0) {
                if (terms_o) {
                    float4 t4 = make_float4(lp_uv, lp_3, lp_45, lp_bt);
                    reinterpret_cast<float4 *>(terms_o)[r] = t4;
                }
                if (logp_o) logp_o[r] = ((lp_uv + lp_3) + lp_45) + lp_bt;
            }
        }
    }
}

// ---------------------------------------------------------------------------
// The same pass with FOUR hypotheses per wavefront (round 3): a hypothesis owns a 16-lane group, its 16 joints sit one per lane
// (Rodrigues, rest joints), its 5 fingers on lanes 0-4 (kinematic chain), its 15 fingertip coordinates on lanes 0-14, the 45 / 48 /
// 63 / 42-wide stages take 3-4 elements per lane.  The one-hypothesis-per-wave kernel above issues ~1,500 wave instructions per
// hypothesis, most of them with 5-16 of the 64 lanes active: at R = 16,384 that is VALU-issue-bound (51 us, 2.5 % of the HBM
// roofline of its 10 MB).  Here every instruction carries four hypotheses.  Per-hypothesis operands that the one-wave form broadcast
// with v_readlane (th45, det) come from the group's LDS scratch instead.  Same arithmetic, same fmaf chains per output; the
// 42 / 45 / 10-term reductions are summed per lane and folded over the group's 16 lanes (different association than the 64-lane
// butterflies: last-bit differences).
constexpr int S16_TH = SCRATCH, S16_DET = SCRATCH + 48, S16_XYZ = SCRATCH + 64;
constexpr int SCRATCH16 = SCRATCH + 144;     // 912 floats: 16 banks past a multiple of 64, so the four groups of a wave sit in four different bank quarters

__device__ __forceinline__ float group_sum16(float v) {
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__global__ __launch_bounds__(256) void mano_joints16_kernel(
    const float *__restrict__ th45_g, const float *__restrict__ det_g, const float *__restrict__ crop_uv,
    const float *__restrict__ vis, const float *__restrict__ tables,
    float *__restrict__ z_o, float *__restrict__ xyz_o, float *__restrict__ uv_o, float *__restrict__ terms_o,
    float *__restrict__ logp_o, float *__restrict__ norms_o, float *__restrict__ jmm_o, float *__restrict__ ws_o,
    int R, int B, float lap_b, float th45_alpha, int inv_norm, float image_size) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *tb = smem;
    for (int i = threadIdx.x; i < JOINT_FLOATS / 4; i += 256)
        reinterpret_cast<float4 *>(tb)[i] = reinterpret_cast<const float4 *>(tables)[i];
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, grp = lane >> 4, sub = lane & 15;
    float *sc = smem + JOINT_FLOATS + (wave * 4 + grp) * SCRATCH16;
    const float log2b = logf(2.f * lap_b);
    const int nrow4 = (R + 3) / 4;                       // row quads; a wave takes quad qd -> rows 4 qd + grp

    for (int qd = blockIdx.x * 4 + wave; qd < nrow4; qd += gridDim.x * 4) {
        const int r_raw = qd * 4 + grp;
        const bool live = r_raw < R;
        const int r = live ? r_raw : R - 1;              // (a dead group re-does the last row and stores nothing)
        const int b = r % B;
        // -- inputs -> the group's scratch
#pragma unroll
        for (int i = 0; i < 3; ++i) { const int k = sub + 16 * i; if (k < 45) sc[S16_TH + k] = th45_g[(size_t)r * 45 + k]; }
        sc[S16_DET + sub] = det_g[b * 16 + sub];
        wave_sync();
        // -- PCA coefficients -> axis-angle (manolayer.py:131-143)
        {   // (the lane's three outputs as three interleaved fmaf chains: one LDS broadcast of th45[k] feeds all of them)
            const int o2 = sub + 32 < 45 ? sub + 32 : 44;
            float a0 = 0.f, a1 = 0.f, a2 = 0.f;
#pragma unroll 9
            for (int k = 0; k < 45; ++k) {
                const float t = sc[S16_TH + k];
                a0 = fmaf(t, tb[COMPS + k * 45 + sub], a0); a1 = fmaf(t, tb[COMPS + k * 45 + sub + 16], a1); a2 = fmaf(t, tb[COMPS + k * 45 + o2], a2);
            }
            sc[S_POSE + 3 + sub] = tb[MEAN + sub] + a0;
            sc[S_POSE + 3 + sub + 16] = tb[MEAN + sub + 16] + a1;
            if (sub + 32 < 45) sc[S_POSE + 3 + sub + 32] = tb[MEAN + o2] + a2;
        }
        if (sub < 3) sc[S_POSE + sub] = sc[S16_DET + sub];
        wave_sync();
        // -- Rodrigues through a unit quaternion, joint = sub (rodrigues_layer.py:43-54, :15-40)
        {
            const float ax = sc[S_POSE + 3 * sub], ay = sc[S_POSE + 3 * sub + 1], az = sc[S_POSE + 3 * sub + 2];
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
            float *rr = sc + S_ROT + 9 * sub;
            rr[0] = w2 + x2 - y2 - z2; rr[1] = 2 * xy - 2 * wz;    rr[2] = 2 * wy + 2 * xz;
            rr[3] = 2 * wz + 2 * xy;   rr[4] = w2 - x2 + y2 - z2;  rr[5] = 2 * yz - 2 * wx;
            rr[6] = 2 * xz - 2 * wy;   rr[7] = 2 * wx + 2 * yz;    rr[8] = w2 - x2 - y2 + z2;
        }
        // -- rest joints: affine in beta (manolayer.py:181-184)
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int e = sub + 16 * i;
            float a = 0.f;
#pragma unroll
            for (int k = 0; k < 10; ++k) a = fmaf(tb[JSD + e * 10 + k], sc[S16_DET + 3 + k], a);
            sc[S_JR + e] = tb[JT + e] + a;
        }
        wave_sync();
        // -- kinematic chain, one finger per lane (manolayer.py:193-229)
        if (sub < 5) {
            float PR[9], Pt[3];
#pragma unroll
            for (int e = 0; e < 9; ++e) PR[e] = sc[S_ROT + e];
#pragma unroll
            for (int c = 0; c < 3; ++c) Pt[c] = sc[S_JR + c];
            if (sub == 0) {
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
                const int j = 1 + 3 * sub + lvl;
                float Rj[9], rel[3], jr[3], CR[9], Ct[3];
#pragma unroll
                for (int e = 0; e < 9; ++e) Rj[e] = sc[S_ROT + 9 * j + e];
#pragma unroll
                for (int c = 0; c < 3; ++c) { jr[c] = sc[S_JR + 3 * j + c]; rel[c] = jr[c] - sc[S_JR + 3 * parent + c]; }
#pragma unroll
                for (int a = 0; a < 3; ++a) {
#pragma unroll
                    for (int bb = 0; bb < 3; ++bb)
                        CR[3 * a + bb] = PR[3 * a] * Rj[bb] + PR[3 * a + 1] * Rj[3 + bb] + PR[3 * a + 2] * Rj[6 + bb];
                    Ct[a] = PR[3 * a] * rel[0] + PR[3 * a + 1] * rel[1] + PR[3 * a + 2] * rel[2] + Pt[a];
                }
#pragma unroll
                for (int e = 0; e < 9; ++e) { sc[S_G + 12 * j + e] = CR[e]; sc[S_GR + 12 * j + e] = CR[e]; PR[e] = CR[e]; }
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    sc[S_G + 12 * j + 9 + c] = Ct[c];
                    sc[S_GR + 12 * j + 9 + c] = Ct[c] - (CR[3 * c] * jr[0] + CR[3 * c + 1] * jr[1] + CR[3 * c + 2] * jr[2]);
                    Pt[c] = Ct[c];
                }
                parent = j;
            }
        }
        // -- pose-corrective blend of the 15 fingertip coordinates (manolayer.py:187-188): 15 joints x 9 rotation entries, three
        //    interleaved partial sums (one per matrix row) so that the 135-term chain is not one dependent fmaf sequence
        {
            const int tcc = sub < 15 ? sub : 14;
            const float *pd = tb + TIP_PD + tcc * 135;
            float p0 = 0.f, p1 = 0.f, p2 = 0.f;
#pragma unroll 5
            for (int j = 0; j < 15; ++j) {
                const float *rt = sc + S_ROT + 9 + 9 * j;
                p0 = fmaf(pd[9 * j], rt[0] - 1.f, p0);     p1 = fmaf(pd[9 * j + 3], rt[3], p1);       p2 = fmaf(pd[9 * j + 6], rt[6], p2);
                p0 = fmaf(pd[9 * j + 1], rt[1], p0);       p1 = fmaf(pd[9 * j + 4], rt[4] - 1.f, p1); p2 = fmaf(pd[9 * j + 7], rt[7], p2);
                p0 = fmaf(pd[9 * j + 2], rt[2], p0);       p1 = fmaf(pd[9 * j + 5], rt[5], p1);       p2 = fmaf(pd[9 * j + 8], rt[8] - 1.f, p2);
            }
            const float part = (p0 + p1) + p2;
            float shaped = 0.f;
#pragma unroll
            for (int k = 0; k < 10; ++k) shaped = fmaf(tb[TIP_SD + tcc * 10 + k], sc[S16_DET + 3 + k], shaped);
            if (sub < 15) sc[S_TIPV + sub] = (shaped + tb[TIP_T + sub]) + part;
        }
        wave_sync();
        // -- skin the tips (manolayer.py:236-246)
        if (sub < 15) {
            const int tip = sub / 3, c = sub % 3;
            float T0 = 0.f, T1 = 0.f, T2 = 0.f, T3 = 0.f;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const float w = tb[TIP_W + tip * 16 + j];
                const float *g = sc + S_GR + 12 * j;
                T0 = fmaf(g[3 * c], w, T0); T1 = fmaf(g[3 * c + 1], w, T1); T2 = fmaf(g[3 * c + 2], w, T2);
                T3 = fmaf(g[9 + c], w, T3);
            }
            const float *v = sc + S_TIPV + 3 * tip;
            sc[S_PRE + 48 + sub] = T0 * v[0] + T1 * v[1] + T2 * v[2] + T3;
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) { const int e = sub + 16 * i; sc[S_PRE + e] = sc[S_G + 12 * (e / 3) + 9 + e % 3]; }
        wave_sync();
        // -- reorder, centre on joint 9, metres -> mm (manolayer.py:260-273, ManoLayer.py:54-56)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = sub + 16 * i;
            if (e < 63) {
                const int c3 = e % 3, src = kJointReorder[kFreihand2Rhd[e / 3]];
                sc[S_J21 + e] = (sc[S_PRE + 3 * src + c3] - sc[S_PRE + 3 * kCenterPre + c3]) * 1000.f;
            }
        }
        wave_sync();
        // -- root-relative, bone-length normalised (hand/utils.py:46-66)
        const float d0 = sc[S_J21 + 3 * kNormIdx] - sc[S_J21 + 3 * kRootIdx];
        const float d1 = sc[S_J21 + 3 * kNormIdx + 1] - sc[S_J21 + 3 * kRootIdx + 1];
        const float d2 = sc[S_J21 + 3 * kNormIdx + 2] - sc[S_J21 + 3 * kRootIdx + 2];
        const float bone = sqrtf(d0 * d0 + d1 * d1 + d2 * d2);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = sub + 16 * i;
            if (e < 63) {
                const float J = sc[S_J21 + e];
                const float xv = (J - sc[S_J21 + 3 * kRootIdx + e % 3]) / bone;
                sc[S16_XYZ + e] = xv;
                if (live && xyz_o) xyz_o[(size_t)r * 63 + e] = xv;
                if (live && jmm_o) jmm_o[(size_t)r * 63 + e] = J;
            }
        }
        // -- what the full-mesh skinning needs of this hypothesis (mhe_mano_decode_f32; the row mano_pose_kernel writes for mhe_mano_verts_f32)
        if (live && ws_o) {
            float *w = ws_o + (size_t)r * WS_STRIDE;
            for (int k = sub; k < 135; k += 16) {
                const int e = k % 9;
                w[WS_PM + k] = sc[S_ROT + 9 + k] - ((e == 0 || e == 4 || e == 8) ? 1.f : 0.f);
            }
            if (sub < 10) w[WS_BT + sub] = sc[S16_DET + 3 + sub];
#pragma unroll
            for (int k = 0; k < 12; ++k) w[WS_GR + sub + 16 * k] = sc[S_GR + sub + 16 * k];
            if (sub < 3) { w[WS_NRM + sub] = sc[S_PRE + 3 * kCenterPre + sub]; w[WS_NRM + 3 + sub] = sc[S_J21 + 3 * kRootIdx + sub]; }
            if (sub == 0) w[WS_NRM + 6] = bone;
        }
        wave_sync();
        // -- orthographic projection (hand/network.py:497-514, ManoLayer.py:150-165) + visibility-masked Laplace (hand/network.py:255-257)
        const float s_cam = expf(sc[S16_DET + 13]);
        float lt = 0.f;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int u = sub + 16 * i;
            if (u < 42) {
                float uv = s_cam * sc[S16_XYZ + 3 * (u >> 1) + (u & 1)] + sc[S16_DET + 14 + (u & 1)];
                if (inv_norm) uv = (uv + 1.f) / 2.f * image_size;
                if (live && uv_o) uv_o[(size_t)r * 42 + u] = uv;
                if (terms_o || logp_o) {
                    const float y = crop_uv[b * 42 + u];
                    const float w = vis[b * 21 + (u >> 1)];
                    const float e = -(fmaxf(fabsf(y - uv) - 1e-4f, 0.f) + 1e-4f) / lap_b - log2b;
                    lt += (w == 1.f) ? e : 0.f;
                }
            }
        }
        if (live && z_o) {       // z = [th3 th45 bt logs t]  (hand/network.py:703-717)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int e = sub + 16 * i;
                if (e < 61) z_o[(size_t)r * 61 + e] = e < 3 ? sc[S16_DET + e] : (e < 48 ? sc[S16_TH + e - 3] : sc[S16_DET + e - 45]);
            }
        }
        // |theta|, |beta| and the soft priors (hand/network.py:787-788, :155-163,429-435)
        float th_sq = 0.f, v45s = 0.f;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int k = sub + 16 * i;
            if (k < 45) {
                const float t = sc[S16_TH + k];
                th_sq += t * t;
                const float v45 = fmaxf(fabsf(t) / 2.f - 1.f, 0.f);
                v45s += th45_alpha * v45 * v45;
            }
        }
        const float dv = sc[S16_DET + sub];
        th_sq += sub < 3 ? dv * dv : 0.f;
        const float bt_l = (sub >= 3 && sub < 13) ? dv * dv : 0.f;
        const float vbt = (sub >= 3 && sub < 13) ? fmaxf(fabsf(dv) / 0.03f - 1.f, 0.f) : 0.f;
        th_sq = group_sum16(th_sq);
        const float bt_sq = group_sum16(bt_l);
        if (live && norms_o && sub == 0) { norms_o[(size_t)r * 2] = sqrtf(th_sq); norms_o[(size_t)r * 2 + 1] = sqrtf(bt_sq); }
        if (terms_o || logp_o) {
            const float lp_uv = group_sum16(lt);
            const float lp_45 = -group_sum16(v45s);
            const float lp_bt = -group_sum16(50.f * vbt * vbt);
            const float t0 = sc[S16_DET], t1 = sc[S16_DET + 1], t2 = sc[S16_DET + 2];
            const float r3 = sqrtf(t0 * t0 + t1 * t1 + t2 * t2);
            const float v3 = fmaxf(r3 / 3.14159265358979323846f - 1.f, 0.f);
            const float lp_3 = -5.f * v3 * v3;
            if (live && sub == 0) {
                if (terms_o) reinterpret_cast<float4 *>(terms_o)[r] = make_float4(lp_uv, lp_3, lp_45, lp_bt);
                if (logp_o) logp_o[r] = ((lp_uv + lp_3) + lp_45) + lp_bt;
            }
        }
        wave_sync();                                     // the next quad's inputs overwrite the scratch
    }
}

// ---------------------------------------------------------------------------
// Full mesh in two launches.
//  (1) mano_pose_kernel: one wavefront per hypothesis runs joint_pass and leaves what skinning needs in a
//      352-float workspace row: pose map (135), beta (10), the 16 skinning transforms (192), centre/root/bone (7).
//  (2) mano_skin_kernel: one thread per vertex, HB hypotheses per workgroup.  Every per-hypothesis quantity
//      is wave-uniform, so it is read with SCALAR loads and enters the FMAs as an SGPR operand; the blend-shape
//      tables (vertex-fastest, coalesced) are fetched once per workgroup and reused from registers across
//      the HB hypotheses (1.5 MB / HB of L2 traffic per hypothesis).  No LDS at all in this kernel: the first
//      version kept the per-hypothesis values in LDS and was LDS-issue bound (1536 broadcast reads per vertex).

__global__ __launch_bounds__(256) void mano_pose_kernel(const float *__restrict__ z_g, const float *__restrict__ tables,
                                                        float *__restrict__ ws, int R) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *tb = smem;
    for (int i = threadIdx.x; i < JOINT_FLOATS / 4; i += 256)
        reinterpret_cast<float4 *>(tb)[i] = reinterpret_cast<const float4 *>(tables)[i];
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float *sc = smem + JOINT_FLOATS + wave * SCRATCH;
    for (int r = blockIdx.x * 4 + wave; r < R; r += gridDim.x * 4) {
        const float *zr = z_g + (size_t)r * 61;
        const float th45 = lane < 45 ? zr[3 + lane] : 0.f;
        // det lane layout [th3 bt logs t] from z = [th3 th45 bt logs t]
        const float det = lane < 3 ? zr[lane] : (lane < 16 ? zr[45 + lane] : 0.f);
        const RowOut o = joint_pass(tb, sc, lane, th45, det);
        float *w = ws + (size_t)r * WS_STRIDE;
        for (int k = lane; k < 135; k += 64) {
            const int e = k % 9;
            w[WS_PM + k] = sc[S_ROT + 9 + k] - ((e == 0 || e == 4 || e == 8) ? 1.f : 0.f);
        }
        if (lane < 10) w[WS_BT + lane] = zr[48 + lane];
        for (int k = lane; k < 192; k += 64) w[WS_GR + k] = sc[S_GR + k];
        if (lane < 3) { w[WS_NRM + lane] = o.center_c; w[WS_NRM + 3 + lane] = o.root_c; }
        if (lane == 0) w[WS_NRM + 6] = o.bone;
        wave_sync();
    }
}

template <int HB>
__global__ __launch_bounds__(256) void mano_skin_kernel(const float *__restrict__ ws, const float *__restrict__ tables,
                                                        float *__restrict__ verts_o, int R, int mm_mode) {
    const int v = blockIdx.x * 256 + threadIdx.x;            // vertex (VP = 832 padded: 4 vertex groups, last partial)
    const int r0 = blockIdx.y * HB;
    const int vc = v < VP ? v : VP - 1;
    const float *Vt = tables + V_T, *Vsd = tables + V_SD, *Vpd = tables + V_PD, *Vw = tables + V_W;
    // wave-uniform row pointers (rows past R are clamped; their results are not stored)
    const float *wrow[HB];
#pragma unroll
    for (int h = 0; h < HB; ++h) wrow[h] = ws + (size_t)(r0 + h < R ? r0 + h : R - 1) * WS_STRIDE;
    float acc[HB][3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float t = Vt[c * VP + vc];
#pragma unroll
        for (int h = 0; h < HB; ++h) acc[h][c] = t;
    }
    // shape blend (manolayer.py:181-183) then pose-corrective blend (:187-188)
#pragma unroll 2
    for (int k = 0; k < 10; ++k) {
        const float s0 = Vsd[(k * 3 + 0) * VP + vc], s1 = Vsd[(k * 3 + 1) * VP + vc], s2 = Vsd[(k * 3 + 2) * VP + vc];
#pragma unroll
        for (int h = 0; h < HB; ++h) {
            const float bk = wrow[h][WS_BT + k];
            acc[h][0] = fmaf(s0, bk, acc[h][0]); acc[h][1] = fmaf(s1, bk, acc[h][1]); acc[h][2] = fmaf(s2, bk, acc[h][2]);
        }
    }
#pragma unroll 5
    for (int k = 0; k < 135; ++k) {
        const float p0 = Vpd[(k * 3 + 0) * VP + vc], p1 = Vpd[(k * 3 + 1) * VP + vc], p2 = Vpd[(k * 3 + 2) * VP + vc];
#pragma unroll
        for (int h = 0; h < HB; ++h) {
            const float m = wrow[h][WS_PM + k];
            acc[h][0] = fmaf(p0, m, acc[h][0]); acc[h][1] = fmaf(p1, m, acc[h][1]); acc[h][2] = fmaf(p2, m, acc[h][2]);
        }
    }
    float w[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) w[j] = Vw[j * VP + vc];
#pragma unroll
    for (int h = 0; h < HB; ++h) {
        // T = sum_j w_j Gr_j ; v' = T [v;1]   (manolayer.py:236-246)
        float T[12];
#pragma unroll
        for (int e = 0; e < 12; ++e) T[e] = 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j)
#pragma unroll
            for (int e = 0; e < 12; ++e) T[e] = fmaf(wrow[h][WS_GR + j * 12 + e], w[j], T[e]);
        const int r = r0 + h;
        if (v < NV && r < R) {
            const float bone = wrow[h][WS_NRM + 6];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float vp = T[3 * c] * acc[h][0] + T[3 * c + 1] * acc[h][1] + T[3 * c + 2] * acc[h][2] + T[9 + c];
                const float mesh = (vp - wrow[h][WS_NRM + c]) * 1000.f;            // manolayer.py:262-273
                verts_o[((size_t)r * NV + v) * 3 + c] = mm_mode ? mesh : (mesh - wrow[h][WS_NRM + 3 + c]) / bone;   // network.py:480
            }
        }
    }
}

// ManoLayer.xyz_from_vertice (hand/ManoLayer.py:141-148,108-139): 16 joints regressed
// from the mesh + 5 tip vertices, FreiHand order, then the RHD reorder (:54-56).
// One workgroup per hypothesis; wave w handles joints w, w+4, ...
__device__ constexpr int kWrapJointMap[16] = {0, 5, 6, 7, 9, 10, 11, 17, 18, 19, 13, 14, 15, 1, 2, 3};   // mano id -> FreiHand id
__device__ constexpr int kWrapTipVert[5] = {744, 320, 443, 555, 672};                                    // FreiHand ids 4,8,12,16,20
__global__ __launch_bounds__(256) void regress_joints_kernel(const float *__restrict__ verts, const float *__restrict__ tables,
                                                             float *__restrict__ joints, int R) {
    __shared__ float kp[21 * 3];
    const int r = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const float *v = verts + (size_t)r * NV * 3;
    for (int j = wave; j < 16; j += 4) {
        float a0 = 0.f, a1 = 0.f, a2 = 0.f;
        for (int i = lane; i < NV; i += 64) {
            const float w = tables[V_JR + j * VP + i];
            a0 = fmaf(v[3 * i], w, a0); a1 = fmaf(v[3 * i + 1], w, a1); a2 = fmaf(v[3 * i + 2], w, a2);
        }
        a0 = wave_sum(a0); a1 = wave_sum(a1); a2 = wave_sum(a2);
        if (lane == 0) { float *o = kp + 3 * kWrapJointMap[j]; o[0] = a0; o[1] = a1; o[2] = a2; }
    }
    if (threadIdx.x < 15) kp[3 * (4 + 4 * (threadIdx.x / 3)) + threadIdx.x % 3] = v[3 * kWrapTipVert[threadIdx.x / 3] + threadIdx.x % 3];
    __syncthreads();
    if (threadIdx.x < 63) joints[(size_t)r * 63 + threadIdx.x] = kp[3 * kFreihand2Rhd[threadIdx.x / 3] + threadIdx.x % 3];
}

__global__ void elbo_reduce_kernel(const float *__restrict__ lp_rows, const float *__restrict__ lq_rows,
                                   float *__restrict__ q_log_p, float *__restrict__ h_o, float *__restrict__ log_p,
                                   int N, int B) {
    // one wavefront per image: lanes stride over the N hypotheses, LDS-free wave reduce
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (wave >= B) return;
    float a = 0.f, q = 0.f;
    for (int n = lane; n < N; n += 64) {
        a += lp_rows[(size_t)n * B + wave];
        q += lq_rows ? -lq_rows[(size_t)n * B + wave] : 0.f;
    }
    a = wave_sum(a) / (float)N;
    q = wave_sum(q) / (float)N;
    if (lane == 0) {
        if (q_log_p) q_log_p[wave] = a;
        if (h_o) h_o[wave] = q;
        if (log_p) log_p[wave] = q + a;
    }
}

}}  // namespace mhe::mano

using namespace mhe;

extern "C" size_t mhe_mano_table_floats(void) { return mano::TOTAL_FLOATS; }

static int joints_launch(const float *th45, const float *det, const float *crop_uv, const float *vis,
                         const float *tables, float *z, float *xyz, float *uv, float *terms, float *log_p,
                         float *norms, float *joints_mm, float *ws_rows, int R, int B, float laplace_b, float th45_alpha,
                         int inv_norm, float image_size, void *stream) {
    MHE_REQUIRE(th45 && det && tables, "mhe_mano_joints_f32: null input");
    MHE_REQUIRE(R > 0 && B > 0 && R % B == 0, "mhe_mano_joints_f32: R=%d must be a positive multiple of B=%d", R, B);
    MHE_REQUIRE(!(terms || log_p) || (crop_uv && vis), "mhe_mano_joints_f32: likelihood outputs need crop_uv and vis");
    MHE_REQUIRE(laplace_b > 0.f, "mhe_mano_joints_f32: laplace_b must be > 0");
    static const int four = getenv("MHE_MANO_FOUR") ? atoi(getenv("MHE_MANO_FOUR")) : 1;      // 0: the one-hypothesis-per-wave kernel (A/B runs)
    if (four) {
        const int quads = (R + 3) / 4, wgs = (quads + 3) / 4;
        const int blocks = wgs < 512 ? wgs : 512;            // two workgroups per CU (67 KiB of LDS each), every wave walks its share of the row quads
        const size_t lds = (mano::JOINT_FLOATS + 16 * mano::SCRATCH16) * sizeof(float);
        hipLaunchKernelGGL(mano::mano_joints16_kernel, dim3(blocks), dim3(256), lds, (hipStream_t)stream, th45, det, crop_uv,
                           vis, tables, z, xyz, uv, terms, log_p, norms, joints_mm, ws_rows, R, B, laplace_b, th45_alpha, inv_norm, image_size);
        return check_launch("mano_joints16_kernel");
    }
    MHE_REQUIRE(!ws_rows, "mhe_mano_decode_f32 runs on the four-hypotheses-per-wave kernel (MHE_MANO_FOUR=1)");
    const int blocks = (R + 3) / 4 < 2048 ? (R + 3) / 4 : 2048;
    const size_t lds = (mano::JOINT_FLOATS + 4 * mano::SCRATCH) * sizeof(float);
    hipLaunchKernelGGL(mano::mano_joints_kernel, dim3(blocks), dim3(256), lds, (hipStream_t)stream, th45, det, crop_uv,
                       vis, tables, z, xyz, uv, terms, log_p, norms, joints_mm, R, B, laplace_b, th45_alpha, inv_norm,
                       image_size);
    return check_launch("mano_joints_kernel");
}

size_t mhe_mano_skin_split_floats();
int mhe_mano_skin_mfma(const float *ws_rows, const float *tables, float *split, float *verts, int R, int mm_mode, hipStream_t stream);      // mano_skin.hip

// [bf16 pieces of the vertex tables (mano_skin.hip) | 352 floats per hypothesis]
extern "C" int mhe_mano_joints_f32(const float *th45, const float *det, const float *crop_uv, const float *vis,
                                   const float *tables, float *z, float *xyz, float *uv, float *terms, float *log_p,
                                   float *norms, float *joints_mm, int R, int B, float laplace_b, float th45_alpha,
                                   int inv_norm, float image_size, void *stream) {
    return joints_launch(th45, det, crop_uv, vis, tables, z, xyz, uv, terms, log_p, norms, joints_mm, nullptr, R, B, laplace_b, th45_alpha,
                         inv_norm, image_size, stream);
}

extern "C" size_t mhe_mano_verts_workspace_floats(int R) { return R > 0 ? mhe_mano_skin_split_floats() + (size_t)R * mano::WS_STRIDE : 0; }

extern "C" int mhe_mano_verts_f32(const float *z, const float *tables, float *verts, float *workspace, int R, int mm_mode,
                                  void *stream) {
    MHE_REQUIRE(z && tables && verts && workspace, "mhe_mano_verts_f32: null pointer");
    MHE_REQUIRE(R > 0, "mhe_mano_verts_f32: R=%d", R);
    constexpr int HB = 16;
    const size_t lds = (mano::JOINT_FLOATS + 4 * mano::SCRATCH) * sizeof(float);
    const int pb = (R + 3) / 4 < 2048 ? (R + 3) / 4 : 2048;
    float *rows = workspace + mhe_mano_skin_split_floats();
    hipLaunchKernelGGL(mano::mano_pose_kernel, dim3(pb), dim3(256), lds, (hipStream_t)stream, z, tables, rows, R);
    if (int rc = check_launch("mano_pose_kernel")) return rc;
    // 1 (default): both products on the matrix cores from bf16 pieces (mano_skin.hip); 0: the round-1 kernel, one thread per vertex (A/B runs)
    static const int mfma = getenv("MHE_MANO_SKIN_MFMA") ? atoi(getenv("MHE_MANO_SKIN_MFMA")) : 1;
    if (mfma) return mhe_mano_skin_mfma(rows, tables, workspace, verts, R, mm_mode, (hipStream_t)stream);
    hipLaunchKernelGGL(mano::mano_skin_kernel<HB>, dim3((mano::VP + 255) / 256, (R + HB - 1) / HB), dim3(256), 0,
                       (hipStream_t)stream, rows, tables, verts, R, mm_mode);
    return check_launch("mano_skin_kernel");
}

// joints + full mesh of the same hypotheses in two launches: the joint pass leaves the skinning operands of every hypothesis in the workspace
// (what mano_pose_kernel recomputes from z for mhe_mano_verts_f32: 135 us at 51,200 hypotheses), the matrix-core skinning follows
extern "C" int mhe_mano_decode_f32(const float *th45, const float *det, const float *crop_uv, const float *vis,
                                   const float *tables, float *z, float *xyz, float *uv, float *terms, float *log_p,
                                   float *norms, float *joints_mm, float *verts, float *workspace, int R, int B, float laplace_b,
                                   float th45_alpha, int inv_norm, float image_size, int mm_mode, void *stream) {
    MHE_REQUIRE(verts && workspace, "mhe_mano_decode_f32: null pointer");
    float *rows = workspace + mhe_mano_skin_split_floats();
    if (int rc = joints_launch(th45, det, crop_uv, vis, tables, z, xyz, uv, terms, log_p, norms, joints_mm, rows, R, B, laplace_b, th45_alpha,
                               inv_norm, image_size, stream))
        return rc;
    return mhe_mano_skin_mfma(rows, tables, workspace, verts, R, mm_mode, (hipStream_t)stream);
}

extern "C" int mhe_mano_regress_joints_f32(const float *verts, const float *tables, float *joints, int R, void *stream) {
    MHE_REQUIRE(verts && tables && joints && R > 0, "mhe_mano_regress_joints_f32: bad arguments");
    hipLaunchKernelGGL(mano::regress_joints_kernel, dim3(R), dim3(256), 0, (hipStream_t)stream, verts, tables, joints, R);
    return check_launch("regress_joints_kernel");
}

extern "C" int mhe_elbo_reduce_f32(const float *log_p_rows, const float *log_q_rows, float *q_log_p, float *h,
                                   float *log_p, int N, int B, void *stream) {
    MHE_REQUIRE(log_p_rows && N > 0 && B > 0, "mhe_elbo_reduce_f32: bad arguments");
    hipLaunchKernelGGL(mano::elbo_reduce_kernel, dim3((B + 3) / 4), dim3(256), 0, (hipStream_t)stream, log_p_rows,
                       log_q_rows, q_log_p, h, log_p, N, B);
    return check_launch("elbo_reduce_kernel");
}

// HO3D input pipeline from the decoded arrays on: what the reference's CPU workers do per sample in
// hand/dataloader/ho3d_dataloader.py:272-459 (`Generate_ho3d_uv.__getitem__`, helpers :32-199, `compute_st`
// hand/dataloader/rhddataloader.py:237-269), batched on the GPU.  Two launches per batch:
//   targets kernel  one 64-lane workgroup per sample: projection of joints / object, hand + object boxes -> crop window, crop
//                   coordinates, the two visibility passes (9x9 windows over the hand mask and the depth map), normalised pose,
//                   augmentation bookkeeping (rotation matrix, rotated pose, transformed uv), joint re-ordering, compute_st,
//                   and the per-sample geometry the image kernel needs (crop window, inverse affine map);
//   image kernel    one thread per output pixel: inverse affine in OpenCV's 10-bit fixed point (INTER_NEAREST), crop + nearest
//                   resize index arithmetic, border fill, colour noise, ToTensor + Normalize; hand / object masks and depth crop.
// Integer decisions (pixel indices, crop window, visibility) use the same arithmetic types as the reference's numpy code
// (float32 where its arrays are float32, float64 where numpy promotes) so that they come out bit-identical; see oracle/ho3d_ref.py.
// HBM-bound byte work: per sample ~0.9 MB of source pixels touched at most, 1.3 MB written.
#include "common.h"

namespace mhe { namespace ho3d {

constexpr int SH = 480, SW = 640, OUT = 256, NJ = 21, NVH = 778;
constexpr double DEPTH_SCALE = 0.00012498664727900177;           // ho3d_vis_utils.py:463
__constant__ int HO3D2RHD[NJ] = {0, 16, 15, 14, 13, 17, 3, 2, 1, 18, 6, 5, 4, 19, 12, 11, 10, 20, 9, 8, 7};     // ho3d_dataloader.py:17
constexpr int GEOM = 12;                                         // doubles per sample: x1 y1 x2 y2 | inverse affine m[6] | aug flag | -

// no fused multiply-add where the reference rounds after every float32 operation
__device__ __forceinline__ float mulf(float a, float b) { return __fmul_rn(a, b); }
__device__ __forceinline__ float addf(float a, float b) { return __fadd_rn(a, b); }
__device__ __forceinline__ float subf(float a, float b) { return __fsub_rn(a, b); }
__device__ __forceinline__ float divf(float a, float b) { return __fdiv_rn(a, b); }

__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// get_bbox_joints (ho3d_dataloader.py:82-92): [cx - dx, cy - dy, cx + dx, cy + dy] as float32 of a float64 difference
__device__ __forceinline__ void bbox_of(float mnx, float mny, float mxx, float mxy, float factor, float *bb) {
    const int cx = (int)divf(addf(mxx, mnx), 2.f), cy = (int)divf(addf(mxy, mny), 2.f);
    const float dx = divf(mulf(subf(mxx, mnx), factor), 2.f), dy = divf(mulf(subf(mxy, mny), factor), 2.f);
    bb[0] = (float)((double)cx - (double)dx); bb[1] = (float)((double)cy - (double)dy);
    bb[2] = (float)((double)cx + (double)dx); bb[3] = (float)((double)cy + (double)dy);
}

struct TargetArgs {
    const float *joints3d, *mesh, *cam, *obj_rot, *obj_trans, *obj_verts;
    const int *obj_count;
    const unsigned char *seg, *depth_png;
    const double *aug;           // [B][7]: pn0 pn1 pn2 scale angle tx ty, or null (evaluation mode)
    float *crop_uv, *vis, *original_pose3d, *verts, *pose3d, *st, *scale, *crop_center, *crop_size, *pose3d_root, *rot_mat_inv, *rot_mat,
          *uvd, *object_verts;
    double *geom;
    int NVmax;
};

__global__ __launch_bounds__(64) void targets_kernel(TargetArgs a) {
    const int b = blockIdx.x, lane = threadIdx.x;
    __shared__ float J[NJ][3], U[NJ][3], Nm[NJ][3];      // joints (mm, flipped), uvd, normalised pose
    __shared__ double UV[NJ][2];
    __shared__ float visf[NJ];
    __shared__ float sc[8];
    __shared__ int ci[2];
    const float *K = a.cam + (size_t)b * 9;
    const float fx = K[0], fy = K[4], fu = K[2], fv = K[5];
    // ---- joints: mm, uvd (xyz2uvd :72-80: flip y and z, pinhole; all float32), coord_change
    if (lane < NJ) {
        const float *p = a.joints3d + ((size_t)b * NJ + lane) * 3;
        const float x = mulf(p[0], 1000.f), y = -mulf(p[1], 1000.f), z = -mulf(p[2], 1000.f);
        J[lane][0] = x; J[lane][1] = y; J[lane][2] = z;
        U[lane][0] = addf(divf(mulf(x, fx), z), fu);
        U[lane][1] = addf(divf(mulf(y, fy), z), fv);
        U[lane][2] = z;
    }
    // ---- hand mesh: mm + coord_change
    for (int v = lane; v < NVH; v += 64) {
        const float *p = a.mesh + ((size_t)b * NVH + v) * 3;
        float *o = a.verts + ((size_t)b * NVH + v) * 3;
        o[0] = mulf(p[0], 1000.f); o[1] = -mulf(p[1], 1000.f); o[2] = -mulf(p[2], 1000.f);
    }
    // ---- object: Rodrigues (float64), pose, projection (float64 -> float32), 2D extent
    double R[9];
    {
        const float *r = a.obj_rot + (size_t)b * 3;
        const double rx = r[0], ry = r[1], rz = r[2], th = sqrt(rx * rx + ry * ry + rz * rz);
        if (th < 2.220446049250313e-16) {
            R[0] = R[4] = R[8] = 1; R[1] = R[2] = R[3] = R[5] = R[6] = R[7] = 0;
        } else {
            const double c = cos(th), s = sin(th), c1 = 1 - c, kx = rx / th, ky = ry / th, kz = rz / th;
            R[0] = c + c1 * kx * kx; R[1] = c1 * kx * ky - s * kz; R[2] = c1 * kx * kz + s * ky;
            R[3] = c1 * ky * kx + s * kz; R[4] = c + c1 * ky * ky; R[5] = c1 * ky * kz - s * kx;
            R[6] = c1 * kz * kx - s * ky; R[7] = c1 * kz * ky + s * kx; R[8] = c + c1 * kz * kz;
        }
    }
    float mnx = INFINITY, mny = INFINITY, mxx = -INFINITY, mxy = -INFINITY;
    {
        const float *t = a.obj_trans + (size_t)b * 3;
        const int n = a.obj_count[b];
        for (int v = lane; v < n; v += 64) {
            const float *p = a.obj_verts + ((size_t)b * a.NVmax + v) * 3;
            const double px = p[0], py = p[1], pz = p[2];
            const double X = (px * R[0] + py * R[1] + pz * R[2] + (double)t[0]) * 1000.0;
            const double Y = -((px * R[3] + py * R[4] + pz * R[5] + (double)t[1]) * 1000.0);
            const double Z = -((px * R[6] + py * R[7] + pz * R[8] + (double)t[2]) * 1000.0);
            const float u = (float)(X * (double)fx / Z + (double)fu), w = (float)(Y * (double)fy / Z + (double)fv);
            mnx = fminf(mnx, u); mxx = fmaxf(mxx, u); mny = fminf(mny, w); mxy = fmaxf(mxy, w);
            if (a.object_verts) {
                float *o = a.object_verts + ((size_t)b * a.NVmax + v) * 3;
                o[0] = (float)X; o[1] = (float)Y; o[2] = (float)Z;
            }
        }
        mnx = wave_min(mnx); mny = wave_min(mny); mxx = wave_max(mxx); mxy = wave_max(mxy);
    }
    __syncthreads();
    float hmnx = lane < NJ ? U[lane][0] : INFINITY, hmny = lane < NJ ? U[lane][1] : INFINITY;
    float hmxx = lane < NJ ? U[lane][0] : -INFINITY, hmxy = lane < NJ ? U[lane][1] : -INFINITY;
    hmnx = wave_min(hmnx); hmny = wave_min(hmny); hmxx = wave_max(hmxx); hmxy = wave_max(hmxy);
    if (lane == 0) {
        // fuse_bbox (:94-108): union of the hand box (factor 1.5) and the object box, clamped (x against 480, y against 640, as written there)
        float bh[4], bo[4];
        bbox_of(hmnx, hmny, hmxx, hmxy, 1.5f, bh);
        bbox_of(mnx, mny, mxx, mxy, 1.0f, bo);
        float lx = fminf(fminf(bh[0], bh[2]), fminf(bo[0], bo[2])), ly = fminf(fminf(bh[1], bh[3]), fminf(bo[1], bo[3]));
        float hx = fmaxf(fmaxf(bh[0], bh[2]), fmaxf(bo[0], bo[2])), hy = fmaxf(fmaxf(bh[1], bh[3]), fmaxf(bo[1], bo[3]));
        lx = fmaxf(0.f, lx); ly = fmaxf(0.f, ly);
        hx = fminf(hx, (float)SH); hy = fminf(hy, (float)SW);
        const int cx = (int)divf(addf(hx, lx), 2.f), cy = (int)divf(addf(hy, ly), 2.f);
        const float size = divf(fmaxf(subf(hx, lx), subf(hy, ly)), 2.f);
        ci[0] = cx; ci[1] = cy; sc[0] = size;
        a.crop_center[b * 2] = (float)cx; a.crop_center[b * 2 + 1] = (float)cy; a.crop_size[b] = size;
        double *g = a.geom + (size_t)b * GEOM;          // imcrop window (:110-114): round half to even of int - float32 in float64
        g[0] = rint((double)cx - (double)size); g[1] = rint((double)cy - (double)size);
        g[2] = rint((double)cx + (double)size); g[3] = rint((double)cy + (double)size);
    }
    __syncthreads();
    const float size = sc[0];
    const float factor = divf(256.f, mulf(size, 2.f));
    // ---- crop coordinates, first visibility pass (:367-384), normalised pose (:154-160)
    if (lane < NJ) {
        UV[lane][0] = (double)(float)(((double)U[lane][0] - (double)ci[0] + (double)size) * (double)factor);
        UV[lane][1] = (double)(float)(((double)U[lane][1] - (double)ci[1] + (double)size) * (double)factor);
        const int u0 = (int)U[lane][0], v0 = (int)U[lane][1];
        const double d = (double)U[lane][2];
        const unsigned char *seg = a.seg + (size_t)b * 120 * 160 * 3, *dp = a.depth_png + (size_t)b * SH * SW * 3;
        bool flag = false;
        for (int u = u0 - 4; u <= u0 + 4 && !flag; ++u)
            for (int v = v0 - 4; v <= v0 + 4; ++v) {
                if (u < 0 || v < 0 || u >= SW || v >= SH) continue;
                if (seg[((v >> 2) * 160 + (u >> 2)) * 3 + 2] <= 200) continue;
                const unsigned char *q = dp + ((size_t)v * SW + u) * 3;
                const double depth = (double)((unsigned)q[2] + (unsigned)q[1] * 256u) * DEPTH_SCALE;
                if (d - depth * 1000.0 < 40.0) { flag = true; break; }
            }
        visf[lane] = flag ? 1.f : 0.f;
    }
    __syncthreads();
    if (lane < NJ) {
        const float rx = subf(J[lane][0], J[4][0]), ry = subf(J[lane][1], J[4][1]), rz = subf(J[lane][2], J[4][2]);
        const float bx = subf(subf(J[4][0], J[4][0]), subf(J[5][0], J[4][0])), by = subf(subf(J[4][1], J[4][1]), subf(J[5][1], J[4][1])),
                    bz = subf(subf(J[4][2], J[4][2]), subf(J[5][2], J[4][2]));
        const float bone = sqrtf(addf(addf(mulf(bx, bx), mulf(by, by)), mulf(bz, bz)));
        Nm[lane][0] = divf(rx, bone); Nm[lane][1] = divf(ry, bone); Nm[lane][2] = divf(rz, bone);
        if (lane == 0) { sc[1] = bone; a.scale[b] = divf(bone, 1000.f); }
    }
    // ---- augmentation bookkeeping (:162-189)
    double rot[6] = {1, 0, 0, 0, 1, 0};
    const bool aug = a.aug != nullptr;
    if (aug) {
        const double *q = a.aug + (size_t)b * 7;
        const double ang = (-180.0 * q[4] / 3.141592653589793) * 3.141592653589793 / 180.0;       // getRotationMatrix2D takes degrees
        const double al = cos(ang) * q[3], be = sin(ang) * q[3];
        rot[0] = al; rot[1] = be; rot[2] = (1 - al) * 128.0 - be * 128.0 + q[5];
        rot[3] = -be; rot[4] = al; rot[5] = be * 128.0 + (1 - al) * 128.0 + q[6];
        if (lane < NJ) {
            // rotate() (:140-152) about the origin: python-float factors times float32 arrays = float32 products
            const float c32 = (float)cos(q[4]), s32 = (float)sin(q[4]);
            const float px = Nm[lane][0], py = Nm[lane][1];
            Nm[lane][0] = subf(mulf(c32, px), mulf(s32, py));
            Nm[lane][1] = addf(mulf(s32, px), mulf(c32, py));
            const double u = UV[lane][0], v = UV[lane][1];
            UV[lane][0] = rot[0] * u + rot[1] * v + rot[2];
            UV[lane][1] = rot[3] * u + rot[4] * v + rot[5];
        }
    }
    __syncthreads();
    // ---- second visibility pass (:396-409): some pixel of the 9x9 window lies inside the crop
    if (lane < NJ) {
        const double u = UV[lane][0], v = UV[lane][1];
        bool any = false;
        for (int du = -4; du <= 4; ++du)
            for (int dv = -4; dv <= 4; ++dv) {
                // float32 + int stays float32 without augmentation (uv is a float32 array then), float64 with it
                const double uu = aug ? u + du : (double)addf((float)u, (float)du), vv = aug ? v + dv : (double)addf((float)v, (float)dv);
                if (!(uu > 255 || vv > 255 || vv < 0 || uu < 0)) any = true;
            }
        if (!any) visf[lane] = 0.f;
    }
    __syncthreads();
    // ---- outputs in the RHD joint order, uv to [-1, 1]
    if (lane < NJ) {
        const int s = HO3D2RHD[lane];
        float cu, cv;
        if (aug) { cu = (float)(UV[s][0] / 256 * 2 - 1); cv = (float)(UV[s][1] / 256 * 2 - 1); }
        else { cu = subf(mulf(divf((float)UV[s][0], 256.f), 2.f), 1.f); cv = subf(mulf(divf((float)UV[s][1], 256.f), 2.f), 1.f); }
        a.crop_uv[(size_t)b * 42 + lane * 2] = cu; a.crop_uv[(size_t)b * 42 + lane * 2 + 1] = cv;
        a.vis[(size_t)b * NJ + lane] = visf[s];
        for (int k = 0; k < 3; ++k) {
            a.original_pose3d[((size_t)b * NJ + lane) * 3 + k] = J[s][k];
            a.pose3d[((size_t)b * NJ + lane) * 3 + k] = Nm[s][k];
        }
        a.uvd[((size_t)b * NJ + lane) * 3] = cu; a.uvd[((size_t)b * NJ + lane) * 3 + 1] = cv; a.uvd[((size_t)b * NJ + lane) * 3 + 2] = Nm[s][2];
        if (lane == 12) for (int k = 0; k < 3; ++k) a.pose3d_root[b * 3 + k] = divf(J[s][k], 1000.f);
    }
    __syncthreads();
    if (lane == 0) {
        // compute_st (rhddataloader.py:237-269, utils.py:502-525) on the re-ordered, normalised uv and pose (order does not matter: sums)
        double t1[2] = {0, 0}, t2[2] = {0, 0};
        double uvn[NJ][2];
        for (int j = 0; j < NJ; ++j) {
            for (int k = 0; k < 2; ++k) {
                uvn[j][k] = aug ? (UV[j][k] / 256 * 2 - 1) : (double)subf(mulf(divf((float)UV[j][k], 256.f), 2.f), 1.f);
                t1[k] += uvn[j][k]; t2[k] += (double)Nm[j][k];
            }
        }
        for (int k = 0; k < 2; ++k) { t1[k] /= NJ; t2[k] /= NJ; }
        double n1 = 0, n2 = 0, M[4] = {0, 0, 0, 0};
        for (int j = 0; j < NJ; ++j) {
            const double a0 = uvn[j][0] - t1[0], a1 = uvn[j][1] - t1[1], b0 = (double)Nm[j][0] - t2[0], b1 = (double)Nm[j][1] - t2[1];
            n1 += a0 * a0 + a1 * a1; n2 += b0 * b0 + b1 * b1;
            M[0] += a0 * b0; M[1] += a0 * b1; M[2] += a1 * b0; M[3] += a1 * b1;
        }
        const double s1 = sqrt(n1) + 1e-8, s2 = sqrt(n2) + 1e-8;
        for (int k = 0; k < 4; ++k) M[k] /= s1 * s2;
        // sum of the singular values of the 2x2 cross-covariance: sqrt(|M|_F^2 + 2 |det M|)
        double s = sqrt(M[0] * M[0] + M[1] * M[1] + M[2] * M[2] + M[3] * M[3] + 2.0 * fabs(M[0] * M[3] - M[1] * M[2]));
        const double tx = -t2[0] / s2 * s * s1 + t1[0], ty = -t2[1] / s2 * s * s1 + t1[1];
        s *= s1 / s2;
        a.st[b * 3] = (float)s; a.st[b * 3 + 1] = (float)tx; a.st[b * 3 + 2] = (float)ty;
        // rot_mat_inv = inv([[L t],[0 1]]^T)[:, :2] = [L^-T ; (-L^-1 t)^T], _rot_mat = L / |L row 0|        (:421-425, :455)
        const double det = rot[0] * rot[4] - rot[1] * rot[3];
        const double i00 = rot[4] / det, i01 = -rot[1] / det, i10 = -rot[3] / det, i11 = rot[0] / det;      // L^-1
        float *ri = a.rot_mat_inv + (size_t)b * 6;
        ri[0] = (float)i00; ri[1] = (float)i10; ri[2] = (float)i01; ri[3] = (float)i11;
        ri[4] = (float)(-(i00 * rot[2] + i01 * rot[5])); ri[5] = (float)(-(i10 * rot[2] + i11 * rot[5]));
        const double nr = sqrt(rot[0] * rot[0] + rot[1] * rot[1]);
        float *rm = a.rot_mat + (size_t)b * 4;
        rm[0] = (float)(rot[0] / nr); rm[1] = (float)(rot[1] / nr); rm[2] = (float)(rot[3] / nr); rm[3] = (float)(rot[4] / nr);
        // inverse map of cv2.warpAffine (imgwarp.cpp), same operation order, for the image kernel
        double *g = a.geom + (size_t)b * GEOM;
        double m[6] = {rot[0], rot[1], rot[2], rot[3], rot[4], rot[5]};
        double D = m[0] * m[4] - m[1] * m[3];
        D = D != 0 ? 1.0 / D : 0.0;
        const double A11 = m[4] * D, A22 = m[0] * D;
        m[0] = A11; m[1] *= -D; m[3] *= -D; m[4] = A22;
        const double b1 = -m[0] * m[2] - m[1] * m[5], b2 = -m[3] * m[2] - m[4] * m[5];
        m[2] = b1; m[5] = b2;
        for (int k = 0; k < 6; ++k) g[4 + k] = m[k];
        g[10] = aug ? 1.0 : 0.0;
    }
}

__global__ __launch_bounds__(256) void image_kernel(const unsigned char *__restrict__ image, const unsigned char *__restrict__ seg,
                                                    const unsigned char *__restrict__ depth_png, const double *__restrict__ geom,
                                                    const double *__restrict__ aug, float *__restrict__ out, unsigned char *__restrict__ hand_mask,
                                                    unsigned char *__restrict__ object_mask, float *__restrict__ depth_out) {
    const int b = blockIdx.y, pix = blockIdx.x * 256 + threadIdx.x;
    const int y = pix >> 8, x = pix & 255;
    const double *g = geom + (size_t)b * GEOM;
    const int x1 = (int)g[0], y1 = (int)g[1], cw = (int)g[2] - x1, ch = (int)g[3] - y1;
    int X = x, Y = y;
    bool inside = true;
    if (g[10] != 0.0) {           // cv2.warpAffine INTER_NEAREST: 10-bit fixed point, round_delta = 512, BORDER_CONSTANT 0
        const long long ad = llrint(g[4] * x * 1024.0), bd = llrint(g[7] * x * 1024.0);
        const long long X0 = llrint((g[5] * y + g[6]) * 1024.0) + 512, Y0 = llrint((g[8] * y + g[9]) * 1024.0) + 512;
        X = (int)((X0 + ad) >> 10); Y = (int)((Y0 + bd) >> 10);
        inside = X >= 0 && X < OUT && Y >= 0 && Y < OUT;
    }
    float rgb[3] = {0.f, 0.f, 0.f}, dep = 0.f;
    unsigned char hm = 0, om = 0;
    if (inside) {
        // cv2.resize INTER_NEAREST of the crop: source index = min(floor(dst * src / dst_size), src - 1)
        int sx = (int)floor((double)X * ((double)cw / 256.0)), sy = (int)floor((double)Y * ((double)ch / 256.0));
        sx = sx < cw - 1 ? sx : cw - 1; sy = sy < ch - 1 ? sy : ch - 1;
        const int u = x1 + sx, v = y1 + sy;
        const bool in_img = u >= 0 && u < SW && v >= 0 && v < SH;
        unsigned char px[3] = {127, 127, 127};
        if (in_img) {
            const unsigned char *p = image + ((size_t)b * SH * SW + (size_t)v * SW + u) * 3;
            px[0] = p[0]; px[1] = p[1]; px[2] = p[2];
            const unsigned char *s = seg + ((size_t)b * 120 * 160 + (v >> 2) * 160 + (u >> 2)) * 3;
            om = s[1] > 200; hm = s[2] > 200;
            const unsigned char *q = depth_png + ((size_t)b * SH * SW + (size_t)v * SW + u) * 3;
            dep = (float)((double)((unsigned)q[2] + (unsigned)q[1] * 256u) * DEPTH_SCALE);
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            unsigned char val = px[c];
            if (aug) {            // rgb_processing (:191-198): per-channel factor, clamped, stored back into the uint8 image
                const double t = fmin(255.0, fmax(0.0, (double)val * aug[(size_t)b * 7 + c]));
                val = (unsigned char)t;
            }
            rgb[c] = (float)val;
        }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c)     // ToTensor, Normalize(0.5, 0.5) in float32
        out[(((size_t)b * 3 + c) * OUT + y) * OUT + x] = __fdiv_rn(__fsub_rn(__fdiv_rn(rgb[c], 255.f), 0.5f), 0.5f);
    hand_mask[((size_t)b * OUT + y) * OUT + x] = hm;
    object_mask[((size_t)b * OUT + y) * OUT + x] = om;
    depth_out[((size_t)b * OUT + y) * OUT + x] = dep;
}

}}  // namespace mhe::ho3d

using namespace mhe;

extern "C" int mhe_ho3d_geom_doubles(void) { return ho3d::GEOM; }

extern "C" int mhe_ho3d_targets(const float *joints3d, const float *mesh, const float *cam, const float *obj_rot, const float *obj_trans,
                                const float *obj_verts, const int *obj_count, int NVmax, const unsigned char *seg,
                                const unsigned char *depth_png, const double *aug, float *crop_uv, float *vis, float *original_pose3d,
                                float *verts, float *pose3d, float *st, float *scale, float *crop_center, float *crop_size,
                                float *pose3d_root, float *rot_mat_inv, float *rot_mat, float *uvd, float *object_verts, double *geom,
                                int B, void *stream) {
    MHE_REQUIRE(joints3d && mesh && cam && obj_rot && obj_trans && obj_verts && obj_count && seg && depth_png, "mhe_ho3d_targets: null input");
    MHE_REQUIRE(crop_uv && vis && original_pose3d && verts && pose3d && st && scale && crop_center && crop_size && pose3d_root && rot_mat_inv &&
                rot_mat && uvd && geom, "mhe_ho3d_targets: null output");
    MHE_REQUIRE(B > 0 && NVmax > 0, "mhe_ho3d_targets: B=%d NVmax=%d", B, NVmax);
    ho3d::TargetArgs a = {joints3d, mesh, cam, obj_rot, obj_trans, obj_verts, obj_count, seg, depth_png, aug, crop_uv, vis, original_pose3d, verts,
                          pose3d, st, scale, crop_center, crop_size, pose3d_root, rot_mat_inv, rot_mat, uvd, object_verts, geom, NVmax};
    hipLaunchKernelGGL(ho3d::targets_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, a);
    return check_launch("ho3d::targets_kernel");
}

extern "C" int mhe_ho3d_images(const unsigned char *image, const unsigned char *seg, const unsigned char *depth_png, const double *geom,
                               const double *aug, float *image_out, unsigned char *hand_mask, unsigned char *object_mask, float *depth_out,
                               int B, void *stream) {
    MHE_REQUIRE(image && seg && depth_png && geom && image_out && hand_mask && object_mask && depth_out, "mhe_ho3d_images: null pointer");
    MHE_REQUIRE(B > 0 && B < 65536, "mhe_ho3d_images: B=%d", B);
    hipLaunchKernelGGL(ho3d::image_kernel, dim3(256, B), dim3(256), 0, (hipStream_t)stream, image, seg, depth_png, geom, aug, image_out, hand_mask,
                       object_mask, depth_out);
    return check_launch("ho3d::image_kernel");
}

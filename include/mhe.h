/*
 * mhe.h - C ABI of the MI355X-native multi-hypothesis hot path (libmhe_hip.so).
 *
 * The reference (GloryyrolG/MHEntropy) has no FFI layer: its boundary for this
 * path is the Python module API (hand/network.py, hand/flows.py,
 * hand/ManoLayer.py, hand/criteria.py).  Each entry point below replaces the
 * arithmetic of one group of those functions; the Python classes in
 * mhentropy_amd/ keep the reference's names/signatures and call in here through
 * ctypes.  Citations are reference file:line.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer (hipMalloc'ed / torch CUDA tensor
 *    storage) unless the name ends in _host; the caller owns all memory, nothing
 *    is allocated or freed inside, workspaces are passed in;
 *  - `stream` is a hipStream_t (pass torch's current stream); all work is
 *    enqueued asynchronously on it, nothing synchronises;
 *  - return 0 on success, non-zero on error (mhe_last_error() gives the text);
 *    no C++ exception crosses the ABI;
 *  - hypothesis rows are SAMPLE-MAJOR like the reference's feat.repeat(N,1)
 *    (hand/network.py:734): row r = n*B + b, R = N*B rows;
 *  - all matrices are dense row-major with the stated shapes.
 */
#ifndef MHE_H
#define MHE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MHE_OK 0
#define MHE_ERR_ARG 1
#define MHE_ERR_LAUNCH 2

enum { MHE_ACT_NONE = 0, MHE_ACT_RELU = 1 };
enum { MHE_F32 = 0, MHE_BF16 = 1 };
enum { MHE_FLOW_FORWARD = 0, MHE_FLOW_INVERSE = 1 };

/* library ---------------------------------------------------------------- */
/* MHE_ABI_VERSION changes whenever a struct layout or an existing signature changes (new entry points alone do not bump it).
 *   1: round 1.   2: mhe_conv_desc gained `tile` and `res_half` (every convolution entry reads them), mhe_conv_wgrad_nhwc
 *   takes the descriptor.   3: statistic accumulators are fixed-point mhe_stat_t words (below) instead of f32; pixel counts are double.
 *   4 (round 5): mhe_bottleneck_tail256_* and the forced convolution variants 5 / 6 / 15 are gone (negative results, pruned).  A caller compiled against another version must not call in: check
 *   mhe_abi_version() == MHE_ABI_VERSION once after loading the library. */
#define MHE_ABI_VERSION 4
int         mhe_abi_version(void);

/* Sharded per-channel accumulators (BatchNorm batch statistics, BatchNorm-reverse sums): 64-bit FIXED-POINT words added with integer
 * atomics, so that a total does not depend on the order in which workgroups arrive (bit-identical runs, eager or graph replay).
 * One unit of C channels = [2 planes][S shards][2 statistics][C] words, S = mhe_conv_stat_shards() = 64:
 *   value = sum_shards (plane0 word * 2^-16 + plane1 word * 2^-56)          (mhe_stat_words(C) words in all)
 * A NaN / Inf / out-of-range partial plants a marker the finalize kernels turn into NaN.  Zeroed by the caller once; the
 * finalize entry points with a clear flag re-zero what they read.  (ABI 3: these were f32 [S,2,C] up to ABI 2.) */
typedef long long mhe_stat_t;
size_t mhe_stat_words(int C);
const char *mhe_last_error(void);

/* dense layers ------------------------------------------------------------
 * Y[M,N] = act(X[M,K] * W[N,K]^T + bias[N]); W is torch.nn.Linear's layout.
 * Replaces: BasicEnc.l1 (hand/network.py:87,121), MHEnt.det_head
 * (hand/network.py:380-383,747) and the per-image conditioning projections
 * c0/c1 of every coupling net (hand/flows.py:108-109) - evaluated once per
 * image instead of once per hypothesis row.  `bias` may be NULL.
 * f32 in / f32 accumulate on v_mfma_f32_16x16x4_f32 (bit-level = fmaf chain). */
int mhe_linear_f32(const float *X, const float *W, const float *bias, float *Y,
                   int M, int N, int K, int act, void *stream);
/* the M <= 256, K % 64 == 0 specialisation mhe_linear_f32 dispatches to (one workgroup per 16 output
 * columns, K split over its 4 waves): the shapes of the per-image heads. */
int mhe_linear_skinny_f32(const float *X, const float *W, const float *bias, float *Y,
                          int M, int N, int K, int act, void *stream);
/* the same, also writing the result as bf16 (Y_bf16 [M,N]): BasicEnc.l1's feature (hand/network.py:121) feeds the flow's conditioning
 * product, which takes bf16 operands in the bf16 mode - no separate cast launch in the step. */
int mhe_linear_f32_bf16copy(const float *X, const float *W, const float *bias, float *Y, void *Y_bf16,
                            int M, int N, int K, int act, void *stream);

/* base noise -----------------------------------------------------------------
 * out[n] ~ N(0, scale^2) i.i.d. (Philox4x32-10 + Box-Muller): the z0 = prior.sample((N*B,)) * temp of RealNVP.sample
 * (hand/flows.py:339; hand/network.py:733-735), drawn on the device inside the step.  state = three uint64 in DEVICE memory
 * {seed, next counter, 0}: the launch advances the counter itself, so a captured HIP graph draws fresh noise on every replay.
 * A device generator cannot reproduce the reference's CPU stream: parity runs supply the noise instead. */
int mhe_randn_f32(float *out, long n, unsigned long long *state, float scale, void *stream);
/* train-mode dropout of the ConditionalGlow's residual blocks (hand/network.py:343-344, `dropout_probability=0.2`; nflows ResidualBlock:
 * activation -> linear -> activation -> dropout -> linear): x <- x * keep / (1 - p_drop) in place, keep ~ Bernoulli(1 - p_drop).
 * draw != 0: the mask is drawn from `state` (the device generator of mhe_randn_f32, advanced by the launch) and its bits - bit k of byte i =
 * element 8 i + k kept - are written to `bits` (optional); draw == 0: the given bits are applied (the reverse pass applies them to the
 * gradient; parity tests feed the oracle the same mask).  n % 8 == 0; dtype MHE_F32 / MHE_BF16. */
int mhe_dropout(void *x, int dtype, unsigned char *bits, long n, float p_drop, unsigned long long *state, int draw, void *stream);
/* BasicEnc's stochastic head (hand/network.py:121-138): sd = exp(l2/2) (sigmoid_act: sigmoid(l2)), z = mn + sd*eps (deterministic:
 * z = mn; eps may then be NULL).  Dead for MHEnt, which keeps only mn (:779,862); built so BasicEnc returns the reference's (z, mn, sd). */
int mhe_reparam_f32(const float *mn, const float *l2, const float *eps, float *sd, float *z, long n, int sigmoid_act,
                    int deterministic, void *stream);

/* conditional RealNVP ----------------------------------------------------- */

/* Geometry of the packed weight stream for one coupling network
 * (dim -> hidden -> hidden -> dim, hand/flows.py:86-89). */
size_t mhe_flow_packed_floats_per_net(int dim, int hidden);

/* Pack one network's three Linear weights (torch layout, W0[hidden,dim],
 * W1[hidden,hidden], W2[dim,hidden]; hand/flows.py:87-89) from HOST memory into
 * the MFMA-fragment-ordered stream the coupling kernel consumes.  Pure data
 * movement, done when weights change.  `out_host` holds
 * mhe_flow_packed_floats_per_net() floats. */
int mhe_flow_pack_net_host(const float *W0_host, const float *W1_host, const float *W2_host,
                           int dim, int hidden, float *out_host);

/* All couplings of RealNVP.forward_p (z -> x, hand/flows.py:210-217) or
 * RealNVP.backward_p (x -> z, hand/flows.py:219-227) in one launch.
 *   in, out      [R, dim]      flow variable before / after (may alias)
 *   cond         [B, 2*ncoup, 2, hidden]  per image, per net (net index = 2*i for
 *                s_i, 2*i+1 for t_i), per hidden layer j: c_j(feat) + c_j.bias +
 *                l_j.bias (hand/flows.py:105-117) - produced with mhe_linear_f32
 *   wstream      [2*ncoup] packed nets (mhe_flow_pack_net_host), same net order
 *   bias2        [2*ncoup, dim]  l_2.bias of every net
 *   mask         [ncoup, dim]    RealNVP.mask (hand/flows.py:153-155,188)
 *   sum_s        [R]  out: sum_i sum_d s_i,d  (forward: log|det dx/dz|;
 *                inverse: equals -log_det_J of hand/flows.py:226)
 *   log_prob     [R]  out: log q(x) = logN(z;0,I) -+ sum_s, i.e. what
 *                RealNVP.log_prob returns (hand/flows.py:314-320); in FORWARD
 *                mode it is evaluated from the sampling pass itself (z = `in`),
 *                SURVEY.md appendix A2(ii).  May be NULL.
 * Row r uses image r % B (sample-major rows).  hidden must be a multiple of 64,
 * dim <= 48. */
int mhe_flow_couplings_f32(const float *in, float *out, const float *cond,
                           const float *wstream, const float *bias2, const float *mask,
                           float *sum_s, float *log_prob,
                           int R, int B, int dim, int hidden, int ncoup, int direction,
                           void *stream);

/* bf16-operand variant (bf16 x bf16 products, f32 accumulate, hidden activations rounded to bf16
 * between layers; flow variable, s, t, exp and log-det in f32): same contract as
 * mhe_flow_couplings_f32 with `wstream` = 2*ncoup nets packed by mhe_flow_pack_net_bf16_host and
 * `bias2` zero-padded to [2*ncoup, 64] (uniform scalar loads, no bounds test in the kernel).
 * v_mfma_f32_32x32x16_bf16, weights DMA'd (global_load_lds) into LDS rings.  hidden in {128, 256, 512}, dim <= 48:
 * 512 runs the unit-split kernel (csrc/flow_ns.hip: 64 rows per workgroup, 64 units per wave), 128 / 256 the
 * row-split one (csrc/flow_bf16.hip: 32 rows per wave); the packed stream's block order follows the kernel. */
size_t mhe_flow_packed_bytes_per_net_bf16(int dim, int hidden);
int mhe_flow_pack_net_bf16_host(const float *W0_host, const float *W1_host, const float *W2_host,
                                int dim, int hidden, void *out_host);
int mhe_flow_couplings_bf16(const float *in, float *out, const float *cond,
                            const void *wstream, const float *bias2, const float *mask,
                            float *sum_s, float *log_prob,
                            int R, int B, int dim, int hidden, int ncoup, int direction,
                            void *stream);
/* the same pass (hidden = 512 only) also writing out what the reverse pass of the train step needs, so that it does not
 * re-evaluate the nets (what autograd keeps for hand/flows.py:105-122): h1, h2 = hidden activations after the leaky-ReLU, bf16
 * [2*ncoup nets][R][512]; o = s / t pre-activations (l2 output incl. bias, before tanh), f32 [2*ncoup][R][64] (cols >= 48 untouched). */
int mhe_flow_couplings_bf16_emit(const float *in, float *out, const float *cond,
                                 const void *wstream, const float *bias2, const float *mask,
                                 float *sum_s, float *log_prob, void *h1, void *h2, float *o,
                                 int R, int B, int dim, int hidden, int ncoup, int direction,
                                 void *stream);

/* HO3D input pipeline from the decoded arrays on (reference: hand/dataloader/ho3d_dataloader.py:272-459 `__getitem__`, helpers
 * :32-199, `compute_st` hand/dataloader/rhddataloader.py:237-269), batched.  mhe_ho3d_targets: one workgroup per sample - projection,
 * hand / object boxes -> crop window, crop uv, both visibility passes, normalised pose, augmentation bookkeeping, RHD joint order,
 * compute_st; writes per-sample geometry (mhe_ho3d_geom_doubles() doubles: crop window, inverse affine map, flag) for
 * mhe_ho3d_images: one thread per output pixel - OpenCV-style fixed-point inverse affine (INTER_NEAREST), crop + nearest resize,
 * border fill, colour noise, ToTensor + Normalize(0.5, 0.5); hand / object masks, depth crop.
 * image u8 [B,480,640,3]; depth_png u8 [B,480,640,3] (BGR, value = R + 256 G); seg u8 [B,120,160,3]; joints3d [B,21,3], mesh
 * [B,778,3] metres (OpenGL camera frame); cam [B,3,3]; obj_verts [B,NVmax,3] + obj_count [B]; aug [B,7] float64 = colour x3, scale,
 * angle, tx, ty (the reference's np.random draws) or NULL (evaluation).  Outputs in the reference's target layout (RHD joint order). */
int mhe_ho3d_geom_doubles(void);
int mhe_ho3d_targets(const float *joints3d, const float *mesh, const float *cam, const float *obj_rot, const float *obj_trans,
                     const float *obj_verts, const int *obj_count, int NVmax, const unsigned char *seg, const unsigned char *depth_png,
                     const double *aug, float *crop_uv, float *vis, float *original_pose3d, float *verts, float *pose3d, float *st,
                     float *scale, float *crop_center, float *crop_size, float *pose3d_root, float *rot_mat_inv, float *rot_mat, float *uvd,
                     float *object_verts /* [B,NVmax,3] or NULL */, double *geom, int B, void *stream);
int mhe_ho3d_images(const unsigned char *image, const unsigned char *seg, const unsigned char *depth_png, const double *geom,
                    const double *aug, float *image_out /* [B,3,256,256] */, unsigned char *hand_mask /* [B,256,256] */,
                    unsigned char *object_mask, float *depth_out, int B, void *stream);

/* MANO decode + likelihood ------------------------------------------------- */

/* Number of floats of the packed MANO table blob (layout: mhentropy_amd/mano_pack.py). */
size_t mhe_mano_table_floats(void);

/* Fused loss-pass decoder: z assembly (hand/network.py:703-717), PCA pose +
 * Rodrigues + kinematic chain + the 21 joints (hand/manopth/manolayer.py:131-273;
 * only the 5 fingertip vertices are skinned), RHD reorder (hand/ManoLayer.py:54-56),
 * root/bone normalisation (hand/utils.py:46-66), orthographic projection
 * (hand/ManoLayer.py:150-165), visibility-masked Laplace log-likelihood and the
 * soft box/ball priors (hand/network.py:155-165,233-258,640-662), parameter
 * norms (hand/network.py:787-788).  One wavefront per hypothesis row.
 *   th45   [R,45]   flow sample            det [B,16] = det_head output
 *   crop_uv [B,42], vis [B,21]  targets (may be NULL when terms == NULL)
 *   tables  packed blob
 * outputs (each may be NULL):
 *   z [R,61], xyz [R,63] normalised joints, uv [R,42],
 *   terms [R,4] = log p(uv|z), log p(th3), log p(th45), log p(bt); log_p [R] their sum,
 *   norms [R,2] = |theta|, |beta|,
 *   joints_mm [R,63] the un-normalised `mano_joints` (mm, centred on joint 9, RHD order).
 * inv_norm != 0 maps uv to pixels ((uv+1)/2*image_size) as MHEnt.sample does. */
int mhe_mano_joints_f32(const float *th45, const float *det, const float *crop_uv, const float *vis,
                        const float *tables,
                        float *z, float *xyz, float *uv, float *terms, float *log_p, float *norms,
                        float *joints_mm, int R, int B, float laplace_b, float th45_alpha, int inv_norm, float image_size,
                        void *stream);

/* Full 778-vertex linear-blend skinning for MHEnt.sample
 * (hand/manopth/manolayer.py:181-246, hand/network.py:480): verts [R,778,3]
 * normalised like xyz ((mesh - root)/bone), or with mm_mode != 0 the `mesh`
 * output of ManoLayer.forward (mm, centred on joint 9).  z [R,61] as written by
 * mhe_mano_joints_f32.  Two launches (per-hypothesis pose pass, per-vertex skinning pass) that hand
 * over through `workspace` (mhe_mano_verts_workspace_floats(R) floats, caller-owned). */
size_t mhe_mano_verts_workspace_floats(int R);
int mhe_mano_verts_f32(const float *z, const float *tables, float *verts, float *workspace, int R, int mm_mode,
                       void *stream);
/* mhe_mano_joints_f32 + mhe_mano_verts_f32 of the same hypotheses (MHEnt.sample with mods = {uv, xyz, verts}: the reference iteration's
 * metrics pass, hand/CrossModalHand.py:357-361, hand/network.py:541-558): the joint pass leaves every hypothesis' skinning operands in the
 * workspace (same size as for mhe_mano_verts_f32), the skinning launch (csrc/mano_skin.hip: both products on the matrix cores from bf16
 * pieces of the f32 operands, f32-class accuracy) follows.  Arguments as in the two entries; verts as there. */
int mhe_mano_decode_f32(const float *th45, const float *det, const float *crop_uv, const float *vis, const float *tables,
                        float *z, float *xyz, float *uv, float *terms, float *log_p, float *norms, float *joints_mm,
                        float *verts, float *workspace, int R, int B, float laplace_b, float th45_alpha, int inv_norm,
                        float image_size, int mm_mode, void *stream);

/* ManoLayer.xyz_from_vertice (hand/ManoLayer.py:108-148) + RHD reorder (:54-56):
 * verts [R,778,3] -> joints [R,21,3] (the wrapper's 'joints' output, unused by MHEnt). */
int mhe_mano_regress_joints_f32(const float *verts, const float *tables, float *joints, int R, void *stream);

/* mean over the N hypotheses of each image and the ELBO
 * (hand/network.py:793,801-808): rows [N*B] -> [B].
 *   q_log_p[b] = mean_n log_p_rows[n*B+b];  h[b] = mean_n -log_q_rows[n*B+b];
 *   log_p[b] = h[b] + q_log_p[b] */
int mhe_elbo_reduce_f32(const float *log_p_rows, const float *log_q_rows,
                        float *q_log_p, float *h, float *log_p, int N, int B, void *stream);

/* image encoder ------------------------------------------------------------
 * NHWC convolution as an implicit GEMM on MFMA, replacing the torchvision
 * ResNet trunk the reference builds (hand/network.py:54-61,110).
 *   x   [B,H,W,Cin]      w [Cout,KH,KW,Cin] (packed from torch's [Cout,Cin,KH,KW], K zero-padded to the
 *       128-byte stage; mhentropy_amd/resnet.py:pack_conv_weight)
 *   y   [B,Ho,Wo,Cout]
 *   in_scale/in_shift [Cin] (optional): the producer's BatchNorm folded to
 *       relu(x*scale+shift) applied while loading (padding stays zero)
 *   out_scale/out_shift [Cout] (optional, eval-mode BN), residual (optional,
 *       [B,Ho,Wo,Cout]), relu flag: y = act(conv*scale+shift + residual)
 *   stats (optional): one mhe_stat_t unit of Cout channels (above): per-channel sum and
 *       sum of squares of the raw conv output (train-mode BatchNorm batch
 *       statistics).  Each workgroup reduces its tile in LDS and adds one value
 *       per channel into shard (workgroup % S); must be zeroed by the caller.
 * dtype is the storage type of x, w, y, residual (MHE_F32 or MHE_BF16);
 * accumulation is always f32. */
typedef struct mhe_conv_desc {
    int B, H, W, Cin, Cout, KH, KW, stride, pad;
    int dtype;
    int relu_in;   /* apply relu after in_scale/in_shift */
    int relu_out;
    int tile;      /* 0 = the launcher chooses the kernel variant; k > 0 forces variant k-1 where the geometry admits it
                    * (0 128x64, 1 128x128, 2 256x256, 3 256x128, 4 256x64,
                    * 7 the 8-phase 256x256 kernel, 8 the streaming 1x1 kernel for 64 / 128 (/ 256) input channels, 9 the row-streaming 3x3 kernel for 64 -> 64 channels,
                    * 10 the 128 x 256 residual-tail kernel with transfer waves for >= 256 output channels, 11 the resident-slab kernel with transfer
                    * waves for 256 / 512 -> >= 512 channels, 13 the 8-phase kernel on a 256 x 128 tile; 5, 6 and 15 - the round-1 LDS-DMA kernels and the
                    * DMA-ring residual tail, measured no faster - left the build in round 5): parity tests and tuning runs
                    * reach every instantiation in-process */
    int res_half;  /* 1: `residual` is at HALF resolution, [B, ceil(Ho/2), ceil(Wo/2), Cout], and is added at the even output positions
                    * only (the data gradient of a stride-2 1x1 shortcut joining the main branch's gradient without being scattered
                    * into a full-resolution tensor first) */
} mhe_conv_desc;

/* Statistics-only form of a 1x1 / stride-1 bf16 convolution with 64 or 128 input channels: the batch statistics (stats, as above)
 * of the output AS IT WOULD BE STORED, without storing it.  With mhe_bottleneck_tail_nhwc below this replaces "write conv3's raw output,
 * read it back in the block tail" for layer1 / layer2 of ResNet-50 (torchvision Bottleneck, hand/network.py:54-61,110). */
int mhe_conv1x1_stats_nhwc(const mhe_conv_desc *d, const void *x, const void *w, const float *in_scale, const float *in_shift,
                           mhe_stat_t *stats, void *stream);
/* ... and cheaper still: the same batch statistics from the moments of the convolution's INPUT.  For y = W a:  sum_p y_c = w_c . m,
 * sum_p y_c^2 = w_c^T G w_c  with m = sum_p a_p and G = sum_p a_p a_p^T (Cb x Cb) - 8x / 4x fewer multiply-adds than the product and no
 * per-output-element work.  mhe_conv1x1_gram_nhwc WRITES G and m of a = relu?(x * in_scale + in_shift) (rounded to bf16, the operand
 * conv3 multiplies) into `gram`, a workspace of mhe_gram_stats_words(Cb) 8-byte words: a count word, then one f32 partial slab per workgroup
 * of the launch (plain stores - nothing to zero, nothing to clear; ONE launch per finalize); mhe_gram_bn_finalize sums the slabs in slab
 * order in f64 (the totals do not depend on the order the workgroups finish in) and turns them into bn's affine like mhe_bn_finalize_step
 * (w = the packed [C][Cb] bf16 weights; workspace = mhe_gram_stats_workspace_bytes(Cb) bytes, left holding the f64 totals [Cb][Cb] | [Cb]).
 * Statistics of the f32 products (not of bf16-rounded outputs). */
size_t mhe_gram_stats_words(int Cb);
size_t mhe_gram_stats_workspace_bytes(int Cb);
int mhe_conv1x1_gram_nhwc(const void *x, const float *in_scale, const float *in_shift, int relu_in, mhe_stat_t *gram, long pixels, int Cb,
                          void *stream);
/* ... that also writes the operand it multiplies, relu(x * in_scale + in_shift) as bf16 [pixels][Cb] (a_out, optional): the train step keeps it
 * for the reverse pass instead of making it in a pass of its own */
int mhe_conv1x1_gram_store_nhwc(const void *x, const float *in_scale, const float *in_shift, int relu_in, mhe_stat_t *gram, void *a_out,
                                long pixels, int Cb, void *stream);
int mhe_gram_bn_finalize(mhe_stat_t *gram, void *workspace, const void *w, const float *gamma, const float *beta, float *running_mean,
                         float *running_var, float *scale, float *shift, float *mean_invstd, int C, int Cb, double count,
                         float momentum, float eps, long long *num_batches_tracked, void *stream);
/* The tail of a bottleneck block with its conv3 re-evaluated, fused with the next block's conv1 (forward-only path; kernel variant 12):
 *   T  = conv1x1(relu(y2 * bn2_scale + bn2_shift), w3)   rounded to the storage type like a stored conv3 output   [B,H,W,Cin]
 *   a  = relu(T * bn3_scale + bn3_shift + (identity * id_scale + id_shift | identity))     -> a_out                 [B,H,W,Cin]
 *   y1 = conv1x1(a, w1) -> y1 [B,H,W,Cout], stats (optional) as in mhe_conv2d_nhwc.
 * d describes the second product (Cin = block width = 4 Cb, Cout = the next bottleneck width); y2 [B,H,W,Cb], w3 packed [Cin][Cb],
 * w1 packed [Cout][Cin].  Results equal conv3 -> mhe_conv1x1_residual_in_nhwc bit for bit (same summation order).
 * mhe_bottleneck_tail_supported: 1 when the geometry is taken (bf16, Cb 64 / 128, Cout 64 / 128, B*H*W % 128 == 0). */
int mhe_bottleneck_tail_supported(const mhe_conv_desc *d, int Cb);
int mhe_bottleneck_tail_nhwc(const mhe_conv_desc *d, int Cb, const void *y2, const float *bn2_scale, const float *bn2_shift,
                             const void *w3, const float *bn3_scale, const float *bn3_shift, const void *identity,
                             const float *id_scale, const float *id_shift, const void *w1, void *a_out, void *y1, mhe_stat_t *stats,
                             void *stream);

int mhe_conv2d_nhwc(const mhe_conv_desc *d, const void *x, const void *w, void *y,
                    const float *in_scale, const float *in_shift,
                    const float *out_scale, const float *out_shift, const void *residual,
                    mhe_stat_t *stats, void *stream);
/* bf16 operands, f32 accumulation, F32 RESULT y_f32 [B,Ho,Wo,Cout] (+ out_shift per channel, optional): products whose result
 * feeds exp / tanh or a long f32 gradient chain (the 64-wide layers of the flow's reverse pass) at the bf16 MFMA rate. */
int mhe_conv2d_f32out_nhwc(const mhe_conv_desc *d, const void *x, const void *w, float *y_f32, const float *out_shift, void *stream);
/* y = (conv(x, w) + residual) * [mask > 0]: the data-gradient form (residual may be NULL; mask is shaped like y) - the
 * ReLU gate of the tensor the gradient is taken with respect to, applied where the gradient is produced.  Optionally the
 * epilogue also accumulates the BatchNorm-reverse statistics of y for up to two BN units whose raw outputs bn_y* are shaped
 * like y (the tail of a residual block feeds bn3 and the downsample's BN): bn_stats*[shard][0][c] += sum y,
 * [1][c] += sum y (bn_y - mean) invstd, exactly what mhe_bn_bwd_reduce_nhwc would compute in a separate pass. */
int mhe_conv2d_masked_nhwc(const mhe_conv_desc *d, const void *x, const void *w, void *y, const void *residual,
                           const void *mask, const void *bn_y0, const float *bn_mean_invstd0, mhe_stat_t *bn_stats0,
                           const void *bn_y1, const float *bn_mean_invstd1, mhe_stat_t *bn_stats1, void *stream);
int mhe_conv_stat_shards(void);
/* kernel variant the launcher picks for a geometry with plain operands (numbering of mhe_conv_desc.tile, minus 1) */
/* mhe_conv1x1_residual_in_nhwc's dual-input operand load (operand = [relu](x*in_scale+in_shift + x2*x2_scale+x2_shift), d->relu_in
 * selects the ReLU; a_out optionally receives the operand) combined with the data-gradient epilogue of mhe_conv2d_masked_nhwc
 * (residual, gate by mask, BatchNorm-reverse sums of one unit).  The train step uses it for the data gradient of a bottleneck's conv3
 * with the BatchNorm reverse APPLIED ON LOAD: x = gated gradient g', x2 = the unit's raw output y, scales = (k2, k1), shift = k0,
 * so that gy = k2 g' + k1 y + k0 is formed in the operand load (and written once to a_out for the weight gradient) instead of by a
 * separate pass over three block-wide tensors. */
int mhe_conv1x1_residual_in_masked_nhwc(const mhe_conv_desc *d, const void *x, const void *x2, const void *w, void *y,
                                        const float *in_scale, const float *in_shift, const float *x2_scale, const float *x2_shift,
                                        void *a_out, const void *residual, const void *mask, const void *bn_y0,
                                        const float *bn_mean_invstd0, mhe_stat_t *bn_stats0, void *stream);

/* Data gradient of a 3x3 / stride 2 / pad 1 convolution (reference: torchvision Bottleneck.conv2 of layer2-4's first block, reached
 * through hand/network.py:54-61) WITHOUT zero-dilating gy: the four output parity classes (2i+py, 2j+px) are four small
 * convolutions of gy [B,Ho,Wo,Cout] with (1+py) x (1+px) taps each (9 taps over 4 pixels instead of 36), written straight to
 * their strided positions of dx [B,2Ho,2Wo,Cin].  w4[2*py+px] = packed [Cin][(th, tw, co)] with forward taps kh = 1 (py = 0) or
 * kh = 2, 0 for th = 0, 1 (py = 1), likewise kw.  residual / mask / bn_* as mhe_conv2d_masked_nhwc, all shaped like dx
 * (mask may be NULL: no gate, no BatchNorm-reverse sums). */
int mhe_conv3x3s2_dgrad_nhwc(int B, int Ho, int Wo, int Cout, int Cin, int dtype, const void *gy, const void *const *w4,
                             void *dx, const void *residual, const void *mask, const void *bn_y0,
                             const float *bn_mean_invstd0, mhe_stat_t *bn_stats0, const void *bn_y1,
                             const float *bn_mean_invstd1, mhe_stat_t *bn_stats1, int tile /* mhe_conv_desc.tile */, void *stream);
int mhe_conv_tile(const mhe_conv_desc *d);
/* the same for an operand-load form: mode 1 = producer BatchNorm on load, 2 = residual-block tail, 0 = plain operands, 3 = plain operands
 * with a residual added in the epilogue (the streaming kernels' forward forms do not take one) */
int mhe_conv_tile_mode(const mhe_conv_desc *d, int mode);

/* 1x1 stride-1 convolution whose operand is the TAIL of the previous residual block evaluated while
 * loading (torchvision Bottleneck: out = relu(bn3(conv3) + identity)):
 *     a = relu(x*in_scale+in_shift + (x2*x2_scale+x2_shift | x2)),   y = conv1x1(a)
 * x = raw conv3 output, x2 = identity (x2_scale/x2_shift NULL) or the raw downsample conv output with its
 * BatchNorm affine.  a_out (optional, same shape as x) receives `a` once - the next identity.  Saves one
 * full read of the block output compared with mhe_bn_act_nhwc followed by mhe_conv2d_nhwc. */
int mhe_conv1x1_residual_in_nhwc(const mhe_conv_desc *d, const void *x, const void *x2, const void *w, void *y,
                                 const float *in_scale, const float *in_shift, const float *x2_scale,
                                 const float *x2_shift, void *a_out, mhe_stat_t *stats, void *stream);

/* ResNet stem: 7x7 stride-2 pad-3 convolution 3 -> 64 (torchvision `conv1`, reference hand/network.py:54-61)
 * read straight from the NCHW f32 image x [B,3,H,W]; w [64][192] with k = 24*kh + 3*kw + c (each kh row of
 * 21 taps zero-padded to 24, then to 192; mhentropy_amd/resnet.py:pack_stem_weight); y [B,Ho,Wo,64] raw
 * conv output (dtype);
 * stats [S,2,64] as in mhe_conv2d_nhwc (optional). */
int mhe_stem_conv7x7s2(const float *x_nchw, const void *w, void *y, mhe_stat_t *stats, int B, int H, int W, int dtype,
                       void *stream);

/* The stem for the forward-only path (bf16, 256 x 256 images): conv1 + batch statistics + the 3x3 / stride-2 / pad-1 max pool in one kernel.
 * pooled [B,64,64,64] receives, per channel, the window MAXIMUM (bn_gamma[c] >= 0) or MINIMUM (bn_gamma[c] < 0) of the raw bf16-rounded conv1
 * outputs: relu(scale*y + shift) is monotone in y and sign(scale) = sign(gamma), so maxpool(relu(bn1(conv1(x)))) (torchvision ResNet's
 * stem, hand/network.py:54-61,110) = relu(scale * pooled + shift), which the consumers apply on their operand load once the statistics
 * (stats, as in mhe_stem_conv7x7s2) are finalised.  The full-resolution output is never written. */
int mhe_stem_pool_supported(int B, int H, int W, int dtype);
int mhe_stem_conv7x7s2_pool(const float *x_nchw, const void *w, const float *bn_gamma, void *pooled, mhe_stat_t *stats, int B, int H, int W,
                            void *stream);

/* BatchNorm batch statistics (the sharded accumulators of mhe_conv2d_nhwc, summed in f64)
 * -> affine (train mode), torch semantics
 * (momentum 0.1, eps 1e-5, unbiased running_var):
 *   mean = sum/n, var = sumsq/n - mean^2; scale = gamma/sqrt(var+eps);
 *   shift = beta - mean*scale; running stats updated in place when non-NULL;
 *   mean_invstd [2,C] (optional) receives mean and 1/sqrt(var+eps) for the reverse pass. */
int mhe_bn_finalize(const mhe_stat_t *stats, const float *gamma, const float *beta,
                    float *running_mean, float *running_var, float *scale, float *shift, float *mean_invstd,
                    int C, double count, float momentum, float eps, void *stream);
/* the form the forward step uses: clear_stats != 0 zeroes the accumulators it has just read (the arena is clean for the next
 * step without a memset launch), num_batches_tracked (optional, int64 on the device) is incremented like nn.BatchNorm2d does. */
int mhe_bn_finalize_step(mhe_stat_t *stats, const float *gamma, const float *beta,
                         float *running_mean, float *running_var, float *scale, float *shift, float *mean_invstd,
                         int C, double count, float momentum, float eps, int clear_stats, long long *num_batches_tracked, void *stream);

/* y = relu?(x*scale+shift (+ r*r_scale+r_shift | + r)) elementwise over NHWC
 * [P,C]; the bottleneck tail bn3 + identity + relu (torchvision Bottleneck). */
int mhe_bn_act_nhwc(const void *x, const float *scale, const float *shift,
                    const void *res, const float *res_scale, const float *res_shift,
                    void *y, long P, int C, int relu, int dtype, void *stream);

/* 3x3/2 max pool with pad 1 over relu(x*scale+shift), NHWC (stem). */
int mhe_maxpool3x3s2_nhwc(const void *x, const float *scale, const float *shift, void *y,
                          int B, int H, int W, int C, int dtype, void *stream);

/* global average pool NHWC [B,HW,C] -> f32 [B,C]. */
int mhe_avgpool_nhwc(const void *x, float *y, int B, int HW, int C, int dtype, void *stream);
/* mhe_bn_act_nhwc followed by mhe_avgpool_nhwc in one pass - the last residual block's tail and the encoder's average pool
 * (torchvision ResNet.forward: layer4 -> avgpool, hand/network.py:54-61,110): y[b][c] = mean_p round(relu?(x*scale+shift (+ res*rscale+rshift | + res))),
 * rounded to `dtype` before the sum and summed in mhe_avgpool_nhwc's order: the same bits as the two launches.  C % 4 == 0. */
int mhe_bn_act_avgpool_nhwc(const void *x, const float *scale, const float *shift, const void *res, const float *rscale,
                            const float *rshift, float *y, int B, int HW, int C, int relu, int dtype, void *stream);

/* NCHW f32 image -> NHWC (dtype), stem input layout change. */
int mhe_nchw_to_nhwc(const float *x, void *y, int B, int C, int H, int W, int dtype, void *stream);
/* ... with the channel padding given (Cp >= C; 3 -> 4 in bf16 = one 8-byte store per pixel: the operand of mhe_conv_wgrad_rect_nhwc for the stem) */
int mhe_nchw_to_nhwc_pad(const float *x_nchw, void *y, int B, int C, int Cp, int H, int W, int dtype, void *stream);

/* evaluation metrics ---------------------------------------------------------
 * MHEntLoss metrics (hand/criteria.py:91-168): xyz [N,B,63] normalised joints,
 * uv [N,B,42] pixels, targets pose3d [B,63], scale [B], crop_uv [B,42], vis [B,21].
 * out [14,B], rows: for sup in (3d,2d): sample, sample_std, vis, vis_std,
 * vis_mean, invis, invis_std  (key eucLoss_{sup}_rgb_{row}). */
int mhe_metrics_f32(const float *xyz, const float *uv, const float *pose3d, const float *scale,
                    const float *crop_uv, const float *vis, float *out, int N, int B, void *stream);

/* Top-Q hypothesis selection of MHEnt.sample (hand/network.py:866-871): per image b keep the Q rows of
 * highest score[n*B+b] in descending order and gather them: idx_out [Q,B] (hypothesis index n),
 * rows_out [Q*B, D] sample-major.  rows [N*B, D] is the flow sample th45. */
int mhe_topk_gather_f32(const float *score, const float *rows, int *idx_out, float *rows_out,
                        int N, int B, int Q, int D, void *stream);

/* ---- train step (hand/CrossModalHand.py:455-470: zero_grad; loss.backward(); clip; Adam) -------------
 * The reference differentiates the path with autograd; here every stage has a hand-written reverse
 * kernel.  Reverse of mhe_mano_joints_f32's log_p output: row r = n*B + b receives
 * d loss / d log_p[r] = g_log_p[b] * row_weight (row_weight = 1/N for the mean over hypotheses,
 * hand/network.py:793), and the kernel returns d/d th45 [R,45] and d/d det per ROW [R,16]
 * (sum over n with mhe_sum_over_hypotheses_f32 to get the per-image det-head gradient). */
int mhe_mano_joints_bwd_f32(const float *th45, const float *det, const float *crop_uv, const float *vis,
                            const float *tables, const float *g_log_p, float *g_th45, float *g_det_rows,
                            int R, int B, float laplace_b, float th45_alpha, float row_weight, void *stream);
/* out[b*out_stride + c] (+)= sum_n rows[(n*B + b)][c] : the adjoint of `.repeat(N,1)` (hand/network.py:734,747);
 * out_stride <= 0 means C (dense). */
int mhe_sum_over_hypotheses_f32(const float *rows, float *out, int N, int B, int C, int accumulate, long out_stride,
                                void *stream);

/* dW[Cout][KH*KW*Cin] += gy^T (*) x : weight gradient of mhe_conv2d_nhwc (x [B,H,W,Cin], gy [B,Ho,Wo,Cout]
 * of d->dtype storage, dW f32 with row pitch ldw >= KH*KW*Cin, 0 = dense).  The caller zeroes dW; partial sums of the
 * pixel-range split are added with f32 atomics HERE (order-dependent in the last bits) - mhe_conv_wgrad_ws_nhwc below takes a workspace
 * and sums them in a fixed order (what the train step uses).  With H = W = KH = KW = 1 it is the weight gradient of a dense
 * layer, dW[N][K] += gy[R,N]^T x[R,K] (torch.nn.Linear layout).  Cin, Cout multiples of 4. */
int mhe_conv_wgrad_nhwc(const mhe_conv_desc *d, const void *x, const void *gy, float *dw, int ldw, void *stream);
/* The same with a workspace of >= mhe_conv_wgrad_workspace_floats(d) floats: every pixel-range slice stores its partial tile
 * into the workspace with plain coalesced stores and a reducer launch adds the slabs to dW (dW += as above; nothing needs
 * zeroing in the workspace).  Falls back to the atomic form when the workspace is NULL or too small. */
size_t mhe_conv_wgrad_workspace_floats(const mhe_conv_desc *d);
int mhe_conv_wgrad_ws_nhwc(const mhe_conv_desc *d, const void *x, const void *gy, float *dw, int ldw, float *workspace,
                           size_t workspace_floats, void *stream);
/* nbatch independent weight gradients of ONE geometry in one launch (grouped GEMM): problem b reads x + b * x_batch_stride and
 * gy + b * gy_batch_stride (elements) and accumulates into dw + b * dw_batch_stride (floats).  bf16 operands only.  The RealNVP reverse
 * pass uses it for the three products of its 24 coupling nets (hand/flows.py:105-122).  workspace as for mhe_conv_wgrad_ws_nhwc, sized by
 * mhe_conv_wgrad_batched_workspace_floats (0 = not needed). */
size_t mhe_conv_wgrad_batched_workspace_floats(const mhe_conv_desc *d, int nbatch);
int mhe_conv_wgrad_batched_nhwc(const mhe_conv_desc *d, int nbatch, const void *x, long x_batch_stride, const void *gy, long gy_batch_stride,
                                float *dw, long dw_batch_stride, int ldw, float *workspace, size_t workspace_floats, void *stream);
/* SEVERAL independent weight gradients in one call (round 5; csrc/wgrad.hip: wgrad_dma_multi_kernel): the problems that share a tile shape of the
 * LDS-DMA kernel are launched together (16 per launch), the chip is filled by the tiles of ALL of them and every problem's pixel range is cut
 * only as far as a common slice length asks - the partial-slab traffic of the one-problem-per-launch form (workgroups x tile bytes) falls by the
 * number of problems that share the chip.  Unsplit problems add their tiles to dW with plain stores, split ones through slabs and ONE fixed-order
 * reducer launch per batch: sums do not depend on the order in which workgroups finish.  Problems the LDS-DMA kernel does not take run as their
 * own launches.  dW += as in mhe_conv_wgrad_nhwc; workspace: mhe_conv_wgrad_multi_workspace_floats(items, n) floats. */
typedef struct mhe_wgrad_item {
    mhe_conv_desc d;
    const void *x, *gy;
    float *dw;
    int ldw;        /* row pitch of dW in floats, 0 = KH * KW * Cin */
} mhe_wgrad_item;
size_t mhe_conv_wgrad_multi_workspace_floats(const mhe_wgrad_item *items, int n);
int mhe_conv_wgrad_multi_nhwc(const mhe_wgrad_item *items, int n, float *workspace, size_t workspace_floats, void *stream);
/* Weight gradient of a convolution whose WIDTH direction has its own stride and left padding and whose output size is given instead of
 * derived (d->stride / d->pad describe the height direction; d->KH x d->KW taps).  Use: the stem's 7x7 / stride-2 / pad-3 convolution
 * (reference: torchvision ResNet.conv1 under hand/CrossModalHand.py:455-470's backward) read as a 7 x 4 / stride (2, 1) / pad (3, 2)
 * convolution over PIXEL PAIRS - x [B, H, W/2, 8] = two neighbouring pixels x (3 channels padded to 4), Ho x Wo = H/2 x W/2 - so that
 * dW' [Cout][7][4][8] has 224 columns instead of the 392 of 3 channels padded to 8; dW'[co][kh][kw'][4 par + c] = dW[co][c][kh][2 kw' + par - 1]. */
/* the kernel instantiation a weight-gradient launch of this geometry takes (Ho / Wo = 0: derived; nbatch = 1: not grouped): BM * 1000 + BN,
 * + 1,000,000 on the LDS-DMA kernel, + 2,000,000 on the register-staged bf16 kernel; -1 for a bad descriptor.  (bench.py names what it times.) */
int mhe_conv_wgrad_variant(const mhe_conv_desc *d, int Ho, int Wo, int nbatch);
size_t mhe_conv_wgrad_rect_workspace_floats(const mhe_conv_desc *d, int Ho, int Wo);
int mhe_conv_wgrad_rect_nhwc(const mhe_conv_desc *d, int stride_w, int pad_w, int Ho, int Wo, const void *x, const void *gy, float *dw,
                             int ldw, float *workspace, size_t workspace_floats, void *stream);
/* out[c] += sum_r rows[r][c] (bias gradients; caller zeroes out); rows f32 or bf16, sums f32, in a FIXED order (no atomics).
 * mhe_colsum_f32: one workgroup per 256 columns walks all rows (R <= 4,096).  mhe_colsum_ws_f32: with a workspace of
 * mhe_colsum_workspace_floats(R, C) floats the rows are cut into slabs (two launches); column c is added to
 * out[(c / out_group_width) * out_group_stride + c % out_group_width] (out_group_width = 0: out[c]). */
int mhe_colsum_f32(const void *rows, float *out, long R, int C, int dtype, void *stream);
size_t mhe_colsum_workspace_floats(long R, int C);
int mhe_colsum_ws_f32(const void *rows, float *out, long R, int C, int dtype, int out_group_width, long out_group_stride,
                      float *workspace, size_t workspace_floats, void *stream);
/* dst[i] = (idx[i] < 0 ? 0 : src[idx[i]]) + (idx2 != NULL && idx2[i] >= 0 ? src[idx2[i]] : 0) ; dst f32 or bf16.  Every weight re-layout of a train step (forward
 * operand packs, transposed / tap-flipped operands of the data-gradient convolutions, un-packing of weight
 * gradients into the flat gradient buffer) is one gather over an index table built once on the host. */
int mhe_gather_f32(const float *src, const int *idx, const int *idx2, void *dst, size_t n, int dst_dtype, void *stream);
/* dst[8 g + k] = bf16(src[base_g + k * stride_g]) for the k whose bit is set in mask[g], else 0 - mhe_gather_f32's bf16 form for operand layouts
 * whose groups of eight destination elements are affine in the source (permuted / padded weight tensors: the train step's operand re-pack,
 * hand/CrossModalHand.py:455-470 has no counterpart - torch keeps one layout).  base_stride: int32 pairs [n / 8][2], mask: bytes [n / 8]. */
int mhe_gather_affine8_bf16(const float *src, const int *base_stride, const unsigned char *mask, void *dst, size_t n, void *stream);

/* Elementwise stages of the RealNVP reverse pass (hand/flows.py:97-122,210-217), f32, flow variable padded
 * to 64 columns where it is a GEMM operand; see csrc/flow_bwd.hip for the formulas. */
int mhe_flow_mask_pad_f32(const float *x, const float *mask, float *xp, long R, int dim, void *stream);
int mhe_flow_cond_lrelu_f32(float *P, const float *cond, long cond_stride, long R, int B, int H, void *stream);
int mhe_flow_lrelu_bwd_f32(float *G, const float *Hact, long n, float slope, void *stream);
int mhe_add_f32(const float *a, const float *b, float *out, long n, void *stream);
/* mixed-precision forms (bf16 performance mode): input f32 or bf16, output f32 and / or bf16 (either may be NULL) */
int mhe_flow_cond_lrelu_mixed(const void *pre, int pre_dtype, const float *cond, long cond_stride, float *out_f32,
                              void *out_bf16, long R, int B, int H, void *stream);
int mhe_flow_lrelu_bwd_mixed(const void *g, int g_dtype, const void *h, int h_dtype, float *out_f32, void *out_bf16,
                             long n, float slope, void *stream);
/* out = g * (h > 0 ? 1 : slope) as f32 and / or bf16 (either may be NULL) AND sum_out[b][c] = sum over the N hypotheses of image b
 * (row r = n*B + b; sum_out row pitch sum_stride floats): the leaky-ReLU reverse fused with the per-image reduction that
 * gives the gradient of the conditioning table. */
int mhe_flow_lrelu_bwd_sum(const void *g, int g_dtype, const void *h, int h_dtype, float *out_f32, void *out_bf16,
                           float *sum_out, long sum_stride, float *sum_out_t, int N, int B, int H, float slope, void *stream);
/* (sum_out_t, optional: the same sums transposed, sum_out_t[c][b] with row pitch B - the operand layout of the split-K product
 * that turns the conditioning table's gradient into the feature's) */
/* mhe_flow_mask_pad_f32 / mhe_flow_couple_bwd_f32 with optional bf16 copies of the GEMM operands they produce */
int mhe_flow_mask_pad_mixed(const float *x, const float *mask, float *xp, void *xp_bf16, long R, int dim, void *stream);
/* The data-gradient chain of ALL couplings in one launch (bf16 mode, hidden 512, 64 hypotheses per image, forward activations kept by
 * mhe_flow_couplings_frag_bf16, which also leaves their signs as sign_bits): what mask_pad -> couple_bwd -> (GO W2, leaky-ReLU reverse + sums, G2 W1, leaky-ReLU reverse + sums, G1 W0) x 2
 * -> couple_accum compute coupling by coupling (13 launches each), with every intermediate kept on chip.  A workgroup owns one image's 64
 * rows.  Outputs: GO / G2 / G1 [nets][R][64 | 512 | 512] and the masked inputs XP [ncoup][R][64] in bf16 (operands of the grouped weight
 * gradients), Gc [B][cond_stride] (the conditioning table's gradient: column (2 net + layer) * hidden + unit; cond_stride % 4 == 0),
 * db2_rows [B][2 ncoup][64] (every image's share of the nets' l2 bias gradients, WRITTEN: their sum over images is a mhe_colsum_ws_f32 away), z0 [R][dim] (the recovered base sample, optional).  w2F, w1F, w0F =
 * net 0's bf16 operands W2^T [512][64], W1^T [512][512], W0^T [64][512] ([out][k]) in FRAGMENT-MAJOR order - element
 * [out][k] at ((out / 16 * (K / 32) + k / 32) * 64 + (k % 32 / 8) * 16 + out % 16) * 8 + k % 8, the order the lanes of
 * v_mfma_f32_16x16x32_bf16 take them, so a fragment is one 1 KiB run; net k at + k * w_net_stride elements. */
/* mhe_conv2d_masked_nhwc with the gate also as bits: mask_bits [pixels][Cout / 8] bytes, bit i of byte j = (mask value of channel 8 j + i) > 0
 * (written by mhe_bottleneck_tail_bits_nhwc).  The streaming 1x1 data-gradient kernel reads the bits instead of the tensor; every other kernel
 * reads mask.  A bn_y equal to mask asks for sum g only (its second sum is then undefined). */
int mhe_conv2d_masked_bits_nhwc(const mhe_conv_desc *d, const void *x, const void *w, void *y, const void *residual,
                                const void *mask, const void *mask_bits, const void *bn_y0, const float *bn_mean_invstd0, mhe_stat_t *bn_stats0,
                                const void *bn_y1, const float *bn_mean_invstd1, mhe_stat_t *bn_stats1, void *stream);
/* mhe_bottleneck_tail_nhwc that also writes the gate bits of its block output (a_bits [pixels][Cin / 8] bytes, optional). */
int mhe_bottleneck_tail_bits_nhwc(const mhe_conv_desc *d, int Cb, const void *y2, const float *bn2_scale, const float *bn2_shift,
                                  const void *w3, const float *bn3_scale, const float *bn3_shift, const void *identity,
                                  const float *id_scale, const float *id_shift, const void *w1, void *a_out, void *a_bits, void *y1,
                                  mhe_stat_t *stats, void *stream);
/* mhe_conv2d_masked_nhwc with a per-channel constant: y = (conv(x, w) + bias + residual) * [mask > 0] (+ the BatchNorm-reverse sums of one
 * consumer).  Register-staged 128-row tiles only.  xcat (optional, bf16 1x1 stride-1 launches): the operand's K range continues on a second
 * tensor - y = [x | xcat] w^T with w [Cout][Cin + cin2] and xcat [pixels][cin2]. */
int mhe_conv2d_masked_bias_nhwc(const mhe_conv_desc *d, const void *x, const void *xcat, int cin2, const void *w, void *y,
                                const void *residual, const void *mask, const float *bias, const void *bn_y0,
                                const float *bn_mean_invstd0, mhe_stat_t *bn_stats0, void *stream);
/* y = [x | xcat] w^T + bias (+ residual): a bf16 1x1 / stride-1 product whose K range continues on a second tensor (w [Cout][Cin + cin2]),
 * ungated - the data gradient of a shortcut convolution whose BatchNorm reverse runs on Gram statistics (mhe_conv3_bn_fold below; torchvision
 * Bottleneck.downsample, hand/network.py:54-61).  bias, residual optional. */
int mhe_conv1x1_cat_bias_nhwc(const mhe_conv_desc *d, const void *x, const void *xcat, int cin2, const void *w, void *y, const void *residual,
                              const float *bias, void *stream);
/* Reverse of conv3 (1x1, C outputs, Cb = 64 | 128 inputs) + train-mode BatchNorm without the convolution's raw output (csrc/conv_fold.hip).
 * In: D [C][Cb] = g^T A accumulated by a weight-gradient launch on the gated gradient g itself (cleared on the way out), w_bf16 [C][Cb] the
 * forward's weights, gram_totals = the f64 totals of mhe_conv1x1_gram_nhwc on conv3's input ([Cb][Cb] Gram matrix, then [Cb] column sums),
 * rev_stats [64][2][C] whose [.][0][c] shards hold sum_p g, gamma, mean_invstd [2][C], count = pixels.  Out: dgamma, dbeta [C]; dW [C][Cb]
 * ACCUMULATED; w_dg_bf16 [Cb][ld_dg] = (k2 W)^T, the data-gradient weights for g; S_bf16 [Cb][ld_S] = W^T diag(k1) W, the 1x1 weights for
 * conv3's input (as the residual of that launch, or - S_bf16 = w_dg_bf16 + C, ld_S = ld_dg = C + Cb - as the second part of ONE
 * K-concatenated launch, mhe_conv2d_masked_bias_nhwc); c0 [Cb] the per-channel constant.  coef_ws: 2 C floats of scratch. */
int mhe_conv3_bn_fold(float *D, const void *w_bf16, const double *gram_totals, const mhe_stat_t *rev_stats, const float *gamma,
                      const float *mean_invstd, double count, float *dgamma, float *dbeta, float *dW, void *w_dg_bf16, int ld_dg,
                      void *S_bf16, int ld_S, float *c0, float *coef_ws, int C, int Cb, void *stream);
/* 3x3 / stride-1 / pad-1 bf16 convolution with the input tile resident in LDS (csrc/conv_halo.hip; conv2 of a torchvision Bottleneck,
 * hand/network.py:54-61,110, and its data gradient): W = 32 or 16, H a multiple of 256 / W, Cin a multiple of 64 (<= 512), Cout of 128.
 * w_halo = mhe_conv3x3_halo_pack_bf16 of the standard pack [Cout][9 Cin] (same byte count).  in_scale / in_shift (+ relu_in): the producer's
 * BatchNorm (+ ReLU) applied once per element on its way into LDS (zero padding stays zero); a_out (optional, with them): the normalised
 * operand [B][H][W][Cin] written once.  Forward form: stats as mhe_conv2d_nhwc.  Data-gradient form (mask given): outputs (+ residual) gated
 * by mask > 0, BatchNorm-reverse sums of one consumer as mhe_conv2d_masked_nhwc. */
int mhe_conv3x3_halo_supported(int B, int H, int W, int Cin, int Cout);
int mhe_conv3x3_halo_pack_bf16(const void *w, void *w_halo, int Cout, int Cin, void *stream);
int mhe_conv3x3_halo_nhwc(int B, int H, int W, int Cin, int Cout, const void *x, const void *w_halo, void *y, const float *in_scale,
                          const float *in_shift, int relu_in, void *a_out, mhe_stat_t *stats, const void *residual, const void *mask,
                          const void *bn_y0, const float *bn_mean_invstd0, mhe_stat_t *bn_stats0, void *stream);
/* Data-gradient form of mhe_conv3x3_halo_nhwc with the BatchNorm reverse of the convolution's own output gradient on the operand load:
 * operand = k2 g + k1 y_raw + k0 per channel (coef = k2 | k1 | k0 [3][Cin], mhe_bn_bwd_finalize; the arithmetic of mhe_bn_bwd_apply_nhwc), written
 * once to gy_out (optional) for the weight gradient; gx = (conv(operand, w_halo) + residual) [mask > 0], BatchNorm-reverse sums of one consumer. */
int mhe_conv3x3_halo_dgrad_bn_nhwc(int B, int H, int W, int Cin, int Cout, const void *g, const void *y_raw, const float *coef, const void *w_halo,
                                   void *gx, void *gy_out, const void *residual, const void *mask, const void *bn_y0,
                                   const float *bn_mean_invstd0, mhe_stat_t *bn_stats0, void *stream);
int mhe_flow_reverse_chain_supported(int R, int B, int dim, int hidden, int ncoup);
/* mhe_flow_couplings_bf16 / _emit on the fragment-streaming skeleton (csrc/flow_fwd.hip): hidden 512, R % B == 0 (a workgroup = 64 rows
 * of one image; a last chunk of fewer rows computes on zeros and stores nothing for them - the metrics pass' N = 200; the form WITH h1, h2, o
 * needs R % (64 B) == 0), at most 32 couplings.  Same results as mhe_flow_couplings_bf16 up to the
 * order of the f32 accumulation.  w0F, w1F, w2F = net 0's W0 [512][64 (dim zero-padded)], W1 [512][512], W2 [64 (padded)][512] as bf16 in
 * FRAGMENT-MAJOR order (see mhe_flow_reverse_chain_bf16), net k at + k * w_net_stride elements; cond [B][cond_stride] with column
 * (2 net + layer) * 512 + unit; bias2 [nets][64].  h1, h2, o: all three or none - the activations the reverse pass reads
 * (bf16 [nets][R][512] x 2, f32 [nets][R][64]); sign_bits (optional, with them): the signs of h1 / h2 as mhe_flow_reverse_chain_bf16
 * reads them - u64 [nets][R / 64][2 layers][8 waves][64 lanes], bit (unit tile * 4 + row tile) * 4 + e of the MFMA accumulator layout. */
int mhe_flow_couplings_frag_supported(int R, int B, int dim, int hidden, int ncoup);
int mhe_flow_couplings_frag_bf16(const float *in, float *out, const float *cond, int cond_stride, const void *w0F, const void *w1F,
                                 const void *w2F, long w_net_stride, const float *bias2, const float *mask, float *sum_s,
                                 float *log_prob, void *h1, void *h2, float *o, void *sign_bits, int R, int B, int dim, int hidden,
                                 int ncoup, int direction, void *stream);
int mhe_flow_reverse_chain_bf16(const float *x_out, const float *g_x, const float *g_logp, float q_weight, const float *mask,
                                const float *o_pre, const void *sign_bits, const void *w2F, const void *w1F,
                                const void *w0F, long w_net_stride, void *GO_bf16, void *G2_bf16, void *G1_bf16, void *XP_bf16,
                                float *Gc, int cond_stride, float *db2_rows, float *z0, int R, int B, int dim,
                                int hidden, int ncoup, void *stream);
/* src [R][C] f32 (row pitch src_stride) -> dst [R][C] bf16 (optional) and dstT [C][R] bf16: the conditioning gradient as the two operands
 * its consumers read (the conditioning layer's weight gradient; the split-K product that gives the feature's gradient). */
int mhe_pack_transpose_bf16(const float *src, long src_stride, void *dst_bf16, void *dstT_bf16, int R, int C, void *stream);
int mhe_flow_couple_bwd_mixed(const float *x_out, const float *Os, const float *Ot, const float *mask,
                              const float *g_out, const float *g_log_p, float q_weight, float *x_in, float *GOs,
                              float *GOt, float *g_part, void *GOs_bf16, void *GOt_bf16,
                              float *db_s, float *db_t /* optional [64] each: += column sums of GOs / GOt = the l2 bias gradients */,
                              long R, int B, int dim, void *stream);
int mhe_flow_couple_bwd_f32(const float *x_out, const float *Os, const float *Ot, const float *mask,
                            const float *g_out, const float *g_log_p, float q_weight, float *x_in, float *GOs,
                            float *GOt, float *g_part, long R, int B, int dim, void *stream);
int mhe_flow_couple_accum_f32(const float *g_part, const float *GXs, const float *GXt, const float *mask,
                              float *g_in, long R, int dim, void *stream);

/* Body-model path of a ProHMR-style head (reference README.md:26-42) ------------------------
 * 6D rotation representation -> rotation matrix: poses6 [n,6] -> rotmats [n,3,3] row-major with columns (x, y, z);
 * robust = 0: x = n(a), z = n(x x b), y = z x x (hand/manopth/rot6d.py:4-24); robust = 1: the symmetric form (:26-51);
 * normalisation n(v) = v / max(|v|, 1e-8) (:54-60).  Reached at hand/manopth/manolayer.py:150-156.  _bwd: reverse of the
 * Gram-Schmidt form, g_poses6 = (d rotmats / d poses6)^T g_rotmats. */
int mhe_rot6d_to_rotmat_f32(const float *poses6, float *rotmats, long n, int robust, void *stream);
int mhe_rot6d_to_rotmat_bwd_f32(const float *poses6, const float *g_rotmats, float *g_poses6, long n, void *stream);
/* Linear-blend skinning for a body model of runtime size - the arithmetic of hand/manopth/manolayer.py:181-246 with J <= 32
 * joints on any kinematic tree (parents[0] = -1, parents[j] < j), nb shape coefficients, 9(J-1) pose-blend coefficients and
 * NV vertices (SMPL: 24 / 10 / 207 / 6,890), from ROTATION MATRICES rotmats [R,J,3,3] and betas [R,nb].
 *   pose:  j_template [J,3] = J_regressor v_template, j_shapedirs [J,3,nb] = J_regressor shapedirs (the joint regression
 *          folded into the tables, SURVEY.md A2 iii) -> one workspace row per hypothesis (mhe_lbs_workspace_floats(R,J,nb)
 *          floats in all): pose map, betas, the J skinning transforms, posed joints; joints [R,J,3] optional copy
 *   skin:  vertex-fastest tables v_template [3][VP], v_shapedirs [nb][3][VP], v_posedirs [9(J-1)][3][VP], v_weights [J][VP]
 *          (VP >= NV: padded pitch) -> verts [R,NV,3] * scale.  Hypotheses are independent: shard R across GPUs freely. */
size_t mhe_lbs_workspace_floats(int R, int J, int nb);
int mhe_lbs_pose_f32(const float *rotmats, const float *betas, const float *j_template, const float *j_shapedirs,
                     const int *parents, float *workspace, float *joints, int R, int J, int nb, void *stream);
int mhe_lbs_skin_f32(const float *workspace, const float *v_template, const float *v_shapedirs, const float *v_posedirs,
                     const float *v_weights, float *verts, int R, int J, int nb, int NV, int VP, float scale, void *stream);
/* mhe_lbs_skin_f32 on the matrix cores (csrc/lbs_skin.hip: the scheme of csrc/mano_skin.hip at runtime sizes - both products as GEMMs on bf16
 * pieces of the f32 operands, f32-class accuracy).  The table pieces are made once per model: `split` = mhe_lbs_split_floats(J, nb, VP) floats,
 * filled by mhe_lbs_split_tables_f32 from the same four vertex tables (VP a multiple of 32); the skinning launch then takes the workspace rows of
 * mhe_lbs_pose_f32 and `split`.  mhe_lbs_skin_mfma_supported: J <= 32, the hypotheses' pieces within the LDS, the output within 4 GiB. */
size_t mhe_lbs_split_floats(int J, int nb, int VP);
int mhe_lbs_split_tables_f32(const float *v_template, const float *v_shapedirs, const float *v_posedirs, const float *v_weights,
                             float *split, int J, int nb, int VP, void *stream);
int mhe_lbs_skin_mfma_supported(int R, int J, int nb, int NV, int VP);
int mhe_lbs_skin_mfma_f32(const float *workspace, const float *split, float *verts, int R, int J, int nb, int NV, int VP, float scale,
                          void *stream);

/* Train-mode BatchNorm(+ReLU) reverse over NHWC activations of storage `dtype` (F.batch_norm backward):
 *   g' = g [a > 0] (a = the unit's post-activation output, NULL = no ReLU);  xhat = (y - mean) invstd
 *   reduce:   stats[shard][0][c] += sum g',  stats[shard][1][c] += sum g' xhat   (mhe_conv_stat_shards() shards, zeroed by caller)
 *   finalize: dbeta, dgamma and coef [3,C] with  gy = coef0 g' + coef1 y + coef2  (= gamma invstd (g' - dbeta/M - xhat dgamma/M))
 *   apply:    gy (and optionally g' itself, the identity branch's gradient of a residual block)
 * mhe_bn_mean_invstd recovers mean / invstd [2,C] from the forward's statistic shards. */
int mhe_bn_mean_invstd(const mhe_stat_t *stats, float *mean_invstd, int C, double count, float eps, void *stream);
int mhe_bn_bwd_reduce_nhwc(const void *g, const void *a, const void *y, const float *mean_invstd, mhe_stat_t *stats,
                           long P, int C, int dtype, void *stream);
int mhe_bn_bwd_finalize(const mhe_stat_t *stats, const float *gamma, const float *mean_invstd, float *dgamma,
                        float *dbeta, float *coef, int C, double count, void *stream);
int mhe_bn_bwd_apply_nhwc(const void *g, const void *a, const void *y, const float *coef, void *gy, void *g_masked,
                          long P, int C, int dtype, void *stream);
/* 3x3/s2/p1 max pool recording the winning tap (first maximum, as torch), and its reverse gather. */
int mhe_maxpool3x3s2_idx_nhwc(const void *x, void *y, unsigned char *idx, int B, int H, int W, int C, int dtype, void *stream);
int mhe_maxpool3x3s2_bwd_nhwc(const void *gy, const unsigned char *idx, void *gx, int B, int H, int W, int C, int dtype, void *stream);
/* The stem's pool with its BatchNorm folded in (torchvision ResNet.forward: `x = self.relu(self.bn1(x)); x = self.maxpool(x)` differentiated
 * by autograd, reference hand/CrossModalHand.py:455-470).  Forward: y = maxpool(relu(x * scale + shift)) + winning taps from the RAW conv
 * output x (the normalised copy is never written; winners are chosen on the values as stored in `dtype`, i.e. the same winners as
 * mhe_maxpool3x3s2_idx_nhwc on the materialised activation).  Reverse: gx = scatter(gy, idx) [relu(y * scale + shift) > 0] and the
 * BatchNorm-reverse sums of gx into stats[shard][2][C] (as mhe_bn_bwd_reduce_nhwc would add them) in one pass; C / (16 bytes of dtype) must divide 256. */
int mhe_maxpool3x3s2_idx_affine_nhwc(const void *x, const float *scale, const float *shift, void *y, unsigned char *idx, int B, int H, int W,
                                     int C, int dtype, void *stream);
/* ... that also keeps the RAW input at every winner (xwin [B,Ho,Wo,C]), and the BatchNorm-reverse sums taken from the pooled tensors alone
 * (round 4): sum g' = sum g [pooled > 0], sum g' xhat = sum g [pooled > 0] (xwin - mean) invstd over the P = B Ho Wo pooled pixels - the
 * scattered gradient is non-zero only at the winners, so the full-resolution tensor is not walked for the sums (a quarter of the
 * elements, no window gathers).  Differs from mhe_maxpool3x3s2_bwd_bn_nhwc's sums in the last bf16 bit where two windows picked one pixel
 * (that walk rounds the sum of their gradients to `dtype` first). */
int mhe_maxpool3x3s2_idx_affine_win_nhwc(const void *x, const float *scale, const float *shift, void *y, unsigned char *idx, void *xwin, int B,
                                         int H, int W, int C, int dtype, void *stream);
int mhe_pooled_bn_sums_nhwc(const void *g, const void *pooled, const void *xwin, const float *mean_invstd, mhe_stat_t *stats, long P, int C,
                            int dtype, void *stream);
int mhe_maxpool3x3s2_bwd_bn_nhwc(const void *gy, const unsigned char *idx, const void *y, const float *scale, const float *shift,
                                 const float *mean_invstd, mhe_stat_t *stats, void *gx, int B, int H, int W, int C, int dtype, void *stream);
/* (gx may be NULL: the sums only.)  The same walk with the finished BatchNorm-reverse coefficients coef = k2 | k1 | k0 [3][C] (mhe_bn_bwd_finalize):
 * gy_out = k2 (scattered, gated gradient) + k1 y + k0 - mhe_bn_bwd_apply_nhwc's result without the scattered gradient ever being stored. */
int mhe_maxpool3x3s2_bwd_bn_apply_nhwc(const void *gy, const unsigned char *idx, const void *y, const float *scale, const float *shift,
                                       const float *mean_invstd, const float *coef, void *gy_out, int B, int H, int W, int C, int dtype,
                                       void *stream);
/* gx[b,p,c] = g[b,c] / HW, zeroed where mask[b,p,c] <= 0 (mask optional: ReLU gate of the pooled tensor). */
int mhe_avgpool_bwd_nhwc(const float *g, const void *mask, void *gx, int B, int HW, int C, int dtype, void *stream);
/* out[b,2i,2j,c] = g[b,i,j,c] (+ base), 0 (+ base) elsewhere; out is [B,H,W,C], g is [B,ceil(H/2),ceil(W/2),C]:
 * data gradient of stride-2 sampling (zero-dilated operand of a 3x3 data-gradient conv, or 1x1 downsample). */
int mhe_upsample2_nhwc(const void *g, const void *base, void *out, int B, int H, int W, int C, int dtype, void *stream);
/* Optimizer tail (hand/CrossModalHand.py:201,463-470): out[0] = |g|^2 in a fixed summation order (replicas must agree
 * bit for bit; workspace of mhe_sqnorm_workspace_floats() floats); tick: step += 1, sqnorm = 0;
 * adam: clip_grad_norm_(max_norm) (max_norm <= 0: none) folded into torch.optim.Adam's default update,
 * g pre-multiplied by grad_scale (1/world after a sum all-reduce).  step/sqnorm live in device memory. */
size_t mhe_sqnorm_workspace_floats(void);
int mhe_sqnorm_f32(const float *g, size_t n, float *workspace, float *out, void *stream);
int mhe_train_tick(int *step, float *sqnorm, void *stream);
int mhe_adam_step_f32(float *p, const float *g, float *m, float *v, size_t n, const float *sqnorm, const int *step,
                      float lr, float beta1, float beta2, float eps, float max_norm, float grad_scale, void *stream);

/* ---- conditional Glow (q_z_giv_i_model == 'glow', hand/network.py:342-344,693-694,736-742) ------------------
 * PARITY UNPINNED: the class is the third-party nkolot/nflows ConditionalGlow (unpinned git dependency,
 * hand/environment.yml:284), absent from the reference tree; these stages follow the published nflows algorithm
 * (oracle/glow_ref.py).  Dense products go through mhe_linear_f32; context-only terms are per image and indexed
 * by image = (row / row_div) % n_img.  The flow variable is carried zero-padded to 64 columns. */
int mhe_glow_add_image_rows_f32(float *H, const float *img, long img_stride, long R, int C, int row_div, int n_img, void *stream);
int mhe_relu_copy_f32(const float *in, void *out, long n, int out_dtype, void *stream);          /* out f32 or bf16 */
int mhe_glow_glu_residual_f32(float *H, const void *T, int t_dtype, const float *gate, long gate_stride, long R, int C,
                              int row_div, int n_img, void *stream);                         /* T f32 or bf16 */
/* The flow variable is carried zero-padded to ld = ceil(dim/64)*64 columns (64 for the 45-D hand flow, 192 for a 144-D body
 * pose; dim <= 256); params [R, ceil(2T/64)*64] = [shift (T) | unconstrained scale (T)]; transform feature j is column
 * first + 2j; logdet accumulates. */
int mhe_glow_coupling_f32(const float *u, const float *params, float *y, float *logdet, long R, int dim, int first,
                          int n_transform, int inverse, void *stream);
int mhe_pad64_f32(const float *x, float *xp, long R, int dim, void *stream);
/* reverse stages of the sampling direction (train step of the Glow branch; formulas in csrc/glow.hip) */
int mhe_glow_coupling_inv_bwd_f32(const float *v, const float *params, const float *g_y, const float *g_log_p, float q_weight,
                                  float *g_v, float *g_params, long R, int B, int dim, int first, int n_transform, void *stream);
int mhe_glow_glu_bwd_f32(const float *g_h, const void *t3, const float *gate, long gate_stride, void *g_t3, float *g_gate_rows,
                         long R, int C, int row_div, int n_img, int t_dtype, void *stream);     /* t3, g_t3: f32 or bf16 */
int mhe_relu_bwd_add_f32(float *acc, const void *g, const float *h, long n, int g_dtype, void *stream);
/* log_prob[r] = log N(z_r; 0, I) + sign * (logdet[r] + logdet_const); optionally un-pads v_padded into v_out [R,dim]. */
int mhe_glow_finish_f32(const float *z_padded, const float *v_padded, const float *logdet, float *v_out, float *log_prob,
                        long R, int dim, float sign, float logdet_const, void *stream);
/* ... with the log-determinant constant read from the device: log_prob = base(z) + sign * logdet[r] + sum(const_parts[0 .. n_parts)) - a captured
 * HIP graph then follows the parameters from step to step */
int mhe_glow_finish_dev_f32(const float *z_padded, const float *v_padded, const float *logdet, float *v_out, float *log_prob,
                            long R, int dim, float sign, const float *const_parts, int n_parts, void *stream);
/* Per-image reverse stages of the one-launch Glow kernel's tape (csrc/glow.hip; sample-major rows r = n B + b, C columns, bf16 tensors [N B][C]):
 *   glu_bwd_sum:     g_t3 = g_h sigmoid(gate[b]) (bf16);  gct[b][c] = s (1 - s) sum_n g_h t3;  bsum[b][c] = s sum_n g_h    (rows of pitch *_stride)
 *   mask_scale_sum:  g <- g * scale * [t2 > 0] in place;  bsum[b][c] = sum_n g (of the stored values)
 * - the gate's gradient, the dropout + ReLU reverse and the per-image rows of the bias gradients without a second pass over [R][C]. */
int mhe_glow_glu_bwd_sum(const float *g_h, const void *t3, const float *gate, long gate_stride, void *g_t3, float *gct, long gct_stride,
                         float *bsum, long bsum_stride, int N, int B, int C, void *stream);
int mhe_glow_mask_scale_sum(void *g, const void *t2, float scale, float *bsum, long bsum_stride, int N, int B, int C, void *stream);
/* relu'(h) reverse with h (or relu(h): same sign) stored as f32 or bf16: acc += g * [h > 0] */
int mhe_relu_bwd_add_mixed(float *acc, const void *g, const void *h, long n, int g_dtype, int h_dtype, void *stream);
/* Dropout mask bits alone (n elements, n % 8 == 0; byte i = keep bits of elements 8 i .. 8 i + 7, as mhe_dropout writes them), drawn from the
 * device-resident Philox state, which advances. */
int mhe_dropout_bits(unsigned char *bits, long n, float p_drop, unsigned long long *state, void *stream);
/* The conditional Glow's sampling direction, ALL layers in one launch (csrc/glow_fwd.hip; hidden 512, 2 residual blocks per layer, dim <= 48;
 * reference call site hand/network.py:736-742, algorithm oracle/glow_ref.py - parity unpinned).  N hypotheses for each of B images; hypothesis n of
 * image b is row n * row_n + b * row_b of every [R = N B] tensor ((B, 1) sample-major or (1, N) batch-major).
 *   noise [R][dim] f32 -> out [R][dim], log_q [R];   ctab [B][ctab_stride] f32: slot l * 3 = initial-layer context term + bias, slots l * 3 + 1, + 2 =
 *   the blocks' GLU gate pre-activations;   weights bf16 in MFMA fragment order ([rows / 16][K / 32][4][16][8], ops.mfma_fragment_major):
 *   wxF [L][512][64] (initial layer on the 64-padded variable), w0F / w1F [L][2][512][512], wsF / wuF [L][64][512] = the final layer's shift /
 *   unconstrained-scale rows placed at the flow variable's own column (row c = parameter of transform column c, other rows zero), bs / bu [L][64]
 *   their biases likewise;   b0 / b1 [L][2][512];   ainvT [L][64][64], cinv [L][64], const_parts [L] from mhe_glow_affine_f64;
 *   drop_bits [L][2][R * 64] bytes (mhe_dropout_bits' format over [R][512]) or NULL = no dropout.
 * Tape (all NULL, or all given for the train step): v_e, y_e, prm_e f32 [L][R][64] (layer input, coupling output, [shift (T) | us (T) | 0]),
 * tb_e, t2_e, t3_e bf16 [L][2][R][512] (relu(h), dropped second activation, W1 t2 + b1), hf_e bf16 [L][R][512] (the final layer's operand).
 * For mhe_glow_reverse_chain_bf16 (all NULL or all given, N % 64 == 0): prmc_e f32 [L][R][128] = [shift | unconstrained scale] in the flow
 * variable's column order, vb_e bf16 [L][R][64] = the layer input as multiplied, bits_e [L][2][2][R / 64][512] x 8 bytes = [stored value != 0] of
 * relu(h) / t2 in the accumulator layout. */
int mhe_glow_layers_supported(int N, int B, int dim, int hidden, int layers, int blocks);
int mhe_glow_layers_bf16(const float *noise, const float *ctab, int ctab_stride, const void *wxF, const void *w0F, const void *w1F, const void *wsF,
                         const void *wuF, const float *b0, const float *b1, const float *bs, const float *bu, const float *ainvT,
                         const float *cinv, const float *const_parts, const unsigned char *drop_bits, float p_drop, float *out, float *log_q,
                         float *v_e, float *y_e, float *prm_e, void *tb_e, void *t2_e, void *t3_e, void *hf_e, float *prmc_e, void *vb_e,
                         void *bits_e, int N, int B, int dim, int hidden, int layers, int blocks, long row_n, long row_b, void *stream);
/* The reverse of mhe_glow_layers_bf16: the data-gradient chain of all layers in one launch (csrc/glow_rev.hip; 64 hypotheses per image,
 * sample-major rows r = n B + b, hidden 512, 2 blocks per layer).  In: g_x [R][dim] = dL / d sample, g_log_p [B] | NULL with q_weight = -1 / K
 * (dL / d log q[r] = g_log_p[r % B] q_weight), the tape (v_e, prmc_e, t3_e, bits_e of mhe_glow_layers_bf16), ctab (the gates), the weights as bf16
 * fragment-major TRANSPOSES (wsT / wuT [L][512][64]: row u, k = flow-variable column; w1T / w0T [L][2][512][512] = W^T; wxT [L][64][512] = Wx^T), ainv
 * [L][64][64] = A^-1.  Out (all written): gv_e f32 [L][R][64] (dL / d layer output: for dA^-1 = gv^T y and dc^-1), gpc_e bf16 [L][R][128] =
 * [g_shift | g_us] in the flow variable's column order, gt3_e / gt2_e bf16 [L][2][R][512], gh0_e bf16 [L][R][512] (operands of the grouped weight-
 * gradient launches), and per-image rows: gct [B][ctab_stride] (gradient of ctab: gates and initial-layer context terms), bsum [B][L * 2 * 2 * 512]
 * (rows of the block bias gradients: [l][blk][b0 | b1]), bfsum [B][L * 128] (rows of the final layer's bias gradient, column order). */
int mhe_glow_reverse_chain_supported(int R, int B, int dim, int hidden, int layers, int blocks);
int mhe_glow_reverse_chain_bf16(const float *g_x, const float *g_log_p, float q_weight, const float *v_e, const float *prmc_e, const void *t3_e,
                                const void *bits_e, const float *ctab, int ctab_stride, const void *wsT, const void *wuT, const void *w1T,
                                const void *w0T, const void *wxT, const float *ainv, float p_drop, float *gv_e, void *gpc_e, void *gt3_e,
                                void *gt2_e, void *gh0_e, float *gct, float *bsum, float *bfsum, int R, int B, int dim, int hidden, int layers,
                                int blocks, void *stream);
/* The ActNorm + LU re-parameterisation on the device (csrc/glow_affine.hip; nflows transforms.ActNorm / LULinear, oracle/glow_ref.py): one
 * workgroup per layer, float64.  param_ptrs: [layers][6] device pointers (log_scale, shift, lower_entries, upper_entries,
 * unconstrained_upper_diag, bias: f32 tensors of `features`, features (features - 1) / 2 entries in numpy's tril_indices(-1) / triu_indices(1)
 * order).  Outputs, f32 zero-padded to 64: A = L U diag(exp(log_scale)) [layers][64][64], c = L U shift + bias [layers][64], A^-1, its
 * transpose, c^-1 = -A^-1 c; const_parts [layers] = sum(log_scale) + sum(log diag U); workspace (mhe_glow_affine_workspace_doubles doubles)
 * keeps the float64 factors for mhe_glow_reparam_bwd_f64. */
size_t mhe_glow_affine_workspace_doubles(int layers, int features);
int mhe_glow_affine_f64(const void *param_ptrs, int layers, int features, float eps, float *A, float *c, float *Ainv, float *AinvT,
                        float *cinv, float *const_parts, double *workspace, void *stream);
/* Gradients of the six small parameter tensors of every layer from dL/dA^-1 (g_ainv_ptrs[l]: f32 [64][64]), dL/dc^-1 (g_cinv_ptrs[l]: f32 [64])
 * and S = q_sign * sum(g_log_p[0 .. n_log_p)) = sum_r dL/dlog q[r] (g_log_p may be NULL); written through grad_ptrs ([layers][6] device pointers,
 * the order of param_ptrs).  Formulas: csrc/glow_affine.hip. */
int mhe_glow_reparam_bwd_f64(const void *g_ainv_ptrs, const void *g_cinv_ptrs, const float *g_log_p, int n_log_p, float q_sign, int layers,
                             int features, const double *workspace, const void *grad_ptrs, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* MHE_H */

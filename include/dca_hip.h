/* C ABI of libdca_hip.so -- the MI355X (gfx950) kernels behind the DCANet cost-volume hot path.
 *
 * The reference (cocowy1/Cost-Volume-Aggregation-in-Stereo-Matching-Revisited) has no FFI of its own
 * on this path: every step is a PyTorch op called from Python.  The boundary a maintainer binds is
 * therefore "one extern-C launcher per op the Python path calls", taking raw device pointers,
 * explicit sizes and the HIP stream to enqueue on.  No allocation, no synchronisation and no
 * torch types inside; the caller owns all memory.  Every function returns a hipError_t as int
 * (0 = hipSuccess; invalid arguments -> hipErrorInvalidValue before anything is launched).
 *
 * Tensors are dense fp32 NC[D]HW (the reference's layout).  File:line citations are into the
 * reference tree.  The ctypes binding that mirrors this header lives in
 * cost-volume-aggregation-in-stereo-matching-revisited_amd/_lib.py; INTEGRATION.md shows how the
 * reference's own modules would call it.
 */
#ifndef DCA_HIP_H
#define DCA_HIP_H

#include <hip/hip_runtime_api.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Bumped whenever an argument list below changes; the ctypes loader (_lib.py) refuses a library built from another
 * version of this header. */
#define DCA_ABI_VERSION 19
int dca_abi_version(void);

/* storage types of the reduced-precision inference path (0 = fp32) */
#define DCA_BF16 1
#define DCA_FP16 2

/* ---- cost volumes ------------------------------------------------------------------------------
 * build_gwc_volume(refimg_fea, targetimg_fea, maxdisp, num_groups)  models/submodule.py:157-167
 * (groupwise_correlation, submodule.py:148-154).  ref,tgt: (B,C,H,W); vol: (B,G,maxdisp,H,W). */
int dca_gwc_volume_fwd(const float* ref, const float* tgt, float* vol, int B, int C, int H, int W, int maxdisp,
                       int num_groups, hipStream_t stream);
/* autograd of the above (the reference back-propagates through maxdisp slice-assign nodes). */
int dca_gwc_volume_bwd(const float* gvol, const float* ref, const float* tgt, float* gref, float* gtgt, int B, int C,
                       int H, int W, int maxdisp, int num_groups, hipStream_t stream);
/* build_concat_volume(refimg_fea, targetimg_fea, maxdisp)  models/submodule.py:134-145; vol: (B,2C,maxdisp,H,W) */
int dca_concat_volume_fwd(const float* ref, const float* tgt, float* vol, int B, int C, int H, int W, int maxdisp,
                          hipStream_t stream);
int dca_concat_volume_bwd(const float* gvol, float* gref, float* gtgt, int B, int C, int H, int W, int maxdisp,
                          hipStream_t stream);

/* Fused builder (volume_fused.hip): gwc volume and, when Cc > 0, the concat volume written into ONE tensor
 * vol (B, num_groups + 2*Cc, maxdisp, H, W) -- no torch.cat((gwc_volume, concat_volume), 1) (models/gwcnet_dca_g.py:217-220)
 * -- from `nseg` (1..3) channel segments of the correlation features: refs[s], tgts[s]: (B, seg_channels[s], H, W), read
 * in place instead of their concatenation gwc_feature = torch.cat((l2, l3, l4), 1) (gwcnet_dca_g.py:60).  refs / tgts /
 * seg_channels are HOST arrays.  cref, ctgt: (B, Cc, H, W) or NULL.  dtype: 0 = fp32 volume, DCA_BF16 / DCA_FP16 = the
 * reduced-precision inference path's storage type.  W % 4 == 0, maxdisp % 4 == 0, 16-byte aligned tensors; every
 * segment width must be a multiple of channels / num_groups. */
int dca_cost_volume_fwd(const float* const* refs, const float* const* tgts, const int* seg_channels, int nseg,
                        const float* cref, const float* ctgt, int Cc, void* vol, int B, int H, int W, int maxdisp,
                        int num_groups, int dtype, unsigned* vmax, hipStream_t stream);
/* vmax (may be null; fp32 volume without concat part, B * H <= DCA_AMAX_CSLOTS): per-channel slots [g][b * H + y] that receive
 * max |volume| of group g for the f16x2 convolution that reads it (no separate dca_cmax_f32 pass) */

/* ---- softmax over dim 1 / disparity_regression ---------------------------------------------------
 * x: (B,K,HW).  mode 0: out (B,K,HW) = F.softmax(x, dim=1)   (models/gwcnet_dca_g.py:238,248,...)
 *               mode 1: out (B,HW)   = disparity_regression(F.softmax(x,1), K)  (gwcnet_dca_g.py:238-239,263-264)
 *               mode 2: out (B,HW)   = disparity_regression(x, K)               (models/submodule.py:127-131)
 * bwd: mode 0: aux = softmax output, g = (B,K,HW); mode 1: aux = logits x, g = (B,HW); mode 2: g = (B,HW). */
int dca_softargmin_fwd(const float* x, float* out, int B, int K, long HW, int mode, hipStream_t stream);
int dca_softargmin_bwd(const float* aux, const float* g, float* gx, int B, int K, long HW, int mode,
                       hipStream_t stream);

/* Fused training head: disparity_regression(F.softmax(F.upsample(logits[:,None], scale_factor=(s,s,s),
 * mode='trilinear').squeeze(1), 1), s*n)  -- models/gwcnet_dca_g.py:261-264 (s = 8) and the four heads of the baseline
 * models/gwcnet.py:219-237 (s = 4).  s in {2,4,8}; logits: (B,n,hc,wc), n <= 64;
 * disp: (B,1,s*hc,s*wc).  bwd needs g1 = B*n*(s*hc)*(s*wc) floats of scratch. */
int dca_up_softargmin_fwd(const float* logits, float* disp, int B, int n, int hc, int wc, int scale, hipStream_t stream);
int dca_up_softargmin_bwd(const float* logits, const float* gdisp, float* g1, float* glogits, int B, int n, int hc,
                          int wc, int scale, hipStream_t stream);

/* ---- 3D convolutions (nn.Conv3d / nn.ConvTranspose3d, bias=False) ----------------------------------
 * models/submodule.py:121-124 (convbn_3d), models/gwcnet_dca_g.py:141-168, models/augment/cva.py:13-55,
 * models/augment/SelfAttention_bn.py:136-160.
 *
 * dca_conv3d_prep_weight re-lays a PyTorch weight out as wt[tap][a][b] (a < Apad contraction channels,
 * b < Bpad output channels, zero padded), K = 27 or 1 taps:
 *   src_ab = 0: src is [B][A][K]  (Conv3d weight (Cout,Cin,k,k,k) used forward: A = Cin, B = Cout)
 *   src_ab = 1: src is [A][B][K]  (ConvTranspose3d weight (Cin,Cout,...) used forward, or a Conv3d
 *                                  weight used for its backward-data pass: A = Cout, B = Cin)
 *   flip = 1 reverses the taps (backward-data of a stride-1 convolution).
 * Padding rules: Apad = Cin rounded up to 8 (ksize 3) or exactly 32/64 (ksize 1);
 *                Bpad = 32 if (ksize 1 | transposed | (stride 1 & Cout <= 32)) else 64.
 * Btotal / b_off: the source has Btotal output channels of which this call lays out the slice [b_off, b_off+B)
 * (layers with more output channels than one launch produces are run as several launches). */
int dca_conv3d_prep_weight(const float* w, float* wt, int A, int B, int Apad, int Bpad, int K, int src_ab, int flip,
                           int Btotal, int b_off, hipStream_t stream);
/* y = epilogue(conv(x [, x2], wt)).  ksize 3: pad 1, stride 1|2, or transposed (stride 2, pad 1,
 * output_padding 1).  ksize 1: Cin = 32 or 64; if x2 != NULL the input is cat([x, x2], dim=1) with 32
 * channels each (cva.py:69 without materialising the cat).  C1 = channels of x when x2 is given.
 * epilogue(v) = act(v*scale[co] + shift[co] + res_pre) + res_post, act(v) = v > 0 ? v : slope*v
 * (slope 1: none, 0: ReLU, 0.1: LeakyReLU); scale/shift/res_pre/res_post may be NULL.
 * One launch produces Cout <= 64 (3x3x3 conv) or <= 32 (transposed, 1x1x1) channels and writes them at channel
 * offset co_off of a CoutTotal-channel y (scale/shift/res_* are indexed in the full tensor).
 * Backward-data passes reuse this entry with re-laid-out weights:
 *   stride-1 conv   -> stride-1 conv,   src_ab = 1, flip = 1
 *   stride-2 conv   -> transposed conv, src_ab = 1, flip = 0
 *   transposed conv -> stride-2 conv,   src_ab = 0, flip = 0 */
int dca_conv3d_forward(const float* x, const float* x2, const float* wt, float* y, const float* scale,
                       const float* shift, const float* res_pre, const float* res_post, float slope, int N, int Cin,
                       int C1, int Cout, int CinPad, int CoutTotal, int co_off, int Di, int Hi, int Wi, int Do, int Ho,
                       int Wo, int ksize, int stride, int transposed, hipStream_t stream);
/* dw[cy*s_cy + cx*s_cx + k] = sum_{n,o} dy[n,cy,o] x[n,cx,stride*o-1+k]  (ksize 3) / sum dy*x (ksize 1).
 * x: (N,Cx,Di,Hi,Wi), dy: (N,Cy,Do,Ho,Wo).  Conv3d: x = input, dy = grad of output, dw layout
 * (Cout,Cin,K).  ConvTranspose3d: x = grad of output (fine), dy = input (coarse), stride 2, dw layout
 * (Cin,Cout,K).  `part` is scratch of dca_conv3d_wgrad_workspace(...) floats. */
long dca_conv3d_wgrad_workspace(int N, int Cx, int Cy, int Do, int Ho, int Wo, int ksize, int stride);
int dca_conv3d_wgrad(const float* x, const float* dy, float* part, float* dw, int N, int Cx, int Cy, int Di, int Hi,
                     int Wi, int Do, int Ho, int Wo, int ksize, int stride, long s_cy, long s_cx, hipStream_t stream);

/* 1x1x1 convolutions with fp32 tensors on the bf16 matrix pipe, fp32-grade (conv1_x3.hip: the exact three-way split of
 * conv3d_bf16x3.hip on an LDS-free data path): same contract as dca_conv3d_forward(ksize 1) -- x (N,C1,S) [, x2 (N,C2,S)],
 * y (N,CoutTotal,S) channels [co_off, co_off+Cout), Cout <= 32, (C1,C2) in {(32,0),(64,0),(32,32)}, S % 4 == 0.
 * wfrag (dca_conv1_x3_weight_bytes(A) bytes) from w read as W[b][a] = src_ab ? w[a*Btotal + b_off + b] : w[(b_off+b)*A + a]. */
long dca_conv1_x3_weight_bytes(int A);
int dca_conv1_x3_prep_weight(const float* w, void* wfrag, int A, int Bn, int src_ab, int Btotal, int b_off,
                             hipStream_t stream);
int dca_conv1_x3_forward(const float* x, const float* x2, const void* wfrag, float* y, const float* scale,
                         const float* shift, const float* res_pre, const float* res_post, float slope, int N, int C1,
                         int C2, int Cout, int CoutTotal, int co_off, long S, hipStream_t stream);

/* Batched weight re-layout: ONE launch for n descriptors of dca_conv3d_prep_weight (kind 0) / dca_conv3d_x3_prep_weight
 * (kind 1) work (prep_many.hip) -- a training step re-packs every conv weight after the optimizer update, and ~130
 * separate 5-us launches cost more than the work.  table: n device-resident 72-byte records
 *   { const float* src; void* dst; int kind, A, Bn, Apad, Bpad, K, src_ab, flip, Btotal, b_off, NCH, pad; long total; }
 * (kind 1 uses A, Bn, src_ab, flip, NCH = ceil(A/16), total = dca_conv3d_x3_weight_bytes/2; kind 0 total = K*Apad*Bpad;
 * kind 3 = dca_conv3d_x2_prep_weight with the fields of kind 1 and total = (dca_conv3d_x2_weight_bytes - 16)/2). */
int dca_conv3d_prep_many(const void* table, int n, hipStream_t stream);

/* "bf16x3" split-precision 3x3x3 / stride-1 / pad-1 convolution (conv3d_bf16x3.hip): every fp32 operand is split exactly
 * into three bf16 terms and the six partial products >= 2^-16 run on the bf16 matrix pipe with fp32 accumulation --
 * fp32-grade results (dropped terms <= 2^-23 relative, no range restriction) at 2.67x fewer matrix-pipe cycles than
 * the fp32 MFMA kernel.  Replaces the same nn.Conv3d calls as dca_conv3d_forward (models/submodule.py:121-124,
 * models/augment/cva.py:13-55) and, with src_ab = 1 / flip = 1, their backward-data.
 *   dca_conv3d_x3_weight_bytes(Cin, Cout): size of the pre-split weight image wx.
 *   dca_conv3d_x3_prep_weight: w is the PyTorch weight; A contraction channels, B output channels;
 *       src_ab ? w[a][b][27] : w[b][a][27]; flip reverses the tap order (backward-data).  wx 16-byte aligned.
 *   dca_conv3d_x3_forward: y (N,Cout,D,H,W) = act((conv(x, w)) * scale[c] + shift[c] + res_pre) + res_post, the epilogue
 *       contract of dca_conv3d_forward; any Cin / Cout (zero padded to 16 / 32 internally); Cin*D*H*W*4 < 2^31. */
long dca_conv3d_x3_weight_bytes(int Cin, int Cout);
int dca_conv3d_x3_prep_weight(const float* w, void* wx, int A, int B, int src_ab, int flip, hipStream_t stream);
int dca_conv3d_x3_forward(const float* x, const void* wx, float* y, const float* scale, const float* shift,
                          const float* res_pre, const float* res_post, float slope, int N, int Cin, int Cout, int D,
                          int H, int W, hipStream_t stream);

/* ConvTranspose3d(k 3, stride 2, padding 1, output_padding 1) with fp32 tensors on the bf16 matrix pipe, same exact
 * three-way split (deconv3d_x3.hip): `cost_agg.conv3` forward (models/augment/cva.py:21-29) and the backward-data of
 * `cost_agg.conv1` (cva.py:16-17).  x (N,Cin,Di,Hi,Wi) -> y (N,Cout<=32,2Di,2Hi,2Wi) = act(deconv * scale + shift +
 * res_pre) + res_post, the epilogue contract of dca_conv3d_forward.  wx = dca_conv3d_x3_prep_weight(w, wx, A = Cin,
 * B = Cout, src_ab, flip 0) (w[a][b][27] for src_ab 1).  Requires Wi % 4 == 0, 16-byte aligned x / wx, 8-byte aligned y and
 * residuals (hipErrorInvalidValue otherwise: callers use dca_conv3d_forward(transposed 1)). */
int dca_deconv3d_x3_forward(const float* x, const void* wx, float* y, const float* scale, const float* shift,
                            const float* res_pre, const float* res_post, float slope, int N, int Cin, int Cout, int Di,
                            int Hi, int Wi, hipStream_t stream);

/* The same convolution without epilogue, fused with the BatchNorm batch statistics of its output (training-mode
 * convbn_3d, models/submodule.py:121-124; csrc/bn_fused_stats.h): part (Cout * nchunk * 4 doubles, nchunk =
 * dca_conv3d_x3_stats_chunks(...)) receives ONE partial per (channel, workgroup), part[(c*nchunk + i)*4 + {0,1,2,3}] =
 * {K, n, sum (y - K), sum (y - K)^2} with a shift K taken from the partial's own data (no loss of variance for
 * |mean| >> std); dca_bn_finalize_centered(part, nchunk, ...) re-centres and sums them in double.  Fixed summation order,
 * a function of the data alone: bitwise reproducible. */
long dca_conv3d_x3_stats_chunks(int N, int Cout, int D, int H, int W);
int dca_conv3d_x3_forward_stats(const float* x, const void* wx, float* y, double* stat_part, int N, int Cin, int Cout,
                                int D, int H, int W, hipStream_t stream);
/* The same for the 1x1x1 convolutions (conv1_x3.hip; part covers all CoutTotal channels, every channel slice of a sliced
 * convolution fills its own channels) and the transposed convolution (deconv3d_x3.hip). */
long dca_conv1_x3_stats_chunks(int N, long S);
int dca_conv1_x3_forward_stats(const float* x, const float* x2, const void* wfrag, float* y, double* stat_part, int N,
                               int C1, int C2, int Cout, int CoutTotal, int co_off, long S, hipStream_t stream);
long dca_deconv3d_x3_stats_chunks(int N, int Di, int Hi, int Wi);
int dca_deconv3d_x3_forward_stats(const float* x, const void* wx, float* y, double* stat_part, int N, int Cin, int Cout,
                                  int Di, int Hi, int Wi, hipStream_t stream);
/* [mean | invstd | scale | shift] (4*C floats) from those partials, with the running-statistics update of training-mode
 * nn.BatchNorm3d (momentum, unbiased variance); running_mean / running_var may both be null. */
int dca_bn_finalize_centered(const double* part, int nchunk, const float* gamma, const float* beta, float* running_mean,
                             float* running_var, float momentum, float eps, float* stats, int* zexps,
                             const unsigned* rpre_slots, int rpre_n, const unsigned* rpost_slots, int rpost_n, int C,
                             hipStream_t stream);

/* Weight gradient of the 3x3x3 / stride-1 / pad-1 convolution on the bf16 matrix pipe with the same exact three-way
 * bf16 split (conv3d_wgrad_bf16x3.hip); replaces dca_conv3d_wgrad for ksize 3, stride 1 (autograd's dW of the nn.Conv3d
 * calls above).  dw[cy*s_cy + cx*s_cx + tap] = sum_{n,voxels} dy[n][cy] * x[n][cx] shifted by the tap; x (N,Cx,D,H,W),
 * dy (N,Cy,D,H,W); part = scratch of dca_conv3d_wgrad_x3_workspace floats.  Requires W % 4 == 0 and 16-byte aligned
 * x / dy (hipErrorInvalidValue otherwise: callers use dca_conv3d_wgrad).  Deterministic, no atomics. */
long dca_conv3d_wgrad_x3_workspace(int N, int Cx, int Cy, int D, int H, int W);
int dca_conv3d_wgrad_x3(const float* x, const float* dy, float* part, float* dw, int N, int Cx, int Cy, int D, int H,
                        int W, long s_cy, long s_cx, hipStream_t stream);

/* "f16x2" split-precision 3x3x3 / stride-1 / pad-1 convolution and weight gradient (conv3d_f16x2.hip,
 * conv3d_wgrad_f16x2.hip, conv3d_wgrad_s2_f16x2.hip) -- the kernels the models run: every CHANNEL of an fp32 operand is
 * scaled by its own power of two 2^exps[c] (scaled maximum in [2^14, 2^15): inside the f16 range whatever the channel's
 * magnitude and whatever the other channels'), split into two f16 terms (|x - h - l| <= 2^-22 |x|) and the three partial
 * products >= 2^-11 run on the f16 matrix pipe with fp32 accumulation; the result is scaled back exactly (v_ldexp_f32).
 * Half the matrix-pipe cycles of the bf16x3 kernels at the same measured error against fp64, per output channel
 * (tests/test_gpu_parity.py::test_f16x2_per_channel_scales).  Same operators as dca_conv3d_x3_forward /
 * dca_conv3d_wgrad_x3 (models/submodule.py:121-124, models/augment/cva.py:13-55) and, with src_ab = 1 / flip = 1, their
 * backward-data.
 *   Per-channel maxima ("slots"): slots[c * DCA_AMAX_CSLOTS + s], s < nslots = the bit patterns of partial maxima of |x| over
 *       channel c, one slot per producing workgroup (plain stores, one writer per slot, nothing to zero-initialise; the
 *       consumer takes the unsigned maximum).  Filled by dca_bn_apply (zmax), dca_bn_backward (dmax),
 *       dca_conv3d_x2_forward (y_cmax) for the tensors they write, or by dca_cmax_f32 (a read pass; nslots =
 *       dca_bn_num_chunks(C, S)).  dca_cmax_exps: exps[c] from the slots.
 *   Packed operand "px2": the fp32 tensor (N,C,D,H,W), C % 8 == 0, as its two scaled f16 terms, per sample
 *       [term 2][C/8][D][H][W][8] f16 (4 bytes per element, like fp32), written by dca_bn_apply_pack / dca_bn_backward_pack
 *       (or dca_bn_apply_pack with stats = null: plain packing) together with the exponents it was scaled by.  Consumers:
 *       dca_conv3d_x2_forward[_stats] (packed = 1) and dca_conv3d_wgrad_x2 (x_packed / dy_packed).
 *   dca_conv3d_x2_prep_weight: packs w for ONE launch over an operand with exponents xexps (it folds 2^-xexps[k] into
 *       the weight rows and gives every output channel its own scale): x_slots == null -> xexps is an input;
 *       else xexps is derived from the slots and written (A ints) for later users of the same operand.
 *   dca_conv3d_x2_forward[_stats]: contracts of dca_conv3d_x3_forward[_stats]; y_cmax (may be null; forces the fused
 *       epilogue variant) = per-channel slots that receive max |y|, nslots = dca_conv3d_x2_stats_chunks(...).
 *   dca_conv3d_wgrad_x2 / _s2_x2: contracts of dca_conv3d_wgrad_x3 / the stride-2 form below; exps of both operands. */
#define DCA_AMAX_CSLOTS 1024
int dca_cmax_f32(const float* x, int N, int C, long S, unsigned* slots, hipStream_t stream);
int dca_cmax_exps(const unsigned* slots, int nslots, int C, int* exps, hipStream_t stream);
long dca_conv3d_x2_weight_bytes(int Cin, int Cout);
int dca_conv3d_x2_prep_weight(const float* w, void* wx, int A, int B, int src_ab, int flip, const unsigned* x_slots,
                              int nslots, int* xexps, hipStream_t stream);
int dca_conv3d_x2_forward(const void* x, int packed, const int* xexps, const void* wx, float* y, const float* scale,
                          const float* shift, const float* res_pre, const float* res_post, float slope, unsigned* y_cmax,
                          int N, int Cin, int Cout, int D, int H, int W, hipStream_t stream);
long dca_conv3d_x2_stats_chunks(int N, int Cout, int D, int H, int W);
int dca_conv3d_x2_forward_stats(const void* x, int packed, const int* xexps, const void* wx, float* y, double* stat_part,
                                int N, int Cin, int Cout, int D, int H, int W, hipStream_t stream);
/* dca_conv3d_wgrad_s2_x2: the same arithmetic for the weight gradient of the STRIDE-2 convolution `cost_agg.conv1` and of
 * the transposed convolution `cost_agg.conv3` (models/augment/cva.py:16-29; conv3d_wgrad_s2_f16x2.hip):
 * dw[cy*s_cy + cx*s_cx + tap] = sum_{n,o} c[n][cy][o] * f[n][cx][2o + tap - 1], f = fine tensor (N,Cx,D,H,W), c = coarse tensor
 * (N,Cy,(D+1)/2,(H+1)/2,(W+1)/2); for the transposed convolution f = dy, c = x.  Requires W % 4 == 0, (W+1)/2 % 4 == 0 and
 * 16-byte aligned tensors (hipErrorInvalidValue otherwise: callers use dca_conv3d_wgrad). */
long dca_conv3d_wgrad_s2_x2_workspace(int N, int Cx, int Cy, int D, int H, int W);
int dca_conv3d_wgrad_s2_x2(const float* f, const int* f_exps, const float* c, const int* c_exps, float* part,
                           float* dw, int N, int Cx, int Cy, int D, int H, int W, long s_cy, long s_cx, hipStream_t stream);
/* The STRIDE-2 3x3x3 convolution itself (padding 1) with the f16x2 arithmetic (conv3d_s2_f16x2.hip): `cost_agg.conv1` =
 * Conv3d(32, 64, 3, stride 2, padding 1) forward (models/augment/cva.py:16-17) and the backward-data of `cost_agg.conv3` =
 * ConvTranspose3d(64, 32, 3, stride 2, ...) (cva.py:21-29: the same operator over dy, weight read as [output][contraction]).
 * x (N,Cin,Di,Hi,Wi) fp32, Wi % 4 == 0, 16-byte aligned -> y (N,Cout,(Di+1)/2,(Hi+1)/2,(Wi+1)/2) = act(conv * scale + shift) +
 * res_post (all optional: the training launches use none; inference folds the BatchNorm; no res_pre); y_cmax (may be null) =
 * per-channel slots [c][s], s < dca_conv3d_s2x2_out_slots(...), that receive max |y| for an f16x2 convolution reading y.  Weights are packed per launch like
 * dca_conv3d_x2_prep_weight (another fragment layout: K-steps of 4 channels x 4 taps, output-channel blocks of 64). */
long dca_conv3d_s2x2_weight_bytes(int Cin, int Cout);
int dca_conv3d_s2x2_prep_weight(const float* w, void* wx, int A, int B, int src_ab, int flip, const unsigned* x_slots,
                                int nslots, int* xexps, hipStream_t stream);
long dca_conv3d_s2x2_out_slots(int N, int Cout, int Di, int Hi, int Wi);
int dca_conv3d_s2x2_forward(const float* x, const int* xexps, const void* wx, float* y, const float* scale, const float* shift,
                            float slope, const float* res_post, unsigned* y_cmax, int N, int Cin, int Cout, int Di, int Hi, int Wi,
                            hipStream_t stream);
long dca_conv3d_wgrad_x2_workspace(int N, int Cx, int Cy, int D, int H, int W);
int dca_conv3d_wgrad_x2(const void* x, int x_packed, const int* xexps, const void* dy, int dy_packed, const int* yexps,
                        float* part, float* dw, int N, int Cx, int Cy, int D, int H, int W, long s_cy, long s_cx,
                        hipStream_t stream);

/* Single-output-channel 3x3x3 convolution (the logit heads: nn.Conv3d(32, 1, 3, padding=1, bias=False),
 * models/gwcnet_dca_g.py:154-168 `classif*.2`, models/augment/cva.py:51-53 `classify.2`).  w is the PyTorch weight
 * (1,C,3,3,3) as is.  The 27 taps become a GEMM axis so forward / weight gradient reuse the matrix-core kernels:
 *   forward: T (N,27,D,H,W) = dca_conv3d_forward(ksize 1, weight laid out [ci][tap]);  y = dca_conv3d_c1_gather(T)
 *   wgrad:   G (N,27,D,H,W) = dca_conv3d_c1_expand(dy);  dW = dca_conv3d_wgrad(x, G, ksize 1, s_cy 1, s_cx 27)
 *   bwd_data: dy (N,1,..) -> dx (N,C,..) directly. */
int dca_conv3d_c1_gather(const float* T, float* y, int N, int D, int H, int W, hipStream_t stream);
/* wgrad without the expanded tensor (round 3): dw[ci*27 + tap] = sum x[ci][v] dy[v - offset(tap)], the tap-shifted views of dy
 * are built while the tiles are fetched.  W % 4 == 0, 16-byte aligned x / dy, N*C*D*H*W*4 < 2^31 (hipErrorInvalidValue
 * otherwise: use the expand form); part = dca_conv3d_wgrad_workspace(N, C, 27, D, H, W, 1, 1) floats. */
int dca_conv3d_c1_wgrad(const float* x, const float* dy, float* part, float* dw, int N, int C, int D, int H, int W,
                        hipStream_t stream);
int dca_conv3d_c1_expand(const float* dy, float* G, int N, int D, int H, int W, hipStream_t stream);
int dca_conv3d_c1_bwd_data(const float* dy, const float* w, float* dx, int N, int C, int D, int H, int W,
                           hipStream_t stream);

/* ---- BatchNorm3d + activation + residual -------------------------------------------------------------
 * nn.BatchNorm3d defaults (eps 1e-5, momentum 0.1) as used by convbn_3d, models/submodule.py:121-124.
 * dca_bn_stats:    part[(c*nchunk+i)*2+{0,1}] = partial (sum, sum of squares) of x - K_c in double, with the per-channel
 *                  shift K_c = x[0,c,0] at part[C*nchunk*2 + c] (no catastrophic cancellation when |mean| >> std);
 *                  nchunk = dca_bn_num_chunks(C, S); part holds C*nchunk*2 + C doubles.
 * dca_bn_finalize: stats = [mean | invstd | scale | shift] (4*C floats); training != 0 uses the batch
 *                  statistics and updates running_mean/var (unbiased), else uses the running stats.
 * dca_bn_apply:    z = act(scale*y + shift + res_pre) + res_post.
 * dca_bn_backward: given dz -> dy (grad of the conv output), dgb = [dgamma | dbeta | ...] (4*C floats),
 *                  optional g_out = grad w.r.t. res_pre (= dz masked by the activation).
 * zmax / dmax (dca_bn_apply: of z, dca_bn_backward: of dy; may be null): per-channel slots (see "f16x2" above; nslots =
 *                  dca_bn_num_chunks(C, S)) that receive max |.| of the tensor written; ymax (dca_bn_apply[_pack]; may be
 *                  null): slots that receive max |y - mean| per channel (it bounds |xhat| for the backward pass' scale).
 * zexps (dca_bn_finalize[_centered]; may be null, training only): per-channel scale exponents for z = act(BN(y) + res_pre) +
 *                  res_post from the bound |z| <= |gamma| sqrt(count) + |beta| + max |res_pre| + max |res_post| (the residual
 *                  tensors' per-channel slots rpre_slots / rpost_slots, may be null) -- known before z is written, so that
 * dca_bn_apply_pack writes z directly in the packed px2 format (stats = null: plain packing of y with zexps) and, with
 *                  zf != null, a second fp32 copy for the other readers of z (zmax: its per-channel slots); nslots of ymax / zmax =
 *                  dca_bn_pack_chunks(C, S).
 * dca_bn_backward_pack: dca_bn_backward (no res_pre / g_out) writing dy in the packed px2 format; dyexps (C ints, out) =
 *                  the exponents it was scaled by (bound from max |g|, this launch's reduce pass, and max |xhat| = invstd *
 *                  ymax); gmax = scratch of C * DCA_AMAX_CSLOTS words. */
int dca_bn_num_chunks(int C, long S);
int dca_bn_pack_chunks(int C, long S);
int dca_bn_stats(const float* x, double* part, int N, int C, long S, hipStream_t stream);
int dca_bn_finalize(const double* part, int nchunk, double count, const float* gamma, const float* beta,
                    float* running_mean, float* running_var, float momentum, float eps, int training, float* stats,
                    int* zexps, const unsigned* rpre_slots, int rpre_n, const unsigned* rpost_slots, int rpost_n, int C,
                    hipStream_t stream);
int dca_bn_apply(const float* y, const float* stats, const float* res_pre, const float* res_post, float* z, int N,
                 int C, long S, float slope, unsigned* zmax, unsigned* ymax, hipStream_t stream);
int dca_bn_apply_pack(const float* y, const float* stats, const int* zexps, void* zp, int N, int C, long S, float slope,
                      unsigned* ymax, const float* res_pre, const float* res_post, float* zf, unsigned* zmax,
                      hipStream_t stream);
int dca_bn_backward(const float* dz, const float* y, const float* res_pre, const float* stats, double* part,
                    float* dgb, float* dy, float* g_out, int N, int C, long S, float slope, int training,
                    unsigned* dmax, hipStream_t stream);
int dca_bn_backward_pack(const float* dz, const float* y, const float* stats, double* part, float* dgb, void* dyp,
                         int* dyexps, unsigned* gmax, const unsigned* ymax, int ymax_slots, int N, int C, long S,
                         float slope, int training, hipStream_t stream);

/* ---- AvgPool3d((3,3,3), stride 2, padding 1) -- models/augment/cva.py:39 ---------------------------- */
int dca_avgpool3d_fwd(const float* x, float* y, long NC, int Di, int Hi, int Wi, hipStream_t stream);
/* res (may be null): another gradient of the pooled tensor's input, added to gx on the way (ops._PoolFork) */
int dca_avgpool3d_bwd(const float* gy, float* gx, const float* res, const float* res2, long NC, int Di, int Hi, int Wi,
                     hipStream_t stream);   /* res / res2 (may be null): further gradients of the same input, added on the way */

/* ---- F.interpolate(scale_factor=(s,s,s), mode='trilinear'), align_corners=False -----------------------
 * models/augment/cva.py:64 (s = 2), models/gwcnet_dca_g.py:251,256 (s = 2), :261 (s = 8). */
int dca_trilinear_fwd(const float* x, float* y, long NC, int Di, int Hi, int Wi, int scale, hipStream_t stream);
int dca_trilinear_bwd(const float* gy, float* gx, long NC, int Di, int Hi, int Wi, int scale, hipStream_t stream);

/* ---- homogeneous-region context injection -- SemanticLevelContext.forward, semantic_level.py:96-126 ---
 * x,key: (B,C,n,HW); preds: (B,n,HW) logits.  key = feats_sl + x.  Side outputs (saved for backward):
 * kstar (B,HW) int32 argmax class, e (B,HW), pm (B,HW) = p[k*], denom (B,n).  part: scratch of
 * B*ceil(HW/256)*n floats (per-workgroup class sums, added in a fixed order: bitwise reproducible). */
int dca_context_inject_fwd(const float* x, const float* preds, float* key, int* kstar, float* e, float* pm,
                           float* denom, float* part, int B, int C, int n, long HW, hipStream_t stream);
int dca_context_inject_bwd(const float* dkey, const float* x, const float* preds, const int* kstar, const float* e,
                           const float* pm, const float* denom, float* dx, float* dpreds, float* dw, float* T,
                           float* part, int B, int C, int n, long HW, hipStream_t stream);

/* ---- per-pixel disparity attention core -- SelfAttentionBlock.forward, SelfAttention_bn.py:70-94 ------
 * q,k,v,out: (B,C,n,HW), heads of 8 channels, softmax(q k^T / sqrt(8)) v over the n bins (n <= 64). */
int dca_disp_attention_fwd(const float* q, const float* k, const float* v, float* out, int B, int C, int n, long HW,
                           hipStream_t stream);
int dca_disp_attention_bwd(const float* q, const float* k, const float* v, const float* dout, float* dq, float* dk,
                           float* dv, int B, int C, int n, long HW, hipStream_t stream);

/* ---- the steps right after the path ------------------------------------------------------
 * Convex x4 up-sampling, PropgationNet_4x.forward after its conv (models/submodule.py:366-373, copy
 * models/gwcnet_dca_g.py:114-124): mask_logits (B,144,h,w) with channel = k*16 + i*4 + j, disp (B,1,h,w) in 1/4-res
 * pixels -> up (B,1,4h,4w) = sum_k softmax_k(mask)[k,i,j] * 4*disp[3x3 zero-padded neighbour k].
 * bwd: glogits (B,144,h,w), gdisp (B,1,h,w); wk = B*9*h*w floats of scratch. */
int dca_convex_up4_fwd(const float* mask_logits, const float* disp, float* up, int B, int h, int w, hipStream_t stream);
int dca_convex_up4_bwd(const float* mask_logits, const float* disp, const float* gup, float* glogits, float* gdisp,
                       float* wk, int B, int h, int w, hipStream_t stream);

/* Stereo focal loss, StereoFocalLoss.loss_per_level + LaplaceDisp2Prob (models/loss.py:206-240, 60-128), for `nlev`
 * estimates of equal shape (B,K,HW) that share one ground truth gt (B,HW) ALREADY scaled/pooled to that resolution
 * (loss.py:210-215 stays on the host: one adaptive pooling per resolution).  ests / gests / weights are HOST arrays of
 * nlev (<= 8) device pointers / floats.  out: nlev+1 floats = the per-level losses (un-weighted, loss.py:238) and
 * sum_l weights[l]*loss_l (focal_loss, loss.py:16-24).  K <= 256.  work: dca_focal_loss_workspace(...) doubles, must be
 * handed unchanged to the backward call, which writes gests[l] = d(out[nlev]) / d(ests[l]) * gloss[0]. */
long dca_focal_loss_workspace(int nlev, int B, long HW);
int dca_focal_loss_fwd(const float* const* ests, const float* weights, int nlev, const float* gt, double* work,
                       float* out, int B, int K, long HW, float focal_coefficient, hipStream_t stream);
int dca_focal_loss_bwd(const float* const* ests, float* const* gests, const float* weights, int nlev, const float* gt,
                       const double* work, const float* gloss, int B, int K, long HW, float focal_coefficient,
                       hipStream_t stream);

/* ---- reduced-precision inference path (BASELINE configs 2 "bf16" and 5 "fp16") ---------------------------------
 * Activations stored as bf16 / fp16, ONE native MFMA product per multiply, fp32 accumulation and fp32 epilogue
 * arithmetic (folded BatchNorm affine, activation, residuals); forward only.  dtype codes: */
/* 3x3x3 stride-1 convolution (convbn_3d + ReLU of models/submodule.py:121-124 in eval mode).  wx: dca_conv3d_lp_weight_bytes
 * bytes, filled by dca_conv3d_lp_prep_weight (A, B, src_ab, flip as in dca_conv3d_prep_weight).  x: (N,Cin,D,H,W) in the
 * 2-byte type, or fp32 when in_f32; y, res_pre, res_post: (N,Cout,D,H,W) in the 2-byte type, or fp32 when out_f32.
 * y = act(conv * scale + shift + res_pre) + res_post. */
long dca_conv3d_lp_weight_bytes(int Cin, int Cout);
int dca_conv3d_lp_prep_weight(const float* w, void* wx, int A, int B, int src_ab, int flip, int dtype,
                              hipStream_t stream);
int dca_conv3d_lp_forward(const void* x, const void* wx, void* y, const float* scale, const float* shift,
                          const void* res_pre, const void* res_post, float slope, int N, int Cin, int Cout, int D, int H,
                          int W, int dtype, int in_f32, int out_f32, hipStream_t stream);

/* 1x1x1 convolution (pointwise GEMM), Cout <= 32, over one or two 2-byte inputs (implicit channel concat: the `fuse` conv
 * of models/augment/cva.py:55,69; `cost_agg.redir`, cva.py:23; the tap-expansion GEMM of the logit heads).  w: (Cout,
 * C1 + C2) fp32 row major -> wfrag (dca_conv1_lp_weight_bytes bytes).  x: (N,C1,S), x2: (N,C2,S) or NULL (C2 = 0);
 * y / res_pre / res_post: (N,Cout,S) in the 2-byte type, or fp32 when out_f32.  S % 4 == 0. */
long dca_conv1_lp_weight_bytes(int C1, int C2);
int dca_conv1_lp_prep_weight(const float* w, void* wfrag, int Cout, int C1, int C2, int dtype, hipStream_t stream);
int dca_conv1_lp_forward(const void* x, const void* x2, const void* wfrag, void* y, const float* scale,
                         const float* shift, const void* res_pre, const void* res_post, float slope, int N, int C1,
                         int C2, int Cout, long S, int dtype, int out_f32, hipStream_t stream);

/* Mixed-storage forms of dca_conv3d_forward (exact-fp32 MFMA arithmetic, wt as for dca_conv3d_forward with Bpad 64 /
 * 32): transposed == 0: 3x3x3 stride-2 conv, x (N,Cin,Di,Hi,Wi) 2-byte -> y (N,Cout<=64,Do,Ho,Wo) fp32, no residuals
 * (cost_agg.conv1, models/augment/cva.py:16-17); transposed == 1: ConvTranspose3d(3,s2,p1,op1), x fp32 -> y, res_pre,
 * res_post (N,Cout<=32,2Di,2Hi,2Wi) 2-byte (cost_agg.conv3 + ReLU(. + redir(x)) [+ outer residual], cva.py:21-29). */
int dca_conv3d_forward_mixed(const void* x, const float* wt, void* y, const float* scale, const float* shift,
                             const void* res_pre, const void* res_post, float slope, int N, int Cin, int Cout, int CinPad,
                             int Di, int Hi, int Wi, int Do, int Ho, int Wo, int transposed, int dtype,
                             hipStream_t stream);
/* Reduced-precision 3x3x3 stride-2 convolution (conv3d_s2_lp.hip; cost_agg.conv1 = convbn_3d(32, 64, 3, 2, 1) + ReLU,
 * cva.py:16-17): x (N,Cin,D,H,W) 2-byte, ONE MFMA product, fp32 accumulation -> y (N,Cout<=64,ceil(D/2),..) fp32 =
 * act(conv * scale + shift).  w: (Cout,Cin,3,3,3) fp32 -> wx (dca_conv3d_s2_lp_weight_bytes(Cin) bytes).  W % 4 == 0. */
long dca_conv3d_s2_lp_weight_bytes(int Cin);
int dca_conv3d_s2_lp_prep_weight(const float* w, void* wx, int Cin, int Cout, int dtype, hipStream_t stream);
int dca_conv3d_s2_lp_forward(const void* x, const void* wx, float* y, const float* scale, const float* shift, float slope,
                             int N, int Cin, int Cout, int D, int H, int W, int dtype, hipStream_t stream);
/* Reduced-precision ConvTranspose3d(3, s2, p1, op1) (deconv3d_lp.hip; cost_agg.conv3 + ReLU(. + redir) [+ outer residual],
 * cva.py:21-29): x (N,Cin<=64,Di,Hi,Wi) fp32 rounded on the fly, ONE MFMA product, fp32 accumulation; y, res_pre, res_post
 * (N,Cout<=32,2Di,2Hi,2Wi) 2-byte.  wx = dca_conv3d_lp_prep_weight(w (Cin,Cout,3,3,3), wx, Cin, Cout, src_ab 1, flip 0, dtype). */
int dca_deconv3d_lp_forward(const float* x, const void* wx, void* y, const float* scale, const float* shift,
                            const void* res_pre, const void* res_post, float slope, int N, int Cin, int Cout, int Di,
                            int Hi, int Wi, int dtype, hipStream_t stream);
/* nn.AvgPool3d((3,3,3), 2, 1) (cva.py:39): x (NC,Di,Hi,Wi) 2-byte -> y (NC,ceil/2...) fp32; Wi % 4 == 0.
 * F.interpolate(scale_factor=(2,2,2), mode='trilinear') (cva.py:64): x (NC,Di,Hi,Wi) fp32 -> y (NC,2Di,2Hi,2Wi) 2-byte. */
int dca_avgpool3d_lp_fwd(const void* x, float* y, long NC, int Di, int Hi, int Wi, int dtype, hipStream_t stream);
int dca_trilinear_up2_lp_fwd(const float* x, void* y, long NC, int Di, int Hi, int Wi, int dtype, hipStream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* DCA_HIP_H */

/*
 * moyolo.h -- C ABI of libmoyolo.so: MI355X (gfx950) kernels for the DecoderTracker per-frame
 * tracking inference path.  Plain C: raw device pointers, integer shapes,
 * a hipStream_t passed as void*.  No torch types.
 *
 * Conventions ("Native op" row):
 *   - every entry point only ENQUEUES work on the caller's stream; no allocation, no sync, no
 *     global state (graph-capturable, re-entrant);
 *   - the caller owns every buffer; outputs need not be zero-initialised;
 *   - return value 0 on success, a negative MOY_E* code otherwise (shape/argument errors are
 *     detected on the host BEFORE anything is launched);
 *   - dtype codes: MOY_F32 = 0, MOY_BF16 = 1, MOY_F16 = 2 (the reference's `half` switch, predictor.py:131).  "T" below means the dtype selected by `dtype`.
 *   - activations are channels-last: an image tensor is [B, H, W, C] with a row (pixel) stride
 *     `ld` in elements, so channel slices of wider (concat) buffers are addressed in place.
 *
 * Each entry cites the reference interface it replaces (paths relative to the reference root).
 */
#ifndef MOYOLO_H
#define MOYOLO_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MOY_F32 0
#define MOY_BF16 1
#define MOY_F16 2
/* moy_gemm only (round 5): fp32 tensors exactly as MOY_F32 (A, W, R, C are float), but the products run on the 16-bit matrix
 * cores in SPLIT precision -- every operand x = hi + lo * 2^-11 with hi = fp16(x), lo = fp16((x - hi) * 2^11), three products
 * hi.hi, hi.lo, lo.hi (v_mfma_f32_16x16x32_f16, fp32 accumulation in two accumulator sets combined at the end); lo.lo is dropped:
 * about 22 mantissa bits per product at 16/3 of the fp32 matrix rate.  Operands must be finite and below 65504 in magnitude.
 * A (and A2, R, C, pre) are float; W is handed over PRE-SPLIT by the caller: fp16 [2][N][Kpad32] (Kpad32 = K rounded up to 32, zero
 * padded), plane 0 = fp16(w), plane 1 = fp16((w - plane 0) * 2^11) -- the same bytes as the fp32 matrix, split once instead of in
 * every tile that reads it. */
#define MOY_F32X3 3

#define MOY_OK 0
#define MOY_EINVAL (-22)   /* bad shape / argument */
#define MOY_ENOSYS (-38)   /* combination not implemented */
#define MOY_ELAUNCH (-5)   /* hipLaunch error */

#define MOY_ACT_NONE 0
#define MOY_ACT_SILU 1
#define MOY_ACT_RELU 2
#define MOY_ACT_SIGMOID 3

/* ABI/version probe. */
int moy_version(void);
/* Human-readable text for a MOY_E* code. */
const char* moy_strerror(int code);

/* --------------------------------------------------------------------------------------------
 * Fused implicit-GEMM:  C[m, n] = epilogue( sum_k A(m, k) * W[n, k] )
 *
 * Replaces, on the hot path: Conv.forward (ultralytics/nn/modules/conv.py:36-38, Conv2d+BN+SiLU),
 * the 1x1/3x3 convs inside C2f/Bottleneck/SPPF (nn/modules/block.py:129-134,178-182,281-283),
 * MYDecoder.input_proj (nn/modules/head.py:838-839,1012-1029) and every nn.Linear of
 * MYDecoder / MOTRDecoderLayer / MSDeformAttn / MLP (head.py:1031-1113, transformer.py:149-161,
 * 246-287, 627-652), each with its BN / bias / activation / residual / LayerNorm fused.
 *
 * A operand (channels-last activations, dtype T):
 *   ksize == 1: row m of A is A[arow(m) * lda + k];  arow(m) = a_rows ? a_rows[m] : m
 *               (a gathered A must span < 2 GiB: it is addressed through one buffer descriptor)
 *   ksize == 3: m = (b, oy, ox) over [B, Hout, Wout]; k = (ky*3+kx)*Cin + c; the element is
 *               A[((b*Hin + oy*stride+ky-1)*Win + ox*stride+kx-1) * lda + c], 0 outside the image
 *               (pad = 1).  Cin % 8 == 0 (16-bit types) / % 4 (f32).
 *   A2 (optional, ksize 1 only): added element-wise to A before the product (q = k = x + pos).
 *   a_mask (optional, ksize 1 only): rows with a_mask[arow(m) % mask_period] == 0 read as zero
 *               (valid_mask * feats, head.py:1039; the mask belongs to the token, gathered or not).
 * W: [N, Kpad] dtype T, K contiguous, Kpad = K rounded up to 64 (bf16) / 32 (f32), zero padded.
 * Epilogue, in this order (all optional, fp32 math):
 *   v = acc * scale[n] + shift[n]   (BN folded to scale/shift, or bias with scale == NULL)
 *   v = act(v)
 *   v += R[m * ldr + n]             (residual, dtype T)
 *   v = LayerNorm_n(v) * ln_g[n] + ln_b[n]   (eps 1e-5; requires N == 256)  [+ optional fused narrow head, see dot_*]
 *   C[m * ldc + n] = v              (dtype T, or fp32 when out_f32 != 0)
 *   C == NULL is allowed when the fused narrow head (dot_*) is given: only dot_out is produced (enc_score_head over all
 *   S tokens without materialising enc_output for them, head.py:1036-1042; the selected rows are recomputed afterwards).
 * 16-bit launches with N % 256 == 0, K >= 512, K % 64 == 0 (3x3: Cin % 64 == 0) and >= 384 tiles of 256 x 256 run the large-tile
 * LDS-DMA kernel (csrc/gemm_dma.hip): results bit-identical to the tiled kernel.
 * 16-bit 1x1 launches with M >= 65536 run a weight-stationary kernel (csrc/gemm_wreg.hip), results bit-identical to the tiled
 * kernel in store mode: N % 256 == 0 with K in {128, 256, 384, 512} (incl. c_rows_per_batch, and `pre` at K == 256), N == 128 with
 * K in {128, 192, 256} (`pre` at K == 128), the value form N % 512 == 0, plane_cols == 32, K in {128, 256} (incl. plane_cols
 * together with c_rows_per_batch: one pyramid level per launch), and the score mode (C == NULL, ln_* + dot_*, N == 256, K in
 * {128, 256}, optionally over row runs run_* with a second A numbering run_a_*), which evaluates the LayerNorm statistics in one
 * pass (E[v^2] - mean^2), i.e. equal up to fp32 rounding.  Refusals the host checks before any launch (MOY_ENOSYS; moy_gemm_query
 * answers the same without launching): M * lda * 2 > 4 GiB with row runs, a remapped C of more than 2 GiB, a head plane of more
 * than 1 GiB, a seed of more than 2 GiB, lda / ldc not a multiple of 8 elements or a base that is not 16-byte aligned.
 * -------------------------------------------------------------------------------------------- */
typedef struct moy_gemm_args {
  const void* A;
  const void* A2;
  const int32_t* a_rows;
  int32_t a_rows_bound;   /* with a_rows: every a_rows[m] < a_rows_bound (rows of the A buffer) */
  const uint8_t* a_mask;
  int32_t mask_period;
  int64_t lda;
  const void* W;
  int32_t M, N, K;        /* K = ksize*ksize*Cin (unpadded) */
  int32_t ksize, stride;  /* ksize 1 or 3; stride 1 or 2 */
  int32_t B, Hin, Win, Hout, Wout, Cin; /* used when ksize == 3 */
  const float* scale;
  const float* shift;
  int32_t act;
  const void* R;
  int64_t ldr;
  const float* ln_g;
  const float* ln_b;
  void* C;
  int64_t ldc;
  int32_t out_f32;
  int32_t dtype;
  /* optional output row remap (0 = off): row m is stored at C row
   * (m / c_rows_per_batch) * c_batch_stride + m % c_rows_per_batch -- lets one launch scatter the
   * [B, h*w] rows of a pyramid level into the level-major [B, S] token buffer (head.py:1023-1028). */
  int32_t c_rows_per_batch;
  int32_t c_batch_stride;
  /* optional narrow head fused behind the LayerNorm epilogue (requires ln_g): for j < dot_n (<= 8)
   *   dot_out[m * dot_n + j] = sum_n LN(v)[m, n] * dot_w[j * 256 + n] + dot_b[j]     (all fp32)
   * -- enc_score_head(enc_output(x)) without re-reading the features (head.py:1041-1042). */
  const float* dot_w;
  const float* dot_b;
  float* dot_out;
  int32_t dot_n;
  /* optional accumulator seed (NULL = zeros), ksize 1 only: the row m = (b, y, x) of a [B, pre_h, pre_w] raster starts from
   *   pre[((b * (pre_h/2) + y/2) * (pre_w/2) + x/2) * ld_pre + n]            (fp32, pre_h and pre_w even)
   * i.e. from the nearest-2x upsampled rows of a half-resolution product.  A 1x1 conv commutes with nn.Upsample(nearest), so
   * Conv1x1(Concat[Upsample(u), s]) = W_u.u (at HALF resolution, seeded here) + W_s.s: the neck's Upsample + Concat
   * (yolo_track.yaml:28-33, conv.py:295-297) need no kernel and no full-resolution copy of u. */
  const float* pre;
  int64_t ld_pre;
  int32_t pre_h, pre_w;
  /* with A2: only output columns n < a2_cols see A + A2, the others see A (0 = all columns; a multiple of 256).  One launch
   * then makes q | k | v of nn.MultiheadAttention: q = k = x + pos, v = x (transformer.py:637-640). */
  int32_t a2_cols;
  /* optional column planes (0 = off; ksize 1, plane_cols a multiple of 256): output column n is stored at
   *   C[(n / plane_cols) * plane_stride + m * ldc + n % plane_cols]
   * -- one launch over concatenated weights writes each 256-column group as its own contiguous [M, 256] matrix (the six
   * value_proj outputs of the decoder layers: a layer's slice is then dense in HBM for the deformable gather). */
  int32_t plane_cols;
  int64_t plane_stride;
  /* optional ROW RUNS (run_levels = 0: off; only with the narrow head and C == NULL, i.e. the score pass; no a_mask; 16-bit types in the
   * weight-stationary score kernel, round 6: MOY_F32 / MOY_F32X3 in the tiled kernel at any launch size):
   * the launch visits only the rows
   *     b * run_period + run_tok0[l] + y * run_pitch[l] + x      b < M / run_period, l < run_levels, y < run_rows[l], x < run_len[l]
   * -- per batch element one rectangle of every pyramid level: the tokens whose anchors are VALID (`_generate_anchors`,
   * nn/modules/head.py:1007).  A masked token's feature is LN(enc_output.bias) whatever the frame shows (head.py:1039), so its
   * score is a constant that the caller writes once; at 1088x608 that is 46 % of the rows.  Rows outside the runs are
   * neither read nor written.  MOY_ENOSYS when a 16-bit shape does not take the weight-stationary score kernel. */
  int32_t run_levels, run_period;
  int32_t run_tok0[4], run_pitch[4], run_len[4], run_rows[4];
  /* round 4: the rows of A may be numbered differently from the rows the scores are written to (A = one pyramid level's own
   * [B, h*w, C] tensor, scores = the level-major [B, S] token raster): A row = b * run_a_period + (token - run_a_off).
   * run_a_period == 0: the same numbering (A row = score row). */
  int32_t run_a_period, run_a_off;
  /* round 4, optional POST 1x1 conv: the conv's ONLY consumer is a 1x1 Conv (C2f.cv1 behind a down-sampling Conv:
   * yolo_track.yaml:19-20, block.py:225-235 `self.cv1(x)`), applied to every finished output tile while it sits on chip:
   *     C [M, post_n] = post_act( (act(conv(A)) rounded to the storage type) . post_W^T * post_scale + post_shift )
   * so the conv's own output never reaches HBM.  post_W: storage type, [post_n][ceil64(N)] like W.  Supported by the stride-2
   * weight-stationary kernel for Cin = 64, N = post_n = 128, SiLU twice; every other shape: MOY_ENOSYS (never ignored). */
  const void* post_W;
  const float* post_scale;
  const float* post_shift;
  int32_t post_n, post_act;
} moy_gemm_args;

int moy_gemm(const moy_gemm_args* args, void* stream);

/* --------------------------------------------------------------------------------------------
 * Stem: fused preprocess + first Conv (3x3, stride 2, Cin = 3) + BN + SiLU.
 * Replaces BasePredictor.preprocess (ultralytics/engine/predictor.py:117-134: BGR->RGB, HWC->CHW,
 * float, /255) fused into layer 0 Conv.forward (conv.py:36-38; yolo_track.yaml:17).
 *   in_fmt 0: uint8  [B, H, W, 3] BGR   (value/255 computed in fp32 exactly as the reference)
 *   in_fmt 1: float32 [B, 3, H, W] RGB in [0, 1] (LoadTensor source, data/loaders.py:316-332)
 *   w: fp32 [27, Cout] (k = (ky*3+kx)*3 + c_rgb, Cout contiguous); scale/shift fp32 [Cout]
 *   out: T [B, H/2, W/2, Cout] with pixel stride ldc.   Cout multiple of 8, <= 64.
 * -------------------------------------------------------------------------------------------- */
int moy_stem_conv(const void* in, int in_fmt, int B, int H, int W, const float* w, const float* scale,
                  const float* shift, int Cout, void* out, int64_t ldc, int dtype, void* stream);

/* Same stem on the matrix cores (bf16 output, uint8 BGR input only): K = 27 padded to 32,
 * wpad: bf16 [Cout, 32] with k = (ky*3+kx)*3 + c_rgb (zeros for k >= 27).  Cout in {16, 32, 64}.
 * Input is converted as bf16(u8 * (1/255)) (the fp32 stem keeps the reference's exact u8/255). */
int moy_stem_conv_mfma(const void* in_u8, int B, int H, int W, const void* wpad, const float* scale, const float* shift,
                       int Cout, void* out, int64_t ldc, void* stream);

/* Round 6: the stem of the split-fp16 engine (MOY_F32X3): uint8 BGR input, fp32 output.  The pixel bytes are exact fp16 values, the weights
 * arrive split -- wsplit: fp16 [2][Cout][32], plane 0 = fp16(w), plane 1 = fp16((w - plane 0) * 2^11), in moy_stem_l1_fused's k order (below;
 * mo_yolo_amd.ops.stem_weights_x3) -- two matrix products per output tile; y = SiLU((sum) * scale / 255 + shift) in fp32
 * (predictor.py:117-134 + conv.py:36-38; the exact fp32 engine keeps moy_stem_conv).  Cout in {16, 32, 64}; out 16-byte aligned, ldc % 4 == 0;
 * MOY_ENOSYS for W % 4 != 0 or B*H*W*3 >= 4 GiB (the caller keeps moy_stem_conv). */
int moy_stem_conv_x3(const void* in_u8, int B, int H, int W, const void* wsplit, const float* scale, const float* shift,
                     int Cout, void* out, int64_t ldc, void* stream);

/* Stem AND the first down-sampling conv in one launch (16-bit engines): uint8 BGR frames -> layer 0 (3x3 s2, 3 -> 32) -> layer 1
 * (3x3 s2, 32 -> 64), each Conv + BN + SiLU (predictor.py:117-134; yolo_track.yaml:17-18; conv.py:36-38).  The 32-channel
 * half-resolution tensor never reaches HBM (csrc/stem_l1.hip).
 *   in_u8 [B, H, W, 3] BGR (H, W multiples of 4, pointer 4-byte aligned)
 *   w0: IEEE half [32][32] for BOTH types T (the frame bytes enter the matrix cores as the halfs 1024 + x, exact; the bias
 *       1024 * sum_k w0[n][k] is removed inside), stem weights in the kernel's k order: k = q*8 + e; q < 3: (ky = q, kx = e / 3, c_bgr = e % 3), e = 0..7;
 *       q = 3: e = 0..2: (ky = e, kx = 2, c_bgr = 2), e >= 3: zero  (c_rgb = 2 - c_bgr; mo_yolo_amd.ops.stem_weights_fused)
 *   scale0 / shift0 fp32 [32] (BN folded; the preprocess' 1/255 is applied inside); w1: T [64][320], k = (ky*3+kx)*32 + c;
 *   scale1 / shift1 fp32 [64]; out: T [B, H/4, W/4, 64] with pixel stride ldc.   T = bf16 / fp16 (fp32: MOY_ENOSYS). */
int moy_stem_l1_fused(const void* in_u8, int B, int H, int W, const void* w0, const float* scale0, const float* shift0, const void* w1,
                      const float* scale1, const float* shift1, void* out, int64_t ldc, int dtype, void* stream);

/* A whole C2f block in one launch (16-bit engines): the first C2f of the backbone, 64 -> [32 | 32] -> 64 channels, n = 1 with
 * shortcut (yolo_track.yaml:19; ultralytics/nn/modules/block.py:219-240 C2f.forward, :271-283 Bottleneck; conv.py:36-38):
 *   y0 | y1 = SiLU(BN(cv1 . x));  z = SiLU(BN(m.cv1 * y1));  y2 = y1 + SiLU(BN(m.cv2 * z));  out = SiLU(BN(cv2 . [y0 | y1 | y2])).
 * y0, y1, z, y2 never reach HBM (csrc/c2f_fused.hip); each is rounded to T as the four-launch path (moy_gemm x 4) stores it.
 *   x   T [B, H, W, 64] with pixel stride ldx;   out T [B, H, W, 64] with pixel stride ldo
 *   w_cv1 T [64][kp_cv1] (k = input channel);  w_m1, w_m2 T [32][kp_m], k = (ky*3+kx)*32 + c;  w_cv2 T [64][kp_cv2], k over
 *   [y0 | y1 | y2];  scale_* / shift_* fp32 (BN folded), [64], [32], [32], [64].   T = bf16 / fp16 (fp32: MOY_ENOSYS). */
typedef struct moy_c2f_args {
  const void* x; int64_t ldx;
  int32_t B, H, W;
  const void* w_cv1; int32_t kp_cv1; const float* scale_cv1; const float* shift_cv1;
  const void* w_m1; const float* scale_m1; const float* shift_m1;
  const void* w_m2; const float* scale_m2; const float* shift_m2; int32_t kp_m;
  const void* w_cv2; int32_t kp_cv2; const float* scale_cv2; const float* shift_cv2;
  void* out; int64_t ldo;
  int32_t dtype;
} moy_c2f_args;

int moy_c2f_fused(const moy_c2f_args* args, void* stream);

/* SPPF pooling: y1 = maxpool5(x), y2 = maxpool5(y1), y3 = maxpool5(y2) (stride 1, pad 2, -inf
 * padding) == windows 5/9/13 of x.  Replaces the three nn.MaxPool2d calls of SPPF.forward
 * (nn/modules/block.py:129-134).  x: T [B,H,W,C] stride ldx; y1..y3: stride ldy. C % 8 == 0. */
int moy_sppf_pool(const void* x, int64_t ldx, int B, int H, int W, int C, void* y1, void* y2, void* y3,
                  int64_t ldy, int dtype, void* stream);

/* Nearest 2x upsample into a channel slice: nn.Upsample(None, 2, 'nearest') + Concat
 * (yolo_track.yaml:28-33; conv.py:295-297).  x [B,H,W,C] -> y [B,2H,2W,C]. C % 8 == 0. */
int moy_upsample2x(const void* x, int64_t ldx, int B, int H, int W, int C, void* y, int64_t ldy, int dtype,
                   void* stream);

/* Narrow linear heads, N <= 8 outputs per row, one wavefront per row (K % 64 == 0, K <= 1024):
 *   y[m, j] = sum_k X[xrow(m)*ldx + k] * Wt[j*K + k] + bias[j]      (X dtype T; Wt, bias, y fp32)
 * mode 0: plain                     (enc_score_head / dec_score_head, head.py:852,856)
 * mode 1: y = sigmoid(y + inverse_sigmoid(ref[m, j])), N == 4
 *         (box refinement, transformer.py:709; inverse_sigmoid eps 1e-5, nn/modules/utils.py:34-38)
 * mode 2: y = y + aux[aux_rows[m], j], N == 4           (enc_bbox_head + anchors, head.py:1045)
 * x_rows (optional) gathers input rows (top-k selected tokens). */
int moy_rowdot(const void* X, int64_t ldx, const int32_t* x_rows, int M, int K, const float* Wt,
               const float* bias, int N, int mode, const float* aux, const int32_t* aux_rows, float* y, int dtype,
               void* stream);

/* Three-layer box head in one launch (MLP(256, 256, 4, num_layers=3), nn/modules/transformer.py:149-161):
 *   y[m, :] = W2 . relu(W1 . relu(W0 . x[xrow(m)] + b0) + b1) + b2, then mode as moy_rowdot (1: box refinement
 *   transformer.py:709, 2: + anchors head.py:1045).  X dtype T (bf16 / fp16; fp32 returns MOY_ENOSYS: use moy_gemm x 2 +
 *   moy_rowdot), W0, W1: T [256, 256] (rows = output channels); b0, b1 fp32 [256]; w2 fp32 [4, 256]; b2 fp32 [4]; y fp32 [M, 4].
 *   The hidden activations are rounded to T exactly where the three separate launches store them. */
int moy_mlp_head(const void* X, int64_t ldx, const int32_t* x_rows, int M, const void* W0, const float* b0, const void* W1,
                 const float* b1, const float* w2, const float* b2, int mode, const float* aux, const int32_t* aux_rows,
                 float* y, int dtype, void* stream);

/* The row-wise tail of a decoder layer in one launch (MOTRDecoderLayer.forward after the deformable sampling,
 * nn/modules/transformer.py:642-652, + the box refinement of MOTRTransformerDecoder.forward, :705-709):
 *   e2      = LayerNorm(samp . Wp^T + bp + e1)                     (cross_attn.output_proj, norm2)
 *   out     = LayerNorm(relu(e2 . W1^T + b1) . W2^T + b2 + e2)     (linear1, linear2, norm3)
 *   ref_out = sigmoid(MLP3(out) + inverse_sigmoid(ref_in))         (dec_bbox_head[i]; eps 1e-5)
 * T = bf16 / fp16 (fp32 returns MOY_ENOSYS: moy_gemm x 3 + moy_mlp_head / moy_rowdot, the parity path).  All weight matrices
 * are T, [out features, in features] row-major with the in-feature pitch of their own width (256, or d_ffn for W2);
 * biases / LayerNorm vectors fp32; w2 fp32 [4, 256]; d_ffn a multiple of 256; hidden width 256.  Every intermediate is rounded
 * to T where the separate launches store it; the LayerNorm statistics are one-pass (E[v^2] - mean^2).  d_ffn <= 2048 (MOY_ENOSYS
 * beyond: linear1's bias is staged on chip whole); every T matrix 16-byte aligned with a row pitch that is a multiple of 8 elements
 * (samp, e1, out, out_xp, qpos: whole 512-byte rows move in 16-byte pieces). */
/* MFMA-fragment order of a row-major 16-bit matrix W [N][K] (N, K multiples of 32): 2 KB blocks [N/32][K/32], a block = the two
 * 16-row halves j of its 32 rows, each 64 lanes x 16 bytes: lane (r = lane & 15, q = lane >> 4) holds W[32 g + 16 j + r][32 pn + 8 q .. + 7].
 * Byte offset of that piece: (((g * (K/32) + pn) * 2 + j) * 64 + lane) * 16.  It is the order in which the row-wise decoder kernels
 * (moy_decoder_tail, moy_decoder_mid, moy_msda_raw0) hold weights in registers: a wave's request for a panel is 2 KB CONTIGUOUS instead
 * of 32 rows x 64 bytes.  Measured on MI355X (tools/probes/l2_segments.hip, every CU streaming the same L2-resident table): 37.5 GB/s per
 * CU for the row-major request shape, 132 GB/s for the contiguous one.  mo_yolo_amd.ops.pack_mfma_a() produces it. */
typedef struct moy_decoder_tail_args {
  const void* samp; int64_t ld_samp;   /* T [M, 256] */
  const void* e1;   int64_t ld_e1;     /* T [M, 256] residual */
  int32_t M;
  const void* Wp; const float* bp; const float* ln2_g; const float* ln2_b;
  const void* W1; const float* b1;     /* [d_ffn, 256], [d_ffn] */
  const void* W2; const float* b2;     /* [256, d_ffn], [256] */
  int32_t d_ffn;
  const float* ln3_g; const float* ln3_b;
  void* out; int64_t ld_out;           /* T [M, 256] */
  const void* B0; const float* c0; const void* B1; const float* c1; const float* w2; const float* c2;
  const float* ref_in; float* ref_out; /* fp32 [M, 4] */
  int32_t dtype;
  /* optional (round 3; NULL = off): out_xp = out + qpos, T [M, 256] -- the q = k operand of the next layer's self-attention
   * (with_pos_embed, transformer.py:637-638), element pairs added in fp32 and rounded once, as moy_gemm forms A + A2 */
  const void* qpos; int64_t ld_qpos;
  void* out_xp; int64_t ld_xp;
  /* round 5: 1 = Wp, W1, W2, B0, B1 (and Wqkv) are given in MFMA-FRAGMENT ORDER (above) instead of row-major; results are identical */
  int32_t w_packed;
  /* round 5, optional (qkv = NULL: off): the in_proj of the NEXT layer's self-attention on the rows while they are on chip --
   * qkv[:, 0:512] = (out + qpos) Wqkv[0:512]^T + bqkv[0:512] (q | k, transformer.py:637-639), qkv[:, 512:768] = out Wqkv[512:768]^T +
   * bqkv[512:768] (v, :640); Wqkv T [768, 256] (in_proj_weight), bqkv fp32 [768], qkv T [M, >= 768]; needs qpos; the same bits as
   * moy_gemm over out / out_xp */
  const void* Wqkv; const float* bqkv; void* qkv; int64_t ld_qkv;
} moy_decoder_tail_args;

int moy_decoder_tail(const moy_decoder_tail_args* args, void* stream);

/* The row-wise MIDDLE of a decoder layer in one launch (round 3; csrc/dec_mid.hip), 16-bit types:
 *   e1    = LayerNorm1(x + attn . Wo^T + bo)            MOTRDecoderLayer.forward, nn/modules/transformer.py:640-641
 *                                                        (self_attn.out_proj, dropout = identity, norm1)
 *   offaw = (e1 + qpos) . Woa^T + boa   (fp32)           MSDeformAttn.forward, transformer.py:262-266: sampling_offsets and
 *                                                        attention_weights of with_pos_embed(embed, query_pos), :644
 * Woa = [sampling_offsets.weight ; attention_weights.weight] as [rows, 256] T with rows = max(256, n_oa) (zero rows past n_oa),
 * boa fp32 [n_oa]; n_oa = 8 heads * levels * 4 points * 3 (a multiple of 32, <= 512).  e1 is rounded to T exactly where the
 * separate launches store it; e1 + qpos is formed as moy_gemm forms its A2 operand.  MOY_ENOSYS for fp32 (moy_gemm x 2).
 * attn, x, qpos, e1: 16-byte aligned, row pitch a multiple of 8 elements. */
typedef struct moy_decoder_mid_args {
  const void* attn; int64_t ld_attn;   /* T [M, 256]: self-attention output before out_proj */
  const void* x;    int64_t ld_x;      /* T [M, 256]: the layer input (residual) */
  const void* qpos; int64_t ld_qpos;   /* T [M, 256]: query position embedding */
  int32_t M;
  const void* Wo; const float* bo; const float* ln_g; const float* ln_b;
  const void* Woa; const float* boa; int32_t n_oa;
  void* e1; int64_t ld_e1;             /* out: T [M, 256] */
  float* offaw; int64_t ld_oa;         /* out: fp32 [M, n_oa] */
  int32_t dtype;
  int32_t w_packed;                    /* round 5: 1 = Wo and Woa in MFMA-fragment order (above moy_decoder_tail_args) */
} moy_decoder_mid_args;

int moy_decoder_mid(const moy_decoder_mid_args* args, void* stream);

/* Query selection: per frame b, indices of the nq largest max_c scores[b, s, c], sorted
 * descending (torch.topk(enc_outputs_scores.max(-1).values, nq), head.py:1048).  Ties: lower
 * token index first.  scores fp32 [B, S, nc].  valid (optional) uint8 [S]: n_masked[b] receives
 * the number of selected tokens with valid == 0 (the +inf-anchor / NaN hazard of SURVEY 0.6).
 * idx_local int32 [B, nq] in [0, S); idx_global int32 [B, nq] = b*S + idx_local. */
int moy_topk(const float* scores, int B, int S, int nc, int nq, const uint8_t* valid, int32_t* idx_local,
             int32_t* idx_global, int32_t* n_masked, void* stream);

/* pos2posemb (nn/modules/transformer.py:183-190): boxes fp32 [M, 4] (logits) -> T [M, 256]. */
int moy_pos2posemb(const float* pos, int M, void* out, int64_t ldo, int dtype, void* stream);

/* Multi-head self-attention core of nn.MultiheadAttention(256, 8) as used by
 * MOTRDecoderLayer.forward (transformer.py:637-640) and QIM (MOTR/models/qim.py:275):
 * qkv T [B, L, 3*E] (q | k | v projections incl. bias), out T [B, L, E];
 * softmax_j(q_i . k_j / sqrt(E/nh)) v_j per head.  E/nh must be 32. */
int moy_mha_core(const void* qkv, int64_t ld_qkv, int B, int L, int nh, int E, void* out, int64_t ldo, int dtype,
                 void* stream);

/* moy_mha_core with the key mask of the temporal mode: key j of sequence b takes part iff j < n_prefix[b]
 * (live track slots) or j >= split (this frame's detect queries); n_prefix int32 [B] in device memory.  A query
 * with no live key gets an all-zero output. */
int moy_mha_core_masked(const void* qkv, int64_t ld_qkv, int B, int L, int nh, int E, const int32_t* n_prefix, int split,
                        void* out, int64_t ldo, int dtype, void* stream);

/* Fused deformable-attention sampling for the decoder (MSDeformAttn.forward transformer.py:267-285
 * after the three projections): softmax over the L*P attention logits, sampling locations
 * loc = ref_xy + off / P * ref_wh * 0.5, bilinear gather (zeros padding, align_corners=False),
 * weighted sum.  M = 8 heads x D = 32, L <= 4 levels, P = 4 points.
 *   value T: channel d of head m of token (b, s) at value[(b*S + s)*ldv + m*head_stride + d] -- [B, S, 256] with
 *   head_stride 32 (ldv >= 256), or head planes [8][B*S][32] with ldv 32 and head_stride B*S*32;
 *   offaw fp32 [B*Lq, ld_oa]: columns
 *   [0, M*L*P*2) = offsets ([m][l][p][xy]) then [.., + M*L*P) = attention logits ([m][l*p]);
 *   ref fp32 [B*Lq, 4] (cx, cy, w, h in [0,1]); shapes int32 [L][2] = (H, W) on the HOST.
 *   out T [B*Lq, ldo]. */
int moy_msda_fused(const void* value, int64_t ldv, int64_t head_stride, int B, int S, const int32_t* shapes_hw, int L,
                   const float* offaw, int64_t ld_oa, const float* ref, int Lq, void* out, int64_t ldo, int dtype, void* stream);

/* Round 5: the same sampling with the FIRST pyramid level gathered RAW and projected afterwards (csrc/msda_raw.hip).  value_proj
 * (transformer.py:255-257), input_proj + BN (head.py:838-839) and the bilinear sum (nn/modules/utils.py:41-78) are all linear, so for
 * level 0:  sum_p a_p . bilinear(W x + c)(loc_p) = W_h . (sum_p a_p . bilinear(x)(loc_p)) + c_h . sum_p a_p . (in-range corner weights)
 * -- the level's projected value planes (54 % of which no sample of a layer touches at 1088x608) are never formed.
 *   x0 T: level 0 as the backbone left it, [B, H0, W0, 128] channels-last with pixel pitch ld0 (>= 128) elements;
 *   wc T [256][128] / bc fp32 [256]: value_proj o BN o input_proj of THIS layer composed by the caller (W = Wv diag(s) Wp, c = Wv t + bv);
 *   planes T: head planes of levels 1 .. L-1, [8][B * S1][32] with head_stride elements between heads, S1 = tokens per frame of
 *   those levels (level-major; head_stride >= B * S1 * 32 or MOY_EINVAL); offaw / ref / out / shapes_hw as moy_msda_fused
 *   (shapes_hw[0] = (H0, W0)).  H0, W0 >= 2; a frame of level 0 below 2 GiB.
 *   Round 6: MOY_F32 / MOY_F32X3 -- fp32 x0 / wc (row-major, wc_packed = 0) / planes / out, every sum in fp32, the projection on the exact
 *   fp32 matrix instruction (the fp32 engines' folded head; the split-fp16 engine takes this exact form too).
 *   Numerics vs moy_msda_fused over projected planes: the level-0 contribution is rounded to T once (the gathered vector, before
 *   its product) instead of once per projected value; sums in fp32. */
typedef struct moy_msda_raw_args {
  const void* x0;
  int64_t ld0;
  const void* wc;
  const float* bc;
  const void* planes;
  int64_t head_stride;
  int32_t S1;
  int32_t B, Lq, L;
  const int32_t* shapes_hw;   /* HOST int32 [L][2] = (H, W) per level */
  const float* offaw;
  int64_t ld_oa;
  const float* ref;
  void* out;
  int64_t ldo;
  int32_t dtype;
  int32_t wc_packed;          /* round 5: 1 = wc in MFMA-fragment order (above moy_decoder_tail_args; N = 256 rows, K = 128) */
  const int32_t* perm;        /* round 6, optional (NULL: query order): device int32 [B, Lq], perm[b][i] = the query of frame b that is
                               * processed i-th (moy_query_order: spatial neighbours share a block, hence L1 / L2 lines).  MUST be a
                               * permutation of 0 .. Lq-1 per frame: a row named twice is written twice, a row not named is not written. */
} moy_msda_raw_args;
int moy_msda_raw0(const moy_msda_raw_args* a, void* stream);

/* ---- Temporal mode: carried track queries in a fixed-size query memory per sequence.
 * The shipped snapshot resets its state every frame and its carried branch crashes, so these entry points
 * follow the branch's visible intent (nn/modules/head.py:206-221, 1055-1064: decoder rows = [track queries | top-k detect
 * queries]) and upstream MOTR (MOTR/models/motr.py:303-325, 545-577; QIM MOTR/models/qim.py:251-301); spec: DESIGN.md §7.
 * Memory of sequence b: trk_embed / trk_qpos T [B, n_max, 256], trk_ref fp32 [B, n_max, 4] (logits), trk_id int64
 * [B, n_max], trk_dis int32 [B, n_max], n_trk int32 [B] (live slots are the first n_trk[b]), max_obj_id int64 [B]. */

/* Decoder input of a frame: rows [b*(n_max+nq) + i] = track slot i (zeros if dead) for i < n_max, detect query i - n_max
 * otherwise.  det_* are the dense [B*nq] buffers of the per-frame path; ref_sig = sigmoid(ref_logit). */
int moy_temporal_assemble(const void* trk_embed, const void* trk_qpos, const float* trk_ref, const int32_t* n_trk,
                          const void* det_embed, int64_t ld_de, const void* det_qpos, int64_t ld_dq, const float* det_ref,
                          int B, int n_max, int nq, void* embed, int64_t ld_e, void* qpos, int64_t ld_q, float* ref_logit,
                          float* ref_sig, int dtype, void* stream);

/* ID lifecycle of one frame (RuntimeTrackerBase.update loop, nn/modules/head.py:1232-1243, on carried state; thresholds
 * head.py:1146: birth 0.4, miss below 0.5, dropped after 5 misses, counters never reset) + compaction of the surviving and
 * newborn rows into memory slots (query order) + predictor rows (predict.py:43-94).
 *   logits fp32 [B, Lq, nc], boxes fp32 [B, Lq, 4] with Lq = n_max + nq <= 1024; max_obj_id is read and advanced by the births.
 *   Out: y [B, Lq, 4+nc], scores [B, Lq], obj_idxes int64 [B, Lq] (-1: no id / dead slot), dis_out int32 [B, Lq],
 *   sel_rows int32 [B, n_max] (global decoder row feeding slot s; dead slots point at a finite dummy row), n_new [B],
 *   n_overflow [B] (live rows that did not fit), rows [B, Lq, 6], track_id int64 [B, Lq], n_rows / n_ids [B] as moy_assign_post. */
int moy_temporal_assign(const float* logits, const float* boxes, int B, int n_max, int nq, int nc, const int64_t* trk_id,
                        const int32_t* trk_dis, const int32_t* n_trk, int64_t* max_obj_id, float score_thresh,
                        float filter_thresh, int miss_tol, float conf, float img_w, float img_h, float* y, float* scores,
                        int64_t* obj_idxes, int32_t* dis_out, int32_t* sel_rows, int32_t* n_new, int32_t* n_overflow,
                        float* rows, int64_t* track_id, int32_t* n_rows, int32_t* n_ids, void* stream);

/* Commit ids, miss counters and reference boxes (inverse_sigmoid(pred_boxes), qim.py:299) of the selected rows; n_trk = n_new. */
int moy_temporal_commit(const int32_t* sel_rows, const int32_t* n_new, const int64_t* obj_idxes, const int32_t* dis_out,
                        const float* boxes, int B, int n_max, int64_t* trk_id, int32_t* trk_dis, float* trk_ref,
                        int32_t* n_trk, void* stream);

/* The reference's own native operator, MultiScaleDeformableAttention.ms_deform_attn_forward
 * (MOTR/models/ops/src/vision.cpp:13-16, src/ms_deform_attn.h:21-40, CUDA kernel
 * src/cuda/ms_deform_im2col_cuda.cuh:237-299), same argument meaning:
 *   value [N, S, M, D], spatial_shapes int64 [L, 2] (H, W) and level_start_index int64 [L] in
 *   DEVICE memory, sampling_loc [N, Lq, M, L, P, 2], attn_weight [N, Lq, M, L, P] -> out [N, Lq, M*D].
 * The reference dispatches fp32/fp64 (ms_deform_attn_cuda.cu:64); bf16 and fp16 (config C5, the reference's own `half`
 * switch, engine/predictor.py:131) are added here: 16-bit operands, fp32 accumulation, one rounding of the result. */
int moy_msda_fwd_f32(const float* value, const int64_t* spatial_shapes, const int64_t* level_start_index,
                     const float* sampling_loc, const float* attn_weight, int N, int S, int M, int D, int L, int Lq,
                     int P, float* out, void* stream);
int moy_msda_fwd_bf16(const void* value, const int64_t* spatial_shapes, const int64_t* level_start_index,
                      const void* sampling_loc, const void* attn_weight, int N, int S, int M, int D, int L, int Lq,
                      int P, void* out, void* stream);
int moy_msda_fwd_f16(const void* value, const int64_t* spatial_shapes, const int64_t* level_start_index,
                     const void* sampling_loc, const void* attn_weight, int N, int S, int M, int D, int L, int Lq,
                     int P, void* out, void* stream);
int moy_msda_fwd_f64(const double* value, const int64_t* spatial_shapes, const int64_t* level_start_index,
                     const double* sampling_loc, const double* attn_weight, int N, int S, int M, int D, int L, int Lq,
                     int P, double* out, void* stream);

/* MultiScaleDeformableAttention.ms_deform_attn_backward (MOTR/models/ops/src/vision.cpp:13-16,
 * src/ms_deform_attn.h:42-62, host src/cuda/ms_deform_attn_cuda.cu:81-153, kernels
 * src/cuda/ms_deform_im2col_cuda.cuh:301-400 + the col2im kernels :403-1326).  Inputs as the forward
 * plus grad_output [N, Lq, M*D]; outputs grad_value [N, S, M, D] (CLEARED by the callee on `stream`,
 * then accumulated with hardware atomics: summation order, hence the last ulp, is not deterministic --
 * as in the reference), grad_sampling_loc [N, Lq, M, L, P, 2] and grad_attn_weight [N, Lq, M, L, P]
 * (every element written).  A sample contributes only if -1 < h_im < H and -1 < w_im < W (cuh:339).
 * fp32 / fp64 as the reference (its own test drives fp64 through gradcheck, ops/test.py:66-86);
 * `im2col_step` has no counterpart here: the op is not chunked. */
int moy_msda_bwd_f32(const float* value, const int64_t* spatial_shapes, const int64_t* level_start_index,
                     const float* sampling_loc, const float* attn_weight, const float* grad_output, int N, int S,
                     int M, int D, int L, int Lq, int P, float* grad_value, float* grad_sampling_loc,
                     float* grad_attn_weight, void* stream);
int moy_msda_bwd_f64(const double* value, const int64_t* spatial_shapes, const int64_t* level_start_index,
                     const double* sampling_loc, const double* attn_weight, const double* grad_output, int N, int S,
                     int M, int D, int L, int Lq, int P, double* grad_value, double* grad_sampling_loc,
                     double* grad_attn_weight, void* stream);

/* Stretch resize of uint8 BGR HWC frames to the network resolution: what TrackPredictor.pre_transform does before
 * the path (ultralytics/models/MOTRtrack/predict.py:96-105 -> LetterBox scaleFill branch, data/augment.py:573-576:
 * `cv2.resize(img, (Wd, Hd), interpolation=cv2.INTER_LINEAR)`, no padding).  Bit-exact restatement of OpenCV's 8-bit
 * INTER_LINEAR (11-bit fixed-point taps; exact 2x shrink -> INTER_AREA).  src [B, Hs, Ws, 3] with explicit row / image
 * pitches in bytes, dst [B, Hd, Wd, 3] dense, Wd % 4 == 0 (network widths are multiples of 32).  The channel order is
 * untouched: BGR->RGB, CHW and /255 stay fused in moy_stem_conv*. */
int moy_resize_linear_u8(const uint8_t* src, int B, int Hs, int Ws, int64_t src_row_bytes, int64_t src_img_bytes,
                         uint8_t* dst, int Hd, int Wd, void* stream);

/* Pairwise IoU similarity of pixel x0y0x1y1 boxes for the HOTA evaluator: TrackValidator._calculate_box_ious
 * (ultralytics/models/MOTRtrack/val.py:517-553, box_format 'x0y0x1y1') for T frames at once.
 *   a fp32 [T, n, 4] (ground truth), b fp32 [T, K, 4] (tracker), padded; na / nb int32 [T] = valid rows per frame
 *   (NULL: all n / K); out fp32 [T, n, K], zero beyond a frame's counts.  Degenerate boxes (area <= eps) give 0. */
int moy_box_iou(const float* a, const float* b, int T, int n, int K, const int32_t* na, const int32_t* nb, float* out,
                void* stream);

/* Per-frame ID assignment + predictor rows.  Replaces the host state machine
 * MOTRTrack._post_process_single_image / RuntimeTrackerBase.update (head.py:300-323,1232-1237) in
 * its shipped per-frame-reset semantics and TrackPredictor.postprocess
 * (ultralytics/models/MOTRtrack/predict.py:43-76) + ops.xywh2xyxy (utils/ops.py:378-393).
 *   logits fp32 [B, nq, nc], boxes fp32 [B, nq, 4] (cx, cy, w, h normalised)
 *   y fp32 [B, nq, 4+nc]   = cat(boxes, sigmoid(logits))              (head.py:235)
 *   scores fp32 [B, nq]     = max_c sigmoid(logits)                    (head.py:310)
 *   obj_idxes int64 [B, nq] = running counter over rows with score >= score_thresh, else -1
 *   rows fp32 [B, nq, 6], track_id int64 [B, nq], n_rows/n_ids int32 [B]:
 *     active rows (id >= 0) in query order; rows keeps those with score > conf as
 *     (x1, y1, x2, y2, score, cls) scaled by (img_w, img_h) (pass 1,1 for the tensor-source
 *     branch); track_id keeps ALL active ids (not conf-filtered, predict.py:61-76).
 *     If no row is active the detection fallback (predict.py:79-94) is produced and n_ids = -1. */
int moy_assign_post(const float* logits, const float* boxes, int B, int nq, int nc, float score_thresh, float conf,
                    float img_w, float img_h, float* y, float* scores, int64_t* obj_idxes, float* rows,
                    int64_t* track_id, int32_t* n_rows, int32_t* n_ids, void* stream);

/* Output-invisible per-sequence side state of the shipped path, kept ON DEVICE:
 *  (1) the *copy* half of RuntimeTrackerBase.update (nn/modules/head.py:1245-1283): active rows ->
 *      greedy O(K^2) suppression of rows whose (cx,cy,w,h)-as-(x,y,w,h) IoU with an earlier kept row
 *      exceeds 0.8 (_filter_tracks :1155-1171, _calculate_iou :1173-1196 incl. its early-outs) ->
 *      renumbering of ids above max_obj_id_pre (= 0 after the per-frame reset);
 *  (2) QueryInteractionModule.forward -> FSQM.online_update (MOTR/models/qim.py:303-340,
 *      MOTR/models/fsqm.py:117-180): update_confidence (indexed BY ID), inject_new_queries
 *      (score > 0.7, first free slot, FIFO id pool), remove_inactive_queries (conf < 0.3 for 3
 *      consecutive updates; frees never-used slots too and recycles their id -1, as shipped).
 * Frames b = 0..B-1 are consecutive frames of ONE sequence and are applied in order.
 *   scores fp32 [B,nq], boxes fp32 [B,nq,4], obj_idxes int64 [B,nq] (from moy_assign_post),
 *   hs T [B*nq, ld_hs] (last decoder layer output, 256 wide)
 *   copy_rows int32 [B,nq], copy_ids int64 [B,nq], n_copy int32 [B]: the filtered/renumbered copy
 * FSQM state (caller-owned, zero/-1 initialised by moy_fsqm_reset), n_slots = 300:
 *   mem fp32 [300,256], conf fp32 [300], ids int64 [300], fboxes fp32 [300,4], low int32 [300],
 *   pool int32 [pool_cap] ring + pool_hc int32 [3] = {head, count, overflow flag}. */
int moy_track_state_update(const float* scores, const float* boxes, const int64_t* obj_idxes, const void* hs, int64_t ld_hs,
                           int B, int nq, int32_t* copy_rows, int64_t* copy_ids, int32_t* n_copy, float* mem, float* conf,
                           int64_t* ids, float* fboxes, int32_t* low, int32_t* pool, int32_t pool_cap, int32_t* pool_hc,
                           int dtype, void* stream);
/* FSQM.reset (fsqm.py:182-190): zero memory, ids = -1, pool = 0..299. */
int moy_fsqm_reset(float* mem, float* conf, int64_t* ids, float* fboxes, int32_t* low, int32_t* pool, int32_t pool_cap,
                   int32_t* pool_hc, void* stream);

/* Config C1 (YOLOv8n detect).
 * moy_detect_decode: Detect.forward decode for ONE pyramid level (nn/modules/head.py:60-77): DFL
 * (softmax over 16 bins . arange, nn/modules/block.py:31-35) -> dist2bbox xywh (utils/tal.py:261-270,
 * anchors at cell + 0.5) * stride, sigmoid(cls).  box T [B*h*w, >=64] (ld_box), cls T [B*h*w, >=nc];
 * y fp32 [B, 4+nc, A] channel-major as the reference returns it; this level fills anchors
 * [a_off, a_off + h*w). */
int moy_detect_decode(const void* box, int64_t ld_box, const void* cls, int64_t ld_cls, int B, int h, int w, int nc,
                      float stride, int a_off, int A, float* y, int dtype, void* stream);
/* moy_nms: ops.non_max_suppression (utils/ops.py:148-283; single-label, class-aware via the
 * max_wh class offset) + torchvision.ops.nms semantics (greedy by descending score, IoU > iou_thres
 * suppresses; ties: lower anchor index first) + ops.scale_boxes / clip_boxes (utils/ops.py:99-129,
 * 285-298) as DetectionPredictor.postprocess applies them (models/yolo/detect/predict.py:12-30).
 *   y fp32 [B, 4+nc, A] -> rows fp32 [B, max_det, 6] = (x1, y1, x2, y2, conf, cls), n_rows int32 [B].
 *   boxes are mapped back with (x - pad_x) / gain, (y - pad_y) / gain and clipped to [0, clip_w] x [0, clip_h]
 *   (pass gain 1, pads 0, clip <= 0 to skip).  A <= 16384. */
int moy_nms(const float* y, int B, int nc, int A, float conf_thres, float iou_thres, int max_det, float max_wh, float gain,
            float pad_x, float pad_y, float clip_w, float clip_h, float* rows, int32_t* n_rows, void* stream);

/* Round 4: input_proj (head.py:838-839, 1012-1029: Conv1x1 + BN, no activation) folded into its consumers -- the value projection
 * and the enc_output score pass take each pyramid level's own tensor with composed weights, and the projected features
 * ("feats") are only formed for the nq SELECTED tokens of a frame (features[batch_ind, topk_ind], head.py:1096):
 * moy_level_rows: token (index into the level-major [S] raster of a frame) -> level[m] and, per level j, rows[j][m] = the token's
 *   row in level j's [B, hw_j, C] tensor if the token lies in level j, else 0 (a dummy row for that level's gathered product).
 * moy_level_select: dst T [M, 256] = G[level[m]][m, :] + shift[level[m]][:] rounded once (G fp32 [n_levels][M, ldg]: the per-level
 *   products, shift fp32 [n_levels][256]: the BN shifts); rows whose token is masked (valid[tok_local[m]] == 0) are zero. */
int moy_level_rows(const int32_t* tok_local, int B, int nq, int n_levels, const int32_t* level_hw, int32_t* rows, int32_t* level,
                   void* stream);
int moy_level_select(const float* G, int64_t level_stride, int64_t ldg, const int32_t* level, const float* shift,
                     const int32_t* tok_local, const uint8_t* valid, int M, int N, void* dst, int64_t ldd, int dtype, void* stream);

/* Round 5: the dispatch of moy_gemm WITHOUT the launch (host only: no stream, nothing is enqueued, no device memory is touched).
 * Runs the argument validation and the kernel eligibility rules of a real call and returns what that call would return before it
 * launches -- MOY_OK, MOY_EINVAL, or MOY_ENOSYS where a form the arguments REQUIRE (row runs, a folded 1x1 consumer) has no kernel for
 * the shape / launch size; *kernel (may be NULL) receives the kernel family that would run (MOY_KERNEL_*), 0 on error.  A planner
 * (mo_yolo_amd/engine.py: the folded head, the conv + 1x1 consumer pair, the score runs) asks here instead of trial-launching
 * on uninitialised buffers.  The reference has no counterpart: cuDNN / cuBLAS pick their kernels behind torch. */
#define MOY_KERNEL_TILED 1        /* gemm_kernel: tiled implicit GEMM (every shape) */
#define MOY_KERNEL_WREG 2         /* gemm_wreg_kernel: weight-stationary 1x1 / value / score forms */
#define MOY_KERNEL_DMA 3          /* gemm_dma_kernel: 256 x 256 tiles, both operands by LDS-DMA */
#define MOY_KERNEL_CONV_WS 4      /* conv_ws_kernel: persistent weight-stationary 3x3, stride 1 */
#define MOY_KERNEL_CONV_S2 5      /* conv_s2_kernel: persistent weight-stationary 3x3, stride 2 (+ folded 1x1 consumer) */
#define MOY_KERNEL_CONV_DIRECT 6  /* conv_direct_kernel: tiled direct 3x3 */
int moy_gemm_query(const moy_gemm_args* a, int* kernel);

/* Round 4: compute-unit budget of the calling host THREAD's following launches.  The persistent kernels behind moy_gemm (the
 * weight-stationary 1x1 / value / score kernel, the weight-stationary 3x3 convolution) size their grids for `n_cus` compute units
 * instead of the device's; 0 = the whole device (the default).  Results never depend on it.  The engine uses it to run the
 * write-bound value projection of the P3 level on one half of the chip beside the matrix-rate-bound P4 / P5 branch of the neck on
 * the other (two branches of the hipGraph; engine.py `_plan_fork`, measured by tools/probes/cu_share.py).  The reference has no
 * counterpart: its launch geometry is cuDNN's / cuBLAS's own.  Returns the previous budget. */
int moy_set_cu_limit(int n_cus);

/* Elementwise helpers. */
/* dst T [M, N] (ldd) = src T [rows[m], :] (lds): row gather (features[batch_ind, topk_ind], head.py:1096). N % 8 == 0. */
int moy_gather_rows(const void* src, int64_t lds, const int32_t* rows, int M, int N, void* dst, int64_t ldd, int dtype,
                    void* stream);
/* dst T [M, N] (ldd) = src fp32 [M, N] (lds): cast, used at fp32->bf16 seams. */
int moy_cast_f32_to(const float* src, int64_t lds, int M, int N, void* dst, int64_t ldd, int dtype, void* stream);
/* out fp32 [M, 4] = sigmoid(in fp32 [M, 4])  (refer_bbox.sigmoid(), transformer.py:694; enc_bboxes head.py:1080) */
int moy_sigmoid_f32(const float* in, int n, float* out, void* stream);

/* Round 6: processing order of a frame's decoder queries for moy_msda_raw0 (`perm`).  ref fp32 [B * Lq, 4] = the queries' reference boxes
 * (cx, cy, w, h in [0, 1], transformer.py:676-728 refines them little from layer to layer); perm int32 [B, Lq]: perm[b][i] = the query
 * of frame b to process i-th = the frame's queries sorted by the Morton code of the (H0 x W0)-grid cell of their centre (ties: query
 * index).  Lq <= 1024.  A permutation of every frame by construction; it changes the order in which the gather walks the queries and
 * nothing else (outputs bit for bit the same rows). */
int moy_query_order(const float* ref, int B, int Lq, int H0, int W0, int32_t* perm, void* stream);

/* Upstream MSDeformAttn.forward between its linears and the native op (MOTR/models/ops/modules/ms_deform_attn.py:98-116; used by
 * MOTRDeformableTransformerEncoderLayer, MOTR/models/deformable_transformer_plus.py:347-386):
 *   offaw fp32 [rows, ld]: columns [col_off, +M*L*P*2) = sampling_offsets(query), [col_aw, +M*L*P) = attention_weights(query)
 *   ref fp32 [rows, L, refdim], refdim 2 (points) or 4 (boxes); shapes_hw HOST int32 [L, 2] (H_l, W_l)
 *   aw[row, m, l, p]  = softmax over the L*P samples of head m (sigmoid_attn != 0: sigmoid)
 *   loc[row, m, l, p] = ref[l] + off / (W_l, H_l)                  (refdim 2)
 *                     = ref[l].xy + off / P * ref[l].wh * 0.5      (refdim 4)
 * loc / aw: T = fp32 or bf16, dense, in the layout moy_msda_fwd_* takes. */
int moy_msda_prep(const float* offaw, int64_t ld, int col_off, int col_aw, const float* ref, int refdim, int rows, int n_heads,
                  int n_levels, int n_points, const int32_t* shapes_hw, int sigmoid_attn, void* loc, void* aw, int dtype, void* stream);
/* x T [M, N] (ld): rows with mask[m] != 0 are zeroed (value.masked_fill_(input_padding_mask[..., None], 0), ms_deform_attn.py:95-96). */
int moy_mask_rows(void* x, int64_t ld, int M, int N, const uint8_t* mask, int dtype, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MOYOLO_H */

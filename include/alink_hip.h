/*
 * alink_hip.h — C-ABI of libalink_hip.so, the MI355X (gfx950) implementation of A-LINK's
 * face-recognition hot path.
 *
 * The reference (iamgroot42/A-LINK) has NO native boundary: its hot path is Python duck typing on
 * top of MXNet / Keras.  Every entry point below therefore cites the reference *call site* it
 * replaces (paths relative to the reference checkout).  A maintainer binds these with ctypes (see
 * INTEGRATION.md); this repo's own host side (a-link_amd/_abi.py) does exactly that.
 *
 * Conventions
 *   - every function returns 0 on success or a negative ALINK_E* code; nothing throws across the ABI;
 *     alink_last_error() returns a thread-local human-readable message for the last failure.
 *   - all `dev_*` pointers are device (HBM) pointers owned by the caller (PyTorch-ROCm in this repo);
 *     `stream` is a hipStream_t passed as void* (NULL = default stream).  No entry point allocates,
 *     frees or synchronises inside the launch path (graph-capture safe), except *_create / *_finalize
 *     / *_destroy / *_profile which are documented as synchronous.
 *   - handles are opaque, one per (process, GPU); a handle is not thread-safe (the reference is
 *     single-threaded: code/ALINK_arc.py:22-25).
 *   - pixels are what the reference feeds its models: float32, RGB, 0..255
 *     (code/readDFW.py:82,124); 512-d embeddings and all head tensors are float32.
 */
#ifndef ALINK_HIP_H
#define ALINK_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ALINK_OK          0
#define ALINK_EINVAL     -1   /* bad argument / shape the kernels do not support            */
#define ALINK_ENOMEM     -2   /* hipMalloc failed or caller workspace too small              */
#define ALINK_EHIP       -3   /* a HIP runtime call failed (message has hipGetErrorString)   */
#define ALINK_ESTATE     -4   /* call order violated (e.g. embed before finalize)            */
#define ALINK_ENOTFOUND  -5   /* unknown tensor name                                         */

/* activation / weight storage type of the backbone GEMMs (accumulation is always f32) */
#define ALINK_DT_BF16 0
#define ALINK_DT_F16  1
#define ALINK_DT_F32  2   /* pair head: alink_head_set_compute_dtype; IR backbone: the float32 precision mode */
#define ALINK_DT_F16X2 3  /* IR backbone only: split precision — every activation and folded weight is an f16 PAIR   */
                          /* hi + lo (22 significant bits, power-of-two scales per tensor), three products hi*hi,     */
                          /* hi*lo, lo*hi on the f16 matrix cores into f32 accumulators: the accuracy of the float32  */
                          /* mode (selection sets identical to the f32 arithmetic) at ~1/3 of the bf16 rate.          */
                          /* Needs alink_backbone_calibrate before the first alink_embed; embed / profile / the small-    */
                          /* batch latency mode, no gradient pass.                                                    */

/* input pixel layouts accepted by alink_embed */
#define ALINK_LAYOUT_NHWC_F32 0   /* what siamese.ArcFace.process receives (code/siamese.py:232-234) */
#define ALINK_LAYOUT_NCHW_F32 1   /* what FaceModel.get_feature receives  (code/face_model.py:83-88)  */
#define ALINK_LAYOUT_NHWC_U8  2   /* extension: raw 8-bit pixels, 4x less input traffic              */

const char* alink_last_error(void);
/* Prepares `device` (function attributes, hardware-contract probe); idempotent; the caller's current device
 * is left unchanged.  DEVICE RULE for the whole library: a handle lives on the device that was current
 * when it was created (one process per GPU: torch.cuda.set_device(LOCAL_RANK) first); every entry point
 * makes its handle's device current for the duration of the call and restores the caller's; entry points
 * without a handle run on the device that owns the buffers they are given.  `stream` arguments must be
 * streams of that same device. */
int  alink_init(int device);
int  alink_version(void);

/* ------------------------------------------------------------------------------------------------
 * Backbone: insightface LResNet-E-IR ("ArcFace") feature extractor.
 * Replaces face_model.get_model (code/face_model.py:28-41: mx.model.load_checkpoint + bind) and
 * FaceModel.get_feature (code/face_model.py:86-93: forward + sklearn normalize).
 * ---------------------------------------------------------------------------------------------- */
typedef struct alink_backbone alink_backbone_t;

typedef struct {
    int units[4];      /* residual units per stage: r100 = {3,13,30,3}, r50 = {3,4,14,3}   */
    int widths[5];     /* {64,64,128,256,512}: stem width then one per stage (multiples of 64) */
    int height, width; /* input size; "112,112" in code/siamese.py:222                       */
    int emb;           /* embedding size (512), multiple of 64                               */
    int dtype;         /* ALINK_DT_BF16 | ALINK_DT_F16 | ALINK_DT_F32 (the reference's own precision:  */
                       /* exact-f32 MFMA GEMMs, ~1/12 of the bf16 rate; embed only, no profile / gradient) */
    float bn_eps;      /* 2e-5 in the insightface symbol                                     */
} alink_ir_cfg;

alink_backbone_t* alink_backbone_create(const alink_ir_cfg* cfg);
void alink_backbone_destroy(alink_backbone_t* bb);

/* Hand over one raw checkpoint tensor (host float32, MXNet layout and MXNet names:
 * conv (O,I,kh,kw); FC (O, C*H*W); BN gamma/beta/moving_mean/moving_var; PReLU gamma).
 * This is the per-tensor body of arg_params/aux_params in code/face_model.py:34,40. */
int alink_backbone_load(alink_backbone_t* bb, const char* name, const float* host, size_t count);
/* number of tensors the configured network expects, and the i-th expected name / element count */
int alink_backbone_num_tensors(const alink_backbone_t* bb);
int alink_backbone_tensor_info(const alink_backbone_t* bb, int i, const char** name, size_t* count);
/* fold BN into weights/biases, convert, upload.  Synchronous; allocates device memory. */
int alink_backbone_finalize(alink_backbone_t* bb);

/* Optional: split one alink_embed call into up to `n` (1..8, default 1 = off) independent image
 * shards of >= 64 images on internal streams (ordered after / before `stream` with events).  Results
 * do not depend on n.  Callers that embed many batches get more from issuing whole 256-image calls
 * round-robin on their own streams (what a-link_amd/backbone.py does): the calls are independent. */
int alink_backbone_set_streams(alink_backbone_t* bb, int n);

size_t alink_backbone_workspace_bytes(const alink_backbone_t* bb, int n_images);

/* N images -> N L2-normalised embeddings (code/face_model.py:86-93, batched).
 * dev_in: pixels in `layout`; dev_out: N x emb float32. */
int alink_embed(alink_backbone_t* bb, const void* dev_in, int layout, int n_images,
                float* dev_out, void* dev_workspace, size_t workspace_bytes, void* stream);

/* Per-launch timing of one alink_embed with HIP events on `stream` (synchronous; for bench.py's
 * roofline).  Every kernel is launched 4 times back to back between its events and the mean is
 * reported, so the events' own cost is not booked as kernel time.  ms[i]/flops[i]/kind[i] describe launch i; kind: 0 stem, 1 implicit-GEMM conv,
 * 2 FC split-K GEMM, 3 FC finish.  *n_launches in: capacity, out: count. */
int alink_embed_profile(alink_backbone_t* bb, const void* dev_in, int layout, int n_images,
                        float* dev_out, void* dev_workspace, size_t workspace_bytes, void* stream,
                        float* ms, double* flops, int* kind, int* n_launches);

/* EXTENSION (not in the reference, which has only a black-box pixel attack: code/attack.py; named by
 * BASELINE.json's north_star and SURVEY.md §8f N1): d(loss)/d(pixels) through the frozen backbone for
 * FGSM / PGD.  alink_backbone_enable_grad (before finalize) also builds the transposed, flipped
 * folded weights (doubles weight memory) and refuses checkpoints with negative PReLU slopes.
 * alink_embed_cached = alink_embed that keeps what the backward pass needs in the (larger) workspace;
 * alink_embed_input_grad then maps dev_demb = d(loss)/d(embedding) (n x emb, w.r.t. the L2-normalised
 * output dev_emb of that forward) to dev_dpix = d(loss)/d(pixel) (n images, float32, NHWC or NCHW). */
int alink_backbone_enable_grad(alink_backbone_t* bb);

/* Latency mode for batches of <= 32 images (FaceModel.get_feature is a batch-1 call: code/face_model.py:86-93).
 * A 3x3 layer at batch 1 has a handful of workgroups that each walk all of K; with the switch on such layers
 * are split over K into f32 partial slabs added in a fixed order (IR-ResNet-100: batch 1 2.3 -> 1.6 ms, batch 16
 * 2.4 -> 2.2 ms).  Default OFF: a split sum rounds differently from the fused one, so an image's
 * embedding would no longer be bit-identical whatever batch it arrives in.  Changes the workspace size:
 * query alink_backbone_workspace_bytes after setting it. */
int alink_backbone_set_small_batch_split(alink_backbone_t* bb, int on);

/* ALINK_DT_F16X2 (split precision) only.  The mode stores every activation as an f16 pair scaled by a power of two
 * per tensor, chosen so that the largest value a tensor takes on the calibration images lands in [1024, 2048) —
 * 32x below the f16 overflow threshold, the lo halves of values down to 2^-13 of it normal f16.  A power-of-two scale is
 * exact for every value whose lo half stays normal; far smaller values (lo subnormal) round differently under another
 * scale, so two calibrations move an embedding by ~2e-7 — far below the mode's own error (2e-6), not zero: embeddings are
 * bit-reproducible for FIXED scales (alink_backbone_get_scales / set_scales carry them between processes, ranks and
 * checkpoints).  The calibration images only have to be "like" later inputs within that 32x.  Synchronous; runs the n_images (<= what the workspace was sized for) through
 * the network layer by layer.  merge != 0 keeps every exponent at or below its current value (re-calibration after
 * alink_backbone_range_flag reported a batch that left the range).  Must run once before the first alink_embed. */
int alink_backbone_calibrate(alink_backbone_t* bb, const void* dev_in, int layout, int n_images,
                             void* dev_workspace, size_t workspace_bytes, int merge, void* stream);
/* 16-bit storage modes (BF16 cannot, F16 and F16X2 can leave their range): 1 if, since the last reset, any embedding
 * written by alink_embed was non-finite.  The word lives in pinned host memory the last kernel of a forward writes;
 * the caller must have synchronised the streams it embedded on.  reset != 0 clears it. */
int alink_backbone_range_flag(alink_backbone_t* bb, int reset);
/* the device this handle lives on (the one current at alink_backbone_create: the device rule at the top of this header); -1 for NULL */
int alink_backbone_device(const alink_backbone_t* bb);
/* The calibration state of the split-precision mode, portable: one scale exponent per tensor (the stem's output, then every
 * convolution launch's output, in launch order; stored value = true value x 2^e).  An embedding is bit-reproducible for
 * FIXED scales only (a different scale moves the lo halves of values far below the tensor's maximum through different
 * f16 roundings: drift ~2e-7, far below the mode's own error, but not zero) — so a checkpoint saved for later
 * (reference code/siamese.py:114-125 is the save / load contract of the models) and the ranks of a one-process-per-GPU
 * job, whose top-k merge assumes replicated arithmetic, must SHARE them: get them on the rank / in the process that
 * calibrated, set them everywhere else.  num_scales is 0 for any other dtype.  set_scales marks the handle calibrated. */
/* Split precision only.  n = 3 (default): every multiplication as X_hi W_hi + X_hi W_lo + X_lo W_hi — the exact mode.
 * n = 1: X_hi W_hi alone — the SCREENING form on the same handle (same folded weights, same calibrated scales, same
 * workspace): one matrix-core product per multiplication like the plain f16 mode, but with nothing to overflow (the scales)
 * and the residual stream still carried and added as hi + lo, so only the rounding of each convolution's operands to 11 bits
 * remains.  For screen-then-settle selection (a-link_amd/settle.py): screen with n = 1, settle with n = 3.  Takes effect
 * for launches enqueued after the call.  Calibration always runs with n = 3. */
int alink_backbone_set_products(alink_backbone_t* bb, int n);
int alink_backbone_num_scales(const alink_backbone_t* bb);
int alink_backbone_get_scales(const alink_backbone_t* bb, int* exponents, int n);
int alink_backbone_set_scales(alink_backbone_t* bb, const int* exponents, int n);
size_t alink_backbone_grad_workspace_bytes(const alink_backbone_t* bb, int n_images);
int alink_embed_cached(alink_backbone_t* bb, const void* dev_in, int layout, int n_images, float* dev_out,
                       void* dev_workspace, size_t workspace_bytes, void* stream);
int alink_embed_input_grad(alink_backbone_t* bb, const float* dev_demb, const float* dev_emb, int layout,
                           int n_images, float* dev_dpix, void* dev_workspace, size_t workspace_bytes,
                           void* stream);

/* Diagnostic / unit-test entry: one fused NHWC convolution launch of the implicit-GEMM kernel
 *   out = [prelu_alpha]( conv(in, w) + bias[class] ) [+ resid]
 * dev_w is (Cout, ksz, ksz, Cin) in `dtype` in NATURAL cout order (the call permutes a private
 * copy — synchronous, test use only); dev_bias is (ncls, Cout) f32 with ncls = 9 if border_cls.
 * `fine`: where the linear-tile kernel applies (3x3 stride 1 on 7/14/28-wide maps), 1 / 0 force its
 * 64- / 128-channel workgroup form, < 0 chooses by grid size exactly as alink_embed does. */
int alink_conv_nhwc(int dtype, const void* dev_in, const void* dev_w, const float* dev_bias,
                    const float* dev_alpha, const void* dev_resid, void* dev_out,
                    int N, int H, int W, int Cin, int Cout, int ksz, int stride, int pad,
                    int border_cls, int fine, void* stream);

/* Split-precision twin (ALINK_DT_F16X2 kernels): every tensor float32 in its natural layout on the device — dev_w
 * (Cout, ksz, ksz, Cin), dev_resid / dev_out (M, Cout) — converted to and from f16 pairs on the host (synchronous,
 * test use only); e_*: the power-of-two scale exponents the stored tensors carry (stored = true x 2^e). */
int alink_conv_nhwc_x2(const float* dev_in, const float* dev_w, const float* dev_bias, const float* dev_alpha,
                       const float* dev_resid, float* dev_out, int N, int H, int W, int Cin, int Cout, int ksz,
                       int stride, int pad, int border_cls, int fine, int e_in, int e_w, int e_out, int e_res,
                       void* stream);

/* ------------------------------------------------------------------------------------------------
 * VGGFace2 ResNet-50 feature extractor: siamese.RESNET50 (code/siamese.py:203-216) =
 * keras_vggface VGGFace(model='resnet50', include_top=False) cut at 'avg_pool', flattened (2048-d),
 * fed through utils.preprocess_input(version=2).  Tensors are handed over with Keras names and
 * layouts: "<layer>/kernel" (kh, kw, in, out), "<layer>/bn/{gamma,beta,moving_mean,moving_variance}",
 * layers conv1/7x7_s2 and conv{2..5}_{u}_{1x1_reduce,3x3,1x1_increase,1x1_proj}.
 * alink_resnet50_embed: dev_in (n, H, W, 3) float32; preprocessed = 0: raw RGB 0..255 pixels (the
 * BGR flip and mean subtraction of preprocess_input happen in the stem kernel's loader);
 * preprocessed = 1: the caller already applied RESNET50.preprocess.  dev_out (n, 2048) float32.
 * ---------------------------------------------------------------------------------------------- */
typedef struct alink_resnet50 alink_resnet50_t;
alink_resnet50_t* alink_resnet50_create(int height, int width, int dtype, float bn_eps);
void alink_resnet50_destroy(alink_resnet50_t* r);
int alink_resnet50_num_tensors(const alink_resnet50_t* r);
int alink_resnet50_tensor_info(const alink_resnet50_t* r, int i, const char** name, size_t* count);
int alink_resnet50_load(alink_resnet50_t* r, const char* name, const float* host, size_t count);
int alink_resnet50_finalize(alink_resnet50_t* r);
size_t alink_resnet50_workspace_bytes(const alink_resnet50_t* r, int n_images);
int alink_resnet50_embed(alink_resnet50_t* r, const float* dev_in, int n_images, int preprocessed,
                         float* dev_out, void* dev_workspace, size_t workspace_bytes, void* stream);
/* dtype = ALINK_DT_F16X2 (split precision: features to float32 accuracy, so that selection through the drivers that use
 * this feature model — code/ALINK.py:67, code/ALINK_MTP.py:84, code/existing_al.py:58 — follows the reference's float32
 * arithmetic): as for the IR backbone, alink_resnet50_calibrate once before the first embed (and again with merge != 0 if
 * alink_resnet50_range_flag reports a batch that left the range: a non-finite feature was written since the last reset). */
int alink_resnet50_calibrate(alink_resnet50_t* r, const float* dev_in, int n_images, int preprocessed,
                             void* dev_workspace, size_t workspace_bytes, int merge, void* stream);
int alink_resnet50_range_flag(alink_resnet50_t* r, int reset);
/* the portable calibration state, as for the IR backbone (one exponent per op of the chain, in op order) */
int alink_resnet50_num_scales(const alink_resnet50_t* r);
int alink_resnet50_get_scales(const alink_resnet50_t* r, int* exponents, int n);
int alink_resnet50_set_scales(alink_resnet50_t* r, const int* exponents, int n);
/* per-op HIP-event timing of one forward (synchronous): ms[i] / flops[i] for op i, name via op_name */
int alink_resnet50_profile(alink_resnet50_t* r, const float* dev_in, int n_images, float* dev_out,
                           void* dev_workspace, size_t workspace_bytes, void* stream, float* ms,
                           double* flops, int* n_ops);
const char* alink_resnet50_op_name(const alink_resnet50_t* r, int i);

/* ------------------------------------------------------------------------------------------------
 * VGGFace VGG-16 feature extractor: siamese.FaceVGG16 (code/siamese.py:187-200) = keras_vggface
 * VGGFace(model='vgg16', include_top=False) cut at 'pool5', flattened ((H/32)(W/32)512 = 25088 at
 * 224 x 224), fed through utils.preprocess_input(version=1).  Tensors: "conv{b}_{l}/kernel"
 * (3, 3, in, out) and "conv{b}_{l}/bias".  dev_in (n, H, W, 3) float32, raw RGB (preprocessed = 0)
 * or already preprocessed (1); dev_out (n, feature_size) float32 in Keras' Flatten order (h, w, c).
 * ---------------------------------------------------------------------------------------------- */
typedef struct alink_vgg16 alink_vgg16_t;
alink_vgg16_t* alink_vgg16_create(int height, int width, int dtype);
void alink_vgg16_destroy(alink_vgg16_t* r);
int alink_vgg16_num_tensors(const alink_vgg16_t* r);
int alink_vgg16_tensor_info(const alink_vgg16_t* r, int i, const char** name, size_t* count);
int alink_vgg16_feature_size(const alink_vgg16_t* r);
int alink_vgg16_load(alink_vgg16_t* r, const char* name, const float* host, size_t count);
int alink_vgg16_finalize(alink_vgg16_t* r);
size_t alink_vgg16_workspace_bytes(const alink_vgg16_t* r, int n_images);
int alink_vgg16_embed(alink_vgg16_t* r, const float* dev_in, int n_images, int preprocessed,
                      float* dev_out, void* dev_workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Siamese pair head: |l - r| -> Dense(h1) ReLU -> Dense(h2) ReLU -> Dense(2) -> softmax.
 * Replaces SiameseNetwork.__init__/predict/finetune/customTrainModel's Keras calls
 * (code/siamese.py:19-35, 52-58, 81-112, 130-131) and committee.Bagging.predict
 * (code/committee.py:13-20).  All f32 (exact-f32 MFMA), Keras-2.1.2 semantics.
 * ---------------------------------------------------------------------------------------------- */
typedef struct alink_head alink_head_t;

alink_head_t* alink_head_create(int d_in, int h1, int h2, float lr, float rho, float eps);
/* out_dim = 2: Dense(2) + softmax (code/siamese.py:31-32); out_dim = 1: Dense(1, sigmoid), the pair
 * scorer of the baseline scripts (code/siamese3.py:25, used by code/existing_al.py) — probabilities,
 * labels and metrics are then (n, 1). */
alink_head_t* alink_head_create_ex(int d_in, int h1, int h2, int out_dim, float lr, float rho, float eps);
void alink_head_destroy(alink_head_t* h);
size_t alink_head_num_params(const alink_head_t* h);   /* W1,b1,W2,b2,W3,b3 flattened */
/* Parameters in Keras order and layout: kernel (in,out) row-major then bias, per layer. */
int alink_head_set_params(alink_head_t* h, const float* host_params, size_t count);
int alink_head_get_params(const alink_head_t* h, float* host_params, size_t count);
int alink_head_reset_optimizer(alink_head_t* h);
/* Compute type of the head's forward / backward GEMMs: ALINK_DT_F32 (default: exact-f32 MFMA, Keras' floatx) or
 * ALINK_DT_BF16 — BASELINE configs[4]'s "bf16 fine-tune", mixed precision: parameters, gradients and Adadelta state stay
 * f32 (get/set_params, params_dev, grads_dev unchanged: the all-reduce buffer is f32); the GEMM operands are bf16 —
 * a 2-byte copy of the weights (refreshed by the update kernels) and activations / activation gradients rounded to
 * bf16 where they are produced — with exact products accumulated in f32; biases are not quantised.  Applies to
 * predict (on the bf16 matrix cores for the 512 / 64 / 2 head: 3.6x the f32 pair-scoring rate), eval and the train step alike.  The reference trains in float32 (code/siamese.py:33-35). */
int alink_head_set_compute_dtype(alink_head_t* h, int dtype);
int alink_head_get_compute_dtype(const alink_head_t* h);
/* keras.callbacks.ReduceLROnPlateau hook (code/siamese.py:54): change Adadelta's lr */
int alink_head_set_lr(alink_head_t* h, float lr);
float alink_head_get_lr(const alink_head_t* h);
/* device pointers of the flat parameter / gradient buffers (for RCCL all-reduce / broadcast by the caller).
 * The forward keeps packed copies of W1 / W2: call alink_head_params_dev again AFTER writing through the
 * pointer (each call marks those copies stale).  The gradient buffer is num_params floats followed by 4
 * spare floats: a data-parallel caller points dev_metrics of alink_head_train_step at them so that ONE
 * all-reduce carries the gradients and {loss, accuracy}. */
float* alink_head_params_dev(alink_head_t* h);
float* alink_head_grads_dev(alink_head_t* h);

/* probs[p] = softmax(head(|L[li[p]] - R[ri[p]]|)) for p < P.  li/ri may be NULL (= identity):
 * that is SiameseNetwork.predict([L,R]) (code/siamese.py:130-131); with index lists the pairs are
 * gathered from embedding matrices instead of being materialised (utilities/generateMatrixDFW.py:25-36). */
int alink_head_forward(alink_head_t* h, const float* dev_L, const float* dev_R,
                       const int32_t* dev_li, const int32_t* dev_ri, int64_t P,
                       float* dev_probs, void* stream);
/* Bagging.predict: mean over n_heads members of alink_head_forward (code/committee.py:13-20). */
int alink_committee_forward(alink_head_t* const* heads, int n_heads, const float* dev_L,
                            const float* dev_R, const int32_t* dev_li, const int32_t* dev_ri,
                            int64_t P, float* dev_probs, void* dev_scratch, void* stream);

/* The same mean when every member scores the pairs on ITS OWN feature matrices (a committee whose members
 * have different feature extractors: BASELINE configs[2], three backbones): dev_L[m] / dev_R[m] are member m's
 * embedding matrices, the index lists are shared. */
int alink_committee_forward_multi(alink_head_t* const* heads, int n_heads, const float* const* dev_L,
                                  const float* const* dev_R, const int32_t* dev_li, const int32_t* dev_ri,
                                  int64_t P, float* dev_probs, void* stream);

/* The N x N verification score matrix of utilities/generateMatrixDFW.py:25-36:
 *   scores[r][j] = mean_m softmax(head_m(|E[row0 + r] - E[j]|))[col],  r < nrows, j < n
 * (the reference keeps out[0], i.e. col = 0, of one model: n_heads = 1).  Pairs are enumerated by the
 * kernel; nothing is materialised.  dev_scores is nrows x n float32. */
int alink_pair_scores_matrix(alink_head_t* const* heads, int n_heads, const float* dev_emb, int n,
                             int row0, int nrows, int col, float* dev_scores, void* stream);

/* One Keras train_on_batch: forward, binary_crossentropy (clip 1e-7, mean over the 2 outputs,
 * sample-weighted mean over the batch), backward, [grads left in alink_head_grads_dev],
 * then Adadelta.  dev_y is (n,2) one-hot, dev_sw (n) sample weights or NULL.
 * dev_metrics receives {loss, binary_accuracy}.  `apply` = 0 computes gradients only (the caller
 * all-reduces them, then calls alink_head_apply_update) — the data-parallel fine-tune step. */
int alink_head_train_step(alink_head_t* h, const float* dev_L, const float* dev_R,
                          const float* dev_y, const float* dev_sw, int n, float grad_scale,
                          int apply, float* dev_metrics, void* stream);
int alink_head_apply_update(alink_head_t* h, void* stream);
/* The same and, in the same launch, the same Adadelta rule (the head's lr, rho, eps) on a second parameter block of n2
 * floats: the tower of an end-to-end model (SmallRes' step ends with one update launch instead of two). */
int alink_head_apply_update_with(alink_head_t* h, float* dev_params2, const float* dev_grads2, float* dev_acc2,
                                 float* dev_dacc2, size_t n2, void* stream);
/* Optional: on a non-default stream, capture the train step into a hipGraph per distinct (operand
 * pointers, n, grad_scale, apply, lr) and replay it.  Default OFF: measured 67.8 us vs 66.2 us of plain
 * launches — the chain is bound by dependency latency between its kernels, not by launch cost. */
int alink_head_set_graph(alink_head_t* h, int on);
/* Gradient of the last alink_head_train_step's loss w.r.t. its inputs (for end-to-end models such as
 * SmallRes, code/siamese.py:158-168): dL, dR are (n, d_in) f32. */
int alink_head_input_grads(alink_head_t* h, const float* dev_L, const float* dev_R, int n,
                           float* dev_dL, float* dev_dR, void* stream);
/* The same for inputs that are the outputs of a ReLU (SmallRes' Dense(2048, relu), code/siamese.py:156):
 * gradients w.r.t. that layer's pre-activations, i.e. the above times (input > 0), in the same launch. */
int alink_head_input_grads_relu(alink_head_t* h, const float* dev_L, const float* dev_R, int n,
                                float* dev_dL, float* dev_dR, void* stream);
/* alink_head_train_step(apply = 0) and alink_head_input_grads (relu_inputs: ..._relu) as ONE call: what an end-to-end
 * model's step needs from its head (SmallRes.trainModel, code/siamese.py:158-180).  For SmallRes' own head shape
 * (128 / 32 / 2 outputs on a multiple of 256 features, at most 32 pairs, float32) the pair is three launches instead
 * of seven; other shapes run the two calls above.  Gradients are left in alink_head_grads_dev, the parameters
 * untouched (alink_head_apply_update applies them).  dev_colsum (optional, d_in floats): the column sums of the
 * 2n x d_in matrix [dL ; dR], rows ascending — the bias gradient of the layer that produced the inputs. */
int alink_head_train_step_input_grads(alink_head_t* h, const float* dev_L, const float* dev_R, const float* dev_y,
                                      const float* dev_sw, int n, float grad_scale, int relu_inputs,
                                      float* dev_dL, float* dev_dR, float* dev_colsum, float* dev_metrics,
                                      void* stream);
/* Keras test_on_batch: {loss, binary_accuracy} without touching parameters. */
int alink_head_eval(alink_head_t* h, const float* dev_L, const float* dev_R, const float* dev_y,
                    int n, float* dev_metrics, void* stream);
/* The inner loop of SiameseNetwork.customTrainModel (code/siamese.py:91-110) for `steps` consecutive steps, enqueued by ONE
 * call: per step train_on_batch on the rows that were not held out, then test_on_batch on the held-out ones — the batches
 * given as ROW INDICES into a device-resident feature table (dev_table, rows of d_in floats: left row li[r] against right row
 * ri[r]), so that no feature crosses the host boundary during an epoch (the reference stacks 2 x n x d_in floats per step).
 * host_desc: 4 int64 per step {offset into dev_idx, offset into dev_vals, n, n_held}; at dev_idx + offset: li[n] then ri[n]
 * (int32), the n_held held-out rows FIRST; at dev_vals + offset: y[n][out_dim] in the same row order, then (with_weights)
 * sample_weight[n - n_held] of the trained rows.  dev_metrics: 4 floats per step {train loss, train accuracy, held-out loss,
 * held-out accuracy}; the last two untouched when n_held = 0.  The same kernels, on the same rows in the same order, as
 * alink_head_train_step + alink_head_eval called once per step on gathered rows (bit-identical parameters and metrics). */
int alink_head_custom_train_steps(alink_head_t* h, const float* dev_table, const int32_t* dev_idx, const float* dev_vals,
                                  const int64_t* host_desc, int steps, int with_weights, float* dev_metrics, void* stream);

/* ------------------------------------------------------------------------------------------------
 * SmallRes: the end-to-end-trained low-resolution siamese CNN (code/siamese.py:134-184):
 *   tower  conv3x3(3->32,same) relu, conv3x3(32->32,valid) relu, maxpool2, dropout.25,
 *          conv3x3(32->64,same) relu, conv3x3(64->64,valid) relu, maxpool2, dropout.25, flatten,
 *          dense(feat) relu        (shared by both inputs)
 *   head   |l-r| -> dense128 relu -> dense32 relu -> dense2 -> softmax, BCE + Adadelta
 * f32 throughout.  Pixels are (n, H, W, 3) f32; `prescale` applies SmallRes.preprocess
 * ((x-128)/128, code/siamese.py:179-181) in the first kernel.
 * Parameters in Keras order: conv kernels (3,3,in,out) + bias x4, dense (in,out) + bias, then the head.
 * ---------------------------------------------------------------------------------------------- */
typedef struct alink_smallres alink_smallres_t;
alink_smallres_t* alink_smallres_create(int img_h, int img_w, int feat, float lr, float rho, float eps);
void alink_smallres_destroy(alink_smallres_t* m);
size_t alink_smallres_num_params(const alink_smallres_t* m);
int alink_smallres_set_params(alink_smallres_t* m, const float* host, size_t count);
int alink_smallres_get_params(const alink_smallres_t* m, float* host, size_t count);
int alink_smallres_set_lr(alink_smallres_t* m, float lr);
float* alink_smallres_grads_dev(alink_smallres_t* m);      /* flat tower+head gradients (all-reduce), followed by 4 spare floats:
                                                            * a data-parallel caller points dev_metrics at them (as for alink_head_grads_dev) */
/* predict: probs (n,2).  n <= 256 per call. */
int alink_smallres_forward(alink_smallres_t* m, const float* dev_L, const float* dev_R, int n, int prescale,
                           float* dev_probs, void* stream);
/* one Keras train_on_batch (apply != 0) or gradients only.  dev_masks: the two dropout keep-masks for
 * the 2n tower passes, u8, laid out [2n*P1*P1*32] then [2n*P2*P2*64] (1 = keep), or NULL = no dropout.
 * dev_metrics: {loss, binary_accuracy}. */
int alink_smallres_train_step(alink_smallres_t* m, const float* dev_L, const float* dev_R, const float* dev_y,
                              const float* dev_sw, int n, int prescale, const uint8_t* dev_masks,
                              float grad_scale, int apply, float* dev_metrics, void* stream);
/* The same with the step's Dropout keep-masks drawn BY the step: dev_masks (2n (a + b) bytes, a and b from
 * alink_smallres_mask_sizes) is first filled with alink_keep_masks(dev_masks, 2n (a + b), 0.75, mask_seed) — by
 * extra workgroups of the step's first launch instead of a launch of its own — then read as above. */
int alink_smallres_train_step_drawn(alink_smallres_t* m, const float* dev_L, const float* dev_R, const float* dev_y,
                                    const float* dev_sw, int n, int prescale, uint8_t* dev_masks, uint64_t mask_seed,
                                    float grad_scale, int apply, float* dev_metrics, void* stream);
/* Keras train_on_batch (code/siamese.py:172-184 hands NumPy arrays) as ONE synchronous call on HOST operands: host_L / host_R
 * (n, H, W, 3) f32, host_y (n, 2), host_sw (n) or NULL.  The operands go through pinned staging owned by the handle (one
 * upload), the step draws its Dropout masks from mask_seed (dropout = 0: no Dropout), applies its update, the stream is
 * synchronised and {loss, accuracy} are returned in host_metrics[2]. */
int alink_smallres_train_on_batch_host(alink_smallres_t* m, const float* host_L, const float* host_R, const float* host_y,
                                       const float* host_sw, int n, int prescale, int dropout, uint64_t mask_seed,
                                       float* host_metrics, void* stream);
/* Optional (SmallResNet turns it on): on a non-default stream, the whole train step — ~40 short launches on two streams — is
 * captured per distinct (operand pointers, n, flags, lr) the second time it is seen and replayed from then on.  The caller keeps its
 * operands at stable addresses (staging buffers) for that to hit.  Same kernels, same order: the same results. */
int alink_smallres_set_graph(alink_smallres_t* m, int on);
int alink_smallres_apply_update(alink_smallres_t* m, void* stream);
int alink_smallres_eval(alink_smallres_t* m, const float* dev_L, const float* dev_R, const float* dev_y, int n,
                        int prescale, float* dev_metrics, void* stream);
/* sizes of the two dropout masks per tower image (elements): P1*P1*32 and P2*P2*64 */
int alink_smallres_mask_sizes(const alink_smallres_t* m, int* per_image_1, int* per_image_2);

/* ------------------------------------------------------------------------------------------------
 * Pool scoring helpers (HBM-bound): uncertainty measures (code/uncertainty.py:15-60) and top-k
 * (modAL multi_argmax as used at code/uncertainty.py:155,183,213; disparity top-k at
 * code/ALINK_arc.py:181).
 * ---------------------------------------------------------------------------------------------- */
#define ALINK_SCORE_UNCERTAINTY 0   /* 1 - max_c p                                  */
#define ALINK_SCORE_MARGIN      1   /* p_(1) - p_(2)                                */
#define ALINK_SCORE_ENTROPY     2   /* -sum p ln p (p renormalised)                 */
#define ALINK_SCORE_DISPARITY   3   /* -|a[:,col] - b[:,col]| (needs dev_b)         */
int alink_score(int kind, const float* dev_probs, const float* dev_b, int col, int64_t P, int C,
                float* dev_scores, void* stream);
/* indices of the k smallest (largest=0) or largest scores, ties broken by lower index, returned
 * sorted by (score, index).  dev_scratch >= alink_topk_scratch_bytes(P, k). */
size_t alink_topk_scratch_bytes(int64_t P, int k);
int alink_topk(const float* dev_scores, int64_t P, int k, int largest, int32_t* dev_idx,
               float* dev_vals, void* dev_scratch, void* stream);

/* ------------------------------------------------------------------------------------------------
 * DFW protocol evaluation (utilities/ROC_precompute.py:19-63): over the strict upper triangle of an
 * n x n score matrix, pairs are genuine / impostor by the protocol mask (codes 1,2 genuine; 3,4
 * impostor) and roc_case (1: codes 1 vs 3; 2: 2 vs 4; 3: {1,2} vs {3,4}).  dev_hist receives
 * [2][n_thr+1] counts: hist[k][c] = number of class-k (0 genuine, 1 impostor) scores with exactly c
 * of the ASCENDING thresholds <= score.  True/false positives at the threshold of rank t are the
 * suffix sums over c > t (a-link_amd/evaluation.py); the call zeroes dev_hist itself.
 * ---------------------------------------------------------------------------------------------- */
int alink_roc_counts(const float* dev_scores, const uint8_t* dev_mask, int n,
                     const double* dev_thr_sorted, int n_thr, int roc_case,
                     unsigned long long* dev_hist, void* stream);

/* ------------------------------------------------------------------------------------------------
 * A2-LINK perturbation stage (code/noise.py, code/attack.py:5-29, code/committee.py:22-37).
 * Images are float32 (n, H, W, C) as everywhere upstream of the models.  Random draws come from a
 * counter-based Philox4x32-10 keyed by (seed, element index): results do not depend on the launch
 * shape, and `offset` (elements; any value) / `first_image` (images) let a caller process a row range of one
 * logical array — a chunk, or one rank's shard of the pair batch (a-link_amd/alink_loop.py with `group`) —
 * and get exactly what the whole array would have drawn there.
 * The reference draws from the unseeded np.random global stream, so only the distributions are
 * contractual.  In-place (dev_out == dev_in) is allowed for every call except alink_resize_bilinear.
 * ---------------------------------------------------------------------------------------------- */
/* noise.Gaussian.addIndividualNoise (code/noise.py:40-45): out = x + N(mean, sigma) */
int alink_noise_gaussian(const float* dev_in, float* dev_out, int64_t count, float mean, float sigma,
                         uint64_t seed, uint64_t offset, void* stream);
/* noise.Speckle.addIndividualNoise (code/noise.py:83-88): out = x + x * N(0,1) / divisor (15) */
int alink_noise_speckle(const float* dev_in, float* dev_out, int64_t count, float divisor,
                        uint64_t seed, uint64_t offset, void* stream);
/* EXTENSION (the random start of noise.PGD; no counterpart in code/noise.py): out = x + U(lo, hi), the element's
 * 24-bit uniform from the same (seed, element) keying */
int alink_noise_uniform(const float* dev_in, float* dev_out, int64_t count, float lo, float hi,
                        uint64_t seed, uint64_t offset, void* stream);
/* Dropout keep-masks for alink_smallres_train_step (Keras Dropout(0.25) after each pool of the SmallRes tower,
 * code/siamese.py:146,153; TensorFlow draws them with an op-level generator that cannot be reproduced — only the keep
 * probability is contractual): dev_out[e] = 1 with probability `keep`, from the same (seed, element) Philox keying. */
int alink_keep_masks(uint8_t* dev_out, int64_t count, float keep, uint64_t seed, void* stream);
/* ... elements first, first + 1, ... of that stream of masks: a rank that trains rows lo : hi of a pair batch draws exactly the
 * masks the whole-batch step draws for those rows (the data-parallel SmallRes step) */
int alink_keep_masks_at(uint8_t* dev_out, int64_t count, float keep, uint64_t seed, uint64_t first, void* stream);
/* noise.SaltPepper.addIndividualNoise (code/noise.py:54-65), tuple-index semantics: per image n_salt
 * elements (r,c,ch) <- 1 then n_pepper elements <- 0, r in [0,H-2], c in [0,W-2], ch in [0,C-2]. */
int alink_noise_saltpepper(const float* dev_in, float* dev_out, int n_images, int H, int W, int C,
                           int n_salt, int n_pepper, uint64_t seed, uint64_t first_image, void* stream);
/* noise.Poisson.addIndividualNoise (code/noise.py:72-76): per image vals = 2^ceil(log2(#unique)),
 * out = Poisson(x * vals) / vals.  dev_vals (optional, n_images) receives vals. */
size_t alink_noise_poisson_scratch_bytes(int n_images, int64_t per_image);
int alink_noise_poisson(const float* dev_in, float* dev_out, int n_images, int64_t per_image,
                        uint64_t seed, uint64_t first_image, void* dev_scratch, size_t scratch_bytes,
                        float* dev_vals, void* stream);
/* noise.Perlin (code/noise.py:95-150): three octaves ns3[0..2] of gradient noise on square
 * size x size images, the same noise added to every channel.  dev_vec holds the unit gradient
 * vectors [n_images][alink_perlin_nodes(size, ns3)][2], octave after octave, row-major grids of
 * (size/ns + 1)^2 nodes (code/noise.py:100-107); alink_perlin_vectors fills it with random ones. */
int alink_perlin_nodes(int size, const int* ns3);
int alink_perlin_vectors(int n_images, int nodes_total, uint64_t seed, uint64_t first_image, float* dev_vec,
                         void* stream);
int alink_noise_perlin(const float* dev_in, float* dev_out, int n_images, int size, int C,
                       const int* ns3, const float* dev_vec, void* stream);
/* committee.Bagging.resize (code/committee.py:22-26): cv2.resize(image, (Wo, Ho)), INTER_LINEAR */
/* EXTENSION (FGSM / PGD, not in the reference whose only attack is the few-pixel search of code/attack.py): one
 * signed-gradient step and the projection back into the eps-ball of the clean image and the pixel range, in place:
 *   adv <- clip(clip(adv + step * sign(grad), clean - eps, clean + eps), lo, hi)      (n float32 elements; 16-byte
 * aligned buffers take 16-byte lane accesses, any other alignment the scalar form; pass -INFINITY / INFINITY for no pixel
 * clip; step < 0 descends). */
int alink_pgd_step(float* dev_adv, const float* dev_clean, const float* dev_grad, int64_t n, float step, float eps,
                   float lo, float hi, void* stream);
int alink_resize_bilinear(const float* dev_in, float* dev_out, int n, int H, int W, int C, int Ho,
                          int Wo, void* stream);
/* attack.perturb_image (code/attack.py:5-29): n candidates x k pixels (x, y, r, g, b) as float64
 * (the DE population, truncated like astype(int)) written into n copies of dev_img (Hc, W, 3).
 * split = 0: dev_out is (n, Hc, W, 3); split = 1: dev_out is [2][n][Hc/2][W][3], the top and
 * bottom halves as two contiguous batches (noise.PredictionWrappedModel.predict, code/noise.py:158-168). */
int alink_perturb_images(const float* dev_img, const double* dev_xs, int n, int k, int Hc, int W,
                         int split, float* dev_out, void* stream);
/* The same for the candidates of SEVERAL searches in one launch (PixelAttacker.attack_all, code/attack.py:91-103, advances
 * K pairs' differential-evolution searches in lock-step): dev_imgs is a table [n_img][Hc][W][3] of stacked pair images,
 * candidates [g * group, (g + 1) * group) perturb image dev_img_of[g] (int32, device; NULL: image 0 for all). */
int alink_perturb_images_multi(const float* dev_imgs, const int* dev_img_of, int group, const double* dev_xs, int n,
                               int k, int Hc, int W, int split, float* dev_out, void* stream);

/* ------------------------------------------------------------------------------------------------
 * EXTENSIONS — named by BASELINE.json's north_star, ABSENT from the reference: the reference
 * cuts the ArcFace checkpoint at fc1_output and never builds the margin head (code/face_model.py:35-36,53),
 * and trains its pair scorer with binary cross-entropy on |l - r| (code/siamese.py:27-35), not with a
 * contrastive loss.  Provided for callers that fine-tune the backbone's embedding space; checked against torch
 * autograd (tests/test_gpu_extensions.py), not against the reference.
 * ---------------------------------------------------------------------------------------------- */
/* Additive angular margin softmax (ArcFace): loss = mean_i CE(softmax(s * cos(theta_ij + m [j == y_i])), y_i) over
 * L2-normalised embeddings (N x D) and class centres (C x D); where theta + m would pass pi the target logit is
 * cos(theta) - m sin(pi - m) (easy_margin: cos(theta) itself for cos(theta) <= 0).  dev_demb (N x D) / dev_dW (C x D)
 * receive the gradients w.r.t. the RAW (un-normalised) inputs, or are NULL. */
size_t alink_arcface_margin_workspace_bytes(int N, int D, int C);
int alink_arcface_margin_loss(const float* dev_emb, const float* dev_W, const int32_t* dev_labels, int N, int D,
                              int C, float s, float m, int easy_margin, float* dev_loss, float* dev_demb,
                              float* dev_dW, void* dev_workspace, size_t workspace_bytes, void* stream);
/* Pairwise-L2 contrastive loss: d_p = sqrt(max(|L_p - R_p|^2, 1e-7)),
 * loss = mean_p [ y_p d_p^2 + (1 - y_p) max(margin - d_p, 0)^2 ]  (y = 1: same identity).
 * dev_pair_loss (P) receives the per-pair terms; dev_dL / dev_dR (P x D) the gradients, or both NULL. */
int alink_contrastive_loss(const float* dev_L, const float* dev_R, const float* dev_y, int64_t P, int D,
                           float margin, float* dev_loss, float* dev_pair_loss, float* dev_dL, float* dev_dR,
                           void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ALINK_HIP_H */

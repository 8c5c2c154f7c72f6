/* libvp_hip.so - C ABI of the MI355X-native voicepuppet hot path.
 *
 * The reference (taylorlu/voicepuppet) has no FFI boundary for its neural path: it is TF1.x graph
 * construction behind Python classes.  Its one native precedent is utils/cython/mesh_core.h:53-77
 * (free functions, raw pointers + explicit int sizes, caller owns every buffer).  This header keeps
 * that convention for the path BASELINE.json names:
 *
 *   vp_pixrefer_*   replaces  voicepuppet/pixrefer/pixrefer.py:59-438   (PixReferNet.build_network,
 *                             add_cost_function, build_train_op, build_inference_op) and
 *                             voicepuppet/pixrefer/vgg_simple.py:96-162 (perceptual trunk)
 *   vp_adam_tf      replaces  tf.train.AdamOptimizer                    (pixrefer.py:398,405)
 *   vp_conv_* etc.  the single ops behind them (tf.layers.conv2d / conv2d_transpose /
 *                             batch_normalization: pixrefer.py:61-101), exported for parity tests
 *   vp_logmel_*     replaces  generator/generator.py:60-80              (DataGenerator.extract_mfcc)
 *   vp_bfmnet_*     replaces  voicepuppet/bfmnet/bfmnet.py:189-213,325-333 + tinynet.py:159-212
 *   vp_render_colors   replaces  utils/cython/mesh_core.h:63 _render_colors_core (mesh_core.cpp:169-231)
 *   vp_bfm_reconstruct replaces  utils/reconstruct_mesh.py:198-223 Reconstruction_rotation + infer_bfmvid.py:92-99
 *
 * Conventions: every function returns 0 on success and a negative vp_status otherwise (never throws);
 * all tensor pointers are DEVICE pointers owned by the caller (NHWC, row-major); nothing is allocated
 * on the device by the library - workspace sizes are queried and the caller passes the buffer; every
 * launch goes to the hipStream_t given (passed as void*), no call synchronises.
 */
#ifndef VP_HIP_H_
#define VP_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum vp_status { VP_OK = 0, VP_ERR_ARG = -1, VP_ERR_HIP = -2, VP_ERR_WORKSPACE = -3, VP_ERR_STATE = -4 };
enum vp_dtype { VP_F32 = 0, VP_BF16 = 1 };
enum vp_act { VP_ACT_NONE = 0, VP_ACT_LRELU = 1, VP_ACT_RELU = 2, VP_ACT_TANH = 3, VP_ACT_SIGMOID = 4, VP_ACT_RELU6 = 5, VP_ACT_LEAKY = 6 };

int vp_version(void);
const char* vp_last_error(void);

/* ------------------------------------------------------------------------------------------------
 * PixReferNet step (pixrefer.py:356-438)
 * ---------------------------------------------------------------------------------------------- */
typedef struct vp_pixrefer_desc {
  int batch;        /* per-device batch N */
  int height;       /* square images, multiple of 256 */
  int ngf, ndf;     /* pixrefer.py:28-29 (64, 64) */
  int dtype;        /* vp_dtype of activations / MFMA operands; master weights, statistics, losses are f32 */
  int training;     /* 1: build_train_op graph (G + 3xD + VGG + losses + grads); 0: build_inference_op */
  float l1_weight;  /* pixrefer.py:30-31 */
  float gan_weight;
  int per_sample_bn; /* inference only: batch-norm statistics per sample (== running N frames through the
                        reference's batch-1 graph, infer_bfmvid.py:152,198, but batched on the device) */
  /* Schedule of a training plan, fixed at vp_pixrefer_create (no reference counterpart: the TF graph has one executor).  0 = default
   * in every field, so a zero-initialised tail keeps the shipped schedule; results are bit-identical under every setting.
   *   streams          0 / 4: the step is spread over the caller's stream + three of the executor's; 3: three in all (a host with a busy
   *                    stream of its own, vp_pixrefer_use_streams); 1: the executor creates no stream, everything runs on the caller's
   *   d_backward_fork  0: default (the discriminator-loss pass starts behind the generator-loss pass through D and the VGG trunk);
   *                    1 / 2 / 3: fork point 0 (at once) / 1 (behind the pass through D) / 2 (the default)
   *   d_beside_vgg     0 / 2: discriminator passes on the branch stream beside the VGG passes (default); 1: on the caller's stream */
  int streams;
  int d_backward_fork;
  int d_beside_vgg;
} vp_pixrefer_desc;

/* ABI rule for this descriptor (the precedent is utils/cython/mesh_core.h:53-77: explicit sizes, caller-owned buffers): the struct only
 * ever GROWS AT THE TAIL (rounds 1-4: the first 9 fields = 36 bytes; round 5 added the three schedule fields = 48 bytes), and every entry
 * point that takes it reads sizeof(vp_pixrefer_desc) bytes OF THE LIBRARY'S BUILD.  A binding that declares the struct itself (ctypes,
 * cgo, JNI) must check vp_pixrefer_desc_size() == its own sizeof at load time and refuse to run on a mismatch - a shorter struct would
 * be over-read (INTEGRATION.md B does; tests/test_host_logic.py executes that stub under AddressSanitizer). */
size_t vp_pixrefer_desc_size(void);

typedef struct vp_pixrefer vp_pixrefer_t;

/* Parameter manifest: TF variable names -> offset/shape in the flat f32 arenas.
 * which: 0 = generator*, 1 = discriminator*, 2 = vgg_16 (conv1_1 .. conv3_3).
 * vp_pixrefer_param_info returns VP_OK and fills the outputs, or VP_ERR_ARG when index is past the end. */
size_t vp_pixrefer_param_count(const vp_pixrefer_desc* d, int which);
int vp_pixrefer_param_info(const vp_pixrefer_desc* d, int which, int index, char* name, int name_cap,
                           size_t* offset, int* ndim, int64_t shape[4]);

size_t vp_pixrefer_workspace_bytes(const vp_pixrefer_desc* d);
/* Host-only self-check of the plan a descriptor produces (no GPU needed): every buffer the kernels would be handed is a carved
 * region of sufficient size inside the workspace, regions do not overlap, parameter / packed-weight ranges lie inside their
 * arenas, split-K slabs fit the scratch.  VP_OK, VP_ERR_ARG (bad descriptor) or VP_ERR_STATE (vp_last_error names the breach).
 * Run under AddressSanitizer / UBSan by the `make host-asan` build of the host layer (tests/test_host_logic.py). */
int vp_pixrefer_validate_plan(const vp_pixrefer_desc* d);

/* params_* / grads_*: flat f32 device arenas laid out as the manifest says (grads may be NULL when
 * training == 0; params_d / params_vgg likewise). */
int vp_pixrefer_create(const vp_pixrefer_desc* d, void* workspace, size_t workspace_bytes,
                       float* params_g, float* params_d, const float* params_vgg,
                       float* grads_g, float* grads_d, void* stream, vp_pixrefer_t** out);
void vp_pixrefer_destroy(vp_pixrefer_t* h);

/* Call after the f32 master parameters changed: _params_changed after the host wrote any arena (checkpoint load, initialisation:
 * all three nets are re-packed), _optimizer_stepped after vp_adam_tf on generator* / discriminator* (the frozen vgg_16 trunk -
 * restored from a checkpoint, never an optimiser variable: pixrefer.py:325-327, 396-407 - keeps its packed weights). */
int vp_pixrefer_params_changed(vp_pixrefer_t* h);
int vp_pixrefer_optimizer_stepped(vp_pixrefer_t* h);

/* inputs [N,H,H,6], fg_inputs [N,H,H,6] (inference: only channels 0:3 are read), targets [N,H,H,3],
 * masks [N,H,H,3] (training only) - float32 in [0,1] exactly as PixReferDataGenerator yields them
 * (generator.py:1011-1019).  Runs the generator, the composite, and when training the three
 * discriminator applications, the VGG trunk and all losses. */
int vp_pixrefer_forward(vp_pixrefer_t* h, const float* inputs, const float* fg_inputs,
                        const float* targets, const float* masks, void* stream);

/* The inference graph as infer_bfmvid.py:202-205 feeds it: fg_inputs3 [N,H,H,3] (build_inference_op reads fg_inputs[..., :3] only,
 * pixrefer.py:281).  Inference plans only (VP_ERR_STATE on a training plan, whose graph reads channels 3:6 too). */
int vp_pixrefer_forward_fg3(vp_pixrefer_t* h, const float* inputs, const float* fg_inputs3, const float* targets, void* stream);

/* Both gradient sets from the forward just run: d(Discrim_loss)/d(discriminator*) -> grads_d,
 * d(Gen_loss)/d(generator*) -> grads_g (pixrefer.py:396-407; pre-update weights for both). */
int vp_pixrefer_backward(vp_pixrefer_t* h, void* stream);
/* vp_pixrefer_backward + both tf.train.AdamOptimizer updates (vp_adam_tf on generator* and discriminator*) + the weight re-pack
 * the next forward would do, as ONE call: every range of an arena is updated as soon as its gradients are final, on the side /
 * branch streams under the rest of the backward pass.  m / v: Adam slot arenas (same layout as the parameter arenas);
 * step_t: 1-based Adam step of each optimiser.  Parameters after the call are bit-identical to the three separate calls.
 * Single-GPU steps only: a data-parallel host must all-reduce the gradients first (vp_pixrefer_backward_g_stage). */
int vp_pixrefer_backward_update(vp_pixrefer_t* h, float* m_g, float* v_g, float* m_d, float* v_d, int step_t_g, int step_t_d,
                                float lr, float beta1, float beta2, float eps, void* stream);

/* The two halves of vp_pixrefer_backward, so a data-parallel host can start the all-reduce of the
 * discriminator gradients while the generator backward runs. */
int vp_pixrefer_backward_d(vp_pixrefer_t* h, void* stream);
int vp_pixrefer_backward_g(vp_pixrefer_t* h, void* stream);
/* vp_pixrefer_backward runs the two (independent) halves CONCURRENTLY: the discriminator-loss pass on a HIP stream the handle
 * owns, forked from and joined into `stream` with events (no host blocking; bit-identical results).  The same for a host with
 * work of its own in between: _fork schedules the discriminator-loss pass (it starts inside stage 0 of the generator backward,
 * behind the generator-loss pass through the discriminator - or at once with vp_pixrefer_set_option(h, "d_backward_fork", 0)), _join makes `stream`
 * wait for it; grads_d is final after the join, which may come after any stage (the later, the more of the pass is hidden).  (vp_pixrefer_desc::streams = 1: both run
 * on `stream`, one after the other.) */
int vp_pixrefer_backward_d_fork(vp_pixrefer_t* h, void* stream);
int vp_pixrefer_backward_d_join(vp_pixrefer_t* h, void* stream);
/* The executor's own side stream (hipStream_t), for a data-parallel host that issues its collectives there instead of on one more
 * stream of its own (the device has few hardware queues; streams beyond them share one).  Work the host enqueues on it runs behind the
 * discriminator-loss pass of the step.  NULL if the plan runs on a single stream (vp_pixrefer_desc::streams = 1). */
void* vp_pixrefer_side_stream(vp_pixrefer_t* h);
/* The executor's three extra HIP streams are process-wide (one device per process), created once - by this call, or by the first
 * training plan - and shared by every plan of the process; they are never destroyed.  A host that also creates an RCCL communicator (or
 * any other busy stream) calls this FIRST: streams created behind other streams get the HIP runtime's leftover hardware queues and the
 * step runs 8 % (32 frames) to 30 % (4 frames) slow (scripts/exp_dp_order.py).  No counterpart in the reference (tf.Session owns its
 * executor, train_pixrefer.py:34). */
int vp_reserve_streams(void);
/* The fourth of those streams, for a host with ONE busy stream of its own (an input prefetcher: generator/device_pipeline.py
 * FramePrefetcher) whose plans keep to three (vp_pixrefer_use_streams(h, 3)): a stream the host created itself would be the process's
 * fifth and share a hardware queue (the PCIe-inclusive step: 9.2 ms instead of 7.4).  NULL on error (vp_last_error). */
void* vp_host_stream(void);
/* Streams a training step is spread over: 4 (default) or 3.  A host that runs a busy stream of its own beside the step (an input
 * prefetcher) asks for 3: the device has few hardware queues, a fifth busy stream shares one with an executor stream. */
int vp_pixrefer_use_streams(vp_pixrefer_t* h, int n);
/* Schedule options of ONE plan (no reference counterpart: the TF graph has one executor): "overlap" 0 / 1 (the whole step on `stream`
 * / spread over the executor's streams, default 1), "d_backward_fork" 0..2 (where vp_pixrefer_backward starts the discriminator-loss
 * pass, default 2), "d_beside_vgg" 0 / 1 (default 1); and "store_first_raw" 0 / 1 (default 0): encoder_1 / encoder_fg_1 / discriminator
 * layer_1 of a bf16 plan write their consumers' activations from the conv epilogue and skip the raw output nobody reads - 1 stores it
 * too (vp_pixrefer_tensor refuses "g/encoder_1" ... otherwise).  Per handle: two plans in one process do not change each other's schedule; the
 * initial values come from the descriptor (vp_pixrefer_desc::streams / d_backward_fork / d_beside_vgg).  Bit-identical results under
 * every setting. */
int vp_pixrefer_set_option(vp_pixrefer_t* h, const char* key, int value);
/* Round 6 keys: "bwd_sums_in_epilogue" 0 / 1 (default 1): the two sums of a batch-norm backward pass (sum dz, sum dz * zhat) come from the
 * epilogue of the launch that completes the tensor's gradient instead of a pass of their own over y and dz (same values up to the order of
 * float32 partial sums); "vgg_real_fork" k (default 3): the VGG pass of the real half starts on the side stream in front of generator layer
 * k (TF scope order; 0 = right behind the input packing).  vp_pixrefer_counter: "bwd_sums_launches" = launches since create that carried
 * such sums (-1: unknown key); for tests. */
long long vp_pixrefer_counter(vp_pixrefer_t* h, const char* key);
/* Node values PixReferNet.execute hands to a caller, formed on the device from the last forward pass into `dst` (device memory,
 * N * H * H * 3 elements): what = 0 Outputs (float32, (x + 1) / 2: pixrefer.py:424 / :380), 1 the same as uint8 frames (clamp, * 255,
 * truncate: the bytes infer_bfmvid.py:243 writes), 2 Alphas (float32, three channels: pixrefer.py:284), 3 Outputs_FG as this plan's
 * graph defines it (build_inference_op: ((Outputs_FG + Alphas - 1) + 1) / 2, pixrefer.py:436; build_train_op: the tensor itself). */
int vp_pixrefer_fetch(vp_pixrefer_t* h, int what, void* dst, void* stream);
/* The same pass in vp_pixrefer_backward_g_stages() = 3 consecutive stages (stage < 0: all).  After stage s a contiguous
 * range of the generator gradient arena is final (0: from generator/merged_decoder_5 to the end; 1: from
 * generator/merged_encoder_2 up to merged_decoder_5; 2: the rest), so a data-parallel host can start that bucket's
 * all-reduce while the next stage computes.
 * STREAM-ORDER CONTRACT for a host that hands an arena range to another stream (a collective): the executor runs parts of a stage
 * on HIP streams of its own, but before vp_pixrefer_backward_g_stage / vp_pixrefer_backward_d_join return they have made `stream`
 * wait (hipStreamWaitEvent) for every kernel that writes the range the call completes.  An event recorded on `stream` right after
 * the call therefore covers the whole range; the collective's stream must wait for THAT event (voicepuppet_amd/parallel.py
 * GradExchange does) - it must not read the arena on the strength of host-side ordering alone. */
int vp_pixrefer_backward_g_stages(void);
int vp_pixrefer_backward_g_stage(vp_pixrefer_t* h, int stage, void* stream);
/* Data parallel: tf.train.AdamOptimizer + weight re-pack of ONE gradient bucket on `stream`, for a host that has just all-reduced
 * it there: which = 0 generator (bucket = the stage that completed it), which = 1 discriminator (bucket 0).  `stream` must be
 * ordered behind that stage (see the contract above).  Bit-identical to vp_adam_tf over the whole arena; after the four buckets of
 * a step the packed weights are current (the next forward re-packs nothing). */
int vp_pixrefer_update_bucket(vp_pixrefer_t* h, int which, int bucket, float* m, float* v, int step_t, float lr, float beta1,
                              float beta2, float eps, void* stream);

/* Named device buffers ("nodes" of pixrefer.py:356-438 and every intermediate):
 *   "Outputs_raw" [N,H,H,3] f32 in [-1,1], "Outputs_FG" [N,H,H,3] f32, "gen_out4" [N,H,H,4] f32,
 *   "Predict" [2,N,h,h] f32 (real, fake), "losses" [8] f32 = {Discrim_loss, Gen_loss_GAN, Gen_loss_L1,
 *   Gen_loss, Perceptual_loss}, "g/<scope>" raw conv outputs, "g/<scope>:dy" their gradients, ...
 * dtype receives the vp_dtype of the buffer. */
int vp_pixrefer_tensor(vp_pixrefer_t* h, const char* name, void** ptr, int64_t shape[4], int* dtype);

/* Per-launch timing of the conv kernels with HIP events on the launch stream (bench.py roofline).
 * vp_profile_collect: JSON array [{name, calls, ms, flops, bytes}] since the last collect; call after
 * synchronising the stream; returns the bytes needed (including the terminator). */
int vp_profile_enable(int on);
size_t vp_profile_collect(char* json, size_t cap);

/* Input pipeline on the device.  Replaces the per-sample host work of PixReferDataGenerator.iterator (generator/generator.py:
 * 956-1019: BGR->RGB, split of the S x 3S triptych into target | 3-D face | matte, random square crop, cv2.resize back to S x S,
 * the 6-channel packing of (example, current) and fg = target * matte).  example_frames / current_frames: [n][S][3S][3] uint8
 * exactly as cv2.imread decodes the training jpgs; crops: [n][2][3] int32 = (rx rows, ry columns, rsize) for the example and the
 * current frame, 0 <= rx, ry and rx + rsize, ry + rsize <= S (drawn by the caller as generator.py:975-977, 994-996); outputs:
 * the four float32 tensors vp_pixrefer_forward takes.  All pointers are device memory; crop values are NOT range-checked. */
int vp_pixrefer_pack_frames(const unsigned char* example_frames, const unsigned char* current_frames, const int* crops,
                            int n, int img_size, float* inputs, float* fg_inputs, float* targets, float* masks, void* stream);

/* Kernel-selection knobs for tests and experiments (they choose between kernels that compute the same result); plans made AFTER
 * the call see the new value.  Keys: "patch_tiles" (bit 0 / 1 / 2: allow the 256- / 128- / 64-row tiles of the stride-1 patch
 * kernel, default 7), "patch_min_blocks" (smallest grid that runs on it, default 192 since round 6 - 384 before; < 0: back to the default), "patch_small_tiles" (bit 0: 16x16-pixel
 * tiles for the 128- / 64-row variants, bit 1: 8x16 for the 256-row variant - two blocks per CU; default 3), "patch_long_k_on_256"
 * (default 1: >= 512-channel layers with K >= 4096 stay on the wave-specialised 256x256 tile - float32 plans; bf16 plans: "patch4"), "c64" / "dc64" (default 1: the
 * register-resident-weights kernels conv_c64.hip / conv_dc64.hip - "dc64" also the forward form conv_dc256_kernel; 0: the patch kernels
 * those layers ran on before), "s2c64" (smallest launch, in 4 x 16-pixel tiles, of the 64 -> 128 stride-2 convolutions that runs on
 * conv_s2c64.hip: default 512, 0 = never), "s2c64_pair" (default 1: its two-output backward-data form for merged2_decoder_2), "patch4"
 * (default 1: 4x4 stride-1 layers on the unrolled patch kernel with 16 tap steps), "bfm_dwproj" (default 1: BFMNet's depthwise + projection
 * in one kernel; read at every forward call).  No counterpart in the reference.  (The step executor's schedule is per plan: vp_pixrefer_desc / vp_pixrefer_set_option.) */
int vp_tune(const char* key, int value);
/* Round-6 plan heuristics: "igemm_small_grid" (default 128: a launch whose 128 x 128 tiling has at most that many blocks per class takes the
 * 64-row x 128-pixel tile - twice the blocks; 0: off), "igemm_splitk_target" (default 64, rounds 2-5: 128: resident blocks a K split aims at;
 * < 0: back to the default); "thin_blocks_cout8" / "thin_blocks_dcout8" / "thin_blocks_cout4" / "thin_blocks_cin8" (defaults 1024 / 512 /
 * 512 / 512: grid caps of the persistent thin-layer kernels conv3x3_cout8_tile / deconv_cout8_tile / deconv_cout4_tile / conv_cin8;
 * values <= 0 are ignored); "cout1_bwd" (default 256: most blocks per batch-norm group of the one-output-channel backward-data kernel
 * conv_cout1.hip - the discriminator's layer_5 -, 0: that layer's backward passes on the generic kernels, < 0: back to the default), "cout1_wgrad_rows"
 * (default 512: most blocks = slabs of its weight-gradient kernel). */
/* Further keys: "smallp_max_pixels" (largest pixel count per parity class that runs on the few-pixel kernel conv_smallp.hip,
 * default 256, 0: off), "phase_marks" (1: the step executor records HIP events on the caller's stream at its phase boundaries).
 * vp_pixrefer_phase_ms: milliseconds between consecutive marks of the last step (synchronises on them): generator forward,
 * discriminator / VGG forward + losses, generator-loss pass through D and VGG + composite backward (including the host gap between
 * the forward and the backward call), generator backward stage 0, 1, 2, join of the discriminator-loss pass.  Returns the number written. */
int vp_pixrefer_phase_ms(vp_pixrefer_t* h, float* ms, int cap);
/* "phase_marks" = 2 adds a mark in front of every generator layer (forward: mark 8 + layer, backward on the caller's stream:
 * 32 + layer, layers in TF scope order); vp_pixrefer_mark_ms: milliseconds between two marks of the last step, -1 if not recorded. */
float vp_pixrefer_mark_ms(vp_pixrefer_t* h, int from, int to);

/* Data parallel, optional bf16 transport of a gradient bucket (the f32 arena stays the master copy): round n floats to bf16
 * (nearest even) into a communication buffer; after the bf16 all-reduce (sum) write them back as f32 times `scale` (= 1 / world).
 * src / dst float pointers 32-byte aligned, the bf16 buffer 16-byte aligned.  No counterpart in the reference (single device). */
int vp_grad_pack_bf16(const float* src, void* dst_bf16, size_t n, void* stream);
int vp_grad_unpack_bf16(const void* src_bf16, float* dst, size_t n, float scale, void* stream);

/* theta -= lr_t * m / (sqrt(v) + eps) with lr_t = lr*sqrt(1-beta2^t)/(1-beta1^t) (TF formulation) */
int vp_adam_tf(float* params, const float* grads, float* m, float* v, size_t n, int step_t,
               float lr, float beta1, float beta2, float eps, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Single ops (parity tests; same kernels the step uses).  x: NHWC in `dtype`.
 * in_scale/in_shift: optional per-channel deferred batch-norm affine of the producer, in_act applied
 * after it (pixrefer.py:182-186: act -> conv -> BN).  workspace: vp_conv_workspace_bytes().
 * ---------------------------------------------------------------------------------------------- */
typedef struct vp_conv_desc {
  int kind;          /* 0: conv2d HWIO (pixrefer.py:61-74, vgg_simple.py:138); 1: conv2d_transpose k4 s2 HWOI (pixrefer.py:85) */
  int n, h, w;       /* input size */
  int cin, cout;     /* cin % 8 == 0 and a power of two; any cout */
  int ksize, stride, pad;
  int dtype;
  int in_act;
  int out_act;       /* fwd only */
} vp_conv_desc;

size_t vp_conv_workspace_bytes(const vp_conv_desc* d);
/* y = out_act(conv(in_act(in_scale*x+in_shift), w) + bias); y in `dtype`, [n,ho,wo,cout] */
int vp_conv_fwd(const vp_conv_desc* d, const void* x, const float* in_scale, const float* in_shift,
                const float* w, const float* bias, void* y, void* workspace, void* stream);
/* dx = d/d(in_act(...) input)  (i.e. w.r.t. the activated tensor the conv reads), [n,h,w,cin] in `dtype` */
int vp_conv_bwd_data(const vp_conv_desc* d, const void* dy, const float* w, void* dx, void* workspace, void* stream);
/* dw in the TF layout of `kind`, f32 */
int vp_conv_bwd_weight(const vp_conv_desc* d, const void* x, const float* in_scale, const float* in_shift,
                       const void* dy, float* dw, void* workspace, void* stream);

/* training-mode batch norm statistics (pixrefer.py:99-101): y [pixels, c] -> scale/shift (z = scale*y+shift),
 * mean, rstd; biased variance, eps inside the sqrt.  workspace: vp_bn_workspace_bytes(). */
size_t vp_bn_workspace_bytes(int pixels, int c, int dtype);
int vp_bn_stats(const void* y, int pixels, int c, int dtype, const float* gamma, const float* beta, float eps,
                float* scale, float* shift, float* mean, float* rstd, void* workspace, void* stream);
/* dy = BN backward of dz (in place allowed), dgamma, dbeta */
int vp_bn_bwd(const void* y, const void* dz, void* dy, int pixels, int c, int dtype, const float* gamma,
              const float* mean, const float* rstd, float* dgamma, float* dbeta, void* workspace, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Audio front-end (f32).
 * vp_logmel_*  : DataGenerator.extract_mfcc (generator/generator.py:60-80): Hann STFT -> |.| -> HTK mel -> log(.+1e-6)
 * vp_bfmnet_*  : BFMNet.build_inference_op (voicepuppet/bfmnet/bfmnet.py:325-333 -> 189-213) with MfccNet
 *                (tinynet.py:159-212) in inference mode; parameters by TF variable name (vp_bfmnet_param_info).
 * ---------------------------------------------------------------------------------------------- */
typedef struct vp_logmel_desc {
  int sample_rate, num_mel_bins, win_length, hop_step, fft_length;   /* config/params.yml:16-21; win == fft */
  float lower_hz, upper_hz;                                          /* 80, 7600 (generator.py:68) */
  int batch, samples;                                                /* pcm [batch, samples] */
} vp_logmel_desc;
typedef struct vp_logmel vp_logmel_t;
size_t vp_logmel_workspace_bytes(const vp_logmel_desc* d);
int vp_logmel_frames(const vp_logmel_desc* d);                       /* 1 + (samples - win)/hop */
int vp_logmel_create(const vp_logmel_desc* d, void* workspace, size_t workspace_bytes, void* stream, vp_logmel_t** out);
void vp_logmel_destroy(vp_logmel_t* h);
/* pcm [batch, samples] f32 in [-1,1] -> out [batch, frames, num_mel_bins] f32 */
int vp_logmel_forward(vp_logmel_t* h, const float* pcm, float* out, void* stream);

typedef struct vp_bfmnet_desc {
  int batch;          /* clips */
  int frames;         /* T video frames per clip; the mel input has 5*T frames (frame_mfcc_scale, generator.py:46-52) */
  int num_mel_bins;   /* 80 */
  int trunk_dtype;    /* VP_F32 (parity path: coefficients 1e-5 vs the float64 oracle) or VP_BF16 (opt-in, 2x the rate): the 6x-expanded
                         tensors and the 1x1-conv operands of MfccNet in bf16, f32 accumulation / residual stream / depthwise /
                         pooling / head; trunk features within 4e-2 rel-L2 of the oracle (tests/test_gpu_audio.py) */
} vp_bfmnet_desc;
typedef struct vp_bfmnet vp_bfmnet_t;
size_t vp_bfmnet_param_count(void);
int vp_bfmnet_param_info(int index, char* name, int name_cap, size_t* offset, int* ndim, int64_t shape[4]);
size_t vp_bfmnet_workspace_bytes(const vp_bfmnet_desc* d);
int vp_bfmnet_create(const vp_bfmnet_desc* d, void* workspace, size_t workspace_bytes, const float* params,
                     void* stream, vp_bfmnet_t** out);
void vp_bfmnet_destroy(vp_bfmnet_t* h);
int vp_bfmnet_params_changed(vp_bfmnet_t* h);
/* ears [B,T,1], mfccs [B,5T,80], seq_len [B] (int32) -> BFMCoeffDecoder [B,T,64]; all device pointers */
int vp_bfmnet_forward(vp_bfmnet_t* h, const float* ears, const float* mfccs, const int* seq_len, float* out, void* stream);
/* Opt-in: the reference's BFMCoeffDecoder applies tf.nn.dropout(keep_prob = 0.75) behind both hidden dense layers unconditionally,
 * i.e. also at inference (voicepuppet/bfmnet/bfmnet.py:114,116: that class's drop_rate is never zeroed).  The default forward omits
 * both (deterministic: the expectation of the reference's output).  With masks set - mask0 [B*T,128], mask1 [B*T,64], device f32,
 * entries 0 or 1 / keep_prob, owned by the caller until cleared with NULLs - every following forward multiplies the two hidden
 * activations by them: one SAMPLE of the reference's inference output for that draw. */
int vp_bfmnet_set_decoder_dropout(vp_bfmnet_t* h, const float* mask0, const float* mask1);
/* "MfccEncoder" [B,T,256], "RNNModule" [B,T,256] of the last forward */
int vp_bfmnet_tensor(vp_bfmnet_t* h, const char* name, void** ptr, int64_t shape[4]);

/* ------------------------------------------------------------------------------------------------
 * Single pointwise / audio ops of the two executors (the entry-point list of SURVEY.md 8b), for parity tests and reuse.
 *   vp_maxpool2x2_*      slim max_pool2d 2x2/2 of vgg_simple.py:141,146 (NHWC).  bwd goes through the pool AND the ReLU of the conv
 *                        that produced x (x is stored post-relu): the gradient lands on the FIRST maximum of a window if it is > 0
 *   vp_composite_fwd     pixrefer.py:279-290 inference compositing: out4 = tanh(gen_out4), alpha = (out4[3]+1)/2,
 *                        outputs = rgb*alpha + (2*targets-1)*(1-alpha), outputs_fg = rgb*alpha + alpha - 1   (all float32)
 *   vp_gan_loss          pixrefer.py:334-347: logits [3][m] = D(real1) | D(real2) | D(fake) -> predict [2][m], losses[0] = Discrim_loss,
 *                        losses[1] = Gen_loss_GAN, and the loss seeds w.r.t. the logits ([3][m][8] / [m][8] of dtype, channel 0)
 *   vp_dwconv7x3_bn_act  tinynet.py depthwise [7,3] stride 1 'same' + folded BatchNorm (bias) + relu6; w [21][c], float32 NHWC
 *   vp_maxpool_hw        tf.layers.max_pooling2d(k, s, 'same') (tinynet.py:178-190, bfmnet.py:35); output ceil(h/sh) x ceil(w/sw)
 *   vp_gru_seq           tf.contrib.rnn.GRUCell under dynamic_rnn (bfmnet.py:53-61), 256 units: xg [b,t,512] / xc [b,t,256] are the input
 *                        projections (+ biases), whg [256][512] / whc [256][256] the recurrent halves of gates / candidate kernels
 * ---------------------------------------------------------------------------------------------- */
int vp_maxpool2x2_fwd(const void* x, void* y, int n, int h, int w, int c, int dtype, void* stream);
int vp_maxpool2x2_bwd(const void* x, const void* dy, void* dx, int n, int h, int w, int c, int dtype, void* stream);
int vp_composite_fwd(const float* gen_out4, const float* targets, float* out4, float* outputs, float* outputs_fg, int n, int hw,
                     void* stream);
int vp_gan_loss(const float* logits, void* seed_d, void* seed_g, float* predict, float* losses, int m, float gan_weight, int dtype,
                void* stream);
int vp_dwconv7x3_bn_act(const float* x, const float* w, const float* bias, float* y, int b, int h, int wd, int c, void* stream);
int vp_maxpool_hw(const float* x, float* y, int b, int h, int w, int c, int kh, int kw, int sh, int sw, void* stream);
int vp_gru_seq(const float* xg, const float* xc, const float* whg, const float* whc, const int* seq_len, float* out, int b, int t,
               void* stream);

/* ------------------------------------------------------------------------------------------------
 * Rasteriser ("next" row): replaces mesh_core_cython.render_colors_core ->
 * _render_colors_core(image, face_mask, vertices, triangles, colors, depth_buffer, ntri, h, w, c)
 * (utils/cython/mesh_core.h:63, mesh_core.cpp:169-231; caller infer_bfmvid.py:100-108).  Same argument order and
 * in-place convention (image / face_mask / depth_buffer are read-modify-write), plus nver, a batch of frames that share
 * `triangles` (vertices [batch,nver,3], colors [batch,nver,c], outputs [batch,h,w,...]), a workspace and a stream.
 * Bit-exact with the reference: deepest mean-depth triangle wins, ties go to the lowest index, colour (int)(c0+c1+c2)/3.
 * ---------------------------------------------------------------------------------------------- */
size_t vp_render_colors_workspace_bytes(int batch, int h, int w);
int vp_render_colors(unsigned char* image, unsigned char* face_mask, const float* vertices, const int* triangles,
                     const float* colors, float* depth_buffer, int ntri, int nver, int h, int w, int c, int batch,
                     void* workspace, void* stream);

/* ------------------------------------------------------------------------------------------------
 * BFM reconstruction for a clip ("next" row): replaces the per-frame numpy of
 * utils/reconstruct_mesh.py Reconstruction_rotation(coeff, facemodel, angles) (:198-223) and the float32 / integer packing
 * of infer_bfmvid.py:92-99.  All pointers are device pointers.  The model is the reference's `BFM` object
 * (utils/bfm_load_data.py:9-21) promoted to float64 with 0-based indices: tri [ntri,3]; point_buf [nver,8] with `ntri`
 * marking "no face" (the reference's appended zero normal, reconstruct_mesh.py:47-49).  `center` = column means of
 * meanshape (:27), `sh` = {a0c0, a1c1, a2c2, a2c2/2/sqrt(3), a2c2/2} of Illumination_layer (:138-155), both evaluated
 * by the host in double; focal / image_center are Projection_layer's 1015 / 112 (:100-101).
 * coeff [frames,257] float32; rotation [frames,9] = Compute_rotation_matrix(angles) (:68-93) row-major, float64.
 * Outputs: vertices [frames,nver,3] = (x, 224-y, z_buffer) float32 and colors [frames,nver,3] = float(int(clip(c,0,255))),
 * i.e. exactly the arrays infer_bfmvid.py:101-103 passes to render_colors_core; the float64 intermediates the reference
 * returns (face_shape, face_texture, face_color, face_projection [.,.,2], z_buffer) are written when non-NULL.
 * shared_texture != 0: the texture coefficients are constant over the clip, computed once from frame 0
 * (face_texture then holds 1 frame).
 * ---------------------------------------------------------------------------------------------- */
typedef struct vp_bfm_model {
  int nver, ntri;
  const double* meanshape; /* [3*nver] */
  const double* idBase;    /* [80,3*nver]  K-MAJOR: the reference's [3*nver,80] transposed once at model load */
  const double* exBase;    /* [64,3*nver] */
  const double* meantex;   /* [3*nver] */
  const double* texBase;   /* [80,3*nver] */
  const int* tri;          /* [ntri,3] */
  const int* point_buf;    /* [nver,8] */
  double center[3];
  double focal, image_center;
  double sh[5];
} vp_bfm_model;
size_t vp_bfm_reconstruct_workspace_bytes(int nver, int ntri, int frames);
int vp_bfm_reconstruct(const vp_bfm_model* m, const float* coeff, const double* rotation, int frames, int shared_texture,
                       double* face_shape, double* face_texture, double* face_color, double* face_projection, double* z_buffer,
                       float* vertices, float* colors, void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * BFMNet TRAINING step (voicepuppet/bfmnet/bfmnet.py:215-323, tinynet.py:7-212): the non-GEMM kernels, float32 NHWC,
 * any channel count that is a multiple of 4.  voicepuppet_amd/bfmnet/train_engine.py chains them with plain GEMMs (rocBLAS).
 *   vp_bn_train_fwd / _bwd   tf.contrib.layers.batch_norm(is_training=True, scale=False, eps 1e-3): batch statistics + the affine that
 *                            normalises (y = x * scale + shift), and its backward (dx, dbeta)
 *   vp_bn_act_train_bwd      the same with the backward of the relu / relu6 that follows folded in (no dz tensor is materialised)
 *   vp_affine_act_fwd        y = act(scale[c] * x + shift[c]) * mask (affine and mask optional: relu / relu6 / leaky-relu, dropout)
 *   vp_act_bwd               dx = dy * mask * act'(y)
 *   vp_dwconv7x3_raw         depthwise [7,3] SAME without bias / activation (forward; backward-data with reversed taps);
 *   vp_dwconv7x3_wgrad       its weight gradient [21][c]
 *   vp_maxpool_hw_bwd        backward of vp_maxpool_hw (first maximum of a window, as TF's MaxPoolGrad)
 *   vp_stem_im2col           the 9x5 stride-(1,2) stem as a [pixels, 48] matrix (GEMM operand for forward and weight gradient)
 *   vp_gru_train_fwd / _bwd  GRUCell recurrence with saved gates, and backward through time to the gate / candidate pre-activations.
 *                            whg [256][512] / whc [256][256]: the recurrent halves of the two kernels (row = h unit).  _bwd takes them
 *                            TRANSPOSED (whg_t [512][256], whc_t [256][256]): its products with the kernels' rows then read coalesced columns
 *   vp_bfm_vertex_loss       add_cost_function on D = face_shape(true) - face_shape(pred): loss partials (f64) and dLoss/dD
 *   vp_sumsq                 sum of squares partials (f64): global-norm clipping
 *   vp_l2_regulariser        tf.losses.get_regularization_loss() over the flat arena: gradient contribution + value partials
 *   vp_adam_tf_clipped       tf.clip_by_global_norm + tf.train.AdamOptimizer (bfmnet.py:313-318) with the step scalars on the device
 *   vp_moving_update         the moving-average updates of every batch_norm in one pass (decay 0.999, tinynet.py:20-27)
 * ---------------------------------------------------------------------------------------------- */
size_t vp_bn_train_workspace_bytes(size_t pixels, int c);
int vp_bn_train_fwd(const float* x, size_t pixels, int c, const float* beta, float eps, float* mean, float* var, float* rstd, float* scale,
                    float* shift, void* workspace, void* stream);
int vp_bn_train_bwd(const float* x, const float* dz, size_t pixels, int c, const float* mean, const float* rstd, float* dx, float* dbeta,
                    void* workspace, void* stream);
int vp_bn_act_train_bwd(const float* x, const float* da, size_t pixels, int c, const float* mean, const float* rstd, const float* shift, int act,
                        float* dx, float* dbeta, void* workspace, void* stream);
int vp_affine_act_fwd(const float* x, const float* scale, const float* shift, const float* mask, size_t pixels, int c, int act, float* y, void* stream);
/* the same pass + the residual branch: y = act(scale * x + shift) * mask + add */
int vp_affine_act_add_fwd(const float* x, const float* scale, const float* shift, const float* mask, const float* add, size_t pixels, int c, int act,
                          float* y, void* stream);
int vp_act_bwd(const float* dy, const float* ya, const float* mask, size_t n, int act, float* dx, void* stream);
int vp_dwconv7x3_raw(const float* x, const float* w, float* y, int b, int h, int wd, int c, void* stream);
/* backward-data of vp_dwconv7x3_raw: dy convolved with the taps of w reversed (no flipped copy of w is needed) */
int vp_dwconv7x3_bwd_data(const float* dy, const float* w, float* dx, int b, int h, int wd, int c, void* stream);
size_t vp_dwconv7x3_wgrad_workspace_bytes(int b, int h, int wd, int c);
int vp_dwconv7x3_wgrad(const float* x, const float* dy, float* dw, int b, int h, int wd, int c, void* workspace, void* stream);
int vp_maxpool_hw_bwd(const float* x, const float* dy, float* dx, int b, int h, int w, int c, int kh, int kw, int sh, int sw, void* stream);
int vp_stem_im2col(const float* x, float* col, int b, int h, int w, void* stream);
int vp_gru_train_fwd(const float* xg, const float* xc, const float* whg, const float* whc, const int* seq_len, float* out, float* r, float* u, float* c,
                     float* hprev, int b, int t, void* stream);
/* glue of the training step (bias gradients = column sums over the B*T rows, the recurrent halves of the GRUCell kernels split and
 * transposed in one launch, element-wise products of the dropout masks / gate products, the ear padding of bfmnet.py:117,210) */
int vp_colsum_f32(const float* x, int rows, int cols, float* out, void* stream);
int vp_gru_split_recurrent(const float* gates_kernel, const float* cand_kernel, float* whg, float* whc, float* whg_t, float* whc_t, void* stream);
int vp_mul_f32(const float* a, const float* b, float* out, size_t n, void* stream);
int vp_add_ears_f32(float* out, const float* ears, int rows, void* stream);
int vp_gru_train_bwd(const float* dout, const float* whg_t, const float* whc_t, const int* seq_len, const float* r, const float* u, const float* c,
                     const float* hprev, float* dag, float* dac, int b, int t, void* stream);
int vp_vertex_loss_partials(int b, int j);
int vp_bfm_vertex_loss(const float* d, const float* vmask, const int* seq_len, int b, int t, int j, float* gd, double* partial, void* stream);
int vp_sumsq_partials(size_t n);
int vp_sumsq(const float* x, size_t n, double* partial, void* stream);
int vp_l2_regulariser(const float* params, const float* mask, float* grads, size_t n, float scale, double* partial, void* stream);
/* The step's scalars on the device (no framework reduction / sqrt / stack on the path): out[0] = (add ? add[0] : 0) + scale * sum(partial[0..n));
 * out3 = [loss_data + half_l2 * reg, loss_data, sqrt(sumsq)]; grads *= clip / max(sqrt(sumsq), clip).  All pointers are device memory. */
int vp_sum_f64(const double* partial, int n, double scale, const double* add, double* out, void* stream);
int vp_bfm_step_report(const double* loss_data, const double* reg, double half_l2, const double* sumsq, double* out3, void* stream);
int vp_clip_scale_f32(float* grads, size_t n, const double* sumsq, float clip, void* stream);
int vp_adam_tf_clipped(float* params, float* grads, float* m, float* v, size_t n, const float* lr_t, const double* sumsq, float clip, float beta1,
                       float beta2, float eps, void* stream);
int vp_moving_update(float* moving, const float* batch, const float* factor, size_t n, float decay, void* stream);

/* ---- plain float32 matrix products on the repo's own MFMA kernels (the BFMNet training step is made of them: bfmnet.py:215-323) ----
 * Row-major, leading dimensions in floats.  vp_mm_fwd_f32: y[P,N] = x[P,K] . w[K,N] (+ bias[N]); w_transposed: w is stored [N,K].
 * vp_mm_bwd_data_f32: dx[P,K] (+)= dy[P,N] . w[K,N]^T.  vp_mm_bwd_weight_f32: dw[k_real,N] = (x[P,K]^T . dy[P,N])[:k_real].
 * The contraction dimension of the first two (K resp. N) is walked in chunks of 16 floats: K % 16 == 0, and dy carries
 * lddy >= round_up(N, 16) columns with zeros beyond N.  workspace: vp_mm_workspace_bytes(P, K, N) device bytes (the packed
 * copy of w, split-K slabs) whose FIRST 256 BYTES ARE ZERO on entry; the calls never write them.  float32 products and
 * accumulation (v_mfma_f32_16x16x4_f32). */
size_t vp_mm_workspace_bytes(int P, int K, int N);
int vp_mm_fwd_f32(const float* x, int ldx, const float* w, int ldw, int w_transposed, const float* bias, float* y, int ldy,
                  int P, int K, int N, void* workspace, void* stream);
int vp_mm_bwd_data_f32(const float* dy, int lddy, const float* w, int ldw, int w_transposed, float* dx, int lddx, int accumulate,
                       int P, int K, int N, void* workspace, void* stream);
int vp_mm_bwd_weight_f32(const float* x, int ldx, const float* dy, int lddy, float* dw, int P, int K, int k_real, int N,
                         void* workspace, void* stream);
/* The same two weight-consuming products on weights packed AHEAD of time (a training step re-packs every matrix once, in one
 * launch, instead of once per product).  dir: 0 = the layout of vp_mm_fwd_f32, 1 = of vp_mm_bwd_data_f32; the layout follows from
 * (P, K, N, dir) alone.  vp_mm_pack_desc fills one host descriptor (vp_mm_pack_desc_bytes() bytes) of a matrix at master + w_off
 * floats whose packed block goes to packed + dst_off floats (blocks of vp_mm_packed_bytes); vp_mm_pack_table packs n of them from
 * a DEVICE copy of the descriptor array. */
size_t vp_mm_packed_bytes(int P, int K, int N, int dir);
size_t vp_mm_pack_desc_bytes(void);
int vp_mm_pack_desc(size_t w_off, int ldw, int w_transposed, int P, int K, int N, int dir, size_t dst_off, void* desc);
int vp_mm_pack_table(const void* device_descs, int n, const float* master, void* packed, void* stream);
int vp_mm_fwd_f32_packed(const float* x, int ldx, const void* packed_w, const float* bias, float* y, int ldy, int P, int K, int N,
                         void* workspace, void* stream);
int vp_mm_bwd_data_f32_packed(const float* dy, int lddy, const void* packed_w, float* dx, int lddx, int accumulate, int P, int K, int N,
                              void* workspace, void* stream);

/* ---- cv2.resize(uint8, INTER_LINEAR) + paste: the last step of render_face (voicepuppet/pixrefer/infer_bfmvid.py:110-121) ----
 * OpenCV's fixed-point bilinear (11-bit coefficients, the two-pass rounding of resize.cpp: see csrc/resize.hip), byte for byte;
 * optionally behind cv2.cvtColor(BGR2RGB).  src [frames][src_h][src_w][3] uint8 is resized to dst_h x dst_w and written into
 * canvas [frames][canvas_h][canvas_w][3] at (y0, x0); canvas pixels outside the pasted rectangle are not touched (the caller zeroes
 * the canvas, as np.zeros does in the reference).  workspace: vp_resize_paste_workspace_bytes(dst_h, dst_w) device bytes.
 * vp_resize_linear_table: the coefficient tables alone (host arrays of dst_size entries), for host callers and tests. */
size_t vp_resize_paste_workspace_bytes(int dst_h, int dst_w);
int vp_resize_paste_u8(const unsigned char* src, int frames, int src_h, int src_w, int dst_h, int dst_w, int swap_rb,
                       unsigned char* canvas, int canvas_h, int canvas_w, int y0, int x0, void* workspace, void* stream);
int vp_resize_linear_table(int src_size, int dst_size, int rows, int* ofs, short* a0, short* a1, int* row1);

/* Host helper: CRC-32C (Castagnoli, the checksum of TensorFlow checkpoint bundles) of `n` bytes, continuing from `crc` (0 to start). */
unsigned vp_crc32c(const void* data, size_t n, unsigned crc);

#ifdef __cplusplus
}
#endif
#endif /* VP_HIP_H_ */

/* stackrl_qnet.h — C-ABI of libstackrl_qnet.so: the hand-written pieces of the Q-network rollout path.
 *
 * The reference evaluates `DeepQSiamFCN` (stackrl/nets/models.py:106-201) with stock TensorFlow ops.  Plain
 * convolutions stay library calls (MIOpen through PyTorch-ROCm); the two ops below are the ones written by hand:
 *
 *  srl_xcorr_forward  = `layers.correlation` (stackrl/nets/layers.py:21-38): per-sample VALID cross-correlation
 *                       `tf.map_fn(tf.nn.conv2d)` of the left features x[b] (C,H,W) with the right features
 *                       w[b] (C,kh,kw) as the kernel, summed over channels -> out[b] (H-kh+1, W-kw+1).
 *  srl_policy_head    = the epsilon-greedy head of `DQN.policy` (stackrl/agents/dqn.py:334-348) on the advantage
 *                       map: argmax_a Q(s,a) == argmax_a A(s,a) (the dueling mean and value are per-row constants,
 *                       models.py:188-192), ties to the lowest index, then `where(u > eps, argmax, random)`.
 *
 * Plain C, device pointers owned by the caller, `stream` is a hipStream_t as void*; returns 0 on success.
 */
#ifndef STACKRL_QNET_H_
#define STACKRL_QNET_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* x float32 [B][C][H][W], w float32 [B][C][kh][kw] (kw in {16, 32}), out float32 [B][H-kh+1][W-kw+1]; contiguous */
int srl_xcorr_forward(const float* x_dev, const float* w_dev, float* out_dev, int32_t B, int32_t C, int32_t H,
                      int32_t W, int32_t kh, int32_t kw, void* stream);

/* The same op and its two gradients on the matrix cores (csrc/xcorr_mfma.hip), for the forward shapes (H, kh) in
 * {(128, 32), (64, 16)}; O = H - kh + 1.  All tensors contiguous; `in` / `kern` are bfloat16 (in_f32 / kern_f32 = 0)
 * or float32 (= 1); out is float32 (fp32 accumulation).
 *   mode 0  forward : in x [B][C][H][H],            kern w [B][C][kh][kh]            -> out [B][O][O]
 *   mode 1  d/dx    : in [B][O+2(kh-1)]^2 = dOut zero-padded by kh-1 on every side, kern = w flipped in both axes
 *                     [B][C][kh][kh]                                                 -> out [B][C][H][H]
 *   mode 2  d/dw    : in x [B][C][H][H],            kern dOut [B][O][O]              -> out [B][C][kh][kh]
 * precision 0: operands rounded to bf16; 1 ("bf16x3", float32 operands only): hi/lo bf16 split, hi*hi + hi*lo + lo*hi.
 * `scratch` is caller-owned device memory of at least srl_xcorr_mfma_scratch_bytes(...) bytes: the per-workgroup partial
 * outputs of a small-batch forward whose channels are spread over workgroups (0 bytes otherwise: scratch may then be NULL).
 * (Until round 3 it also held the materialised Toeplitz fragments; they are now built in the kernel from LDS.) */
int64_t srl_xcorr_mfma_scratch_bytes(int32_t mode, int32_t precision, int32_t B, int32_t C, int32_t H, int32_t kh);
int srl_xcorr_mfma(int32_t mode, int32_t precision, const void* in_dev, int32_t in_f32, const void* kern_dev,
                   int32_t kern_f32, float* out_dev, void* scratch_dev, int64_t scratch_bytes, int32_t B, int32_t C,
                   int32_t H, int32_t kh, void* stream);
const char* srl_xcorr_mfma_last_error(void);

/* Fused element-wise passes of the U-Nets' inference forward (csrc/epilogue.hip), replacing Conv2D bias + ReLU,
 * MaxPool2D and Concatenate of `layers.unet` (stackrl/nets/layers.py:135-259) around the library convolutions.
 * Tensors are bfloat16, channels-last ([pixel][channel] in memory), C a multiple of 8.
 *   srl_bias_act      : y = act(x + bias[c]); x [npix][C]; y goes to out[pix * out_stride + out_offset + c] (a channel
 *                       slice of a wider channels-last buffer; in place with out = in, out_stride = C, out_offset = 0)
 *                       or, when nchw_hw > 0, to [npix / nchw_hw][C][nchw_hw] (channel-major, for the cross-correlation)
 *   srl_bias_act_pool : x [B][H][W][C] -> skip = relu(x + bias) (channel slice as above) and its 2 x 2 max-pool
 *                       pooled [B][H/2][W/2][C] */
int srl_bias_act(const void* in_dev, void* out_dev, const float* bias_dev, int64_t npix, int32_t C, int32_t out_stride,
                 int32_t out_offset, int32_t nchw_hw, int32_t relu, void* stream);
int srl_bias_act_pool(const void* in_dev, void* skip_dev, void* pooled_dev, const float* bias_dev, int32_t B, int32_t H,
                      int32_t W, int32_t C, int32_t skip_stride, int32_t skip_offset, void* stream);
/* The same two passes on float32 tensors (the fp32 rollout, the reference's dtype). */
int srl_bias_act_f32(const float* in_dev, float* out_dev, const float* bias_dev, int64_t npix, int32_t C,
                     int32_t out_stride, int32_t out_offset, int32_t nchw_hw, int32_t relu, void* stream);
int srl_bias_act_pool_f32(const float* in_dev, float* skip_dev, float* pooled_dev, const float* bias_dev, int32_t B,
                          int32_t H, int32_t W, int32_t C, int32_t skip_stride, int32_t skip_offset, void* stream);
/* Backward of the float32 bias + activation pass for the update path (agents/dqn.py:466-469): gx = gy * (y > 0) (relu)
 * or gy, and gbias[c] = sum over pixels of gx[.][c], summed in a fixed order.  gy, y (the activation's output), gx:
 * float32 channels-last [npix][C]; C a multiple of 8 that divides 2,048, at most 256; scratch:
 * srl_bias_act_bwd_scratch_floats(npix, C) floats. */
int64_t srl_bias_act_bwd_scratch_floats(int64_t npix, int32_t C);
int srl_bias_act_bwd_f32(const float* gy_dev, const float* y_dev, float* gx_dev, float* gbias_dev, float* scratch_dev,
                         int64_t npix, int32_t C, int32_t relu, void* stream);
/* 2 x 2 max-pool (`MaxPool2D` of layers.unet) of a finished activation that may be a channel slice of a wider
 * channels-last buffer (pixel stride in_stride, channel offset in_offset) into a contiguous [B][H/2][W/2][C] tensor;
 * f32 = 0: bfloat16, 1: float32. */
int srl_pool2x2(const void* in_dev, void* pooled_dev, int32_t B, int32_t H, int32_t W, int32_t C, int32_t in_stride,
                int32_t in_offset, int32_t f32, void* stream);
const char* srl_epilogue_last_error(void);

/* 3 x 3 convolution (stride 1, SAME) + bias + ReLU on the matrix cores for the thin, wide layers of `layers.unet`
 * (csrc/conv_mfma.hip; inference).  in bfloat16 channels-last [B][H][W][cin], H and W multiples of 16, cin in {16, 32, 64}, cout in
 * {16, 32}; wfrag = the weights packed in MFMA A-fragment order (srl_conv3x3_wfrag_elems(cin, cout) bfloat16 elements:
 * [k-step][cout / 16][lane][8], element = w[16 mt + lane % 16][ci][tap / 3][tap % 3] with, for cin = 16,
 * tap = 2 ks + k / 16, ci = k % 16 (tap 9 = zero) and for cin = 32 m, tap = ks / m, ci = 32 (ks % m) + k, where k = 8 (lane / 16) + j).
 * out: channel slice [out_offset, out_offset + cout) of a channels-last buffer with out_stride channels per pixel, or
 * (nchw = 1) a [B][cout][H][W] tensor; pooled (may be NULL): the 2 x 2 max-pooled result [B][H/2][W/2][cout]. */
int32_t srl_conv3x3_wfrag_elems(int32_t cin, int32_t cout);
int srl_conv3x3_bias_relu(const void* in_dev, const void* wfrag_dev, const float* bias_dev, void* out_dev, void* pooled_dev,
                          int32_t B, int32_t H, int32_t W, int32_t cin, int32_t cout, int32_t out_stride,
                          int32_t out_offset, int32_t nchw, void* stream);
/* The same layer in fp32-class precision (the fp32 rollout, the reference's dtype): float32 channels-last in and out,
 * cin in {16, 32, 64}, cout in {16, 32}; every product is hi hi + hi lo + lo hi of bfloat16 halves on the matrix cores, fp32 accumulation
 * ("bf16x3").  wfrag: 2 x srl_conv3x3_wfrag_elems(cin, cout) bfloat16 elements — the fragments of bf16(w), then those of
 * bf16(w - bf16(w)), each in the order described above.  pooled (may be NULL) float32 [B][H/2][W/2][cout]. */
int srl_conv3x3_bias_relu_f32(const float* in_dev, const void* wfrag_dev, const float* bias_dev, float* out_dev,
                              float* pooled_dev, int32_t B, int32_t H, int32_t W, int32_t cin, int32_t cout,
                              int32_t out_stride, int32_t out_offset, int32_t nchw, void* stream);
/* The deep levels of `layers.unet` (64 / 128 / 256 output channels at 32^2 / 16^2 / 8^2) as an implicit GEMM on the
 * matrix cores (csrc/conv_gemm.hip; inference): in channels-last [B][W][W][cin], out: the channel slice
 * [out_offset, out_offset + cout) of a channels-last buffer with out_stride channels per pixel; f32 = 0: bfloat16 tensors;
 * f32 = 1: float32 tensors, fp32-class products (bf16x3), wfrag then holds the hi fragments followed by the lo fragments.
 * Layers: srl_conv3x3_gemm_supported(cin, cout, W).  wfrag: srl_conv3x3_gemm_wfrag_elems(cin, cout) bfloat16 elements
 * [cin / 32][tap][cout / 16][lane][8], element = w[16 mt + lane % 16][32 cb + 8 (lane / 16) + j][tap / 3][tap % 3]. */
int32_t srl_conv3x3_gemm_supported(int32_t cin, int32_t cout, int32_t W);
int64_t srl_conv3x3_gemm_wfrag_elems(int32_t cin, int32_t cout);
int32_t srl_conv3x3_gemm_batch_multiple(int32_t cout, int32_t W);   /* maps per workgroup: B must be a multiple of it */
int srl_conv3x3_gemm_bias_relu(const void* in_dev, const void* wfrag_dev, const float* bias_dev, void* out_dev, int32_t B,
                               int32_t W, int32_t cin, int32_t cout, int32_t out_stride, int32_t out_offset, int32_t f32,
                               void* stream);
const char* srl_conv_gemm_last_error(void);
/* The transposed convolutions of the deep levels (128 -> 64, 256 -> 128; any map size) as a GEMM on the matrix cores
 * (csrc/conv_gemm.hip): in channels-last [B][H][W][cin] -> the channel slice [out_offset, out_offset + cout) of a
 * channels-last buffer [B][2H][2W][out_stride]; f32 = 0 bfloat16, 1 float32 (fp32-class, wfrag = hi set then lo set).
 * wfrag: the order of srl_convt2x2_wfrag_elems. */
int32_t srl_convt2x2_gemm_supported(int32_t cin, int32_t cout);
int srl_convt2x2_gemm_bias_relu(const void* in_dev, const void* wfrag_dev, const float* bias_dev, void* out_dev, int32_t B,
                                int32_t H, int32_t W, int32_t cin, int32_t cout, int32_t out_stride, int32_t out_offset,
                                int32_t f32, void* stream);
/* The thin first layers (1 or 2 input channels -> 16) on the vector ALU: in uint8 (in_dtype 0: the env's observation
 * bytes, scaled by 1/255 as in models.py:144-147) or float32 (in_dtype 1) channels-last [B][H][W][cin]; w float32
 * [16][3][3][cin] (the Conv2d weight with the input channel innermost: a pixel's channels pair up with adjacent weights),
 * bias float32 [16]; out bfloat16 channels-last [B][Hp][Wp][16] with Hp >= H, Wp >= W (only the H x W
 * region is written: a zero-initialised margin stays zero). */
int srl_conv3x3_thin(const void* in_dev, int32_t in_dtype, const float* w_dev, const float* bias_dev, void* out_dev,
                     int32_t B, int32_t H, int32_t W, int32_t cin, int32_t Hp, int32_t Wp, void* stream);
/* The same with float32 output [B][Hp][Wp][16] (the fp32 rollout; the arithmetic is fp32 in both). */
int srl_conv3x3_thin_f32(const void* in_dev, int32_t in_dtype, const float* w_dev, const float* bias_dev, float* out_dev,
                         int32_t B, int32_t H, int32_t W, int32_t cin, int32_t Hp, int32_t Wp, void* stream);
/* conv3x3 16 -> 16 + bias + ReLU followed by the 1 x 1 convolution to one channel, in one kernel (the tail of
 * `pos_layers`, layers.py:439-472): in bfloat16 [B][H][W][16] (H, W multiples of 16), out float32 [B][Hv][Wv]. */
int srl_conv3x3_relu_project(const void* in_dev, const void* wfrag_dev, const float* bias_dev, const float* proj_w_dev,
                             float proj_b, float* out_dev, int32_t B, int32_t H, int32_t W, int32_t Hv, int32_t Wv,
                             void* stream);
/* The same in fp32-class precision: in float32 [B][H][W][16], wfrag as for srl_conv3x3_bias_relu_f32 (16 -> 16). */
int srl_conv3x3_relu_project_f32(const float* in_dev, const void* wfrag_dev, const float* bias_dev, const float* proj_w_dev,
                                 float proj_b, float* out_dev, int32_t B, int32_t H, int32_t W, int32_t Hv, int32_t Wv,
                                 void* stream);
/* The thin layer and the 16 -> 16 layer behind it in one kernel, fp32-class (`convdw00`, `convdw01` [+ `down0`] of layers.unet,
 * layers.py:196-209: Conv2D(16, 3, relu) twice [+ MaxPool2D]): srl_conv3x3_thin_f32 followed by srl_conv3x3_bias_relu_f32 without the
 * 16-channel intermediate's round trip through HBM; results equal the two calls' bit for bit.  in [B][H][W][cin] uint8
 * (in_dtype 0, scaled by 1/255) or float32 (1), cin in {1, 2}, H and W multiples of 16; w1 [16][3][3][cin] (as for srl_conv3x3_thin), b1 [16] float32;
 * wfrag / bias / out / pooled / out_stride / out_offset / nchw as for srl_conv3x3_bias_relu_f32 with cin = cout = 16. */
int srl_thin_conv3x3_bias_relu_f32(const void* in_dev, int32_t in_dtype, int32_t cin, const float* w1_dev, const float* b1_dev,
                                   const void* wfrag_dev, const float* bias_dev, float* out_dev, float* pooled_dev, int32_t B,
                                   int32_t H, int32_t W, int32_t out_stride, int32_t out_offset, int32_t nchw, void* stream);
/* `pos_layers` whole (layers.py:439-472: Conv2D(16, 3, relu) twice, Conv2D(1, 1)) in one kernel, fp32-class:
 * srl_conv3x3_thin_f32 into a zero-margined map followed by srl_conv3x3_relu_project_f32, bit for bit.
 * in float32 [B][H][W] (any H, W >= 1), out float32 [B][H][W]. */
int srl_thin_conv3x3_relu_project_f32(const float* in_dev, const float* w1_dev, const float* b1_dev, const void* wfrag_dev,
                                      const float* bias_dev, const float* proj_w_dev, float proj_b, float* out_dev, int32_t B,
                                      int32_t H, int32_t W, void* stream);
/* Transposed convolution 2 x 2, stride 2 + bias + ReLU on the matrix cores (`up{i}` of layers.unet, layers.py:222-229),
 * (cin, cout) in {(32, 16), (64, 32)}: in bfloat16 [B][H][W][cin] (W a multiple of 16) -> the channel slice
 * [out_offset, out_offset + cout) of a channels-last buffer [B][2H][2W][out_stride].  wfrag: the ConvTranspose2d weight
 * w[ci][co][dy][dx] in A-fragment order, srl_convt2x2_wfrag_elems(cin, cout) bfloat16 elements:
 * [k-step][m-tile][lane][8], element = w[32 ks + 8 (lane / 16) + j][co][dy][dx] for GEMM row 16 mt + lane % 16 =
 * (2 dy + dx) cout + co. */
int32_t srl_convt2x2_wfrag_elems(int32_t cin, int32_t cout);
int srl_convt2x2_bias_relu(const void* in_dev, const void* wfrag_dev, const float* bias_dev, void* out_dev, int32_t B,
                           int32_t H, int32_t W, int32_t cin, int32_t cout, int32_t out_stride, int32_t out_offset,
                           void* stream);
/* The same in fp32-class precision (bf16x3 products): float32 in and out; wfrag: 2 x srl_convt2x2_wfrag_elems(cin, cout)
 * bfloat16 elements, the fragments of bf16(w), then those of bf16(w - bf16(w)). */
int srl_convt2x2_bias_relu_f32(const float* in_dev, const void* wfrag_dev, const float* bias_dev, float* out_dev, int32_t B,
                               int32_t H, int32_t W, int32_t cin, int32_t cout, int32_t out_stride, int32_t out_offset,
                               void* stream);
const char* srl_conv_last_error(void);

/* adv float32 [B][A]; u float32 [B] uniform(0,1); rnd int64 [B] uniform {0..A-1}; actions int64 [B] */
int srl_policy_head(const float* adv_dev, const float* u_dev, const int64_t* rnd_dev, float epsilon,
                    int64_t* actions_dev, int32_t B, int32_t A, void* stream);

/* Heuristic baseline policies (stackrl/baselines.py), the reference's yardstick and optional initial-collect policy
 * (training.py:256-263).  method: 1 correlate (:141-143), 2 height (:28-43), 3 difference (:45-77), 4 corrcoef
 * (:79-114).  obs_map uint8 [B][H][H][2], obs_obj uint8 [B][h][h][1] as the env returns them; values float64
 * [B][(H-h+1)^2]; mask (may be NULL) uint8 [B][(H-h+1)^2] = goal_overlap (:152-156) with `threshold`. */
int srl_heuristic(int32_t method, const uint8_t* obs_map_dev, const uint8_t* obs_obj_dev, double* values_dev,
                  uint8_t* mask_dev, int32_t B, int32_t H, int32_t h, int32_t difference_exponent,
                  int32_t weights_exponent, int32_t localized, double threshold, void* stream);

/* `Baseline.call` (baselines.py:201-217): arg-min of the values over the goal mask (if use_goal), restricted to local
 * minima of a (1 + 2 minorder)^2 window when any exist; actions int64 [B]; neg_values (may be NULL) float64 [B][A] is
 * the second return value (-values outside the mask replaced by -(max masked value + 0.001)). */
int srl_baseline_select(const double* values_dev, const uint8_t* mask_dev, int32_t use_goal, int32_t minorder,
                        int64_t* actions_dev, double* neg_values_dev, int32_t B, int32_t OH, void* stream);

const char* srl_qnet_last_error(void);
/* "SRL_BUILD_INFO<variant|hash of the sources and flags>" (stackrl_amd/build.py) */
const char* srl_qnet_build_info(void);

/* ---- update path (csrc/learner.hip).  Device pointers, contiguous, `stream` a hipStream_t; nothing is allocated, so a
 * captured hipGraph can replay every call.
 *
 * srl_td_epilogue = the loss of `DQN.train` (stackrl/agents/dqn.py:408-469) and its gradient in one pass over the
 * minibatch: y = r [x reward_scale] + (terminal ? 0 : gamma Q_target(s', a*)), a* = argmax_a Q_online(s', a)
 * (use_double, ties to the lowest index) or argmax_a Q_target(s', a); td = Q(s, action) - y; Huber with `huber_delta`
 * (< 0: plain 0.5 td^2) times the importance weight (weights may be NULL); outputs: loss = mean, mtd = mean td,
 * td_abs[mb], logits[mb] = log(|td| + prio_eps) (the new replay priorities, memory.py:272), and grad_q[mb][A] (may be
 * NULL) = d loss / d Q(s, .): zero except at the taken action.  scratch: 2 mb floats; ticket: one int32, zero before the
 * first call (the kernel leaves it zero). */
int srl_td_epilogue(const float* q_dev, const float* q_next_online_dev, const float* q_next_target_dev,
                    const int64_t* actions_dev, const float* rewards_dev, const uint8_t* terminal_dev,
                    const float* weights_dev, float gamma, float huber_delta, float reward_scale, int32_t use_double,
                    float prio_eps, int32_t mb, int32_t A, float* loss_dev, float* mean_td_dev, float* td_abs_dev,
                    float* logits_dev, float* grad_q_dev, float* scratch_dev, int32_t* ticket_dev, void* stream);

/* Keras Adam (`optimizer.apply_gradients`, dqn.py:473; config.gin:90-93) over one flat fp32 bucket of n elements
 * (16-byte aligned): m += (g - m)(1 - beta1); v += (g^2 - v)(1 - beta2); p -= lr_t m / (sqrt(v) + eps) with
 * lr_t = lr sqrt(1 - beta2^t) / (1 - beta1^t).  state: 4 device floats {t, beta1^t, beta2^t, lr_t}, {0, 1, 1, 0} before
 * the first step, advanced on the device by every call. */
int srl_adam_step(float* params_dev, const float* grads_dev, float* m_dev, float* v_dev, int64_t n, float* state_dev,
                  float lr, float beta1, float beta2, float eps, void* stream);

/* K7 — prioritised sampling without replacement (`ReplayMemory.sample`, memory.py:220-223): the k largest of
 * key_i = alpha logit_i - log(-log(u_i)) over n slots (logit = -inf: not sampleable), indices in descending key order,
 * the lower index first among equal keys; out_key = the keys (-inf where fewer than k slots were sampleable).
 * alpha is read from device memory.  scratch: srl_gumbel_topk_scratch_bytes(n, k) bytes. */
int64_t srl_gumbel_topk_scratch_bytes(int64_t n, int32_t k);
int srl_gumbel_topk(const float* logits_dev, const float* u_dev, const float* alpha_dev, int64_t n, int32_t k,
                    int64_t* out_idx_dev, float* out_key_dev, void* scratch_dev, int64_t scratch_bytes, void* stream);

/* K8 — `ReplayMemory.add` (memory.py:153-161): transition b of the B collected ones goes to row b part_len + slot of
 * the two state tensors (rows of bytes0 / bytes1 bytes, multiples of 16) and of reward / terminal / action; its logit
 * becomes -inf (not sampleable until its successor exists). */
int srl_replay_scatter(const uint8_t* state0_dev, const uint8_t* state1_dev, int64_t bytes0, int64_t bytes1,
                       const float* reward_dev, const uint8_t* terminal_dev, const int64_t* action_dev, int32_t B,
                       int64_t slot, int64_t part_len, uint8_t* mem0_dev, uint8_t* mem1_dev, float* mem_reward_dev,
                       uint8_t* mem_terminal_dev, int64_t* mem_action_dev, float* mem_logits_dev, void* stream);
/* K8 — the minibatch of `ReplayMemory.sample` (memory.py:232-260) for mb sampled rows: state and next state (the row
 * n_steps on inside the partition of part_len rows; literal_next = 1: the formula of memory.py:239-242 as written),
 * action, reward and terminal flag of the next row, and (weight_dev != NULL) the importance weight
 * exp(beta alpha (min_logit - logit)); alpha, beta, min_logit are device scalars; next_dev (may be NULL) receives the
 * next-row indices. */
int srl_replay_gather(const int64_t* idx_dev, int32_t mb, int64_t part_len, int64_t n_steps, int32_t literal_next,
                      int64_t* next_dev, const uint8_t* mem0_dev, const uint8_t* mem1_dev, int64_t bytes0, int64_t bytes1,
                      const float* mem_reward_dev, const uint8_t* mem_terminal_dev, const int64_t* mem_action_dev,
                      const float* mem_logits_dev, const float* alpha_dev, const float* beta_dev,
                      const float* min_logit_dev, uint8_t* state0_dev, uint8_t* state1_dev, uint8_t* next0_dev,
                      uint8_t* next1_dev, int64_t* action_dev, float* reward_dev, uint8_t* terminal_dev,
                      float* weight_dev, void* stream);
/* Priority trackers of `ReplayMemory` (memory.py:164-177, :282-316): out_v2 = {max logit, min finite logit (+inf if none)},
 * out_i2 = {lowest index of the max, lowest index of the min (0 if none)} over logits[n].  Two launches with per-block
 * partials in `scratch` (srl_logit_extrema_scratch_bytes): no cross-workgroup reduction inside a kernel. */
int64_t srl_logit_extrema_scratch_bytes(void);
int srl_logit_extrema(const float* logits_dev, int64_t n, float* out_v2_dev, int64_t* out_i2_dev, void* scratch_dev, void* stream);
const char* srl_learner_last_error(void);

/* ---- update-path convolutions (csrc/train_conv.hip): the forward (with saved activations), data-gradient and
 * weight-gradient convolutions of `DQN.train` (stackrl/agents/dqn.py:466-473 differentiates `DeepQSiamFCN`,
 * stackrl/nets/models.py:106-201 / layers.py:135-259) in true float32 on the matrix cores (v_mfma_f32_16x16x4_f32).
 * Tensors are float32, channels-last: pixel p of [B][H][W] holds its channels at p * stride + offset (a channel slice of a
 * wider buffer, e.g. the decoder's concatenation buffer).  Nothing is allocated; results are bit-identical on repetition
 * (no atomics: fixed-order partial sums).
 *
 * srl_tconv: y[p][co] = act(bias[co] + sum_{t, ci} x[p + d(t)][ci] wp[t][ci][co]); taps = 9: 3 x 3, stride 1, SAME;
 * taps = 1: 1 x 1.  wp = packed weights [taps][cin_p][cout], cin_p = cin rounded up to 4 (srl_trepack); cout a multiple of
 * 16; bias may be NULL.  The data gradient of a 3 x 3 layer is the same call on the flipped / transposed packing (kind 1).
 * d2s > 0 (taps = 1, cout = 4 d2s): the 2 x 2 stride-2 transposed convolution — output channel q d2s + co is stored at
 * pixel (2 y + q / 2, 2 x + q % 2), channel co of y [B][2H][2W], bias indexed by co. */
int srl_tconv(const float* x_dev, int32_t x_stride, int32_t x_off, const float* wp_dev, const float* bias_dev, float* y_dev,
              int32_t y_stride, int32_t y_off, int32_t B, int32_t H, int32_t W, int32_t cin, int32_t cout, int32_t taps,
              int32_t relu, int32_t d2s, void* stream);
/* srl_twrw: gw = sum_p x[p + d(t)][ci] gz[p][co] in the framework's parameter layout — Conv2d [cout][cin][3][3] (or 1 x 1),
 * or (convt = 1, taps = 1, cout = 4 cout_t, gz stored space-to-depth) ConvTranspose2d [cin][cout_t][2][2].  gz contiguous
 * [B][H][W][cout]; scratch: srl_twrw_scratch_floats(...) floats.  bias_partial (may be NULL): the bias-gradient partials
 * srl_tact_bwd left in ITS scratch (bias_nblk = srl_tact_bwd_blocks(...) rows of bias_C floats); the finishing launch then
 * also writes gbias[bias_C] — one launch per layer for both gradients. */
int64_t srl_twrw_scratch_floats(int32_t B, int32_t H, int32_t W, int32_t cin, int32_t cout, int32_t taps);
int srl_twrw(const float* x_dev, int32_t x_stride, int32_t x_off, const float* gz_dev, float* gw_dev, float* scratch_dev,
             int32_t B, int32_t H, int32_t W, int32_t cin, int32_t cout, int32_t taps, int32_t convt,
             const float* bias_partial_dev, int32_t bias_nblk, int32_t bias_C, float* gbias_dev, void* stream);
/* srl_tact_bwd: gz[p][c] = (g[p][c] + the gradient gpool[B][H/2][W/2][C] routed back through the 2 x 2 max-pool of y to the
 * first maximal element of each window, if gpool != NULL) * [y[p][c] > 0] (relu), contiguous [B][H][W][C] or (s2d = 1)
 * space-to-depth [B][H/2][W/2][4 C] with channel (2 (y % 2) + x % 2) C + c; gbias (may be NULL) = sum over pixels of gz in
 * a fixed order.  C a multiple of 4 with C / 4 dividing 256, C <= 256; scratch (may be NULL: no bias gradient):
 * srl_tact_bwd_scratch_floats(B H W, C) floats = srl_tact_bwd_blocks(B H W, C) rows of per-block partial sums, always
 * written when given; gbias NULL leaves finishing them to srl_twrw. */
int64_t srl_tact_bwd_scratch_floats(int64_t npix, int32_t C);
int32_t srl_tact_bwd_blocks(int64_t npix, int32_t C);
int srl_tact_bwd(const float* g_dev, int32_t g_stride, int32_t g_off, const float* y_dev, int32_t y_stride, int32_t y_off,
                 const float* gpool_dev, float* gz_dev, float* gbias_dev, float* scratch_dev, int32_t B, int32_t H, int32_t W,
                 int32_t C, int32_t relu, int32_t s2d, void* stream);
/* srl_trepack: every packed weight layout of every layer from the flat parameter bucket in one launch.  desc: int64
 * [nlayers][8] = {src_off, dst_off, cin, cout, taps, kind, cin_pad, 0}, destination ranges consecutive from 0 to `total`;
 * kind 0 conv forward [t][cin_p][cout]; 1 conv data gradient [t][cout][cin_pad] (taps flipped; cin_pad = cin, or cin rounded
 * up to 16 with zero columns so that a thin layer's data gradient has a supported channel count); 2 transposed conv forward
 * [cin][4 cout] (k = q cout + co); 3 transposed conv data gradient [4 cout][cin]. */
int srl_trepack(const float* flat_dev, float* packed_dev, const int64_t* desc_dev, int32_t nlayers, int64_t total, void* stream);
/* The dueling head of `DeepQSiamFCN` in the update (models.py:179-192; `tf.GradientTape` through it, agents/dqn.py:466-469):
 * z [B][A][16] the last position map (channels last, A = O x O), pw[16] / pb[1] the 1 x 1 projection, v[B] the state value:
 *   srl_thead_fwd:  q[b][p] = (z[b][p][:] . pw + pb) - mean_p(...) + v[b]
 *   srl_thead_bwd:  for the first n samples, from gq[n][A]: gv[b] = sum_p gq, ga = gq - mean_p gq, gz[b][p][c] = ga pw[c],
 *                   gpw[c] = sum z ga, gpb = sum ga; scratch: 17 n floats.
 * One workgroup per sample and fixed-order sums: no cross-workgroup reduction inside a kernel, no atomics (the framework's
 * multi-block reductions returned garbage under the concurrent env step). */
int srl_thead_fwd(const float* z_dev, const float* pw_dev, const float* pb_dev, const float* v_dev, float* q_dev, int32_t B,
                  int32_t A, void* stream);
int srl_thead_bwd(const float* z_dev, const float* pw_dev, const float* gq_dev, float* gz_dev, float* gv_dev, float* gpw_dev,
                  float* gpb_dev, float* scratch_dev, int32_t n, int32_t A, void* stream);
/* The value branch of the dueling head (`layers.value`, layers.py:424-436; models.py:179-186): global average pool of the
 * bottom features x0 [B][P][C] (channels last, P = h x w), Dense(U) + ReLU (W1 [U][C], b1 [U]), Dense(1) (W2 [U], b2 [1]):
 *   srl_tvalue_fwd:  v[b]; pooled [B][C] and h [B][U] (may be NULL) are kept for the backward
 *   srl_tvalue_bwd:  for the first n samples, from gv[n]: gW1, gb1, gW2, gb2 (written, not accumulated) and
 *                    gx[b][p][c] = gin[b][p][c] + (d v / d pooled)[b][c] / P — the gradient wrt x0, added to the one that
 *                    arrives through the decoder (gin); scratch: n U floats.
 * One workgroup per sample / hidden unit, every sum in index order, no atomics; replaces the framework's GEMM, bias, ReLU and
 * reduction kernels of rounds 1-3. */
int srl_tvalue_fwd(const float* x0_dev, const float* W1_dev, const float* b1_dev, const float* W2_dev, const float* b2_dev,
                   float* pooled_dev, float* h_dev, float* v_dev, int32_t B, int32_t P, int32_t C, int32_t U, void* stream);
int srl_tvalue_bwd(const float* gv_dev, const float* h_dev, const float* pooled_dev, const float* W1_dev, const float* W2_dev,
                   const float* gin_dev, float* gx_dev, float* gW1_dev, float* gb1_dev, float* gW2_dev, float* gb2_dev,
                   float* scratch_dev, int32_t n, int32_t P, int32_t C, int32_t U, void* stream);
/* Layout passes around the cross-correlation (`layers.correlation`, layers.py:21-38: its kernels read channel-major maps, the
 * U-Nets of the update work channels-last): srl_tlayout copies a channels-last activation (pixel stride / channel offset) to
 * [B][C][HW] (to_nhwc = 0) or a channel-major tensor to [B][HW][C] (to_nhwc = 1); srl_tcorr_grad takes channel 0 of a
 * channels-last gradient [n][O][O][g_stride] as the plain map [n][O][O] and as its zero-padded copy [n][O + 2 pad][O + 2 pad];
 * srl_tflip reverses each of `planes` planes of hw elements (the flipped kernels of the data gradient); srl_tu8_to_f32 is the
 * network's input scaling uint8 -> float32 / 255 (models.py:144-147; n a multiple of 4). */
int srl_tlayout(const float* src_dev, int32_t src_stride, int32_t src_off, float* dst_dev, int32_t B, int32_t HW, int32_t C,
                int32_t to_nhwc, void* stream);
int srl_tcorr_grad(const float* g_dev, int32_t g_stride, float* plain_dev, float* padded_dev, int32_t n, int32_t O, int32_t pad,
                   void* stream);
int srl_tflip(const float* in_dev, float* out_dev, int64_t planes, int32_t hw, void* stream);
int srl_tu8_to_f32(const uint8_t* in_dev, float* out_dev, int64_t n, void* stream);
const char* srl_train_conv_last_error(void);

#ifdef __cplusplus
}
#endif
#endif

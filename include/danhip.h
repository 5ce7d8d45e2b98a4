/* libdanhip — C ABI of the MI355X-native detector hot path (drop-in boundary).
 *
 * What it replaces in the reference (HiKapok/DAN; paths relative to the reference checkout):
 *   - the TF custom-op plugin loaded by utility/custom_op.py:33-63 (tf.load_op_library of
 *     cpp/Deform/build/libdeform.so and cpp/ExtraLib/build/libextra_lib.so):
 *       DeformConvOp / DeformConvBackpropOp   cpp/Deform/deform_conv.cc:51,170
 *       SmallMiningMatch                      cpp/ExtraLib/small_mining_match.cc:31
 *       DynamicAnchorRouting                  cpp/ExtraLib/dynamic_anchor_routing.cc:32
 *   - the TF built-in kernels the graphs in net/<model>.py call (conv2d, pools, resize_bilinear, softmax-CE,
 *     top_k, non_max_suppression, Momentum) whose source is not in the reference tree.
 *
 * Conventions (all entry points):
 *   - extern "C"; returns 0 (DANHIP_OK) or a negative DANHIP_E* code; never throws / aborts;
 *     danhip_last_error() returns a thread-local message for the last failure.
 *   - every tensor pointer is a caller-owned DEVICE pointer (16-byte aligned); nothing is allocated
 *     inside; scratch comes from a caller-supplied workspace where a *_workspace_bytes() query exists.
 *   - every call takes a hipStream_t (passed as void*) and is asynchronous on it; no host sync.
 *   - re-entrant and thread-safe.  The ONE piece of process-wide mutable state is the table of kernel-FORM switches behind
 *     danhip_set_option (below): filled from the environment exactly once, every value an atomic, read once per call by a launcher;
 *     whatever it says, results agree up to fp32 summation order.  (Bound once and read-only afterwards: the RCCL entry points of
 *     danhip_comm_*.)  Nothing else outlives a call.
 *   - activations are NHWC, 16-bit, passed as raw uint16.  The 16-bit type is a property of the library build:
 *     libdanhip.so = bf16 (default; every "bf16" below), libdanhip_f16.so = IEEE fp16 (same entry points, same layouts,
 *     v_mfma_f32_16x16x32_f16; BASELINE.json configs[4] "fp16 + MFMA").  danhip_act_dtype() tells which one is loaded;
 *     wherever an out_dtype argument says DANHIP_BF16 it means "the build's 16-bit type" (DANHIP_F16 is accepted as an
 *     alias in the fp16 build).  Accumulation, master weights, gradients of weights and all box / loss arithmetic are fp32.
 */
#ifndef DANHIP_H_
#define DANHIP_H_
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DANHIP_OK 0
#define DANHIP_EINVAL (-1)   /* bad argument / unsupported shape            */
#define DANHIP_ELAUNCH (-2)  /* kernel launch failed                        */
#define DANHIP_EWORKSPACE (-3) /* workspace too small                        */

#define DANHIP_F32 0
#define DANHIP_BF16 1
#define DANHIP_F16 2
#define DANHIP_SPLIT3 3      /* out_dtype of danhip_conv2d_fwd[_ws] in the fp16 build: y = the [hi | lo | hi] half-limb map [N,Ho,Wo,3*Cout] (split_infer.hip) */

const char* danhip_last_error(void);
int danhip_version(void);
/* Process-wide kernel-FORM switches (A/B and test hooks; defaults from the environment variables in brackets, read once; values are
 * atomics and a launcher reads a switch once per call):
 *   "splitk"     [DANHIP_SPLITK, 1]      0: never split K (danhip_conv2d_workspace_bytes answers 0)
 *   "wgrad_slab" [DANHIP_WGRAD_SLAB, 1]  0: weight-gradient partial sums always by fp32 atomics; 2: always stores + combine pass; 1: by launch length
 *   "halo_b2"    [DANHIP_HALO_B2, 0]     1: a second workgroup barrier per step in csrc/conv_halo.hip (the round-2 form; timing A/B only)
 *   "wgrad_b2"   [DANHIP_WGRAD_B2, 0]    1: the same for csrc/conv_wgrad_rows.hip and csrc/conv_wgrad_pw.hip
 *   "deform_bwd_form" [DANHIP_DEFORM_BWD_FORM, 0]  deformable backward: 0 form by the offsets' statistics (on the device), 1 always the
 *                                           gather form with a +-2 px window, 2 always the fp32-atomics scatter form, 3 gather form, +-1 px
 *   "deform_dx_untiled" [DANHIP_DEFORM_DX_UNTILED, 0]  1: the +-1 px gather of the deformable backward as the wave-per-input-pixel kernel instead
 *                                           of the LDS-staged tile kernel (same candidates, weights and summation order)
 *   "halo_general_epilogue" [DANHIP_HALO_GENERAL_EPILOGUE, 0]  1: csrc/conv_halo.hip always takes its general epilogue (A/B of the lean one)
 *   "wgrad_c8"   [DANHIP_WGRAD_C8, 1]    0: the first layer's weight gradient (3x3, 8-channel image, 64 outputs) on the general kernel instead of
 *                                           csrc/conv_wgrad_c8.hip
 *   "pw_dgrad_ld_bn" [DANHIP_PW_DGRAD_LD_BN, 128]  256: csrc/conv_pointwise.hip's data gradient with a ReLU mask / accumulation may take 256-wide tiles
 * Results agree up to fp32 summation order whatever the setting.  danhip_set_option returns DANHIP_EINVAL for an unknown name. */
int danhip_set_option(const char* name, int value);
int danhip_get_option(const char* name);
int danhip_act_dtype(void);   /* DANHIP_BF16 or DANHIP_F16 */

/* ------------------------------------------------------------------------------------------------
 * Dense convolution (tf.layers.conv2d, padding='same'; net/sfd_net.py:81-89 conv_relu and every
 * tf.layers.conv2d call in net/<model>.py).  bf16 NHWC activations, fp32 accumulate on MFMA.
 *
 * Weight packing: the TF kernel variable is HWIO fp32 [kh,kw,Cin,Cout].  The forward kernel consumes
 * wf = bf16 [Cout_pad][Kpad] with k = (i*kw+j)*Cin_pad + c (K-contiguous per output channel); the data
 * gradient consumes wb = bf16 [Cin_pad][Kpad_b] with k = (i*kw+j)*Cout_pad8 + co.  Both are produced by
 * danhip_pack_conv_weight (sizes via danhip_conv_packed_elems).
 * ------------------------------------------------------------------------------------------------ */
typedef struct {
  int32_t N, H, W, Cin;      /* input  [N,H,W,Cin]   (Cin multiple of 8: pad the tensor otherwise)     */
  int32_t Ho, Wo, Cout;      /* output [N,Ho,Wo,Cout]                                                  */
  int32_t kh, kw, stride;    /* Ho/Wo = ceil(in/s): TF 'same' (pad_before = max((Ho-1)*s+kh-H,0)/2 derived);
                                Ho/Wo = floor((in-k)/s)+1: 'valid' (no padding)                          */
} danhip_conv_desc;

/* rows/cols of the packed forward (which=0) or data-gradient (which=1) weight matrix */
int danhip_conv_packed_dims(const danhip_conv_desc* d, int which, int64_t* rows, int64_t* cols);
/* w_hwio fp32 [kh,kw,CinReal,Cout] (CinReal <= d->Cin; extra input channels get zero weights) */
int danhip_pack_conv_weight(const danhip_conv_desc* d, const float* w_hwio, int32_t cin_real,
                            uint16_t* wf_packed, uint16_t* wb_packed, void* stream);

/* All conv weights of a model packed by ONE launch (after each optimizer update).  The caller fills a host array of
 * entries with danhip_pack_entry_init (which also returns how many workgroups the entry gets; first_block is the running
 * sum), copies it to the device and passes it here with the total.  Same packing as danhip_pack_conv_weight. */
typedef struct {
  const float* w_hwio;
  uint16_t* wf_packed;
  uint16_t* wb_packed;       /* NULL: forward packing only */
  int32_t kh, kw, cin, cin_real, cout, rows_f, cols_f, rows_b, cols_b, co8, first_block, pad_;
} danhip_pack_entry;
int danhip_pack_entry_init(danhip_pack_entry* e, const danhip_conv_desc* d, const float* w_hwio, int32_t cin_real,
                           uint16_t* wf_packed, uint16_t* wb_packed, int32_t first_block, int32_t* blocks);
int danhip_pack_conv_weights_batched(const danhip_pack_entry* table_dev, int32_t n, int32_t total_blocks, void* stream);

/* y = act(conv(x, w) + bias).  x bf16; y bf16 (out_dtype=DANHIP_BF16) or fp32; bias fp32[Cout] or NULL.
 * Pointwise (1x1, stride 1) convolutions with channel counts that are multiples of 64 run on the streaming GEMM kernel
 * (csrc/conv_pointwise.hip; bias / ReLU, or mask / accumulate for danhip_conv2d_bwd_data, in its epilogue).  No vendor GEMM or
 * convolution library is linked.
 * relu: 0/1.  residual: optional bf16 tensor of y's shape added AFTER the activation (DAN context modules,
 * net/danet.py:912-918) or NULL. */
int danhip_conv2d_fwd(const danhip_conv_desc* d, const uint16_t* x, const uint16_t* wf_packed, const float* bias,
                      void* y, int out_dtype, int relu, const uint16_t* residual, void* stream);

/* conv_relu + the 2x2 / stride-2 'same' max-pool that follows every VGG block (net/sfd_net.py:128-143) in one call:
 * y = relu(conv(x, w) + bias) [N,Ho,Wo,Cout] and pool_y = maxpool2x2(y) [N,ceil(Ho/2),ceil(Wo/2),Cout], both in the build's
 * 16-bit type.  3x3 kernels that own whole row pairs per wave pool in their epilogue; other shapes run the pool kernel. */
int danhip_conv2d_fwd_pool(const danhip_conv_desc* d, const uint16_t* x, const uint16_t* wf_packed, const float* bias, uint16_t* y,
                           uint16_t* pool_y, void* stream);
/* The same calls that ALSO write the pool's 2-bit arg-max codes (pool_arg: [N*ceil(Ho/2)*ceil(Wo/2)][Cout/4] bytes, channel c in bits
 * 2(c%4).. of byte c/4, code = 2*dh + dw of the FIRST maximum of the window in row-major order - TF's MaxPoolGrad rule), from the epilogue
 * registers where the kernel pools there, from the pool kernel otherwise.  danhip_maxpool2x2_bwd_arg scatters the pooled gradient through
 * the codes: it reads a quarter-size gradient and 2 bits per element instead of the full-resolution activation (1.28 instead of 2.25
 * map-sized HBM passes).  pool_arg == NULL: the plain calls. */
int danhip_conv2d_fwd_pool_arg(const danhip_conv_desc* d, const uint16_t* x, const uint16_t* wf_packed, const float* bias, uint16_t* y,
                               uint16_t* pool_y, uint8_t* pool_arg, void* stream);
int danhip_conv2d_fwd_relu_bits_arg(const danhip_conv_desc* d, const uint16_t* x, const uint16_t* wf_packed, const float* bias, uint16_t* y,
                                    uint8_t* y_bits, uint16_t* pool_y, uint8_t* pool_bits, uint8_t* pool_arg, void* stream);
/* 1 when danhip_conv2d_fwd_pool may be called with y == NULL for this descriptor: only the pooled map is produced and the full-resolution
 * activation is never written (inference: nothing but the pool reads conv1_2's / conv2_2's output - 839 + 419 MB of stores per batch of
 * 16 at 640 x 640).  0: the pool is a separate kernel for this shape, y is needed. */
int danhip_conv2d_fwd_pool_only(const danhip_conv_desc* d);

/* dx = conv_transpose(dy, w) (* (relu_mask > 0) if relu_mask != NULL: fuses the ReLU backward of the layer
 * that produced x).  dy bf16 [N,Ho,Wo,Cout_pad8]; dx bf16 [N,H,W,Cin]. accumulate: dx += instead of = . */
int danhip_conv2d_bwd_data(const danhip_conv_desc* d, const uint16_t* dy, const uint16_t* wb_packed,
                           const uint16_t* relu_mask, uint16_t* dx, int accumulate, void* stream);

/* The same two calls with a caller-provided scratch buffer.  Maps with too few output tiles to fill the chip (the 40x40 ... 5x5 pyramid
 * levels; every level at 2-4 images per GPU, the per-rank shape of an 8-GPU strong-scaling run) otherwise walk K = kh*kw*Cin serially in a
 * handful of workgroups: with >= danhip_conv2d_workspace_bytes(d, which) bytes (which: 0 forward, 1 data gradient; 0 = the shape does not
 * split) K is split over workgroups, the fp32 partial outputs go to the scratch buffer and a second small kernel sums them and applies
 * the epilogue.  Results equal the single-pass form up to fp32 summation order.  ws == NULL behaves as the plain call. */
size_t danhip_conv2d_workspace_bytes(const danhip_conv_desc* d, int which);
int danhip_conv2d_fwd_ws(const danhip_conv_desc* d, const uint16_t* x, const uint16_t* wf_packed, const float* bias, void* y,
                         int out_dtype, int relu, const uint16_t* residual, void* ws, size_t ws_bytes, void* stream);
int danhip_conv2d_bwd_data_ws(const danhip_conv_desc* d, const uint16_t* dy, const uint16_t* wb_packed, const uint16_t* relu_mask,
                              uint16_t* dx, int accumulate, void* ws, size_t ws_bytes, void* stream);
/* The ReLU mask as one bit per element: bits[m][j] bit i = x[m][8j + i] > 0, rows of C/8 bytes (C % 8 == 0).  A data-gradient kernel that
 * keeps its tile's mask in LDS reads 1/16 of the bytes, and not from its epilogue: danhip_conv2d_bwd_data_takes_bits(d) says whether the
 * kernel chosen for descriptor d does (then call danhip_conv2d_bwd_data_bits with the bits of the conv's INPUT activation). */
int danhip_relu_bits(const uint16_t* x, uint8_t* bits, int64_t M, int32_t C, void* stream);
/* conv_relu (+ the block's fused 2x2 max-pool when pool_y != NULL) that also writes those bit masks for y (and pool_y) from its epilogue
 * registers, so the next convolution's data gradient needs no pass over the activation: danhip_conv2d_fwd_emits_bits(d, with_pool) == 1
 * where the forward kernel of descriptor d can (3x3 / stride-1 'same' on the 128-wide halo tiles, Cout % 128 == 0). */
int danhip_conv2d_fwd_emits_bits(const danhip_conv_desc* d, int with_pool);
int danhip_conv2d_fwd_relu_bits(const danhip_conv_desc* d, const uint16_t* x, const uint16_t* wf_packed, const float* bias, uint16_t* y,
                                uint8_t* y_bits, uint16_t* pool_y, uint8_t* pool_bits, void* stream);
int danhip_conv2d_bwd_data_takes_bits(const danhip_conv_desc* d);
/* The data gradient of the SECOND layer (conv1_2: 3x3, 64 -> 64) with the FIRST layer's weight / bias gradient folded into it (round 6).  conv1_1 has
 * no data gradient (its input is the image), so the dX this call would store is read by nothing but conv1_1's weight gradient: the kernel keeps each
 * dX tile in LDS, multiplies it with the tile's patch of the 8-channel image x8 ([N,H,W,8], cin_real <= 4 real channels) and ADDS the result to
 * dw8 [3,3,cin_real,64] / db8 [64] (fp32, may be NULL) — dX never reaches HBM and the first layer's own weight-gradient launch disappears
 * (train_sfd.py's graph: net/sfd_net.py:128 conv1 block).  relu_bits = the bit mask of conv1_1's output (danhip_conv2d_bwd_data_bits).
 * danhip_conv2d_bwd_data_first_supported(d) == 1 where the kernel takes descriptor d (else: danhip_conv2d_bwd_data_bits + danhip_conv2d_bwd_weight). */
int danhip_conv2d_bwd_data_first_supported(const danhip_conv_desc* d);
int danhip_conv2d_bwd_data_bits_first(const danhip_conv_desc* d, const uint16_t* dy, const uint16_t* wb_packed, const uint8_t* relu_bits,
                                      const uint16_t* x8, int32_t cin_real, float* dw8, float* db8, void* stream);
int danhip_conv2d_bwd_data_bits(const danhip_conv_desc* d, const uint16_t* dy, const uint16_t* wb_packed, const uint8_t* relu_bits,
                                uint16_t* dx, int accumulate, void* stream);

/* Forward 1x1 / stride-1 convolution over the CHANNEL CONCATENATION of two NHWC tensors that is never written:
 *   y = relu([x1 | x2] . W + b),  d->Cin = c1 + c2,  wf_packed = the forward packing of the [1, 1, c1 + c2, Cout] kernel.
 * Replaces tf.concat + conv2d where the reference feeds a 1x1 from two maps; with a block-diagonal kernel it is two 1x1 convolutions of
 * different inputs written side by side into one map - DAN's stage-2 input mix (/root/reference/net/danet.py:944-950:
 * concat([conv1x1(stop_gradient(stage1), C/3), conv1x1(feature, C - C/3)]), whose ragged 85 / 171 column counts no tile fits).
 * x1 holds c1 channels, x2 holds Cin - c1, both with pixel pitch src_pitch elements; c1, Cin - c1 and Cout multiples of 64.
 * _supported(): 1 when the streaming GEMM takes this shape (else concatenate on the host side and call danhip_conv2d_fwd). */
int danhip_conv2d_fwd_concat2_supported(const danhip_conv_desc* d, int32_t c1, int32_t src_pitch);
int danhip_conv2d_fwd_concat2(const danhip_conv_desc* d, const uint16_t* x1, const uint16_t* x2, int32_t c1, int32_t src_pitch,
                              const uint16_t* wf_packed, const float* bias, uint16_t* y, int relu, void* stream);

/* Channel-slice views (round 4; net/danet.py:842-918: every branch of the context block is a channel slice of a wider tensor - the fused
 * input 1x1's output, the concat buffer).  A slice [.., c0 : c0 + C] of an NHWC tensor whose pixels are `ld` elements apart is its base
 * pointer + c0 and pitch ld: x_pitch / y_pitch are the pitches of the call's input / output operand (forward: x, y; data gradient: dy,
 * dx; weight gradient: x, dy), aux_pitch that of the data gradient's relu_mask (0 = dx's channel count).  Multiples of 8 elements.
 * These calls run on the streaming GEMM / flat-M kernels and, for 3x3 / stride-1 64 -> 64 on maps the 8 x 32 tiles cover well, on the
 * register-resident 64 -> 64 kernel (the halo kernel addresses dense tensors); the weight gradient also on the row-streaming kernel;
 * 16-bit output, no residual.
 * relu_channels (forward): ReLU applies to output channels < relu_channels only (a fused block of 1x1 convolutions whose last columns
 * stay linear); pass Cout for a plain conv_relu, anything with relu = 0 for none. */
typedef struct { int32_t x_pitch, y_pitch, aux_pitch; } danhip_conv_pitch;
int danhip_conv2d_fwd_strided(const danhip_conv_desc* d, const uint16_t* x, const uint16_t* wf_packed, const float* bias, uint16_t* y,
                              int relu, int32_t relu_channels, const danhip_conv_pitch* pitch, void* ws, size_t ws_bytes, void* stream);
int danhip_conv2d_bwd_data_strided(const danhip_conv_desc* d, const uint16_t* dy, const uint16_t* wb_packed, const uint16_t* relu_mask,
                                   uint16_t* dx, int accumulate, const danhip_conv_pitch* pitch, void* ws, size_t ws_bytes, void* stream);
int danhip_conv2d_bwd_weight_strided(const danhip_conv_desc* d, const uint16_t* x, const uint16_t* dy, float* dw_hwio, float* db,
                                     int32_t cin_real, const danhip_conv_pitch* pitch, void* ws, size_t ws_bytes, void* stream);

/* dw_hwio fp32 [kh,kw,Cin,Cout] += sum_pixels x (x) dy   (atomic fp32 accumulation: zero it first).
 * db (optional) fp32 [Cout] += sum_pixels dy  (bias gradient, computed by the same kernel: no extra pass over dy).
 * cin_real: number of leading input channels that exist in dw (dw is [kh,kw,cin_real,Cout]). */
int danhip_conv2d_bwd_weight(const danhip_conv_desc* d, const uint16_t* x, const uint16_t* dy, float* dw_hwio, float* db,
                             int32_t cin_real, void* stream);
/* The same with a caller-provided scratch buffer: where danhip_conv2d_bwd_weight_workspace_bytes(d) > 0 and `ws` holds at least that many
 * bytes, the split partial sums leave the kernel as plain coalesced stores and a second small kernel combines them into dw_hwio (+=),
 * instead of fp32 atomics (4-5x the bytes per second; the atomic tail dominated the launch at <= 4 images per GPU).  ws == NULL: atomics. */
size_t danhip_conv2d_bwd_weight_workspace_bytes(const danhip_conv_desc* d);
int danhip_conv2d_bwd_weight_ws(const danhip_conv_desc* d, const uint16_t* x, const uint16_t* dy, float* dw_hwio, float* db,
                                int32_t cin_real, void* ws, size_t ws_bytes, void* stream);

/* In place: dy *= (y > 0) (ReLU backward) when y != NULL; db[c] += sum over pixels of the masked dy.
 * dy bf16 [M, C]; y bf16 [M, C] or NULL; db fp32 [C] or NULL. */
int danhip_relu_bwd_bias_grad(uint16_t* dy, const uint16_t* y, float* db, int64_t M, int32_t C, void* stream);

/* Kernel-instance label a forward (which=0) / data-gradient (which=1; which=5 when relu_mask is passed) / forward-with-fused-pool
 * (which=4, danhip_conv2d_fwd_pool) call of this descriptor launches (the demangled name rocprofv3 reports) — lets bench.py attribute
 * measured time to a kernel.  The answer is for the *_ws calls given their scratch buffer; which | 16: for the plain calls (no split-K). */
const char* danhip_conv_kernel_label(const danhip_conv_desc* d, int which);
/* Same for the weight-gradient call of this descriptor. */
const char* danhip_conv_wgrad_kernel_label(const danhip_conv_desc* d);

/* ------------------------------------------------------------------------------------------------
 * HBM-bound layer kernels (bf16 NHWC, 16-byte vectors, wave reductions).
 * ------------------------------------------------------------------------------------------------ */
/* tf.layers.max_pooling2d([2,2],[2,2],'same') — net/sfd_net.py:132.  y [N,ceil(H/2),ceil(W/2),C]; C % 8 == 0.
 * Backward: the gradient goes to the FIRST maximal element in window order (TF CPU kernel). */
int danhip_maxpool2x2_fwd(const uint16_t* x, uint16_t* y, int32_t N, int32_t H, int32_t W, int32_t C, void* stream);
int danhip_maxpool2x2_bwd(const uint16_t* x, const uint16_t* dy, uint16_t* dx, int32_t N, int32_t H, int32_t W, int32_t C,
                          int accumulate, void* stream);
/* The 2 x 2 max-pool with its arg-max codes (layout: danhip_conv2d_fwd_pool_arg) and the backward through them: dx (+)= scatter(dy, arg). */
int danhip_maxpool2x2_fwd_arg(const uint16_t* x, uint16_t* y, uint8_t* arg, int32_t N, int32_t H, int32_t W, int32_t C, void* stream);
int danhip_maxpool2x2_bwd_arg(const uint8_t* arg, const uint16_t* dy, uint16_t* dx, int32_t N, int32_t H, int32_t W, int32_t C,
                              int accumulate, void* stream);
/* VGG16Backbone.l2_normalize — net/sfd_net.py:68-79: y = x * rsqrt(max(sum_c x^2, 1e-10)) * gamma_c.
 * x,y bf16 [M,C], gamma fp32 [C], C in {64,128,256,512,1024}.  bwd: dgamma += (atomic fp32), dx (=|+= if accumulate);
 * relu_mask: x is a ReLU output, so dx is also multiplied by (x > 0) (the producer's ReLU backward folded in). */
int danhip_l2norm_fwd(const uint16_t* x, const float* gamma, uint16_t* y, int64_t M, int32_t C, void* stream);
int danhip_l2norm_bwd(const uint16_t* x, const float* gamma, const uint16_t* dy, uint16_t* dx, float* dgamma, int64_t M,
                      int32_t C, int accumulate, int relu_mask, void* stream);
/* tf.image.resize_bilinear(up, size(lateral)) (TF1 legacy mapping src = dst*(in/out), align_corners=False) fused with the
 * LFPN lateral add: out = lateral + resize(up) (lateral may be NULL) — net/pb_net.py:209-217, net/danet.py:363-371.
 * up bf16 [N,Hi,Wi,C], lateral/out bf16 [N,Ho,Wo,C], C % 8 == 0.  bwd: d_up (=|+= if accumulate) from d_out (the lateral's
 * gradient is d_out itself). */
int danhip_resize_bilinear_add_fwd(const uint16_t* up, const uint16_t* lateral, uint16_t* out, int32_t N, int32_t Hi, int32_t Wi,
                                   int32_t Ho, int32_t Wo, int32_t C, void* stream);
int danhip_resize_bilinear_add_bwd(const uint16_t* dout, uint16_t* dup, int32_t N, int32_t Hi, int32_t Wi, int32_t Ho, int32_t Wo,
                                   int32_t C, int accumulate, void* stream);
/* tf.layers.average_pooling2d((2,2), 1, 'same') — net/danet.py:854: pad (0,1), divisor = number of valid taps. */
int danhip_avgpool2x2s1_same_fwd(const uint16_t* x, uint16_t* y, int32_t N, int32_t H, int32_t W, int32_t C, void* stream);
/* x_mask (may be NULL): the pooled tensor when it is a ReLU output — dx is then multiplied by (x > 0). */
int danhip_avgpool2x2s1_same_bwd(const uint16_t* dy, const uint16_t* x_mask, uint16_t* dx, int32_t N, int32_t H, int32_t W, int32_t C,
                                 int accumulate, void* stream);
/* The residual sum of the DAN context block (net/danet.py:913-918), y = relu(conv(hyper)) + features: out = a + b over n 16-bit elements
 * (n % 8 == 0), and its backward in ONE pass over dy: dr = dy * (r > 0) (the conv's ReLU backward, materialised for its two gradient
 * kernels) and, when dx != NULL, dx (+)= dy * (x_mask > 0, or all when x_mask == NULL) (the skip path into the producer's gradient). */
int danhip_add16(const uint16_t* a, const uint16_t* b, uint16_t* out, int64_t n, void* stream);
int danhip_residual_bwd(const uint16_t* dy, const uint16_t* r, const uint16_t* x_mask, uint16_t* dr, uint16_t* dx, int accumulate, int64_t n,
                        void* stream);
/* The same pool on channel-slice views (pitches in elements, multiples of 8), optionally followed by ReLU, and its backward (dy -> dx,
 * overwritten; a ReLU's backward is the caller's: dy arrives masked).  The DAN context block (net/danet.py:854-861) runs branch 2's 1x1
 * convolution IN FRONT of the (linear) pool, on the 64 output channels instead of the C input channels. */
int danhip_avgpool2x2s1_same_fwd_strided(const uint16_t* x, int32_t x_pitch, uint16_t* y, int32_t y_pitch, int32_t N, int32_t H, int32_t W,
                                         int32_t C, int relu, void* stream);
int danhip_avgpool2x2s1_same_bwd_strided(const uint16_t* dy, int32_t y_pitch, uint16_t* dx, int32_t x_pitch, int32_t N, int32_t H, int32_t W,
                                         int32_t C, void* stream);
/* Backward of tf.concat(axis=-1) / of a residual add (net/danet.py:911-918), one input at a time:
 *   out[m][c] (+)= (mask == NULL || mask[m*ldm + c] > 0) ? dy[m*ldy + c0 + c] : 0      for c < C,   out[m][c] = 0 for C <= c < Cpad (first write)
 * out rows have Cpad (>= C, multiple of 8) elements: the channel-padded gradient layout danhip_conv2d_bwd_* take for a ragged Cout. */
int danhip_slice_deliver(const uint16_t* dy, int32_t ldy, int32_t c0, int32_t C, const uint16_t* mask, int32_t ldm, uint16_t* out, int32_t Cpad,
                         int accumulate, int64_t M, void* stream);
/* tf.layers.batch_normalization over the channel axis of [M,C] (M = N*H*W) — the conv_bn_relu / bn_relu / conv_bn surface of
 * net/sfd_net.py:91-119 (momentum 0.997, eps 1e-5).  Training: batch statistics (biased variance), optional moving-average
 * update, optional fused ReLU; workspace = 2*C floats.  Inference: caller passes mean and rstd = rsqrt(var + eps).
 * Backward: dgamma, dbeta are overwritten. */
int danhip_batchnorm_fwd_train(const uint16_t* x, const float* gamma, const float* beta, uint16_t* y, float* save_mean,
                               float* save_rstd, float* moving_mean, float* moving_var, int64_t M, int32_t C, float eps,
                               float momentum, int relu, float* workspace, void* stream);
int danhip_batchnorm_fwd_infer(const uint16_t* x, const float* gamma, const float* beta, const float* mean, const float* rstd,
                               uint16_t* y, int64_t M, int32_t C, int relu, void* stream);
int danhip_batchnorm_bwd(const uint16_t* x, const uint16_t* dy, const float* gamma, const float* save_mean, const float* save_rstd,
                         uint16_t* dx, float* dgamma, float* dbeta, int64_t M, int32_t C, void* stream);
/* preprocess_for_eval arithmetic (preprocessing/dan_preprocessing.py:55-57,755-758): uint8 RGB [npix,3] ->
 * bf16 [npix,8] = (B-103.94, G-116.78, R-123.68, 0,0,0,0,0). */
int danhip_preprocess_u8(const uint8_t* img_rgb, uint16_t* out, int64_t npix, void* stream);
/* fp32 [rows,c_src] -> bf16 [rows,c_dst] zero padded (gradient of the fp32 head outputs / of ragged-Cout convs).
 * relu_y (optional, bf16 [rows,c_src]): the layer's ReLU output — elements where it is <= 0 are zeroed (ReLU backward applied on
 * the unpadded layout, before the channel padding). */
int danhip_cast_pad_f32_to_bf16(const float* src, const uint16_t* relu_y, uint16_t* dst, int64_t rows, int32_t c_src, int32_t c_dst,
                                void* stream);

/* ------------------------------------------------------------------------------------------------
 * Detection heads glue, hard-negative mining, losses, optimizer (fp32 / int32).
 * ------------------------------------------------------------------------------------------------ */
/* Max-out + reshape_pred for one level with one anchor per cell (net/sfd_net.py:175-216; train_sfd.py:293-304):
 * h fp32 [B*HW, Ch], channels [loc(4) | neg(nneg) | pos(npos)] -> loc [B,A,4], cls [B,A,2] rows [off, off+HW). */
int danhip_head_split_fwd(const float* h, float* loc, float* cls, int32_t B, int32_t HW, int32_t Ch, int32_t nneg, int32_t npos,
                          int32_t A, int32_t anchor_offset, void* stream);
int danhip_head_split_bwd(const float* h, const float* dloc, const float* dcls, float* dy, int32_t B, int32_t HW, int32_t Ch,
                          int32_t nneg, int32_t npos, int32_t A, int32_t anchor_offset, void* stream);
/* Per-image hard-negative mining (train_sfd.py:350-384, train_dan.py:286-324): score = label==0 ? -softmax(cls)[0] : -1;
 * k = min(int(ratio*n_pos), n_neg) (at_least_one: max(k,1)); thr[b] = exact k-th largest score of row b (radix select),
 * +inf when k == 0.  cls fp32 [B,A,2], labels int32 [B,A] in {1,0,-1}; score [B,A], counts int32 [B,2], thr [B], k_out [B]. */
int danhip_hard_neg_select(const float* cls, const int32_t* labels, float* score, int32_t* counts, float* thr, int32_t* k_out,
                           int32_t B, int32_t A, float negative_ratio, int at_least_one, void* stream);
/* acc4 = [sum CE over selected, #selected, sum smooth-L1 over positives, #positives]; sel uint8 [B,A] (0/1 neg/2 pos)
 * (train_sfd.py:386-417).  bwd: dcls = (softmax-onehot)*ce_scale/#selected, dloc = dSmoothL1*loc_scale/#positives. */
int danhip_detection_loss_fwd(const float* cls, const float* loc, const int32_t* labels, const float* loc_targets,
                              const float* score, const float* thr, uint8_t* sel, float* acc4, int32_t B, int32_t A,
                              void* stream);
int danhip_detection_loss_bwd(const float* cls, const float* loc, const float* loc_targets, const uint8_t* sel,
                              const float* acc4, float* dcls, float* dloc, float ce_scale, float loc_scale, int32_t B,
                              int32_t A, void* stream);
/* Fused multi-tensor momentum SGD over flat fp32 buffers (train_sfd.py:419-447): for element i of segment s
 * (seg_starts[s] <= i < seg_starts[s+1]): g' = (g*grad_scale + wd_coef[s]*w) * gmult[s]; v = m*v + g'; w -= lr*v.
 * l2_out (optional) += sum 0.5*wd_coef[s]*w^2 (the L2 loss term at the pre-update weights).
 * Layout contract: buffers 16-byte aligned, every seg_starts[s] and total a multiple of 4 (the trainer pads each variable to 64
 * elements with zeros, which stay zero), so the kernel moves float4s that never straddle two variables. */
int danhip_sgd_momentum_flat(float* w, const float* g, float* v, const int64_t* seg_starts, const float* gmult,
                             const float* wd_coef, int32_t nseg, int64_t total, float lr, float momentum, float grad_scale,
                             float* l2_out, void* stream);
/* The same update under a DYNAMIC loss scale kept on the device (fp16 build; no host round trip, capturable in a hipGraph).
 * loss_scale_state: 4 floats {scale, clean steps so far, growth interval, scratch flag}.  One call = non-finite check of g; the
 * update with g / scale, skipped entirely (w, v untouched) when the check fired; then torch.cuda.amp.GradScaler's rule: scale x0.5
 * after a skipped step, x2 after `interval` clean ones.  The loss terms multiply their gradients by state[0] on the device. */
int danhip_sgd_momentum_flat_dynamic(float* w, const float* g, float* v, const int64_t* seg_starts, const float* gmult,
                                     const float* wd_coef, int32_t nseg, int64_t total, float lr, float momentum,
                                     float* loss_scale_state, float* l2_out, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Anchor / box index-compare kernels (fp32 + int32, bit-exact vs the oracle; utility/anchor_manipulator.py).
 * ------------------------------------------------------------------------------------------------ */
/* generate_anchors_by_offset + center2point (anchor_manipulator.py:125-127,163-198) for one level, order (y,x,depth). */
int danhip_anchors_generate(float* ymin, float* xmin, float* ymax, float* xmax, const float* anchor_h, const float* anchor_w,
                            int32_t depth, int32_t layer_h, int32_t layer_w, float stride, float offset_h, float offset_w,
                            int32_t out_offset, void* stream);
/* iou_matrix (anchor_manipulator.py:24-52), optionally multiplied by inside_mask[a] (encode_anchors :287). [A,G]. */
int danhip_iou_matrix(const float* ymin, const float* xmin, const float* ymax, const float* xmax, const uint8_t* inside_mask,
                      const float* gt_boxes, float* overlaps, int32_t A, int32_t G, void* stream);
size_t danhip_match_workspace_bytes(int32_t A, int32_t G);
/* do_dual_max_match (anchor_manipulator.py:54-105, gt_max_first=True). */
int danhip_dual_max_match(const float* overlaps, int32_t A, int32_t G, float low_thres, float high_thres, int ignore_between,
                          int32_t* match_indices, float* match_scores, void* workspace, size_t workspace_bytes, void* stream);
/* SmallMiningMatch custom op (cpp/ExtraLib/small_mining_match.cc:31-54,68-222): same inputs, attrs, outputs. */
int danhip_small_mining_match(const float* overlaps, int32_t A, int32_t G, float negative_low_thres, float negative_high_thres,
                              float positive_thres, int32_t min_match, float stop_positive_thres, int32_t* match_indices,
                              float* match_scores, void* workspace, size_t workspace_bytes, void* stream);
/* Tail of encode_anchors / encode_pa_anchors after matching (anchor_manipulator.py:294-326, :358-387). */
int danhip_encode_anchors(const float* ymin, const float* xmin, const float* ymax, const float* xmax, const float* gt_boxes,
                          const int32_t* match_indices, float* targets, int32_t* labels, float* matched_gt, int32_t A,
                          float ps0, float ps1, float ps2, float ps3, float scale, void* stream);
/* anchor_encoder_fn over a whole batch in one call (the reference maps encode_anchors / encode_pa_anchors per image inside tf.data:
 * train_sfd.py:206, train_dan.py:243-249; anchor_manipulator.py:275-387): IoU x inside_mask -> small_mining_match (match_mining=1:
 * negative_low_thres, ignore_thres, positive_thres, min_match, stop_positive_thres as the op's attrs) or do_dual_max_match
 * (match_mining=0: low=ignore_thres, high=positive_thres) -> encode.  gt_boxes [total_gt,4] is the concatenation of the images' gt rows,
 * gt_offsets [B+1] int32 (device) their row ranges; every image needs >= 1 row (the reference substitutes [[0,0,1,1]] for an empty
 * list, anchor_manipulator.py:286 / :347).  match_* = anchors used for matching (NULL: the encode anchors; encode_pa_anchors passes the
 * shrunk set and `scale`).  Outputs targets [B,A,4], labels [B,A] in {1,0,-1}, scores [B,A], matched_gt [B,A,4] or NULL.
 * Images are independent, so the B sequential hard-face compensation passes run side by side (one workgroup each). */
size_t danhip_encode_anchors_batched_workspace_bytes(int32_t B, int32_t A, int32_t total_gt);
int danhip_encode_anchors_batched(const float* ymin, const float* xmin, const float* ymax, const float* xmax, const float* match_ymin,
                                  const float* match_xmin, const float* match_ymax, const float* match_xmax, const uint8_t* inside_mask,
                                  const float* gt_boxes, const int32_t* gt_offsets, int32_t B, int32_t A, int32_t total_gt, int32_t max_gt,
                                  int32_t match_mining, float negative_low_thres, float ignore_thres, float positive_thres,
                                  int32_t min_match, float stop_positive_thres, float ps0, float ps1, float ps2, float ps3, float scale,
                                  float* targets, int32_t* labels, float* scores, float* matched_gt, void* workspace,
                                  size_t workspace_bytes, void* stream);
/* batch_decode_anchors / decode_anchors (anchor_manipulator.py:389-424). pred [B,A,4] -> boxes [B,A,4]. */
int danhip_decode_anchors(const float* pred, const float* ymin, const float* xmin, const float* ymax, const float* xmax,
                          float* boxes, int32_t B, int32_t A, float ps0, float ps1, float ps2, float ps3, void* stream);

/* Face score of two-way logits: score[i] = softmax(cls[i, 0:2])[1] (tf.nn.softmax(cls_pred)[:, -1]: eval_dan.py:356,371, eval_sfd.py:281) and /
 * or the easy-anchor mask (score > threshold) as int32 (train_dan.py:438-439, eval_dan.py:386).  cls fp32 [n, 2]; score / mask [n], either
 * may be NULL. */
int danhip_face_scores(const float* cls, float* score, int32_t* mask, float threshold, int64_t n, void* stream);


/* ------------------------------------------------------------------------------------------------
 * Deformable convolution (cpp/Deform: DeformConvOp deform_conv.cc:51-167,392-535; DeformConvBackpropOp :170-189,635-771;
 * Python names utility/custom_op.py:62-63).  The reference computes, per sample, deformable_im2col + GEMM; here the
 * im2col is danhip_deform_sample_fwd (batched, NHWC bf16, sampling rules of deform_conv.cu:91-126,229-275) and the GEMM is
 * danhip_conv2d_{fwd,bwd_data,bwd_weight} run as a 1x1 convolution over S [N,Ho,Wo,kh*kw*C] with the OIHW filter variable
 * viewed as HWIO [1,1,kh*kw*C,Cout] (k = tap*C + c).  Backward: dS from danhip_conv2d_bwd_data, then
 * danhip_deform_sample_bwd = deformable_col2im_coord (d_offsets) + deformable_col2im (dx; fp32 atomics as the reference).
 * offsets: NHWC [N,Ho,Wo,dg*2*kh*kw], channel (g*kh*kw + tap)*2 + {0: dh, 1: dw} (deform_conv.cu:251-259) — the NHWC
 * transpose of the reference's NCHW offset tensor.  SAME padding from the undilated kernel (deform_conv.cc:473-479).
 * ------------------------------------------------------------------------------------------------ */
int danhip_deform_sample_fwd(const uint16_t* x, const uint16_t* offsets, uint16_t* S, int32_t N, int32_t H, int32_t W, int32_t C,
                             int32_t kh, int32_t kw, int32_t stride, int32_t dilation, int32_t deformable_group, void* stream);
/* workspace: at least (N*H*W*C + 64) * sizeof(float) bytes = danhip_deform_sample_bwd_workspace_bytes (fp32 scatter target + the
 * far-corner statistic that selects the gather or the scatter form on the device).  ABI version 2 (danhip_version): the call takes
 * workspace_bytes and returns DANHIP_EWORKSPACE for a smaller buffer (version 1 took no size and sized the buffer N*H*W*C floats). */
size_t danhip_deform_sample_bwd_workspace_bytes(int32_t N, int32_t H, int32_t W, int32_t C);
int danhip_deform_sample_bwd(const uint16_t* x, const uint16_t* offsets, const uint16_t* dS, uint16_t* dx, uint16_t* d_offsets,
                             int32_t N, int32_t H, int32_t W, int32_t C, int32_t kh, int32_t kw, int32_t stride, int32_t dilation,
                             int32_t deformable_group, int accumulate, float* workspace, size_t workspace_bytes, void* stream);

/* DeformConvOp / DeformConvBackpropOp as single calls (the bindings custom_op.py:62-63 would make): same tensors and attrs as
 * the TF ops (strides / rates collapsed to one int each, num_groups = 1 as every call site uses), NHWC bf16, all samples in
 * one batch.  The im2col buffer lives in the workspace for the duration of the call only (deform_conv.cc:497-503
 * allocate_temp); the backward re-runs im2col as the reference does (:744-748) instead of keeping it from the forward.
 *   filter : the OIHW variable [Cout,C,kh,kw] viewed as HWIO [1,1,kh*kw*C,Cout] (k = tap*C + c) and packed with
 *            danhip_pack_conv_weight for the 1x1 descriptor {N,Ho,Wo,kh*kw*C -> Cout}: wf_packed (forward), wb_packed (backward).
 *   dw     : fp32 [kh*kw*C, Cout] in that same view, ACCUMULATED (zero it first); db fp32 [Cout] accumulated, or NULL.
 *   dy     : bf16 [N,Ho,Wo,Cout] gradient w.r.t. the op's output (after the caller's ReLU backward if relu was fused), Cout % 8 == 0.
 *   dx = | += (accumulate_dx) bf16 [N,H,W,C]; d_offsets bf16 like offsets (overwritten). */
size_t danhip_deform_conv_workspace_bytes(int32_t N, int32_t H, int32_t W, int32_t C, int32_t kh, int32_t kw, int32_t stride, int backward);
/* 1 when danhip_deform_conv_fwd runs this shape as ONE kernel (csrc/deform_fused.hip: 3x3 taps, C / deformable_group == 64, Cout in
 * {64, 128, 256}) — sampling feeds the GEMM through LDS.  The forward then takes workspace = NULL (no column buffer exists at all:
 * inference), or a workspace of danhip_deform_conv_workspace_bytes(.., 0) into which the kernel ALSO writes the column buffer for a caller
 * that keeps it for danhip_deform_conv_bwd_with_col.  Other shapes: sampling kernel + GEMM, workspace required. */
int danhip_deform_conv_fused(int32_t N, int32_t H, int32_t W, int32_t C, int32_t Cout, int32_t kh, int32_t kw, int32_t stride, int32_t deformable_group);
int danhip_deform_conv_fwd(const uint16_t* x, const uint16_t* wf_packed, const float* bias, const uint16_t* offsets, uint16_t* y,
                           int32_t N, int32_t H, int32_t W, int32_t C, int32_t Cout, int32_t kh, int32_t kw, int32_t stride,
                           int32_t dilation, int32_t deformable_group, int relu, void* workspace, size_t workspace_bytes, void* stream);
int danhip_deform_conv_bwd(const uint16_t* x, const uint16_t* wb_packed, const uint16_t* offsets, const uint16_t* dy, uint16_t* dx,
                           uint16_t* d_offsets, float* dw, float* db, int32_t N, int32_t H, int32_t W, int32_t C, int32_t Cout,
                           int32_t kh, int32_t kw, int32_t stride, int32_t dilation, int32_t deformable_group, int accumulate_dx,
                           void* workspace, size_t workspace_bytes, void* stream);
/* Same, with the im2col buffer that danhip_deform_conv_fwd left at the start of ITS workspace (kept alive by the caller; same x and
 * offsets): skips the reference's re-im2col (deform_conv.cc:744-748) — on a 288 GB part the 9x-activation buffer is cheaper to keep
 * than to regenerate.  col_saved = NULL behaves as danhip_deform_conv_bwd. */
int danhip_deform_conv_bwd_with_col(const uint16_t* x, const uint16_t* wb_packed, const uint16_t* offsets, const uint16_t* dy,
                                    const uint16_t* col_saved, uint16_t* dx, uint16_t* d_offsets, float* dw, float* db, int32_t N,
                                    int32_t H, int32_t W, int32_t C, int32_t Cout, int32_t kh, int32_t kw, int32_t stride,
                                    int32_t dilation, int32_t deformable_group, int accumulate_dx, void* workspace,
                                    size_t workspace_bytes, void* stream);
/* Same, with the input gradient handed over the way the convolutions hand theirs to the layer below: relu_x != 0 multiplies it by (x > 0)
 * - x is then a ReLU output (DAN-Deform: the 1x1 'down' convolution, /root/reference/net/danet_deform.py:267-290) and its ReLU backward is
 * folded into this call - before accumulate_dx adds what dx already holds (the offset convolution's contribution). */
int danhip_deform_conv_bwd_deliver(const uint16_t* x, const uint16_t* wb_packed, const uint16_t* offsets, const uint16_t* dy,
                                   const uint16_t* col_saved, uint16_t* dx, uint16_t* d_offsets, float* dw, float* db, int32_t N,
                                   int32_t H, int32_t W, int32_t C, int32_t Cout, int32_t kh, int32_t kw, int32_t stride,
                                   int32_t dilation, int32_t deformable_group, int accumulate_dx, int relu_x, void* workspace,
                                   size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * DynamicAnchorRouting custom op (cpp/ExtraLib/dynamic_anchor_routing.cc:32-65 op def, :188-518 kernel; Python name
 * utility/custom_op.py:52) — same tensors / scalars / attrs as the TF op, plus a leading batch B (the reference maps the op
 * over the batch with tf.map_fn, train_dan.py:382): every array carries B images of N = feat_height*feat_width*anchor_depth
 * rows.  img_height / img_width of the TF op are unused by its kernel and therefore not taken.
 *   eval  (trainging=false, :328-408): mask_out in {0,1}, decode_out = stage-2 boxes decoded against the routed stage-1 box.
 *   train (trainging=true,  :203-327): mask_out in {1,0,-1}, decode_out = stage-2 regression targets.  The reference draws
 *          from std::mt19937(std::random_device) (not reproducible); here u(b,i) = splitmix64(seed, counter0 + b*N + i).
 *          counter_dev (nullable): device uint64 added to counter0 at run time, so a launch recorded in a hipGraph draws a
 *          fresh stream on every replay (the caller advances it with a device-side add).
 * ------------------------------------------------------------------------------------------------ */
size_t danhip_routing_workspace_bytes(int64_t N, int32_t B, int training);
int danhip_dynamic_anchor_routing_eval(const float* anchors, const float* gt_targets, const float* labels, const int32_t* mask_in,
                                       int64_t N, int32_t feat_height, int32_t feat_width, int32_t anchor_depth, int32_t feat_strides,
                                       int32_t B, int32_t* mask_out, float* decode_out, void* workspace, size_t workspace_bytes,
                                       void* stream);
int danhip_dynamic_anchor_routing_train(const float* anchors, const float* gt_targets, const float* labels, const int32_t* mask_in,
                                        int64_t N, int32_t feat_height, int32_t feat_width, int32_t anchor_depth, int32_t feat_strides,
                                        int32_t B, float thres, float ignore_thres, uint64_t seed, uint64_t counter0,
                                        const uint64_t* counter_dev, int32_t* mask_out, float* decode_out, void* workspace,
                                        size_t workspace_bytes, void* stream);

/* tf.image.non_max_suppression as used by utility/bbox_util.py:75-91: greedy over candidates already ordered by descending
 * score (stable), suppress when IoU > iou_threshold (raw areas, no +1, corners normalised by min/max).
 * boxes_sorted fp32 [B,K,4]; keep_idx int32 [B,max_out] = kept positions (-1 padded); num_keep int32 [B]. */
int danhip_nms(const float* boxes_sorted, int32_t B, int32_t K, int32_t max_out, float iou_threshold, int32_t* keep_idx,
               int32_t* num_keep, void* stream);

/* Stable descending arg-sort of one fp32 vector (the ordering step of utility/bbox_util.py:61-73 tf.nn.top_k and :75-91 ahead of
 * tf.image.non_max_suppression; eval_dan.py:255 `argsort()[::-1]` with ties_high_index_first = 1): idx_out int32 [n] = positions by
 * descending score, equal scores (+0 == -0) by ascending index (descending with ties_high_index_first).  Bitonic network on unique
 * 64-bit (score, index) keys; workspace = danhip_argsort_workspace_bytes(n) bytes (8 per element of the next power of two >= 8192). */
size_t danhip_argsort_workspace_bytes(int64_t n);
int danhip_argsort_desc_f32(const float* scores, int64_t n, int32_t ties_high_index_first, int32_t* idx_out, void* workspace,
                            size_t workspace_bytes, void* stream);

/* DeformPSROIPool / DeformPSROIPoolGrad (cpp/Deform/deform_psroi_pooling_op.cc:37-97; utility/custom_op.py:93-126):
 * the TF op's tensors and attributes as they are — data fp32 NCHW [B,C,H,W], rois fp32 [R,5] (batch index, x1, y1, x2, y2), trans fp32
 * [R,2*num_classes,part_size,part_size] (NULL allowed when no_trans) -> top_data, mapping_channel (sample count) fp32
 * [R,output_dim,pooled_size,pooled_size]; num_classes = no_trans ? 1 : trans.dim(1)/2.  The backward zeroes data_diff [B,C,H,W] /
 * trans_diff and scatters with fp32 atomics like the reference.  Buffers 16-byte aligned. */
int danhip_deform_psroi_pool_fwd(const float* data, const float* rois, const float* trans, float* top_data, float* mapping_channel, int32_t R,
                                 int32_t C, int32_t H, int32_t W, int32_t output_dim, int32_t group_size, int32_t pooled_size, int32_t part_size,
                                 int32_t sample_per_part, float spatial_scale, float trans_std, int32_t no_trans, int32_t num_classes, void* stream);
int danhip_deform_psroi_pool_bwd(const float* top_diff, const float* mapping_channel, const float* data, const float* rois, const float* trans,
                                 float* data_diff, float* trans_diff, int32_t B, int32_t R, int32_t C, int32_t H, int32_t W, int32_t output_dim,
                                 int32_t group_size, int32_t pooled_size, int32_t part_size, int32_t sample_per_part, float spatial_scale,
                                 float trans_std, int32_t no_trans, int32_t num_classes, void* stream);

/* tf.layers.max_pooling2d([3,3],[2,2],'same') — the ResNet stem's pool_1 (net/resnet_danet.py:129).
 * y [N,ceil(H/2),ceil(W/2),C]; backward routes each window's gradient to its first maximum (gather form, dx overwritten). */
int danhip_maxpool3x3s2_same_fwd(const uint16_t* x, uint16_t* y, int32_t N, int32_t H, int32_t W, int32_t C, void* stream);
int danhip_maxpool3x3s2_same_bwd(const uint16_t* x, const uint16_t* dy, uint16_t* dx, int32_t N, int32_t H, int32_t W, int32_t C,
                                 void* stream);

/* --------------------------------------------------------------------------------------------------
 * Test-time pipeline of eval_dan.py / eval_sfd.py
 *   resize_u8_linear : cv2.resize(image, None, None, fx, fy, INTER_LINEAR) of eval_dan.py:96-97 on an 8-bit HWC image
 *                      (generic fixed-point path of OpenCV); Ho/Wo = round-half-even(H*fy / W*fx), computed by the caller.
 *   bbox_vote        : eval_dan.py:201-241.  det float64 [B,Nmax,5] rows (xmin,ymin,xmax,ymax,score) in the order of
 *                      det[:,4].argsort()[::-1] (:202-203), counts int32 [B]; IoU with +1, merge when >= iou_threshold,
 *                      clusters of one dropped, float64 sums, float32 results.  out fp32 [B,max_out,5], num_out int32 [B].
 * ------------------------------------------------------------------------------------------------ */
int danhip_resize_u8_linear(const uint8_t* src, int32_t H, int32_t W, uint8_t* dst, int32_t Ho, int32_t Wo, int32_t C, double fx,
                            double fy, void* stream);
size_t danhip_bbox_vote_workspace_bytes(int32_t B, int32_t Nmax);
int danhip_bbox_vote(const double* det, const int32_t* counts, int32_t B, int32_t Nmax, double iou_threshold, int32_t max_out,
                     float* out, int32_t* num_out, void* workspace, size_t workspace_bytes, void* stream);

/* --------------------------------------------------------------------------------------------------
 * On-device training input pipeline: the image half of preprocess_for_train
 * (preprocessing/dan_preprocessing.py:677-733) in one pass from the decoded uint8 image [H,W,3] (RGB) to the network input
 * [out_h,out_w,8] (16-bit, B-mean, G-mean, R-mean, 0 x5):  [0,1] conversion -> distort_color (:98-150) -> crop window with
 * mean-colour fill (:410-565; the window may leave the image) -> legacy bilinear resize -> flip -> uint8 saturate -> means -> BGR.
 * op_codes: 0 brightness (value = delta), 1 saturation (factor), 2 hue (delta), 3 contrast (factor; at most one), applied in
 * the given order; the random draws themselves are the caller's (dan_amd/preprocessing/dan_preprocessing.py).
 * ------------------------------------------------------------------------------------------------ */
size_t danhip_augment_workspace_bytes(void);
int danhip_augment_preprocess(const uint8_t* src, int32_t H, int32_t W, int32_t nops, const int32_t* op_codes, const float* op_values,
                              int32_t win_y, int32_t win_x, int32_t win_h, int32_t win_w, int32_t flip, uint16_t* dst, int32_t out_h,
                              int32_t out_w, void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Data-parallel gradient exchange (csrc/comm.cpp): the all-reduce(sum) of tf_replicate_model_fn.py:633-645
 * (_compute_sum_on_device: add_n over the towers' gradients, after _scale_loss :615-631 put 1/N into every tower's loss) issued
 * straight on RCCL (rccl.h: ncclCommInitRank / ncclAllReduce / ncclReduceScatter / ncclAllGather) from this library, on the caller's
 * stream — no framework process group, no watchdog / heartbeat thread, so the calls can be recorded into a hipGraph like any kernel
 * (RCCL's collectives are stream-ordered device work; SURVEY 8b: "directly via rccl.h from C++ for the overlapped bucket path").
 * librccl is NOT a link-time dependency: it is dlopen'ed at the first danhip_comm_* call (danhip_comm_load names the file, e.g. the
 * copy PyTorch ships; NULL = the copy already in the process, else librccl.so[.1] on the loader path).  One communicator = one rank =
 * one GPU (the device current at danhip_comm_create); the 128-byte unique id is produced by rank 0 and carried to the other ranks by
 * the caller (any host channel: a TCP store, a file, MPI).  A communicator must be used by one thread at a time (RCCL's rule).
 * dtype: DANHIP_F32 | DANHIP_BF16 | DANHIP_F16 (here DANHIP_BF16 means bf16 whatever the build's activation type is).
 * ------------------------------------------------------------------------------------------------ */
#define DANHIP_ECOMM (-4)      /* RCCL missing / an RCCL call failed (danhip_last_error carries ncclGetErrorString) */
#define DANHIP_COMM_ID_BYTES 128
int danhip_comm_load(const char* librccl_path);                        /* optional; idempotent once a copy is bound */
int danhip_comm_rccl_version(int* version);                            /* ncclGetVersion: e.g. 22606 */
int danhip_comm_unique_id(void* id128);                                /* host buffer of DANHIP_COMM_ID_BYTES (rank 0) */
/* collective over the nranks callers; device >= 0: hipSetDevice(device) first (RCCL binds a communicator to the calling thread's
 * current device), -1: the current device */
int danhip_comm_create(const void* id128, int32_t nranks, int32_t rank, int32_t device, void** comm_out);
int danhip_comm_destroy(void* comm);                                   /* after the streams that carry its collectives have drained */
int danhip_comm_info(void* comm, int32_t* nranks, int32_t* rank, int32_t* device);
/* Failure detection (ncclCommGetAsyncError / ncclCommAbort; ProcessGroupNCCL's watchdog did this from a background thread, here the
 * CALLER polls): *err = 0 while the communicator is healthy, else the RCCL error code of a collective that failed asynchronously (a
 * peer died, a link dropped) with its text in danhip_last_error().  danhip_comm_abort tears a communicator down WITHOUT waiting for
 * its outstanding collectives (the only way out when a peer is gone; danhip_comm_destroy would wait for ever). */
int danhip_comm_async_error(void* comm, int32_t* err);
int danhip_comm_abort(void* comm);
/* buf[i] = sum over ranks of buf[i], in place, asynchronous on stream */
int danhip_comm_allreduce_sum(void* comm, void* buf, int64_t count, int dtype, void* stream);
/* recv[0:recvcount] = sum over ranks of send[rank*recvcount : (rank+1)*recvcount]   (send holds nranks*recvcount elements; recv may be
 * the caller's own slice of send: in place) */
int danhip_comm_reduce_scatter_sum(void* comm, const void* send, void* recv, int64_t recvcount, int dtype, void* stream);
/* recv[r*sendcount : (r+1)*sendcount] = rank r's send   (send may be the caller's slice of recv: in place) */
int danhip_comm_allgather(void* comm, const void* send, void* recv, int64_t sendcount, int dtype, void* stream);

/* ---- split-operand inference (csrc/split_infer.hip): fp32-accurate evaluation at the 16-bit MFMA rate, for the same north-star bound as the
 * fp32 path below (eval_dan.py:299-404 box outputs within 1e-4).  An fp32 map [M, C] is carried as IEEE-half limbs in a 3C-channel NHWC
 * map X3 = [hi | lo | hi] (hi = half(x), lo = half(x - hi)), C3 = 3C rounded up to 8; with weights W3 = [hi | hi | lo] along Cin the
 * ordinary 16-bit convolution of the fp16 build (libdanhip_f16.so: danhip_conv2d_fwd with Cin = C3, fp32 output) computes
 * hi.hi + lo.hi + hi.lo with exact products and fp32 accumulation.  These entry points are identical in both builds (always IEEE half).
 * relu != 0: max(x, 0) first.  danhip_maxpool2x2_split3: tf.layers.max_pooling2d([2,2],[2,2],'same') on the 3C layout (C % 8 == 0). */
int danhip_split3_f32(const float* x, uint16_t* y3, int64_t M, int32_t C, int32_t C3, int relu, void* stream);
int danhip_unsplit3_f32(const uint16_t* x3, float* y, int64_t M, int32_t C, int32_t C3, void* stream);
int danhip_maxpool2x2_split3(const uint16_t* x3, uint16_t* y3, int32_t N, int32_t H, int32_t W, int32_t C, void* stream);
/* l2_normalize (net/sfd_net.py:68-79) on the limb layout -> limb layout; the arithmetic of unsplit -> danhip_l2norm_fwd_f32 -> split in one pass */
int danhip_l2norm_split3(const uint16_t* x3, const float* gamma, uint16_t* y3, int64_t M, int32_t C, void* stream);
/* The first layer of the split-operand path (conv1_1: 3x3 / stride 1 / 'same', x fp32 [N,H,W,3], w HWIO fp32 [3,3,3,64]) as an exact fp32 FMA chain,
 * stored straight in the next convolution's limb layout y3 [N,H,W,192]; bias may be NULL. */
int danhip_conv3x3_c3_f32_split3(const float* x, const float* w_hwio, const float* bias, uint16_t* y3, int32_t N, int32_t H, int32_t W, int32_t Cout,
                                 int relu, void* stream);

/* ---- fp32 inference path (csrc/f32_infer.hip): the evaluation graphs of eval_sfd.py:232-283 / eval_pb.py / eval_dan.py:299-404 with fp32
 * storage and arithmetic end to end, for the north-star tolerance "eval box outputs within 1e-4 of the reference".  NHWC fp32
 * activations, the TF variables as they are (HWIO fp32 kernel, no packing; Cin need not be padded), forward only.  Same semantics as
 * the 16-bit entry points above: TF 'same' / 'valid' sizes through the descriptor, optional bias / ReLU / residual-after-activation. */
int danhip_conv2d_fwd_f32(const danhip_conv_desc* d, const float* x, const float* w_hwio, const float* bias, float* y, int relu,
                          const float* residual, void* stream);
int danhip_maxpool2x2_fwd_f32(const float* x, float* y, int32_t N, int32_t H, int32_t W, int32_t C, void* stream);
int danhip_l2norm_fwd_f32(const float* x, const float* gamma, float* y, int64_t M, int32_t C, void* stream);
int danhip_resize_bilinear_add_fwd_f32(const float* up, const float* lateral, float* out, int32_t N, int32_t Hi, int32_t Wi, int32_t Ho,
                                       int32_t Wo, int32_t C, void* stream);
int danhip_avgpool2x2s1_same_fwd_f32(const float* x, float* y, int32_t N, int32_t H, int32_t W, int32_t C, void* stream);
/* deformable im2col of DeformConvOp (cpp/Deform/deform_conv.cu:229-275) in fp32: S[N,Ho,Wo,kh*kw*C]; the GEMM is a 1x1
 * danhip_conv2d_fwd_f32 over S */
int danhip_deform_sample_fwd_f32(const float* x, const float* offsets, float* S, int32_t N, int32_t H, int32_t W, int32_t C, int32_t kh,
                                 int32_t kw, int32_t stride, int32_t dilation, int32_t deformable_group, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DANHIP_H_ */

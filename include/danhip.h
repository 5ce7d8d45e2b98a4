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
 *   - re-entrant and thread-safe: no mutable globals.
 *   - activations are NHWC; bf16 storage is raw uint16 (upper half of the fp32 pattern).
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

const char* danhip_last_error(void);
int danhip_version(void);

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
  int32_t kh, kw, stride;    /* TF SAME padding is derived: pad_before = max((Ho-1)*s+kh-H,0)/2        */
} danhip_conv_desc;

/* rows/cols of the packed forward (which=0) or data-gradient (which=1) weight matrix */
int danhip_conv_packed_dims(const danhip_conv_desc* d, int which, int64_t* rows, int64_t* cols);
/* w_hwio fp32 [kh,kw,CinReal,Cout] (CinReal <= d->Cin; extra input channels get zero weights) */
int danhip_pack_conv_weight(const danhip_conv_desc* d, const float* w_hwio, int32_t cin_real,
                            uint16_t* wf_packed, uint16_t* wb_packed, void* stream);

/* y = act(conv(x, w) + bias).  x bf16; y bf16 (out_dtype=DANHIP_BF16) or fp32; bias fp32[Cout] or NULL.
 * relu: 0/1.  residual: optional bf16 tensor of y's shape added AFTER the activation (DAN context modules,
 * net/danet.py:912-918) or NULL. */
int danhip_conv2d_fwd(const danhip_conv_desc* d, const uint16_t* x, const uint16_t* wf_packed, const float* bias,
                      void* y, int out_dtype, int relu, const uint16_t* residual, void* stream);

/* dx = conv_transpose(dy, w) (* (relu_mask > 0) if relu_mask != NULL: fuses the ReLU backward of the layer
 * that produced x).  dy bf16 [N,Ho,Wo,Cout_pad8]; dx bf16 [N,H,W,Cin]. accumulate: dx += instead of = . */
int danhip_conv2d_bwd_data(const danhip_conv_desc* d, const uint16_t* dy, const uint16_t* wb_packed,
                           const uint16_t* relu_mask, uint16_t* dx, int accumulate, void* stream);

/* dw_hwio fp32 [kh,kw,Cin,Cout] += sum_pixels x (x) dy   (atomic fp32 accumulation: zero it first).
 * cin_real: number of leading input channels that exist in dw (dw is [kh,kw,cin_real,Cout]). */
int danhip_conv2d_bwd_weight(const danhip_conv_desc* d, const uint16_t* x, const uint16_t* dy, float* dw_hwio,
                             int32_t cin_real, void* stream);

/* In place: dy *= (y > 0) (ReLU backward) when y != NULL; db[c] += sum over pixels of the masked dy.
 * dy bf16 [M, C]; y bf16 [M, C] or NULL; db fp32 [C] or NULL. */
int danhip_relu_bwd_bias_grad(uint16_t* dy, const uint16_t* y, float* db, int64_t M, int32_t C, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DANHIP_H_ */

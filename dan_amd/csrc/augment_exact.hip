// On-device training input pipeline: the image half of preprocess_for_train
// (preprocessing/dan_preprocessing.py:677-733) as ONE pass from the decoded uint8 image to the network's input tensor:
//   convert to [0,1] -> distort_color (:98-150: brightness / saturation / hue / contrast in one of four orders, clip)
//   -> crop window with mean-colour fill (:410-565) -> tf.image.resize_images(BILINEAR, align_corners=False) -> left-right flip
//   -> convert_image_dtype(uint8, saturate) -> mean subtraction -> BGR -> 16-bit NHWC padded to 8 channels (what conv1_1 reads).
// Nothing is materialised in between: an output pixel pulls its four source pixels, distorts each (the colour ops are
// per-pixel; the contrast op needs the per-channel mean of the image at that point of the chain = one reduction pass first),
// fills with the mean colour outside the image, and interpolates.  The random decisions arrive as plain parameters (drawn on
// the host by dan_amd/preprocessing/dan_preprocessing.py); the colour formulas are restated in oracle/preprocess.py.
#include "common.h"

namespace {

enum { OP_BRIGHTNESS = 0, OP_SATURATION = 1, OP_HUE = 2, OP_CONTRAST = 3 };

struct AugOps {
  int nops;
  int op[4];
  float val[4];
};

__device__ __forceinline__ void rgb2hsv(float r, float g, float b, float& h, float& s, float& v) {
  const float M = fmaxf(fmaxf(r, g), b), m = fminf(fminf(r, g), b), c = M - m;
  h = 0.f;
  if (c > 0.f) {
    if (M == r) { h = (g - b) / c; if (h < 0.f) h += 6.f; }
    else if (M == g) h = (b - r) / c + 2.f;
    else h = (r - g) / c + 4.f;
    h = h / 6.f;
  }
  s = M > 0.f ? c / M : 0.f;
  v = M;
}

__device__ __forceinline__ void hsv2rgb(float h, float s, float v, float& r, float& g, float& b) {
  const float c = s * v, m = v - c, dh = h * 6.f;
  const float x = c * (1.f - fabsf(fmodf(dh, 2.f) - 1.f));
  int k = (int)dh;
  k = k > 5 ? 5 : k;
  r = (k == 0 || k == 5) ? c : ((k == 1 || k == 4) ? x : 0.f);
  g = (k == 1 || k == 2) ? c : ((k == 0 || k == 3) ? x : 0.f);
  b = (k == 3 || k == 4) ? c : ((k == 2 || k == 5) ? x : 0.f);
  r += m; g += m; b += m;
}

// ops [first, last) of the chain on one pixel; `mean` = per-channel mean for the contrast op
__device__ __forceinline__ void apply_ops(const AugOps& o, int first, int last, const float* __restrict__ mean, float& r, float& g, float& b) {
  for (int i = first; i < last; ++i) {
    const float v = o.val[i];
    switch (o.op[i]) {
      case OP_BRIGHTNESS: r += v; g += v; b += v; break;
      case OP_SATURATION: {
        float h, s, x;
        rgb2hsv(r, g, b, h, s, x);
        s = fminf(fmaxf(s * v, 0.f), 1.f);
        hsv2rgb(h, s, x, r, g, b);
        break;
      }
      case OP_HUE: {
        float h, s, x;
        rgb2hsv(r, g, b, h, s, x);
        h = h + v;
        h = h - floorf(h);
        hsv2rgb(h, s, x, r, g, b);
        break;
      }
      default:
        r = (r - mean[0]) * v + mean[0]; g = (g - mean[1]) * v + mean[1]; b = (b - mean[2]) * v + mean[2];
    }
  }
}

// per-channel sum of the image after ops [0, upto) — the contrast op's mean (fp32 partials per thread, fp64 atomics)
__global__ void aug_mean_kernel(const uint8_t* __restrict__ src, long npix, AugOps o, int upto, double* __restrict__ sums) {
  double s0 = 0., s1 = 0., s2 = 0.;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += (long)gridDim.x * blockDim.x) {
    float r = (float)src[i * 3] * (1.0f / 255), g = (float)src[i * 3 + 1] * (1.0f / 255), b = (float)src[i * 3 + 2] * (1.0f / 255);
    apply_ops(o, 0, upto, nullptr, r, g, b);
    s0 += r; s1 += g; s2 += b;
  }
  for (int off = 32; off > 0; off >>= 1) {
    s0 += __shfl_down(s0, off); s1 += __shfl_down(s1, off); s2 += __shfl_down(s2, off);
  }
  if ((threadIdx.x & 63) == 0) { atomicAdd(sums, s0); atomicAdd(sums + 1, s1); atomicAdd(sums + 2, s2); }
}

__global__ void aug_mean_finish_kernel(const double* __restrict__ sums, long npix, float* __restrict__ mean) {
  if (threadIdx.x < 3) mean[threadIdx.x] = (float)(sums[threadIdx.x] / (double)npix);
}

__global__ void augment_kernel(const uint8_t* __restrict__ src, int H, int W, AugOps o, const float* __restrict__ mean, int wy, int wx, int wh,
                               int ww, int flip, bf16_t* __restrict__ dst, int Ho, int Wo) {
  const float sy = (float)wh / (float)Ho, sx = (float)ww / (float)Wo;
  const float fill[3] = {123.68f / 255.f, 116.78f / 255.f, 103.94f / 255.f};
  const long total = (long)Ho * Wo;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int oy = (int)(idx / Wo), ox = (int)(idx % Wo);
    const int rx = flip ? Wo - 1 - ox : ox;                    // the flip acts on the resized image
    const float iy = (float)oy * sy, ix = (float)rx * sx;
    const int y0 = (int)floorf(iy), x0 = (int)floorf(ix);
    const int y1 = min(y0 + 1, wh - 1), x1 = min(x0 + 1, ww - 1);
    const float ly = iy - (float)y0, lx = ix - (float)x0;
    float px[4][3];
    const int ys[2] = {y0, y1}, xs[2] = {x0, x1};
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const int Y = ys[a] + wy, X = xs[c] + wx;              // window -> image coordinates
        float r, g, b;
        if ((unsigned)Y < (unsigned)H && (unsigned)X < (unsigned)W) {
          const uint8_t* p = src + ((long)Y * W + X) * 3;
          r = (float)p[0] * (1.0f / 255); g = (float)p[1] * (1.0f / 255); b = (float)p[2] * (1.0f / 255);
          apply_ops(o, 0, o.nops, mean, r, g, b);
          r = fminf(fmaxf(r, 0.f), 1.f); g = fminf(fmaxf(g, 0.f), 1.f); b = fminf(fmaxf(b, 0.f), 1.f);
        } else {
          r = fill[0]; g = fill[1]; b = fill[2];               // tf.pad(.., constant_values = mean colour) of the distorted image
        }
        px[a * 2 + c][0] = r; px[a * 2 + c][1] = g; px[a * 2 + c][2] = b;
      }
    float out[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const float top = px[0][k] + (px[1][k] - px[0][k]) * lx;
      const float bot = px[2][k] + (px[3][k] - px[2][k]) * lx;
      const float v = top + (bot - top) * ly;
      out[k] = truncf(fminf(fmaxf(v * 255.5f, 0.f), 255.f));   // convert_image_dtype(float -> uint8, saturate=True), back to float
    }
    uint4 pk;
    pk.x = pack2bf(out[2] - 103.94f, out[1] - 116.78f);        // B, G
    pk.y = pack2bf(out[0] - 123.68f, 0.f);                     // R, pad
    pk.z = 0u; pk.w = 0u;
    *reinterpret_cast<uint4*>(dst + idx * 8) = pk;
  }
}

}  // namespace

extern "C" size_t danhip_augment_workspace_bytes(void) { return 64; }

extern "C" int danhip_augment_preprocess(const uint8_t* src, int32_t H, int32_t W, int32_t nops, const int32_t* op_codes, const float* op_values,
                                         int32_t win_y, int32_t win_x, int32_t win_h, int32_t win_w, int32_t flip, uint16_t* dst, int32_t out_h,
                                         int32_t out_w, void* workspace, size_t workspace_bytes, void* stream) {
  DH_REQUIRE(src && dst && workspace && H > 0 && W > 0 && out_h > 0 && out_w > 0 && win_h > 0 && win_w > 0, DANHIP_EINVAL,
             "augment_preprocess: bad arguments");
  DH_REQUIRE(nops >= 0 && nops <= 4 && (nops == 0 || (op_codes && op_values)), DANHIP_EINVAL, "augment_preprocess: at most 4 colour ops");
  DH_REQUIRE(workspace_bytes >= danhip_augment_workspace_bytes(), DANHIP_EWORKSPACE, "augment_preprocess: workspace too small");
  AugOps o{};
  o.nops = nops;
  int contrast_at = -1;
  for (int i = 0; i < nops; ++i) {
    DH_REQUIRE(op_codes[i] >= 0 && op_codes[i] <= 3, DANHIP_EINVAL, "augment_preprocess: unknown colour op %d", op_codes[i]);
    o.op[i] = op_codes[i];
    o.val[i] = op_values[i];
    if (op_codes[i] == OP_CONTRAST) {
      DH_REQUIRE(contrast_at < 0, DANHIP_EINVAL, "augment_preprocess: one contrast op per chain");
      contrast_at = i;
    }
  }
  hipStream_t s = (hipStream_t)stream;
  double* sums = (double*)workspace;
  float* mean = (float*)((char*)workspace + 32);
  if (contrast_at >= 0) {
    int rc = danhip_zero_async(workspace, 64, s);
    if (rc) return rc;
    const long npix = (long)H * W;
    long blocks = (npix + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(aug_mean_kernel, dim3((unsigned)blocks), dim3(256), 0, s, src, npix, o, contrast_at, sums);
    DH_LAUNCH_CHECK();
    hipLaunchKernelGGL(aug_mean_finish_kernel, dim3(1), dim3(64), 0, s, sums, npix, mean);
    DH_LAUNCH_CHECK();
  }
  const long total = (long)out_h * out_w;
  long blocks = (total + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(augment_kernel, dim3((unsigned)blocks), dim3(256), 0, s, src, H, W, o, mean, win_y, win_x, win_h, win_w, flip, dst, out_h, out_w);
  DH_LAUNCH_CHECK();
  return DANHIP_OK;
}

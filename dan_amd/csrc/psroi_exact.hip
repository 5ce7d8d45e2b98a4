// Deformable position-sensitive ROI pooling (cpp/Deform/deform_psroi_pooling_op.cc:37-97 op defs, deform_psroi_pooling_op_gpu.cu:47-125
// forward, :187-300 backward; Python names utility/custom_op.py:93-126) — SURVEY §8f row 4.  Same tensors and attributes as the TF op:
// data fp32 NCHW [B,C,H,W], rois fp32 [R,5] (batch index, x1, y1, x2, y2), trans fp32 [R, 2*num_classes, part, part];
// top_data / mapping_channel (= sample count) fp32 [R, output_dim, pooled, pooled].  Arithmetic follows the reference statement by
// statement (float data, the double-typed literals it mixes in, C round()); compiled with -ffp-contract=off so the forward pass is
// bit-exact against oracle/deform.py.  The backward pass scatters with fp32 atomics exactly as the reference does.
#include "common.h"

namespace {

struct PsroiArgs {
  int R, C, H, W, output_dim, group_size, pooled, part_size, sample_per_part, num_classes, no_trans;
  float spatial_scale, trans_std;
};

struct PsroiBin {
  int roi_batch_ind, part_h, part_w, class_id, gw, gh;
  float roi_width, roi_height, wstart, hstart, sub_w, sub_h, trans_x, trans_y;
};

__device__ __forceinline__ PsroiBin psroi_bin(const PsroiArgs& a, const float* __restrict__ rois, const float* __restrict__ trans, int n, int ctop,
                                              int ph, int pw, bool backward) {
  PsroiBin b;
  const float* r = rois + n * 5;
  b.roi_batch_ind = (int)r[0];
  const float roi_start_w = (float)((double)((float)round(r[1]) * a.spatial_scale) - 0.5);
  const float roi_start_h = (float)((double)((float)round(r[2]) * a.spatial_scale) - 0.5);
  const float roi_end_w = (float)((double)((float)(round((double)r[3]) + 1.) * a.spatial_scale) - 0.5);
  const float roi_end_h = (float)((double)((float)(round((double)r[4]) + 1.) * a.spatial_scale) - 0.5);
  if (backward) {                                    // :225-226 compare against float(0.1), the forward (:78-79) against the double 0.1
    b.roi_width = fmaxf(roi_end_w - roi_start_w, 0.1f);
    b.roi_height = fmaxf(roi_end_h - roi_start_h, 0.1f);
  } else {
    b.roi_width = (float)fmax((double)(roi_end_w - roi_start_w), 0.1);
    b.roi_height = (float)fmax((double)(roi_end_h - roi_start_h), 0.1);
  }
  const float bin_h = b.roi_height / (float)a.pooled, bin_w = b.roi_width / (float)a.pooled;
  b.sub_h = bin_h / (float)a.sample_per_part;
  b.sub_w = bin_w / (float)a.sample_per_part;
  b.part_h = (int)floorf((float)ph / (float)a.pooled * (float)a.part_size);
  b.part_w = (int)floorf((float)pw / (float)a.pooled * (float)a.part_size);
  const int channels_each_class = a.output_dim / a.num_classes;
  b.class_id = ctop / channels_each_class;
  b.trans_x = a.no_trans ? 0.f : trans[(((n * a.num_classes + b.class_id) * 2) * a.part_size + b.part_h) * a.part_size + b.part_w] * a.trans_std;
  b.trans_y = a.no_trans ? 0.f : trans[(((n * a.num_classes + b.class_id) * 2 + 1) * a.part_size + b.part_h) * a.part_size + b.part_w] * a.trans_std;
  b.wstart = (float)pw * bin_w + roi_start_w;
  b.wstart += b.trans_x * b.roi_width;
  b.hstart = (float)ph * bin_h + roi_start_h;
  b.hstart += b.trans_y * b.roi_height;
  int gw = (int)floorf((float)pw * (float)a.group_size / (float)a.pooled);
  int gh = (int)floorf((float)ph * (float)a.group_size / (float)a.pooled);
  b.gw = min(max(gw, 0), a.group_size - 1);
  b.gh = min(max(gh, 0), a.group_size - 1);
  return b;
}

// sample position (:108-117 / :262-269): false when the sample lies outside the half-pixel border
__device__ __forceinline__ bool psroi_sample(const PsroiArgs& a, const PsroiBin& b, int ih, int iw, float& w, float& h) {
  w = b.wstart + (float)iw * b.sub_w;
  h = b.hstart + (float)ih * b.sub_h;
  if ((double)w < -0.5 || (double)w > (double)a.W - 0.5 || (double)h < -0.5 || (double)h > (double)a.H - 0.5) return false;
  w = (float)fmin(fmax((double)w, 0.), (double)a.W - 1.);
  h = (float)fmin(fmax((double)h, 0.), (double)a.H - 1.);
  return true;
}

__global__ void psroi_fwd_kernel(const float* __restrict__ data, const float* __restrict__ rois, const float* __restrict__ trans, PsroiArgs a,
                                 float* __restrict__ top, float* __restrict__ top_count) {
  const long count = (long)a.R * a.output_dim * a.pooled * a.pooled;
  for (long index = (long)blockIdx.x * blockDim.x + threadIdx.x; index < count; index += (long)gridDim.x * blockDim.x) {
    const int pw = (int)(index % a.pooled), ph = (int)((index / a.pooled) % a.pooled);
    const int ctop = (int)((index / a.pooled / a.pooled) % a.output_dim), n = (int)(index / a.pooled / a.pooled / a.output_dim);
    const PsroiBin b = psroi_bin(a, rois, trans, n, ctop, ph, pw, false);
    const float* base = data + (long)b.roi_batch_ind * a.C * a.H * a.W;
    float sum = 0.f;
    int cnt = 0;
    for (int ih = 0; ih < a.sample_per_part; ++ih)
      for (int iw = 0; iw < a.sample_per_part; ++iw) {
        float w, h;
        if (!psroi_sample(a, b, ih, iw, w, h)) continue;
        const int c = (ctop * a.group_size + b.gh) * a.group_size + b.gw;
        const float* d = base + (long)c * a.H * a.W;
        const int x1 = (int)floorf(w), x2 = (int)ceilf(w), y1 = (int)floorf(h), y2 = (int)ceilf(h);
        const float dx = w - (float)x1, dy = h - (float)y1;
        const float v11 = d[y1 * a.W + x1], v12 = d[y2 * a.W + x1], v21 = d[y1 * a.W + x2], v22 = d[y2 * a.W + x2];
        const float val = (1 - dx) * (1 - dy) * v11 + (1 - dx) * dy * v12 + dx * (1 - dy) * v21 + dx * dy * v22;      // :42-43
        sum += val;
        ++cnt;
      }
    top[index] = cnt == 0 ? 0.f : sum / (float)cnt;
    top_count[index] = (float)cnt;
  }
}

__global__ void psroi_bwd_kernel(const float* __restrict__ top_diff, const float* __restrict__ top_count, const float* __restrict__ data,
                                 const float* __restrict__ rois, const float* __restrict__ trans, PsroiArgs a, float* __restrict__ data_diff,
                                 float* __restrict__ trans_diff) {
  const long count = (long)a.R * a.output_dim * a.pooled * a.pooled;
  for (long index = (long)blockIdx.x * blockDim.x + threadIdx.x; index < count; index += (long)gridDim.x * blockDim.x) {
    const int pw = (int)(index % a.pooled), ph = (int)((index / a.pooled) % a.pooled);
    const int ctop = (int)((index / a.pooled / a.pooled) % a.output_dim), n = (int)(index / a.pooled / a.pooled / a.output_dim);
    if (top_count[index] <= 0) continue;
    const PsroiBin b = psroi_bin(a, rois, trans, n, ctop, ph, pw, true);
    const float diff_val = top_diff[index] / top_count[index];
    const long boff = (long)b.roi_batch_ind * a.C * a.H * a.W;
    for (int ih = 0; ih < a.sample_per_part; ++ih)
      for (int iw = 0; iw < a.sample_per_part; ++iw) {
        float w, h;
        if (!psroi_sample(a, b, ih, iw, w, h)) continue;
        const int c = (ctop * a.group_size + b.gh) * a.group_size + b.gw;
        const int x0 = (int)floorf(w), x1 = (int)ceilf(w), y0 = (int)floorf(h), y1 = (int)ceilf(h);
        const float dx = w - (float)x0, dy = h - (float)y0;
        const long cb = boff + (long)c * a.H * a.W;
        atomicAdd(data_diff + cb + y0 * a.W + x0, (1 - dx) * (1 - dy) * diff_val);
        atomicAdd(data_diff + cb + y1 * a.W + x0, (1 - dx) * dy * diff_val);
        atomicAdd(data_diff + cb + y0 * a.W + x1, dx * (1 - dy) * diff_val);
        atomicAdd(data_diff + cb + y1 * a.W + x1, dx * dy * diff_val);
        if (a.no_trans) continue;
        const float U00 = data[cb + y0 * a.W + x0], U01 = data[cb + y1 * a.W + x0], U10 = data[cb + y0 * a.W + x1], U11 = data[cb + y1 * a.W + x1];
        float gx = (U11 * dy + U10 * (1 - dy) - U01 * dy - U00 * (1 - dy)) * a.trans_std * diff_val;
        gx *= b.roi_width;
        float gy = (U11 * dx + U01 * (1 - dx) - U10 * dx - U00 * (1 - dx)) * a.trans_std * diff_val;
        gy *= b.roi_height;
        atomicAdd(trans_diff + (((n * a.num_classes + b.class_id) * 2) * a.part_size + b.part_h) * a.part_size + b.part_w, gx);
        atomicAdd(trans_diff + (((n * a.num_classes + b.class_id) * 2 + 1) * a.part_size + b.part_h) * a.part_size + b.part_w, gy);
      }
  }
}

int psroi_check(PsroiArgs* a, int R, int C, int H, int W, int output_dim, int group_size, int pooled, int part_size, int spp, float scale,
                float trans_std, int no_trans, int num_classes, const char* who) {
  DH_REQUIRE(R > 0 && C > 0 && H > 0 && W > 0 && output_dim > 0 && group_size > 0 && pooled > 0 && part_size >= 0 && spp > 0, DANHIP_EINVAL,
             "%s: non-positive dimension / attribute", who);
  DH_REQUIRE(num_classes > 0 && output_dim % num_classes == 0, DANHIP_EINVAL, "%s: output_dim %d not divisible by num_classes %d", who, output_dim, num_classes);
  DH_REQUIRE(C >= output_dim * group_size * group_size, DANHIP_EINVAL, "%s: %d channels < output_dim * group_size^2 = %d", who, C,
             output_dim * group_size * group_size);
  a->R = R; a->C = C; a->H = H; a->W = W; a->output_dim = output_dim; a->group_size = group_size; a->pooled = pooled; a->part_size = part_size;
  a->sample_per_part = spp; a->num_classes = num_classes; a->no_trans = no_trans; a->spatial_scale = scale; a->trans_std = trans_std;
  return DANHIP_OK;
}

inline int psroi_grid(long total) {
  long b = (total + 255) / 256;
  return (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}

}  // namespace

extern "C" int danhip_deform_psroi_pool_fwd(const float* data, const float* rois, const float* trans, float* top_data, float* mapping_channel,
                                            int32_t R, int32_t C, int32_t H, int32_t W, int32_t output_dim, int32_t group_size, int32_t pooled_size,
                                            int32_t part_size, int32_t sample_per_part, float spatial_scale, float trans_std, int32_t no_trans,
                                            int32_t num_classes, void* stream) {
  DH_REQUIRE(data && rois && top_data && mapping_channel && (no_trans || trans), DANHIP_EINVAL, "deform_psroi_pool_fwd: null pointer");
  PsroiArgs a;
  int rc = psroi_check(&a, R, C, H, W, output_dim, group_size, pooled_size, part_size, sample_per_part, spatial_scale, trans_std, no_trans, num_classes,
                       "deform_psroi_pool_fwd");
  if (rc) return rc;
  hipLaunchKernelGGL(psroi_fwd_kernel, dim3(psroi_grid((long)R * output_dim * pooled_size * pooled_size)), dim3(256), 0, (hipStream_t)stream, data,
                     rois, trans, a, top_data, mapping_channel);
  DH_LAUNCH_CHECK();
  return DANHIP_OK;
}

/* data_diff fp32 [B,C,H,W] and trans_diff fp32 (trans's shape) are zeroed inside, then accumulated with fp32 atomics. */
extern "C" int danhip_deform_psroi_pool_bwd(const float* top_diff, const float* mapping_channel, const float* data, const float* rois,
                                            const float* trans, float* data_diff, float* trans_diff, int32_t B, int32_t R, int32_t C, int32_t H,
                                            int32_t W, int32_t output_dim, int32_t group_size, int32_t pooled_size, int32_t part_size,
                                            int32_t sample_per_part, float spatial_scale, float trans_std, int32_t no_trans, int32_t num_classes,
                                            void* stream) {
  DH_REQUIRE(top_diff && mapping_channel && data && rois && data_diff && (no_trans || (trans && trans_diff)) && B > 0, DANHIP_EINVAL,
             "deform_psroi_pool_bwd: null pointer");
  PsroiArgs a;
  int rc = psroi_check(&a, R, C, H, W, output_dim, group_size, pooled_size, part_size, sample_per_part, spatial_scale, trans_std, no_trans, num_classes,
                       "deform_psroi_pool_bwd");
  if (rc) return rc;
  hipStream_t s = (hipStream_t)stream;
  rc = danhip_zero_async(data_diff, sizeof(float) * (size_t)B * C * H * W, s);
  if (rc) return rc;
  if (!no_trans) {
    rc = danhip_zero_async(trans_diff, sizeof(float) * (size_t)R * 2 * num_classes * part_size * part_size, s);
    if (rc) return rc;
  }
  hipLaunchKernelGGL(psroi_bwd_kernel, dim3(psroi_grid((long)R * output_dim * pooled_size * pooled_size)), dim3(256), 0, s, top_diff, mapping_channel,
                     data, rois, trans, a, data_diff, trans_diff);
  DH_LAUNCH_CHECK();
  return DANHIP_OK;
}

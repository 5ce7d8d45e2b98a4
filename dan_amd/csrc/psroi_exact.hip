// Deformable position-sensitive ROI pooling (cpp/Deform/deform_psroi_pooling_op.cc:37-97 op defs, deform_psroi_pooling_op_gpu.cu:47-125
// forward, :187-300 backward; Python names utility/custom_op.py:93-126) — SURVEY §8f row 4.  Same tensors and attributes as the TF op:
// data fp32 NCHW [B,C,H,W], rois fp32 [R,5] (batch index, x1, y1, x2, y2), trans fp32 [R, 2*num_classes, part, part];
// top_data / mapping_channel (= sample count) fp32 [R, output_dim, pooled, pooled].  The per-sample arithmetic (psroi_bin, psroi_sample,
// the bilinear term) follows the reference expression by expression (float data, the double-typed literals it mixes in, C round());
// compiled with -ffp-contract=off so the forward pass is bit-exact against oracle/deform.py.  The work decomposition is this file's own
// (see above the kernels): wave per (roi, bin, class) with channels in the lanes, atomic-free deterministic shift gradient.
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

// ---- work decomposition (MI355X): the geometry of a bin — ROI corners, bin size, the learned shift of its part cell, its group cell —
// depends on (roi, bin, class) only, while the op's channels map onto planes that nothing else shares (c = (ctop*G + gh)*G + gw).  So:
//   forward : ONE WAVE per (roi, bin, class); the lanes are the class's output channels.  The geometry is wave-uniform, the sample loop has
//             no divergence, and a lane adds its samples in the reference's (ih, iw) order — the pooled value stays bit-exact.
//   backward: ONE WAVE per (roi, class, part cell); it walks the bins that share the cell's shift, lanes = channels again.  The shift
//             gradient of the cell is then a per-lane sum followed by one wave reduction and ONE plain store: no atomics and no zero-fill
//             for trans_diff, and the same bits on every run (the reference issues two atomics per sample and channel).  data_diff keeps
//             the atomics — ROIs overlap arbitrarily — as in the reference.
__device__ __forceinline__ float wave_sum64(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

__global__ __launch_bounds__(256) void psroi_fwd_kernel(const float* __restrict__ data, const float* __restrict__ rois, const float* __restrict__ trans,
                                                        PsroiArgs a, float* __restrict__ top, float* __restrict__ top_count) {
  const int lane = threadIdx.x & 63;
  const int cec = a.output_dim / a.num_classes;
  const long units = (long)a.R * a.pooled * a.pooled * a.num_classes;
  for (long u = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); u < units; u += (long)gridDim.x * (blockDim.x >> 6)) {
    const int cls = (int)(u % a.num_classes);
    long r = u / a.num_classes;
    const int pw = (int)(r % a.pooled);
    r /= a.pooled;
    const int ph = (int)(r % a.pooled), n = (int)(r / a.pooled);
    const PsroiBin b = psroi_bin(a, rois, trans, n, cls * cec, ph, pw, false);
    const float* base = data + (long)b.roi_batch_ind * a.C * a.H * a.W;
    for (int k = lane; k < cec; k += 64) {
      const int ctop = cls * cec + k;
      const float* d = base + (long)((ctop * a.group_size + b.gh) * a.group_size + b.gw) * a.H * a.W;
      float sum = 0.f;
      int cnt = 0;
      for (int ih = 0; ih < a.sample_per_part; ++ih)
        for (int iw = 0; iw < a.sample_per_part; ++iw) {
          float w, h;
          if (!psroi_sample(a, b, ih, iw, w, h)) continue;              // (wave-uniform)
          const int x1 = (int)floorf(w), x2 = (int)ceilf(w), y1 = (int)floorf(h), y2 = (int)ceilf(h);
          const float dx = w - (float)x1, dy = h - (float)y1;
          const float v11 = d[y1 * a.W + x1], v12 = d[y2 * a.W + x1], v21 = d[y1 * a.W + x2], v22 = d[y2 * a.W + x2];
          sum += (1 - dx) * (1 - dy) * v11 + (1 - dx) * dy * v12 + dx * (1 - dy) * v21 + dx * dy * v22;      // :42-43
          ++cnt;
        }
      const long index = (((long)n * a.output_dim + ctop) * a.pooled + ph) * a.pooled + pw;
      top[index] = cnt == 0 ? 0.f : sum / (float)cnt;
      top_count[index] = (float)cnt;
    }
  }
}

__global__ __launch_bounds__(256) void psroi_bwd_kernel(const float* __restrict__ top_diff, const float* __restrict__ top_count,
                                                        const float* __restrict__ data, const float* __restrict__ rois, const float* __restrict__ trans,
                                                        PsroiArgs a, float* __restrict__ data_diff, float* __restrict__ trans_diff) {
  const int lane = threadIdx.x & 63;
  const int cec = a.output_dim / a.num_classes;
  const int cells = a.no_trans ? a.pooled : a.part_size;                // cell grid: the part cells, or the bins themselves without shifts
  const long units = (long)a.R * a.num_classes * cells * cells;
  for (long u = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); u < units; u += (long)gridDim.x * (blockDim.x >> 6)) {
    const int cw = (int)(u % cells);
    long r = u / cells;
    const int chh = (int)(r % cells);
    r /= cells;
    const int cls = (int)(r % a.num_classes), n = (int)(r / a.num_classes);
    float gx_sum = 0.f, gy_sum = 0.f;
    for (int ph = 0; ph < a.pooled; ++ph) {
      const int cell_h = a.no_trans ? ph : (int)floorf((float)ph / (float)a.pooled * (float)a.part_size);
      if (cell_h != chh) continue;
      for (int pw = 0; pw < a.pooled; ++pw) {
        const int cell_w = a.no_trans ? pw : (int)floorf((float)pw / (float)a.pooled * (float)a.part_size);
        if (cell_w != cw) continue;
        const PsroiBin b = psroi_bin(a, rois, trans, n, cls * cec, ph, pw, true);
        const long boff = (long)b.roi_batch_ind * a.C * a.H * a.W;
        for (int k = lane; k < cec; k += 64) {
          const int ctop = cls * cec + k;
          const long index = (((long)n * a.output_dim + ctop) * a.pooled + ph) * a.pooled + pw;
          if (top_count[index] <= 0) continue;
          const float diff_val = top_diff[index] / top_count[index];
          const long cb = boff + (long)((ctop * a.group_size + b.gh) * a.group_size + b.gw) * a.H * a.W;
          for (int ih = 0; ih < a.sample_per_part; ++ih)
            for (int iw = 0; iw < a.sample_per_part; ++iw) {
              float w, h;
              if (!psroi_sample(a, b, ih, iw, w, h)) continue;
              const int x0 = (int)floorf(w), x1 = (int)ceilf(w), y0 = (int)floorf(h), y1 = (int)ceilf(h);
              const float dx = w - (float)x0, dy = h - (float)y0;
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
              gx_sum += gx;
              gy_sum += gy;
            }
        }
      }
    }
    if (a.no_trans) continue;
    gx_sum = wave_sum64(gx_sum);
    gy_sum = wave_sum64(gy_sum);
    if (lane == 0) {
      trans_diff[(((long)(n * a.num_classes + cls) * 2) * a.part_size + chh) * a.part_size + cw] = gx_sum;
      trans_diff[(((long)(n * a.num_classes + cls) * 2 + 1) * a.part_size + chh) * a.part_size + cw] = gy_sum;
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
  hipLaunchKernelGGL(psroi_fwd_kernel, dim3(psroi_grid((long)R * pooled_size * pooled_size * num_classes * 64)), dim3(256), 0, (hipStream_t)stream, data,
                     rois, trans, a, top_data, mapping_channel);
  DH_LAUNCH_CHECK();
  return DANHIP_OK;
}

/* data_diff fp32 [B,C,H,W] is zeroed inside, then accumulated with fp32 atomics; trans_diff fp32 (trans's shape) is written once per cell. */
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
  const int cells = no_trans ? pooled_size : part_size;                 // (every trans_diff cell is written exactly once: no zero-fill)
  DH_REQUIRE(cells > 0, DANHIP_EINVAL, "deform_psroi_pool_bwd: part_size must be positive when shifts are learned");
  hipLaunchKernelGGL(psroi_bwd_kernel, dim3(psroi_grid((long)R * num_classes * cells * cells * 64)), dim3(256), 0, s, top_diff, mapping_channel,
                     data, rois, trans, a, data_diff, trans_diff);
  DH_LAUNCH_CHECK();
  return DANHIP_OK;
}

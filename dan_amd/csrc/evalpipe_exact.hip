// Test-time pipeline kernels of eval_dan.py / eval_sfd.py, compiled with -ffp-contract=off:
//   * resize_u8_linear : cv2.resize(image, None, None, fx, fy, INTER_LINEAR) on an 8-bit HWC image (eval_dan.py:96-97) — the
//                        generic fixed-point path of OpenCV's resize (11-bit coefficients, two integer passes), restated in
//                        oracle/evalpipe.py:cv2_resize_linear_u8.
//   * bbox_vote        : the score-weighted box voting of eval_dan.py:201-241 (IoU with the +1 convention, merge >= threshold,
//                        singleton clusters dropped, float64 arithmetic as numpy does it, float32 results).
#include "common.h"

namespace {

inline int grid_for(long total, int block, int cap = 4096) {
  long g = (total + block - 1) / block;
  return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

__device__ __forceinline__ int cv_round_to_short(float v) {      // saturate_cast<short>(float) = cvRound (half to even) + clamp
  int r = __float2int_rn(v);
  return r < -32768 ? -32768 : (r > 32767 ? 32767 : r);
}

// one thread per (dy, dx, c): both passes recomputed from the four source bytes (the integer formulas make that exact)
__global__ void resize_u8_linear_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, int H, int W, int Ho, int Wo, int C,
                                        double scale_x, double scale_y) {
  const long total = (long)Ho * Wo;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int dx = (int)(idx % Wo), dy = (int)(idx / Wo);
    float fx = (float)(((double)dx + 0.5) * scale_x - 0.5);
    int sx = (int)floorf(fx);
    fx -= (float)sx;
    bool xedge = false;                                          // dx >= xmax: horizontal pass copies S[sx] * ONE
    if (sx < 0) { fx = 0.f; sx = 0; }
    if (sx >= W - 1) { fx = 0.f; sx = W - 1; xedge = true; }
    float fy = (float)(((double)dy + 0.5) * scale_y - 0.5);
    int sy = (int)floorf(fy);
    fy -= (float)sy;
    const int a0 = cv_round_to_short((1.f - fx) * 2048.f), a1 = cv_round_to_short(fx * 2048.f);
    const int b0 = cv_round_to_short((1.f - fy) * 2048.f), b1 = cv_round_to_short(fy * 2048.f);
    int sy0 = sy, sy1 = sy + 1;                                  // rows are clamped (BORDER_REPLICATE), coefficients kept
    sy0 = sy0 < 0 ? 0 : (sy0 > H - 1 ? H - 1 : sy0);
    sy1 = sy1 < 0 ? 0 : (sy1 > H - 1 ? H - 1 : sy1);
    const int sx1 = xedge ? sx : sx + 1;
    for (int c = 0; c < C; ++c) {
      const int p00 = src[((long)sy0 * W + sx) * C + c], p01 = src[((long)sy0 * W + sx1) * C + c];
      const int p10 = src[((long)sy1 * W + sx) * C + c], p11 = src[((long)sy1 * W + sx1) * C + c];
      const int r0 = xedge ? p00 * 2048 : p00 * a0 + p01 * a1;
      const int r1 = xedge ? p10 * 2048 : p10 * a0 + p11 * a1;
      const int v = (((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2;
      dst[idx * C + c] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
    }
  }
}

// ---------------------------------------------------------------------------------------------------------- bbox_vote
constexpr int kVoteThreads = 1024;

struct VoteAcc {
  double sx1, sy1, sx2, sy2, ss, smax;
  int cnt;
};

__device__ __forceinline__ void vote_combine(VoteAcc& a, const VoteAcc& b) {
  a.sx1 += b.sx1; a.sy1 += b.sy1; a.sx2 += b.sx2; a.sy2 += b.sy2; a.ss += b.ss;
  a.smax = b.smax > a.smax ? b.smax : a.smax;
  a.cnt += b.cnt;
}

__device__ __forceinline__ double shfl_down_f64(double v, int off) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __shfl_down(lo, off);
  hi = __shfl_down(hi, off);
  return __hiloint2double(hi, lo);
}

// one block per image; det rows (xmin, ymin, xmax, ymax, score) float64, already ordered as eval_dan.py:202-203 orders them
__global__ void __launch_bounds__(kVoteThreads) bbox_vote_kernel(const double* __restrict__ det_all, const int* __restrict__ counts, int Nmax,
                                                                double thr, int max_out, float* __restrict__ out_all, int* __restrict__ num_out,
                                                                uint8_t* __restrict__ alive_all) {
  const int b = blockIdx.x;
  const double* det = det_all + (long)b * Nmax * 5;
  uint8_t* alive = alive_all + (long)b * Nmax;
  float* out = out_all + (long)b * max_out * 5;
  const int N = counts[b];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  __shared__ VoteAcc red[kVoteThreads / 64];
  __shared__ int s_head;
  for (int i = tid; i < N; i += kVoteThreads) alive[i] = 1;
  int produced = 0, head = 0;
  __syncthreads();
  while (produced < max_out) {
    if (tid == 0) {
      int h = head;
      while (h < N && !alive[h]) ++h;
      s_head = h;
    }
    __syncthreads();
    head = s_head;
    if (head >= N) break;
    const double hx1 = det[head * 5], hy1 = det[head * 5 + 1], hx2 = det[head * 5 + 2], hy2 = det[head * 5 + 3];
    const double harea = (hx2 - hx1 + 1) * (hy2 - hy1 + 1);
    VoteAcc acc = {0., 0., 0., 0., 0., -INFINITY, 0};
    for (int i = head + tid; i < N; i += kVoteThreads) {
      if (!alive[i]) continue;
      const double x1 = det[i * 5], y1 = det[i * 5 + 1], x2 = det[i * 5 + 2], y2 = det[i * 5 + 3], s = det[i * 5 + 4];
      const double area = (x2 - x1 + 1) * (y2 - y1 + 1);
      const double xx1 = fmax(hx1, x1), yy1 = fmax(hy1, y1), xx2 = fmin(hx2, x2), yy2 = fmin(hy2, y2);
      const double w = fmax(0.0, xx2 - xx1 + 1), h = fmax(0.0, yy2 - yy1 + 1);
      const double inter = w * h;
      const double o = inter / (harea + area - inter);
      if (o >= thr) {
        alive[i] = 0;
        acc.sx1 += x1 * s; acc.sy1 += y1 * s; acc.sx2 += x2 * s; acc.sy2 += y2 * s; acc.ss += s;
        acc.smax = s > acc.smax ? s : acc.smax;
        acc.cnt += 1;
      }
    }
    for (int off = 32; off > 0; off >>= 1) {
      VoteAcc o;
      o.sx1 = shfl_down_f64(acc.sx1, off); o.sy1 = shfl_down_f64(acc.sy1, off); o.sx2 = shfl_down_f64(acc.sx2, off);
      o.sy2 = shfl_down_f64(acc.sy2, off); o.ss = shfl_down_f64(acc.ss, off); o.smax = shfl_down_f64(acc.smax, off);
      o.cnt = __shfl_down(acc.cnt, off);
      vote_combine(acc, o);
    }
    if (lane == 0) red[wid] = acc;
    __syncthreads();
    if (tid == 0) {
      VoteAcc t = red[0];
      for (int w = 1; w < kVoteThreads / 64; ++w) vote_combine(t, red[w]);
      if (t.cnt == 0) alive[head] = 0;                           // eval_dan.py:221-222: nothing merged (degenerate head) -> drop the head
      if (t.cnt >= 2) {                                          // :223-229 singletons are discarded
        float* o = out + produced * 5;
        o[0] = (float)(t.sx1 / t.ss); o[1] = (float)(t.sy1 / t.ss); o[2] = (float)(t.sx2 / t.ss); o[3] = (float)(t.sy2 / t.ss);
        o[4] = (float)t.smax;
      }
      red[0].cnt = t.cnt;
    }
    __syncthreads();
    if (red[0].cnt >= 2) ++produced;
    __syncthreads();                                             // red[] / alive[] settled before the next round
  }
  if (tid == 0) num_out[b] = produced;
}

}  // namespace

extern "C" int danhip_resize_u8_linear(const uint8_t* src, int32_t H, int32_t W, uint8_t* dst, int32_t Ho, int32_t Wo, int32_t C, double fx,
                                       double fy, void* stream) {
  DH_REQUIRE(src && dst && H > 0 && W > 0 && Ho > 0 && Wo > 0 && C > 0 && fx > 0 && fy > 0, DANHIP_EINVAL, "resize_u8_linear: bad arguments");
  const long total = (long)Ho * Wo;
  hipLaunchKernelGGL(resize_u8_linear_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, src, dst, H, W, Ho, Wo, C, 1. / fx,
                     1. / fy);
  DH_LAUNCH_CHECK();
  return DANHIP_OK;
}

extern "C" size_t danhip_bbox_vote_workspace_bytes(int32_t B, int32_t Nmax) { return (size_t)B * (size_t)Nmax; }

extern "C" int danhip_bbox_vote(const double* det, const int32_t* counts, int32_t B, int32_t Nmax, double iou_threshold, int32_t max_out,
                                float* out, int32_t* num_out, void* workspace, size_t workspace_bytes, void* stream) {
  DH_REQUIRE(det && counts && out && num_out && workspace && B > 0 && Nmax > 0 && max_out > 0, DANHIP_EINVAL, "bbox_vote: bad arguments");
  DH_REQUIRE(workspace_bytes >= danhip_bbox_vote_workspace_bytes(B, Nmax), DANHIP_EWORKSPACE, "bbox_vote: workspace too small");
  hipLaunchKernelGGL(bbox_vote_kernel, dim3(B), dim3(kVoteThreads), 0, (hipStream_t)stream, det, counts, Nmax, iou_threshold, max_out, out, num_out,
                     (uint8_t*)workspace);
  DH_LAUNCH_CHECK();
  return DANHIP_OK;
}

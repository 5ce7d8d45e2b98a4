// DynamicAnchorRouting (cpp/ExtraLib/dynamic_anchor_routing.cc:188-518) and greedy NMS (tf.image.non_max_suppression as
// called from utility/bbox_util.py:77,82) as HIP index/compare kernels.  Compiled with -ffp-contract=off; arithmetic types
// (float vs double promotions) follow the reference statement by statement so results are bit-exact against
// oracle/extra_lib.cpp (exp/log excepted: device libm, <= 1 ulp).
//
// The reference kernel is a single sequential loop whose result depends on visiting order.  Parallel restatement:
//   eval : a source i may claim cell n iff it is valid and (n is not easy-background or i < n); the winner is the claimed
//          source with the largest label, ties -> smallest index, and only labels > 0 ever win (prior starts at 0)
//          => one 64-bit atomicMax per source on key (label bits << 32 | ~i), then a per-cell decode.
//   train: per-cell reservoir sampling consumes the sources of a cell in index order
//          => bucket the eligible sources per cell (count, scan, scatter), sort each small bucket by index, replay.
#include "common.h"

namespace {

// splitmix64 stream shared bit-exactly with oracle_uniform (oracle/extra_lib.cpp)
__device__ __forceinline__ double uniform01(unsigned long long seed, unsigned long long counter) {
  unsigned long long z = seed + 0x9E3779B97F4A7C15ull * (counter + 1);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z = z ^ (z >> 31);
  return (double)(z >> 11) * (1.0 / 9007199254740992.0);
}

// dynamic_anchor_routing.cc:262-279 / :351-371 — cell of the box centre; false = skip this source
__device__ __forceinline__ bool route_cell(float ymin, float xmin, float ymax, float xmax, int fh, int fw, int depth, int stride, long index,
                                           long* cell) {
  long cx = (long)round((double)(xmin + xmax) / (2. * (double)stride));
  if (xmin / (float)stride < -1 || (double)(xmax / (float)stride) > (double)fw + 1 - 1.) return false;
  cx = min(cx, (long)(fw - 1));
  cx = max(cx, 0l);
  long cy = (long)round((double)(ymin + ymax) / (2. * (double)stride));
  if (ymin / (float)stride < -1 || (double)(ymax / (float)stride) > (double)fh + 1 - 1.) return false;
  cy = min(cy, (long)(fh - 1));
  cy = max(cy, 0l);
  *cell = (cy * fw + cx) * depth + index % depth;
  return true;
}

// ------------------------------------------------------------------------------------------------ eval mode
__global__ void route_eval_claim_kernel(const float* __restrict__ anchors, const float* __restrict__ labels, const int* __restrict__ mask_in,
                                        unsigned long long* __restrict__ key, long N, int fh, int fw, int depth, int stride) {
  const long b = blockIdx.y;
  anchors += b * N * 4; labels += b * N; mask_in += b * N; key += b * N;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (long)gridDim.x * blockDim.x) {
    if (mask_in[i] < 1) continue;
    const float ymin = anchors[i * 4], xmin = anchors[i * 4 + 1], ymax = anchors[i * 4 + 2], xmax = anchors[i * 4 + 3];
    if (xmax - xmin < 1 || ymax - ymin < 1) continue;
    long n;
    if (!route_cell(ymin, xmin, ymax, xmax, fh, fw, depth, stride, i, &n)) continue;
    const float l = labels[i];
    if (!(l > 0.f)) continue;                                   // prior_prob starts at 0: only labels > 0 can ever win
    if (mask_in[n] < 1 && !(i < n)) continue;                   // an easy-background cell can only be claimed before its own turn
    const unsigned long long k = ((unsigned long long)__float_as_uint(l) << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)i);
    atomicMax(key + n, k);
  }
}

__global__ void route_eval_decode_kernel(const float* __restrict__ anchors, const float* __restrict__ gt_targets, const int* __restrict__ mask_in,
                                         const unsigned long long* __restrict__ key, int* __restrict__ mask_out, float* __restrict__ decode_out,
                                         long N) {
  const long b = blockIdx.y;
  anchors += b * N * 4; gt_targets += b * N * 4; mask_in += b * N; key += b * N; mask_out += b * N; decode_out += b * N * 4;
  for (long n = (long)blockIdx.x * blockDim.x + threadIdx.x; n < N; n += (long)gridDim.x * blockDim.x) {
    const unsigned long long k = key[n];
    float ymin = 0.f, xmin = 0.f, ymax = 0.f, xmax = 0.f;
    int m = 0;
    if (k != 0ull) {
      const long i = (long)(0xFFFFFFFFu - (unsigned)(k & 0xFFFFFFFFull));
      ymin = anchors[i * 4]; xmin = anchors[i * 4 + 1]; ymax = anchors[i * 4 + 2]; xmax = anchors[i * 4 + 3];
      m = mask_in[n] < 1 ? 0 : 1;
    }
    mask_out[n] = m;
    // dynamic_anchor_routing.cc:382-408 (float variables, double literals)
    const float pcy = (float)((double)(ymin + ymax) / 2.);
    const float pcx = (float)((double)(xmin + xmax) / 2.);
    const float ph = (float)((double)(ymax - ymin) + 1.);
    const float pw = (float)((double)(xmax - xmin) + 1.);
    float ty = gt_targets[n * 4], tx = gt_targets[n * 4 + 1], th = gt_targets[n * 4 + 2], tw = gt_targets[n * 4 + 3];
    th = expf(th) * ph;
    tw = expf(tw) * pw;
    ty = ty * ph + pcy;
    tx = tx * pw + pcx;
    decode_out[n * 4] = (float)((double)ty - ((double)th - 1.) / 2.);
    decode_out[n * 4 + 1] = (float)((double)tx - ((double)tw - 1.) / 2.);
    decode_out[n * 4 + 2] = (float)((double)ty + ((double)th - 1.) / 2.);
    decode_out[n * 4 + 3] = (float)((double)tx + ((double)tw - 1.) / 2.);
  }
}

// ------------------------------------------------------------------------------------------------ train mode
// state per source (int): -1 = not a reservoir candidate, else its target cell.
__global__ void route_train_pass1_kernel(const float* __restrict__ anchors, const float* __restrict__ gt, const float* __restrict__ labels,
                                         const int* __restrict__ mask_in, int* __restrict__ matched, int* __restrict__ ignore_flag,
                                         int* __restrict__ cand_cell, int* __restrict__ count, long N, int fh, int fw, int depth, int stride,
                                         float thres, float ignore_thres) {
  const long b = blockIdx.y;
  anchors += b * N * 4; gt += b * N * 4; labels += b * N; mask_in += b * N;
  matched += b * N; ignore_flag += b * N; cand_cell += b * N; count += b * N;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (long)gridDim.x * blockDim.x) {
    const float gy0 = gt[i * 4], gx0 = gt[i * 4 + 1], gy1 = gt[i * 4 + 2], gx1 = gt[i * 4 + 3];
    const float lab = labels[i];
    // ---- pass 1 (:205-243): the cell containing the gt centre is positive
    if (lab > 0.f && !(gx1 - gx0 < 1 || gy1 - gy0 < 1)) {
      long cx = (long)round((double)(gx0 + gx1) / (2. * (double)stride));
      long cy = (long)round((double)(gy0 + gy1) / (2. * (double)stride));
      const bool okx = !(cx < -stride || (double)cx > (double)fw + (double)stride - 1.);
      const bool oky = !(cy < -stride || (double)cy > (double)fh + (double)stride - 1.);
      if (okx && oky) {
        cx = max(min(cx, (long)(fw - 1)), 0l);
        cy = max(min(cy, (long)(fh - 1)), 0l);
        matched[(cy * fw + cx) * depth + i % depth] = 1;                    // benign race: every writer stores 1
      }
    }
    // ---- pass 2 prefix (:245-305): decide ignore flag and reservoir candidacy of source i
    int cell = -1, ign = 0;
    const float ymin = anchors[i * 4], xmin = anchors[i * 4 + 1], ymax = anchors[i * 4 + 2], xmax = anchors[i * 4 + 3];
    if (mask_in[i] < 1) ign = 1;
    else if (xmax - xmin < 1 || ymax - ymin < 1) ign = 1;
    else {
      long n;
      if (route_cell(ymin, xmin, ymax, xmax, fh, fw, depth, stride, i, &n) && lab > 0.f) {
        const float iy0 = fmaxf(ymin, gy0), ix0 = fmaxf(xmin, gx0), iy1 = fminf(ymax, gy1), ix1 = fminf(xmax, gx1);
        const float h = (float)fmax((double)(iy1 - iy0) + 1., 0.);
        const float w = (float)fmax((double)(ix1 - ix0) + 1., 0.);
        const float inter = h * w;
        const float area_a = (float)(((double)(gy1 - gy0) + 1.) * ((double)(gx1 - gx0) + 1.));
        const float area_b = (float)(((double)(ymax - ymin) + 1.) * ((double)(xmax - xmin) + 1.));
        const float uni = area_a + area_b - inter;
        if (!((double)fabsf(uni) <= 1.) && !(inter / uni <= ignore_thres)) {
          if (inter / uni < thres) ign = 1;
          cell = (int)n;
        }
      }
    }
    ignore_flag[i] = ign;
    cand_cell[i] = cell;
    if (cell >= 0) atomicAdd(count + cell, 1);
  }
}

// exclusive scan of count[N] -> offset[N] (one block of 1024 threads per image)
__global__ void route_scan_kernel(const int* __restrict__ count, int* __restrict__ offset, long N) {
  const long b = blockIdx.x;
  count += b * N; offset += b * N;
  __shared__ int part[1024];
  const long per = (N + 1023) / 1024;
  const long s = threadIdx.x * per, e = min(N, s + per);
  int sum = 0;
  for (long i = s; i < e; ++i) sum += count[i];
  part[threadIdx.x] = sum;
  __syncthreads();
  for (int o = 1; o < 1024; o <<= 1) {
    const int v = threadIdx.x >= o ? part[threadIdx.x - o] : 0;
    __syncthreads();
    part[threadIdx.x] += v;
    __syncthreads();
  }
  int run = part[threadIdx.x] - sum;
  for (long i = s; i < e; ++i) { offset[i] = run; run += count[i]; }
}

__global__ void route_scatter_kernel(const int* __restrict__ cand_cell, const int* __restrict__ offset, int* __restrict__ cursor,
                                     int* __restrict__ bucket, long N) {
  const long b = blockIdx.y;
  cand_cell += b * N; offset += b * N; cursor += b * N; bucket += b * N;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (long)gridDim.x * blockDim.x) {
    const int c = cand_cell[i];
    if (c >= 0) bucket[offset[c] + atomicAdd(cursor + c, 1)] = (int)i;
  }
}

// one thread per cell: sort its bucket by source index, replay the reservoir (:306-324), write outputs
__global__ void route_train_cell_kernel(const float* __restrict__ anchors, const float* __restrict__ gt, const int* __restrict__ matched,
                                        const int* __restrict__ ignore_flag, const int* __restrict__ count, const int* __restrict__ offset,
                                        int* __restrict__ bucket, int* __restrict__ mask_out, float* __restrict__ decode_out, long N,
                                        unsigned long long seed, unsigned long long counter0, const unsigned long long* __restrict__ counter_dev) {
  const long b = blockIdx.y;
  if (counter_dev) counter0 += *counter_dev;                            // device-resident part of the counter (replayable launches)
  anchors += b * N * 4; gt += b * N * 4; matched += b * N; ignore_flag += b * N; count += b * N; offset += b * N; bucket += b * N;
  mask_out += b * N; decode_out += b * N * 4;
  for (long n = (long)blockIdx.x * blockDim.x + threadIdx.x; n < N; n += (long)gridDim.x * blockDim.x) {
    int m = matched[n];
    int positive = m;
    float d0 = 0.f, d1 = 0.f, d2 = 0.f, d3 = 0.f;
    const int cnt = count[n];
    int* bk = bucket + offset[n];
    for (int a = 1; a < cnt; ++a) {                                      // insertion sort (buckets are tiny)
      const int v = bk[a];
      int p = a - 1;
      while (p >= 0 && bk[p] > v) { bk[p + 1] = bk[p]; --p; }
      bk[p + 1] = v;
    }
    for (int a = 0; a < cnt; ++a) {
      const long i = bk[a];
      const double u = uniform01(seed, counter0 + (unsigned long long)(b * N + i));
      if (u <= 1. / (double)(m + 1)) {
        m += 1;
        positive = 1;
        const float ymin = anchors[i * 4], xmin = anchors[i * 4 + 1], ymax = anchors[i * 4 + 2], xmax = anchors[i * 4 + 3];
        const float gy0 = gt[i * 4], gx0 = gt[i * 4 + 1], gy1 = gt[i * 4 + 2], gx1 = gt[i * 4 + 3];
        const float pcy = (float)((double)(ymin + ymax) / 2.), pcx = (float)((double)(xmin + xmax) / 2.);
        const float ph = (float)((double)(ymax - ymin) + 1.), pw = (float)((double)(xmax - xmin) + 1.);
        const float gcy = (float)((double)(gy0 + gy1) / 2.), gcx = (float)((double)(gx0 + gx1) / 2.);
        const float gh = (float)((double)(gy1 - gy0) + 1.), gw = (float)((double)(gx1 - gx0) + 1.);
        d0 = (gcy - pcy) / ph;
        d1 = (gcx - pcx) / pw;
        d2 = logf(fmaxf(gh / ph, 1.1920928955078125e-07f));
        d3 = logf(fmaxf(gw / pw, 1.1920928955078125e-07f));
      }
    }
    mask_out[n] = positive ? 1 : (ignore_flag[n] ? -1 : 0);
    decode_out[n * 4] = d0; decode_out[n * 4 + 1] = d1; decode_out[n * 4 + 2] = d2; decode_out[n * 4 + 3] = d3;
  }
}

// ------------------------------------------------------------------------------------------------ NMS
// One 1024-thread block per image over K score-sorted boxes: walk the candidates in order; a kept box suppresses every later
// box with IoU > thr (raw areas, no +1; corners normalised by min/max).  keep_idx gets the kept candidate positions.
__global__ void nms_kernel(const float* __restrict__ boxes, int* __restrict__ keep_idx, int* __restrict__ num_keep, int K, int max_out, float thr) {
  const int b = blockIdx.x;
  boxes += (long)b * K * 4; keep_idx += (long)b * max_out; num_keep += b;
  extern __shared__ unsigned char sup[];                         // suppressed flags
  for (int j = threadIdx.x; j < K; j += blockDim.x) sup[j] = 0;
  __syncthreads();
  int kept = 0;
  for (int i = 0; i < K && kept < max_out; ++i) {
    if (sup[i]) continue;                                        // uniform: sup[i] is final once every j < i has been processed
    if (threadIdx.x == 0) keep_idx[kept] = i;
    ++kept;
    const float by1 = fminf(boxes[i * 4], boxes[i * 4 + 2]), by2 = fmaxf(boxes[i * 4], boxes[i * 4 + 2]);
    const float bx1 = fminf(boxes[i * 4 + 1], boxes[i * 4 + 3]), bx2 = fmaxf(boxes[i * 4 + 1], boxes[i * 4 + 3]);
    const float ai = (by2 - by1) * (bx2 - bx1);
    for (int j = i + 1 + threadIdx.x; j < K; j += blockDim.x) {
      if (sup[j]) continue;
      const float y1 = fminf(boxes[j * 4], boxes[j * 4 + 2]), y2 = fmaxf(boxes[j * 4], boxes[j * 4 + 2]);
      const float x1 = fminf(boxes[j * 4 + 1], boxes[j * 4 + 3]), x2 = fmaxf(boxes[j * 4 + 1], boxes[j * 4 + 3]);
      const float aj = (y2 - y1) * (x2 - x1);
      float iou = 0.f;
      if (!(ai <= 0.f || aj <= 0.f)) {
        const float ih = fmaxf(fminf(by2, y2) - fmaxf(by1, y1), 0.f);
        const float iw = fmaxf(fminf(bx2, x2) - fmaxf(bx1, x1), 0.f);
        const float inter = ih * iw;
        iou = inter / (ai + aj - inter);
      }
      if (iou > thr) sup[j] = 1;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) *num_keep = kept;
  for (int j = kept + threadIdx.x; j < max_out; j += blockDim.x) keep_idx[j] = -1;
}

inline int grid1d(long n, int block = 256, int cap = 2048) {
  long b = (n + block - 1) / block;
  if (b > cap) b = cap;
  if (b < 1) b = 1;
  return (int)b;
}

int check_route(const void* a, const void* g, const void* l, const void* m, const void* mo, const void* d, int64_t N, int fh, int fw, int depth,
                int stride, int B, const char* what) {
  DH_REQUIRE(a && g && l && m && mo && d, DANHIP_EINVAL, "%s: null pointer", what);
  DH_REQUIRE(B > 0 && N > 0 && fh > 0 && fw > 0 && depth > 0 && stride > 0, DANHIP_EINVAL, "%s: non-positive dims", what);
  DH_REQUIRE((int64_t)fh * fw * depth == N, DANHIP_EINVAL, "%s: N=%lld != feat_height*feat_width*anchor_depth", what, (long long)N);
  DH_REQUIRE(N < (1ll << 31), DANHIP_EINVAL, "%s: N too large", what);
  return DANHIP_OK;
}

}  // namespace

extern "C" size_t danhip_routing_workspace_bytes(int64_t N, int32_t B, int training) {
  const size_t n = (size_t)N * (size_t)(B > 0 ? B : 1);
  return training ? n * 7 * sizeof(int) : n * sizeof(unsigned long long);
}

extern "C" int danhip_dynamic_anchor_routing_eval(const float* anchors, const float* gt_targets, const float* labels, const int32_t* mask_in,
                                                  int64_t N, int32_t feat_height, int32_t feat_width, int32_t anchor_depth, int32_t feat_strides,
                                                  int32_t B, int32_t* mask_out, float* decode_out, void* workspace, size_t workspace_bytes,
                                                  void* stream) {
  int rc = check_route(anchors, gt_targets, labels, mask_in, mask_out, decode_out, N, feat_height, feat_width, anchor_depth, feat_strides, B,
                       "dynamic_anchor_routing_eval");
  if (rc) return rc;
  DH_REQUIRE(workspace && workspace_bytes >= danhip_routing_workspace_bytes(N, B, 0), DANHIP_EWORKSPACE, "dynamic_anchor_routing_eval: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  unsigned long long* key = reinterpret_cast<unsigned long long*>(workspace);
  { const int zrc = danhip_zero_async(key, sizeof(unsigned long long) * (size_t)N * B, s); if (zrc) return zrc; }
  dim3 grid((unsigned)grid1d(N), (unsigned)B);
  hipLaunchKernelGGL(route_eval_claim_kernel, grid, dim3(256), 0, s, anchors, labels, mask_in, key, (long)N, feat_height, feat_width, anchor_depth,
                     feat_strides);
  DH_LAUNCH_CHECK();
  hipLaunchKernelGGL(route_eval_decode_kernel, grid, dim3(256), 0, s, anchors, gt_targets, mask_in, key, mask_out, decode_out, (long)N);
  DH_LAUNCH_CHECK();
  return DANHIP_OK;
}

/* u(i) of image b, source i = splitmix64(seed, counter0 + b*N + i) — the stream oracle_uniform() replays on the CPU. */
extern "C" int danhip_dynamic_anchor_routing_train(const float* anchors, const float* gt_targets, const float* labels, const int32_t* mask_in,
                                                   int64_t N, int32_t feat_height, int32_t feat_width, int32_t anchor_depth, int32_t feat_strides,
                                                   int32_t B, float thres, float ignore_thres, uint64_t seed, uint64_t counter0,
                                                   const uint64_t* counter_dev, int32_t* mask_out, float* decode_out, void* workspace,
                                                   size_t workspace_bytes, void* stream) {
  int rc = check_route(anchors, gt_targets, labels, mask_in, mask_out, decode_out, N, feat_height, feat_width, anchor_depth, feat_strides, B,
                       "dynamic_anchor_routing_train");
  if (rc) return rc;
  DH_REQUIRE(thres >= 0.f && thres < 1.f && ignore_thres >= 0.f && ignore_thres < 1.f, DANHIP_EINVAL,
             "dynamic_anchor_routing_train: thresholds must be in [0,1) (dynamic_anchor_routing.cc:526-530)");
  DH_REQUIRE(workspace && workspace_bytes >= danhip_routing_workspace_bytes(N, B, 1), DANHIP_EWORKSPACE, "dynamic_anchor_routing_train: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  const size_t n = (size_t)N * B;
  int* ws = reinterpret_cast<int*>(workspace);
  int *matched = ws, *ignore_flag = ws + n, *cand = ws + 2 * n, *count = ws + 3 * n, *offset = ws + 4 * n, *cursor = ws + 5 * n, *bucket = ws + 6 * n;
  { const int zrc = danhip_zero_async(ws, sizeof(int) * 7 * n, s); if (zrc) return zrc; }
  dim3 grid((unsigned)grid1d(N), (unsigned)B);
  hipLaunchKernelGGL(route_train_pass1_kernel, grid, dim3(256), 0, s, anchors, gt_targets, labels, mask_in, matched, ignore_flag, cand, count, (long)N,
                     feat_height, feat_width, anchor_depth, feat_strides, thres, ignore_thres);
  DH_LAUNCH_CHECK();
  hipLaunchKernelGGL(route_scan_kernel, dim3(B), dim3(1024), 0, s, count, offset, (long)N);
  DH_LAUNCH_CHECK();
  hipLaunchKernelGGL(route_scatter_kernel, grid, dim3(256), 0, s, cand, offset, cursor, bucket, (long)N);
  DH_LAUNCH_CHECK();
  hipLaunchKernelGGL(route_train_cell_kernel, grid, dim3(256), 0, s, anchors, gt_targets, matched, ignore_flag, count, offset, bucket, mask_out,
                     decode_out, (long)N, (unsigned long long)seed, (unsigned long long)counter0, (const unsigned long long*)counter_dev);
  DH_LAUNCH_CHECK();
  return DANHIP_OK;
}

/* boxes [B,K,4] already ordered by descending score (stable); keep_idx int32 [B,max_out] (positions into K, -1 padded),
 * num_keep int32 [B]. */
extern "C" int danhip_nms(const float* boxes_sorted, int32_t B, int32_t K, int32_t max_out, float iou_threshold, int32_t* keep_idx,
                          int32_t* num_keep, void* stream) {
  DH_REQUIRE(boxes_sorted && keep_idx && num_keep, DANHIP_EINVAL, "nms: null pointer");
  DH_REQUIRE(B > 0 && K > 0 && max_out > 0 && K <= 65536, DANHIP_EINVAL, "nms: bad dims (K <= 65536)");
  hipLaunchKernelGGL(nms_kernel, dim3(B), dim3(1024), (size_t)K, (hipStream_t)stream, boxes_sorted, keep_idx, num_keep, K, max_out, iou_threshold);
  DH_LAUNCH_CHECK();
  return DANHIP_OK;
}

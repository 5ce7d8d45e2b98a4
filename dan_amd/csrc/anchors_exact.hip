// Anchor / box index-compare kernels (gfx950), bit-exact against the fp32 oracle (compiled with -ffp-contract=off:
// no FMA contraction, IEEE divide).  fp32 + int32 only.
//
//   anchors_generate        <- utility/anchor_manipulator.py:163-198 (generate_anchors_by_offset, center2point)
//   iou_matrix              <- utility/anchor_manipulator.py:24-52
//   dual_max_match          <- utility/anchor_manipulator.py:54-105
//   small_mining_match      <- cpp/ExtraLib/small_mining_match.cc:68-222 (SmallMiningMatch custom op)
//   encode_anchors          <- utility/anchor_manipulator.py:294-326 / :358-387 (tail after matching)
//   decode_anchors          <- utility/anchor_manipulator.py:389-424
#include "common.h"

namespace {

// ------------------------------------------------------------------ anchors
// one thread per anchor of one level; order (y, x, depth); out = 4 separate arrays at offset `off`
__global__ void anchors_generate_kernel(float* ymin, float* xmin, float* ymax, float* xmax, const float* __restrict__ ah,
                                        const float* __restrict__ aw, int depth, int lh, int lw, float stride, float offset_h, float offset_w,
                                        int off) {
  const int total = lh * lw * depth;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int d = i % depth, cell = i / depth;
    const int x = cell % lw, y = cell / lw;
    const float cy = ((float)y + offset_h) * stride;
    const float cx = ((float)x + offset_w) * stride;
    const float hh = (ah[d] - 1.f) / 2.f, hw = (aw[d] - 1.f) / 2.f;
    ymin[off + i] = cy - hh;
    xmin[off + i] = cx - hw;
    ymax[off + i] = cy + hh;
    xmax[off + i] = cx + hw;
  }
}

// ------------------------------------------------------------------ IoU matrix [A,G] (+1 box convention), x inside_mask
__device__ __forceinline__ float iou_value(float ay0, float ax0, float ay1, float ax1, const float* __restrict__ gt, int g, bool in) {
  const float gy0 = gt[g * 4], gx0 = gt[g * 4 + 1], gy1 = gt[g * 4 + 2], gx1 = gt[g * 4 + 3];
  const float h = fmaxf(fminf(ay1, gy1) - fmaxf(ay0, gy0) + 1.f, 0.f);
  const float w = fmaxf(fminf(ax1, gx1) - fmaxf(ax0, gx0) + 1.f, 0.f);
  const float inter = h * w;
  const float area_a = (ax1 - ax0 + 1.f) * (ay1 - ay0 + 1.f);
  const float area_g = (gx1 - gx0 + 1.f) * (gy1 - gy0 + 1.f);
  const float uni = area_a + area_g - inter;
  float v = (uni == 0.f) ? 0.f : inter / uni;
  if (!in) v = v * 0.f;
  return v;
}

__device__ __forceinline__ void iou_body(const float* __restrict__ ymin, const float* __restrict__ xmin, const float* __restrict__ ymax,
                                         const float* __restrict__ xmax, const unsigned char* __restrict__ inside, const float* __restrict__ gt,
                                         float* __restrict__ ov, int A, int G) {
  const long total = (long)A * G;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int a = (int)(idx / G), g = (int)(idx % G);
    const float v = iou_value(ymin[a], xmin[a], ymax[a], xmax[a], gt, g, inside ? inside[a] != 0 : true);
    ov[idx] = v;
  }
}

__global__ void iou_matrix_kernel(const float* __restrict__ ymin, const float* __restrict__ xmin, const float* __restrict__ ymax,
                                  const float* __restrict__ xmax, const unsigned char* __restrict__ inside, const float* __restrict__ gt,
                                  float* __restrict__ ov, int A, int G) {
  iou_body(ymin, xmin, ymax, xmax, inside, gt, ov, A, G);
}

// order-preserving float <-> int key (for atomicMax on floats of any sign)
__device__ __forceinline__ int fkey(float f) { int i = __float_as_int(f); return i >= 0 ? i : i ^ 0x7fffffff; }
__device__ __forceinline__ float fkey_inv(int k) { return __int_as_float(k >= 0 ? k : k ^ 0x7fffffff); }

// column maxima colmax[g] = max_a ov[a,g]   (ws_colkey must be pre-set to INT_MIN)
__device__ __forceinline__ void colmax_body(const float* __restrict__ ov, int* __restrict__ colkey, int A, int G, int* sk) {
  for (int g = threadIdx.x; g < G; g += blockDim.x) sk[g] = INT_MIN;
  __syncthreads();
  const long total = (long)A * G;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x)
    atomicMax(&sk[idx % G], fkey(ov[idx]));
  __syncthreads();
  for (int g = threadIdx.x; g < G; g += blockDim.x) atomicMax(&colkey[g], sk[g]);
}
__global__ void colmax_kernel(const float* __restrict__ ov, int* __restrict__ colkey, int A, int G) {
  extern __shared__ int sk[];
  colmax_body(ov, colkey, A, G, sk);
}

// ------------------------------------------------------------------ dual-max matching (one thread per anchor)
// element (a, g) = ov[a * rs + g * gs]: (rs, gs) = (G, 1) for the [A,G] matrix, (1, A) for the batched path's [G,A] copy
struct RowView {
  const float* p;
  long gs;
  __device__ __forceinline__ float operator[](int g) const { return p[g * gs]; }
};
__device__ __forceinline__ void dual_max_match_body(const float* __restrict__ ov, long rs, long gs, const int* __restrict__ colkey,
                                                    int* __restrict__ midx, float* __restrict__ mscore, int A, int G, float low, float high,
                                                    int ignore_between) {
  for (int a = blockIdx.x * blockDim.x + threadIdx.x; a < A; a += gridDim.x * blockDim.x) {
    const RowView row{ov + (long)a * rs, gs};
    int best = 0;
    float bv = row[0];
    for (int g = 1; g < G; ++g) if (row[g] > bv) { bv = row[g]; best = g; }      // tf.argmax: first maximum
    const bool less = bv < low, between = (bv < high) && (bv >= low);
    const bool neg = ignore_between ? less : between, ign = ignore_between ? between : less;
    int idx = neg ? -1 : best;
    if (ign) idx = -2;
    // gt side: every anchor that equals a column maximum is forced to the highest-IoU such gt
    bool any = false;
    int pick = 0;
    float pv = 0.f;
    bool first = true;
    for (int g = 0; g < G; ++g) {
      const bool tie = row[g] == fkey_inv(colkey[g]);
      any = any || tie;
      const float v = row[g] * (tie ? 1.f : 0.f);
      if (first || v > pv) { pv = v; pick = g; first = false; }
    }
    if (any) { midx[a] = pick; mscore[a] = row[pick]; }
    else { midx[a] = idx; mscore[a] = row[best]; }
  }
}

__global__ void dual_max_match_kernel(const float* __restrict__ ov, const int* __restrict__ colkey, int* __restrict__ midx,
                                      float* __restrict__ mscore, int A, int G, float low, float high, int ignore_between) {
  dual_max_match_body(ov, G, 1, colkey, midx, mscore, A, G, low, high, ignore_between);
}

// ------------------------------------------------------------------ small-mining matching
// phase 1 + 2 (one thread per anchor; per-gt counters via atomics: increments commute, so the totals equal the
// reference's sequential loop)
__device__ __forceinline__ void smm_phase12_body(const float* __restrict__ ov, long rs, long gs, const int* __restrict__ colkey,
                                                 int* __restrict__ midx, float* __restrict__ mscore, int* __restrict__ cnt, int A, int G,
                                                 float neg_low, float neg_high, float pos_thres) {
  const float eps = 1.1920928955078125e-07f;  // std::numeric_limits<float>::epsilon()
  for (int a = blockIdx.x * blockDim.x + threadIdx.x; a < A; a += gridDim.x * blockDim.x) {
    const RowView row{ov + (long)a * rs, gs};
    int best = 0;
    float bs = -3.4028234663852886e+38f;
    for (int g = 0; g < G; ++g) if (row[g] > bs) { best = g; bs = row[g]; }
    float sc = bs;
    int idx;
    if (bs >= neg_low && bs < neg_high) idx = -1;
    else if (bs >= pos_thres) idx = best;
    else idx = -2;
    // phase 2, restricted to this anchor's row: gts in ascending order, later gts overwrite earlier ones.
    // candidate <=> |ov - colmax| < eps (the running-max pre-filter of the reference is implied by this test)
    for (int g = 0; g < G; ++g) {
      const float s = row[g];
      if (fabsf(s - fkey_inv(colkey[g])) < eps) { sc = s; idx = g; }
    }
    // net effect of phase 1's increment and phase 2's decrement/increment pairs: the final gt counts this anchor once
    if (idx > -1) atomicAdd(&cnt[idx], 1);
    midx[a] = idx;
    mscore[a] = sc;
  }
}

__global__ void smm_phase12_kernel(const float* __restrict__ ov, const int* __restrict__ colkey, int* __restrict__ midx,
                                   float* __restrict__ mscore, int* __restrict__ cnt, int A, int G, float neg_low, float neg_high,
                                   float pos_thres) {
  smm_phase12_body(ov, G, 1, colkey, midx, mscore, cnt, A, G, neg_low, neg_high, pos_thres);
}

// phase 3 (hard-face compensation): ONE workgroup, gts processed in order because each gt's picks remove anchors from
// the later gts' candidate sets.  Candidates are compacted in ascending anchor order and pushed/popped through a binary
// max-heap with exactly libstdc++'s std::push_heap / std::pop_heap element moves (comparator: a < b <=> b.dist > a.dist),
// so equal-IoU candidates come out in the same order as the reference's std::priority_queue.
struct HeapItem { int anchor; float dist; };
__device__ __forceinline__ bool heap_less(const HeapItem& a, const HeapItem& b) { return b.dist > a.dist; }
__device__ void heap_push(HeapItem* h, int len_after, HeapItem v) {     // element already counted in len_after
  int hole = len_after - 1;
  int parent = (hole - 1) / 2;
  while (hole > 0 && heap_less(h[parent], v)) { h[hole] = h[parent]; hole = parent; parent = (hole - 1) / 2; }
  h[hole] = v;
}
__device__ void heap_pop(HeapItem* h, int len_before) {                  // removes the top; heap shrinks by one
  const int len = len_before - 1;
  if (len <= 0) return;
  const HeapItem v = h[len];
  int hole = 0, child = 0;
  while (child < (len - 1) / 2) {
    child = 2 * (child + 1);
    if (heap_less(h[child], h[child - 1])) --child;
    h[hole] = h[child];
    hole = child;
  }
  if ((len & 1) == 0 && child == (len - 2) / 2) {
    child = 2 * (child + 1);
    h[hole] = h[child - 1];
    hole = child - 1;
  }
  int parent = (hole - 1) / 2;                                            // __push_heap(first, hole, 0, v)
  while (hole > 0 && heap_less(h[parent], v)) { h[hole] = h[parent]; hole = parent; parent = (hole - 1) / 2; }
  h[hole] = v;
}

__global__ void smm_phase3_kernel(const float* __restrict__ ov, int* __restrict__ midx, float* __restrict__ mscore, int* __restrict__ cnt,
                                  HeapItem* __restrict__ heap, int A, int G, int min_match, float stop_pos) {
  __shared__ int s_scan[1024];
  __shared__ int s_total;
  const int T = blockDim.x;
  const int per = (A + T - 1) / T;
  const int a0 = threadIdx.x * per, a1 = min(A, a0 + per);
  for (int g = 0; g < G; ++g) {
    __syncthreads();
    if (cnt[g] >= min_match) continue;                     // uniform: cnt[g] is only written by thread 0 below, behind a barrier
    int n = 0;
    for (int a = a0; a < a1; ++a) n += (midx[a] < 0 && ov[(long)a * G + g] > stop_pos);
    s_scan[threadIdx.x] = n;
    __syncthreads();
    if (threadIdx.x == 0) {                                 // exclusive scan (T <= 1024 ints; serial is fine here)
      int run = 0;
      for (int t = 0; t < T; ++t) { const int v = s_scan[t]; s_scan[t] = run; run += v; }
      s_total = run;
    }
    __syncthreads();
    int w = s_scan[threadIdx.x];
    for (int a = a0; a < a1; ++a) {
      const float s = ov[(long)a * G + g];
      if (midx[a] < 0 && s > stop_pos) { heap[w].anchor = a; heap[w].dist = s; ++w; }
    }
    __threadfence_block();
    __syncthreads();
    if (threadIdx.x == 0) {
      const int ncand = s_total;
      // build by successive pushes in ascending anchor order (the list is reused in place as the heap)
      for (int i = 1; i <= ncand; ++i) { const HeapItem v = heap[i - 1]; heap_push(heap, i, v); }
      int len = ncand, c = cnt[g];
      while (len > 0 && c < min_match) {
        const HeapItem top = heap[0];
        ++c;
        mscore[top.anchor] = top.dist;
        midx[top.anchor] = g;
        heap_pop(heap, len);
        --len;
      }
      cnt[g] = c;
    }
    __threadfence_block();
  }
}

// ------------------------------------------------------------------ encode / decode
__device__ __forceinline__ void encode_anchors_body(const float* __restrict__ ymin, const float* __restrict__ xmin,
                                                    const float* __restrict__ ymax, const float* __restrict__ xmax,
                                                    const float* __restrict__ gt, const int* __restrict__ midx, float* __restrict__ targets,
                                                    int* __restrict__ labels, float* __restrict__ matched, int A, float ps0, float ps1,
                                                    float ps2, float ps3, float scale) {
  for (int a = blockIdx.x * blockDim.x + threadIdx.x; a < A; a += gridDim.x * blockDim.x) {
    const int m = midx[a];
    const bool pos = m > -1;
    const int mi = pos ? m : 0;
    labels[a] = (pos ? 1 : 0) + (m < -1 ? -1 : 0);
    const float gy0 = gt[mi * 4], gx0 = gt[mi * 4 + 1], gy1 = gt[mi * 4 + 2], gx1 = gt[mi * 4 + 3];
    const float gh = gy1 - gy0 + 1.f, gw = gx1 - gx0 + 1.f;
    const float gcy = (gy0 + gy1) / 2.f, gcx = (gx0 + gx1) / 2.f;
    const float ah = ymax[a] - ymin[a] + 1.f, aw = xmax[a] - xmin[a] + 1.f;
    const float acy = (ymin[a] + ymax[a]) / 2.f, acx = (xmin[a] + xmax[a]) / 2.f;
    const float t0 = (gcy - acy) / ah / ps0;
    const float t1 = (gcx - acx) / aw / ps1;
    const float t2 = logf(gh * scale / ah) / ps2;
    const float t3 = logf(gw * scale / aw) / ps3;
    const float f = pos ? 1.f : 0.f;
    *reinterpret_cast<float4*>(targets + (long)a * 4) = make_float4(f * t0, f * t1, f * t2, f * t3);
    if (matched) *reinterpret_cast<float4*>(matched + (long)a * 4) = make_float4(gy0 * f, gx0 * f, gy1 * f, gx1 * f);
  }
}

__global__ void encode_anchors_kernel(const float* __restrict__ ymin, const float* __restrict__ xmin, const float* __restrict__ ymax,
                                      const float* __restrict__ xmax, const float* __restrict__ gt, const int* __restrict__ midx,
                                      float* __restrict__ targets, int* __restrict__ labels, float* __restrict__ matched, int A, float ps0,
                                      float ps1, float ps2, float ps3, float scale) {
  encode_anchors_body(ymin, xmin, ymax, xmax, gt, midx, targets, labels, matched, A, ps0, ps1, ps2, ps3, scale);
}

// ------------------------------------------------------------------ batched encode (blockIdx.y = image; ragged gt lists through goff[B+1])
// Per-image views: gt rows [goff[b], goff[b+1]), ovT at goff[b]*A floats, colkey / cnt at goff[b] ints, heap / outputs at b*A.
struct BatchArgs {
  const float *ymin, *xmin, *ymax, *xmax;      // anchors the targets are encoded against
  const float *mymin, *mxmin, *mymax, *mxmax;  // anchors used for matching (encode_pa_anchors shrinks them); same pointers otherwise
  const unsigned char* inside;
  const float* gt;
  const int* goff;
  float* ovT;                                  // IoU [G_b, A] per image at goff[b]*A floats
  int *colkey, *cnt;
  void* heap;
  int* midx;
  float *targets, *scores, *matched;
  int* labels;
  int A;
};

__global__ void batch_init_kernel(int* __restrict__ colkey, int* __restrict__ cnt, int n) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) { colkey[i] = INT_MIN; cnt[i] = 0; }
}
// IoU in the [G,A] layout only (anchor index fastest: coalesced anchor reads and stores; every later pass walks it that way)
__global__ void batch_iou_kernel(BatchArgs p) {
  const int b = blockIdx.y, g0 = p.goff[b], G = p.goff[b + 1] - g0, A = p.A;
  const float* gt = p.gt + (long)g0 * 4;
  float* ovT = p.ovT + (long)g0 * A;
  const long total = (long)A * G;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int g = (int)(idx / A), a = (int)(idx % A);
    ovT[idx] = iou_value(p.mymin[a], p.mxmin[a], p.mymax[a], p.mxmax[a], gt, g, p.inside ? p.inside[a] != 0 : true);
  }
}
// column maxima: grid (anchor chunks, gt, image); wave max by shuffles, one atomic per wave (max is order-independent)
__global__ void batch_colmax_kernel(BatchArgs p) {
  const int b = blockIdx.z, g0 = p.goff[b], G = p.goff[b + 1] - g0, A = p.A, g = blockIdx.y;
  if (g >= G) return;
  const float* col = p.ovT + ((long)g0 + g) * A;
  int k = INT_MIN;
  for (int a = blockIdx.x * blockDim.x + threadIdx.x; a < A; a += gridDim.x * blockDim.x) k = max(k, fkey(col[a]));
  for (int o = 32; o > 0; o >>= 1) k = max(k, __shfl_xor(k, o));
  if ((threadIdx.x & 63) == 0) atomicMax(&p.colkey[g0 + g], k);
}
__global__ void batch_phase12_kernel(BatchArgs p, int mining, float low, float high, float pos) {
  const int b = blockIdx.y, g0 = p.goff[b], G = p.goff[b + 1] - g0;
  const float* ovT = p.ovT + (long)g0 * p.A;
  if (mining) smm_phase12_body(ovT, 1, p.A, p.colkey + g0, p.midx + (long)b * p.A, p.scores + (long)b * p.A, p.cnt + g0, p.A, G, low, high, pos);
  else dual_max_match_body(ovT, 1, p.A, p.colkey + g0, p.midx + (long)b * p.A, p.scores + (long)b * p.A, p.A, G, low, high, 1);
}

// phase 3, one workgroup per image.  Same semantics as smm_phase3_kernel; the candidate list of a gt is compacted in ascending anchor
// order by giving every wave a contiguous anchor range and ranking lanes with a ballot (coalesced reads of the [G,A] copy).
constexpr int P3_LDS_HEAP = 2048;     // candidate lists up to this size are heaped in LDS (thread 0's dependent loads), longer ones in HBM
__host__ __device__ inline int p3_chunks(int A) { return ((((A + 15) / 16) + 511) / 512) * 8; }   // 64-anchor chunks per wave (16 waves), x8
__global__ __launch_bounds__(1024) void batch_phase3_kernel(BatchArgs p, int min_match, float stop_pos) {
  extern __shared__ unsigned long long s_dyn[];
  unsigned long long* s_mask = s_dyn;                                   // [16][nchunk] candidate ballots of the current gt
  __shared__ HeapItem s_heap[P3_LDS_HEAP];
  __shared__ int s_wave[16];
  __shared__ int s_total;
  const int b = blockIdx.x, g0 = p.goff[b], G = p.goff[b + 1] - g0, A = p.A;
  const int nchunk = p3_chunks(A);
  int* s_cnt = reinterpret_cast<int*>(s_dyn + 16 * nchunk);            // [G] per-gt match counts of this image
  const float* ovT = p.ovT + (long)g0 * A;
  int* midx = p.midx + (long)b * A;
  float* mscore = p.scores + (long)b * A;
  HeapItem* gheap = reinterpret_cast<HeapItem*>(p.heap) + (long)b * A;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int a0 = wave * nchunk * 64, a1 = min(A, a0 + nchunk * 64);
  unsigned long long* my_mask = s_mask + wave * nchunk;
  for (int g = threadIdx.x; g < G; g += blockDim.x) s_cnt[g] = p.cnt[g0 + g];
  __syncthreads();
  for (int g = 0; g < G; ++g) {
    if (s_cnt[g] >= min_match) continue;                    // uniform: s_cnt[g] is only written in iteration g (thread 0, behind barriers)
    __syncthreads();                                        // thread 0's midx / mscore updates of the previous processed gt
    const float* col = ovT + (long)g * A;
    int n = 0;
    for (int c0 = 0; c0 < nchunk; c0 += 8) {                // 8 chunks of loads in flight per wave
      float sc[8];
      int mi[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int ac = min(a0 + (c0 + k) * 64 + lane, A - 1);
        sc[k] = col[ac];
        mi[k] = midx[ac];
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int a = a0 + (c0 + k) * 64 + lane;
        const unsigned long long m = __ballot(a < a1 && mi[k] < 0 && sc[k] > stop_pos);
        n += __popcll(m);
        if (lane == 0) my_mask[c0 + k] = m;
      }
    }
    if (lane == 0) s_wave[wave] = n;
    __syncthreads();
    if (threadIdx.x == 0) {
      int run = 0;
      for (int t = 0; t < 16; ++t) { const int v = s_wave[t]; s_wave[t] = run; run += v; }
      s_total = run;
    }
    __syncthreads();
    const int ncand = s_total;
    HeapItem* heap = ncand <= P3_LDS_HEAP ? s_heap : gheap;
    if (n > 0) {                                            // wave-uniform
      int w = s_wave[wave];
      for (int ch = 0; ch < nchunk; ++ch) {                 // ascending anchor order: waves own contiguous ranges, lanes ranked by the ballot
        const unsigned long long m = my_mask[ch];
        if (m == 0) continue;
        const int a = a0 + ch * 64 + lane;
        if ((m >> lane) & 1ull) { const int r = w + __popcll(m & ((1ull << lane) - 1ull)); heap[r].anchor = a; heap[r].dist = col[a]; }
        w += __popcll(m);
      }
    }
    __threadfence_block();
    __syncthreads();
    if (threadIdx.x == 0) {
      for (int i = 1; i <= ncand; ++i) { const HeapItem v = heap[i - 1]; heap_push(heap, i, v); }
      int len = ncand, c = s_cnt[g];
      while (len > 0 && c < min_match) {
        const HeapItem top = heap[0];
        ++c;
        mscore[top.anchor] = top.dist;
        midx[top.anchor] = g;
        heap_pop(heap, len);
        --len;
      }
      s_cnt[g] = c;
      __threadfence_block();
    }
  }
}
__global__ void batch_encode_kernel(BatchArgs p, float ps0, float ps1, float ps2, float ps3, float scale) {
  const int b = blockIdx.y, g0 = p.goff[b] < p.goff[b + 1] ? p.goff[b] : 0;   // an (unsupported) empty row range must not read past the buffer
  encode_anchors_body(p.ymin, p.xmin, p.ymax, p.xmax, p.gt + (long)g0 * 4, p.midx + (long)b * p.A, p.targets + (long)b * p.A * 4,
                      p.labels + (long)b * p.A, p.matched ? p.matched + (long)b * p.A * 4 : nullptr, p.A, ps0, ps1, ps2, ps3, scale);
}

__global__ void decode_anchors_kernel(const float* __restrict__ pred, const float* __restrict__ ymin, const float* __restrict__ xmin,
                                      const float* __restrict__ ymax, const float* __restrict__ xmax, float* __restrict__ out, int B, int A,
                                      float ps0, float ps1, float ps2, float ps3) {
  const long total = (long)B * A;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int a = (int)(i % A);
    const float4 p = *reinterpret_cast<const float4*>(pred + i * 4);
    const float ah = ymax[a] - ymin[a] + 1.f, aw = xmax[a] - xmin[a] + 1.f;
    const float acy = (ymin[a] + ymax[a]) / 2.f, acx = (xmin[a] + xmax[a]) / 2.f;
    const float ph = expf(p.z * ps2) * ah;
    const float pw = expf(p.w * ps3) * aw;
    const float pcy = p.x * ps0 * ah + acy;
    const float pcx = p.y * ps1 * aw + acx;
    *reinterpret_cast<float4*>(out + i * 4) =
        make_float4(pcy - (ph - 1.f) / 2.f, pcx - (pw - 1.f) / 2.f, pcy + (ph - 1.f) / 2.f, pcx + (pw - 1.f) / 2.f);
  }
}

// tf.nn.softmax(cls_pred)[:, -1] of two-way logits (eval_dan.py:356,371, eval_sfd.py:281, train_dan.py:438): exp(x - max) / sum in fp32,
// the order TF / Eigen evaluate it in; optionally also the "easy" mask score > thr as int32 (train_dan.py:439, eval_dan.py:386).
__global__ void face_scores_kernel(const float* __restrict__ cls, float* __restrict__ score, int* __restrict__ mask, float thr, long total) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const float2 l = *reinterpret_cast<const float2*>(cls + i * 2);
    const float m = fmaxf(l.x, l.y);
    const float e0 = expf(l.x - m), e1 = expf(l.y - m);
    const float p = e1 / (e0 + e1);
    if (score) score[i] = p;
    if (mask) mask[i] = p > thr ? 1 : 0;
  }
}

__global__ void fill_int_kernel(int* p, int v, int n) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) p[i] = v;
}

inline int grid_for(long total, int block, int cap = 2048) {
  long b = (total + block - 1) / block;
  if (b > cap) b = cap;
  if (b < 1) b = 1;
  return (int)b;
}

}  // namespace

extern "C" int danhip_anchors_generate(float* ymin, float* xmin, float* ymax, float* xmax, const float* anchor_h, const float* anchor_w,
                                       int32_t depth, int32_t layer_h, int32_t layer_w, float stride, float offset_h, float offset_w,
                                       int32_t out_offset, void* stream) {
  DH_REQUIRE(ymin && xmin && ymax && xmax && anchor_h && anchor_w && depth > 0 && layer_h > 0 && layer_w > 0 && out_offset >= 0, DANHIP_EINVAL,
             "anchors_generate: bad arguments");
  hipLaunchKernelGGL(anchors_generate_kernel, dim3(grid_for((long)layer_h * layer_w * depth, 256)), dim3(256), 0, (hipStream_t)stream, ymin, xmin,
                     ymax, xmax, anchor_h, anchor_w, depth, layer_h, layer_w, stride, offset_h, offset_w, out_offset);
  DH_LAUNCH_CHECK();
  return DANHIP_OK;
}

extern "C" int danhip_iou_matrix(const float* ymin, const float* xmin, const float* ymax, const float* xmax, const uint8_t* inside_mask,
                                 const float* gt_boxes, float* overlaps, int32_t A, int32_t G, void* stream) {
  DH_REQUIRE(ymin && xmin && ymax && xmax && gt_boxes && overlaps && A > 0 && G > 0, DANHIP_EINVAL, "iou_matrix: bad arguments");
  hipLaunchKernelGGL(iou_matrix_kernel, dim3(grid_for((long)A * G, 256, 4096)), dim3(256), 0, (hipStream_t)stream, ymin, xmin, ymax, xmax,
                     inside_mask, gt_boxes, overlaps, A, G);
  DH_LAUNCH_CHECK();
  return DANHIP_OK;
}

extern "C" size_t danhip_match_workspace_bytes(int32_t A, int32_t G) { return sizeof(int) * (size_t)(2 * G + 16) + sizeof(HeapItem) * (size_t)A; }

static int prep_colmax(const float* ov, int* colkey, int A, int G, hipStream_t s) {
  hipLaunchKernelGGL(fill_int_kernel, dim3(grid_for(G, 256)), dim3(256), 0, s, colkey, INT_MIN, G);
  hipLaunchKernelGGL(colmax_kernel, dim3(grid_for((long)A * G, 256, 512)), dim3(256), sizeof(int) * G, s, ov, colkey, A, G);
  DH_LAUNCH_CHECK();
  return DANHIP_OK;
}

extern "C" int danhip_dual_max_match(const float* overlaps, int32_t A, int32_t G, float low_thres, float high_thres, int ignore_between,
                                     int32_t* match_indices, float* match_scores, void* workspace, size_t workspace_bytes, void* stream) {
  DH_REQUIRE(overlaps && match_indices && match_scores && workspace && A > 0 && G > 0, DANHIP_EINVAL, "dual_max_match: bad arguments");
  DH_REQUIRE(G <= 8192, DANHIP_EINVAL, "dual_max_match: G=%d > 8192", G);
  DH_REQUIRE(workspace_bytes >= danhip_match_workspace_bytes(A, G), DANHIP_EWORKSPACE, "dual_max_match: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  int* colkey = reinterpret_cast<int*>(workspace);
  int rc = prep_colmax(overlaps, colkey, A, G, s);
  if (rc) return rc;
  hipLaunchKernelGGL(dual_max_match_kernel, dim3(grid_for(A, 128)), dim3(128), 0, s, overlaps, colkey, match_indices, match_scores, A, G, low_thres,
                     high_thres, ignore_between);
  DH_LAUNCH_CHECK();
  return DANHIP_OK;
}

// SmallMiningMatch op: same inputs/attrs/outputs as cpp/ExtraLib/small_mining_match.cc:31-54.
extern "C" int danhip_small_mining_match(const float* overlaps, int32_t A, int32_t G, float negative_low_thres, float negative_high_thres,
                                         float positive_thres, int32_t min_match, float stop_positive_thres, int32_t* match_indices,
                                         float* match_scores, void* workspace, size_t workspace_bytes, void* stream) {
  DH_REQUIRE(overlaps && match_indices && match_scores && workspace && A > 0 && G > 0, DANHIP_EINVAL, "small_mining_match: bad arguments");
  // attribute validation as in SmallMiningMatchOp's constructor (small_mining_match.cc:291-306)
  DH_REQUIRE(negative_low_thres >= 0.f && negative_low_thres < 1.f, DANHIP_EINVAL, "small_mining_match: negative_low_thres must be in [0,1)");
  DH_REQUIRE(negative_high_thres > 0.f && negative_high_thres < 1.f, DANHIP_EINVAL, "small_mining_match: negative_high_thres must be in (0,1)");
  DH_REQUIRE(positive_thres > 0.f && positive_thres < 1.f, DANHIP_EINVAL, "small_mining_match: positive_thres must be in (0,1)");
  DH_REQUIRE(stop_positive_thres >= 0.f && stop_positive_thres < 1.f, DANHIP_EINVAL, "small_mining_match: stop_positive_thres must be in [0,1)");
  DH_REQUIRE(min_match >= 0, DANHIP_EINVAL, "small_mining_match: min_match must be >= 0");
  DH_REQUIRE(G <= 8192, DANHIP_EINVAL, "small_mining_match: G=%d > 8192", G);
  DH_REQUIRE(workspace_bytes >= danhip_match_workspace_bytes(A, G), DANHIP_EWORKSPACE, "small_mining_match: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  int* colkey = reinterpret_cast<int*>(workspace);
  int* cnt = colkey + G;
  HeapItem* heap = reinterpret_cast<HeapItem*>(colkey + 2 * G + 16 - ((2 * G) % 2));
  int rc = prep_colmax(overlaps, colkey, A, G, s);
  if (rc) return rc;
  hipLaunchKernelGGL(fill_int_kernel, dim3(grid_for(G, 256)), dim3(256), 0, s, cnt, 0, G);
  hipLaunchKernelGGL(smm_phase12_kernel, dim3(grid_for(A, 128)), dim3(128), 0, s, overlaps, colkey, match_indices, match_scores, cnt, A, G,
                     negative_low_thres, negative_high_thres, positive_thres);
  DH_LAUNCH_CHECK();
  hipLaunchKernelGGL(smm_phase3_kernel, dim3(1), dim3(1024), 0, s, overlaps, match_indices, match_scores, cnt, heap, A, G, min_match,
                     stop_positive_thres);
  DH_LAUNCH_CHECK();
  return DANHIP_OK;
}

extern "C" int danhip_encode_anchors(const float* ymin, const float* xmin, const float* ymax, const float* xmax, const float* gt_boxes,
                                     const int32_t* match_indices, float* targets, int32_t* labels, float* matched_gt, int32_t A,
                                     float ps0, float ps1, float ps2, float ps3, float scale, void* stream) {
  DH_REQUIRE(ymin && xmin && ymax && xmax && gt_boxes && match_indices && targets && labels && A > 0, DANHIP_EINVAL,
             "encode_anchors: bad arguments");
  hipLaunchKernelGGL(encode_anchors_kernel, dim3(grid_for(A, 256)), dim3(256), 0, (hipStream_t)stream, ymin, xmin, ymax, xmax, gt_boxes,
                     match_indices, targets, labels, matched_gt, A, ps0, ps1, ps2, ps3, scale);
  DH_LAUNCH_CHECK();
  return DANHIP_OK;
}

static inline size_t al256(size_t v) { return (v + 255) & ~(size_t)255; }
extern "C" size_t danhip_encode_anchors_batched_workspace_bytes(int32_t B, int32_t A, int32_t total_gt) {
  if (B < 1 || A < 1 || total_gt < 1) return 0;
  return al256(sizeof(float) * (size_t)total_gt * A) + 2 * al256(sizeof(int) * (size_t)total_gt) + al256(sizeof(HeapItem) * (size_t)B * A) +
         al256(sizeof(int) * (size_t)B * A);
}

// One call = anchor_encoder_fn over a whole batch (train_sfd.py:206 / train_dan.py:243-249 map it per image in tf.data): IoU, matching
// (small-mining or dual-max), target encoding.  Images are independent: blockIdx.y (phase 3: blockIdx.x) is the image.
extern "C" int danhip_encode_anchors_batched(const float* ymin, const float* xmin, const float* ymax, const float* xmax, const float* match_ymin,
                                             const float* match_xmin, const float* match_ymax, const float* match_xmax,
                                             const uint8_t* inside_mask, const float* gt_boxes, const int32_t* gt_offsets, int32_t B, int32_t A,
                                             int32_t total_gt, int32_t max_gt, int32_t match_mining, float negative_low_thres,
                                             float ignore_thres, float positive_thres, int32_t min_match, float stop_positive_thres, float ps0,
                                             float ps1, float ps2, float ps3, float scale, float* targets, int32_t* labels, float* scores,
                                             float* matched_gt, void* workspace, size_t workspace_bytes, void* stream) {
  DH_REQUIRE(ymin && xmin && ymax && xmax && gt_boxes && gt_offsets && targets && labels && scores && workspace, DANHIP_EINVAL,
             "encode_anchors_batched: null argument");
  DH_REQUIRE(B > 0 && A > 0 && total_gt >= B && max_gt > 0 && max_gt <= total_gt, DANHIP_EINVAL,
             "encode_anchors_batched: B=%d A=%d total_gt=%d max_gt=%d (every image needs at least one gt row)", B, A, total_gt, max_gt);
  DH_REQUIRE(max_gt <= 8192, DANHIP_EINVAL, "encode_anchors_batched: max_gt=%d > 8192", max_gt);
  DH_REQUIRE(min_match >= 0, DANHIP_EINVAL, "encode_anchors_batched: min_match must be >= 0");
  DH_REQUIRE(workspace_bytes >= danhip_encode_anchors_batched_workspace_bytes(B, A, total_gt), DANHIP_EWORKSPACE,
             "encode_anchors_batched: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  BatchArgs p;
  p.ymin = ymin; p.xmin = xmin; p.ymax = ymax; p.xmax = xmax;
  p.mymin = match_ymin ? match_ymin : ymin; p.mxmin = match_xmin ? match_xmin : xmin;
  p.mymax = match_ymax ? match_ymax : ymax; p.mxmax = match_xmax ? match_xmax : xmax;
  p.inside = inside_mask; p.gt = gt_boxes; p.goff = gt_offsets;
  char* w = reinterpret_cast<char*>(workspace);
  const size_t ovb = al256(sizeof(float) * (size_t)total_gt * A), gb = al256(sizeof(int) * (size_t)total_gt);
  p.ovT = reinterpret_cast<float*>(w); w += ovb;
  p.colkey = reinterpret_cast<int*>(w); w += gb;
  p.cnt = reinterpret_cast<int*>(w); w += gb;
  p.heap = w; w += al256(sizeof(HeapItem) * (size_t)B * A);
  p.midx = reinterpret_cast<int*>(w);
  p.targets = targets; p.scores = scores; p.matched = matched_gt; p.labels = labels; p.A = A;
  hipLaunchKernelGGL(batch_init_kernel, dim3(grid_for(total_gt, 256)), dim3(256), 0, s, p.colkey, p.cnt, total_gt);
  hipLaunchKernelGGL(batch_iou_kernel, dim3(grid_for((long)A * max_gt, 256, 1024), B), dim3(256), 0, s, p);
  hipLaunchKernelGGL(batch_colmax_kernel, dim3(grid_for(A, 256 * 8), max_gt, B), dim3(256), 0, s, p);
  if (match_mining)
    hipLaunchKernelGGL(batch_phase12_kernel, dim3(grid_for(A, 128), B), dim3(128), 0, s, p, 1, negative_low_thres, ignore_thres, positive_thres);
  else
    hipLaunchKernelGGL(batch_phase12_kernel, dim3(grid_for(A, 128), B), dim3(128), 0, s, p, 0, ignore_thres, positive_thres, 0.f);
  if (match_mining) {
    const size_t lds = sizeof(unsigned long long) * 16 * (size_t)p3_chunks(A) + sizeof(int) * (size_t)max_gt;
    DH_REQUIRE(lds <= 120 * 1024, DANHIP_EINVAL, "encode_anchors_batched: A=%d / max_gt=%d exceed the phase-3 LDS plan", A, max_gt);
    if (lds > 48 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(batch_phase3_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(batch_phase3_kernel, dim3(B), dim3(1024), lds, s, p, min_match, stop_positive_thres);
  }
  hipLaunchKernelGGL(batch_encode_kernel, dim3(grid_for(A, 256), B), dim3(256), 0, s, p, ps0, ps1, ps2, ps3, scale);
  DH_LAUNCH_CHECK();
  return DANHIP_OK;
}

extern "C" int danhip_decode_anchors(const float* pred, const float* ymin, const float* xmin, const float* ymax, const float* xmax, float* boxes,
                                     int32_t B, int32_t A, float ps0, float ps1, float ps2, float ps3, void* stream) {
  DH_REQUIRE(pred && ymin && xmin && ymax && xmax && boxes && B > 0 && A > 0, DANHIP_EINVAL, "decode_anchors: bad arguments");
  hipLaunchKernelGGL(decode_anchors_kernel, dim3(grid_for((long)B * A, 256)), dim3(256), 0, (hipStream_t)stream, pred, ymin, xmin, ymax, xmax, boxes,
                     B, A, ps0, ps1, ps2, ps3);
  DH_LAUNCH_CHECK();
  return DANHIP_OK;
}

extern "C" int danhip_face_scores(const float* cls, float* score, int32_t* mask, float threshold, int64_t n, void* stream) {
  DH_REQUIRE(cls && (score || mask) && n > 0, DANHIP_EINVAL, "face_scores: bad arguments");
  hipLaunchKernelGGL(face_scores_kernel, dim3(grid_for((long)n, 256)), dim3(256), 0, (hipStream_t)stream, cls, score, mask, threshold, (long)n);
  DH_LAUNCH_CHECK();
  return DANHIP_OK;
}

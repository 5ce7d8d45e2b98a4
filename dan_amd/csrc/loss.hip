// Prediction re-layout, hard-negative mining, losses and the optimizer step (gfx950; HBM-bound fp32/int kernels).
//
//   head_split_{fwd,bwd}   <- max-out + reshape_pred: net/sfd_net.py:175-216, train_sfd.py:293-304, train_dan.py:340-355
//   hard_neg_scores / kth_largest_rows / detection_loss_{fwd,bwd}
//                          <- train_sfd.py:350-417, train_dan.py:286-324,470-478 (per-image top-k replaced by an exact
//                             per-row radix select of the k-th largest score; same '>=' tie semantics)
//   sgd_momentum_flat      <- train_sfd.py:419-447 (L2 term, bias gradient x2, tf.train.MomentumOptimizer)
#include "common.h"

namespace {

__device__ __forceinline__ float block_sum(float v, float* sh) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  if ((threadIdx.x & 63) == 0) sh[w] = v;
  __syncthreads();
  float t = 0.f;
  for (int i = 0; i < nw; ++i) t += sh[i];
  __syncthreads();
  return t;
}

// ---------------------------------------------------------------- head split (depth = 1 anchor per cell)
// h fp32 [B*HW, Ch] with channels [loc(4) | neg(nneg) | pos(npos)] -> loc [B, A, 4], cls [B, A, 2] at anchor offset `off`
__global__ void head_split_fwd_kernel(const float* __restrict__ h, float* __restrict__ loc, float* __restrict__ cls, int B, int HW, int Ch,
                                      int nneg, int npos, int A, int off) {
  const long total = (long)B * HW;
  for (long m = (long)blockIdx.x * blockDim.x + threadIdx.x; m < total; m += (long)gridDim.x * blockDim.x) {
    const int b = (int)(m / HW), p = (int)(m % HW);
    const float* r = h + m * Ch;
    const long a = (long)b * A + off + p;
    *reinterpret_cast<float4*>(loc + a * 4) = make_float4(r[0], r[1], r[2], r[3]);
    float ng = r[4];
    for (int i = 1; i < nneg; ++i) ng = fmaxf(ng, r[4 + i]);
    float ps = r[4 + nneg];
    for (int i = 1; i < npos; ++i) ps = fmaxf(ps, r[4 + nneg + i]);
    *reinterpret_cast<float2*>(cls + a * 2) = make_float2(ng, ps);
  }
}

// gradient of reduce_max is shared equally between tied maxima (TF / torch.amax semantics); dy fp32 [B*HW, Ch]
__global__ void head_split_bwd_kernel(const float* __restrict__ h, const float* __restrict__ dloc, const float* __restrict__ dcls,
                                      float* __restrict__ dy, int B, int HW, int Ch, int nneg, int npos, int A, int off) {
  const long total = (long)B * HW;
  for (long m = (long)blockIdx.x * blockDim.x + threadIdx.x; m < total; m += (long)gridDim.x * blockDim.x) {
    const int b = (int)(m / HW), p = (int)(m % HW);
    const float* r = h + m * Ch;
    const long a = (long)b * A + off + p;
    float* o = dy + m * Ch;
    for (int i = 0; i < 4; ++i) o[i] = dloc[a * 4 + i];
    for (int grp = 0; grp < 2; ++grp) {
      const int s = grp == 0 ? 4 : 4 + nneg, n = grp == 0 ? nneg : npos;
      float mx = r[s];
      for (int i = 1; i < n; ++i) mx = fmaxf(mx, r[s + i]);
      int cnt = 0;
      for (int i = 0; i < n; ++i) cnt += (r[s + i] == mx);
      const float g = dcls[a * 2 + grp] / (float)cnt;
      for (int i = 0; i < n; ++i) o[s + i] = r[s + i] == mx ? g : 0.f;
    }
  }
}

// ---------------------------------------------------------------- hard-negative mining
// score = label==0 ? -softmax(cls)[0] : -1 ;  counts[b*2+0] += (label>0), counts[b*2+1] += (label==0)
__global__ void hard_neg_scores_kernel(const float* __restrict__ cls, const int* __restrict__ labels, float* __restrict__ score,
                                       int* __restrict__ counts, int A) {
  __shared__ int sp, sn;
  if (threadIdx.x == 0) { sp = 0; sn = 0; }
  __syncthreads();
  const int b = blockIdx.y;
  int np = 0, nn = 0;
  for (int a = blockIdx.x * blockDim.x + threadIdx.x; a < A; a += gridDim.x * blockDim.x) {
    const long i = (long)b * A + a;
    const float2 l = *reinterpret_cast<const float2*>(cls + i * 2);
    const float mx = fmaxf(l.x, l.y);
    const float e0 = expf(l.x - mx), e1 = expf(l.y - mx);
    const float pbg = e0 / (e0 + e1);
    const int lab = labels[i];
    score[i] = lab == 0 ? 0.f - pbg : -1.f;
    np += lab > 0;
    nn += lab == 0;
  }
  atomicAdd(&sp, np);
  atomicAdd(&sn, nn);
  __syncthreads();
  if (threadIdx.x == 0) { atomicAdd(counts + b * 2, sp); atomicAdd(counts + b * 2 + 1, sn); }
}

__device__ __forceinline__ unsigned f2key(float f) {  // monotone float -> uint
  unsigned u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key2f(unsigned k) {
  return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

// One block per row: exact k-th largest value.  k[b] = min(int(ratio*npos), nneg) (DAN: max(.,1)).  k == 0 -> thr = +inf (no negative
// selected; the reference indexes -1 there: undefined).
// Round 5: BIT-WISE selection on the monotone 32-bit keys, most significant bit first - "how many of the still-matching keys have this bit
// set" is a ballot + population count per 64 keys and one 16-word LDS exchange per bit, no atomics at all.  The round-3 form built 256-bin
// histograms with LDS atomics in four passes: almost all scores of a row share their top bytes (negatives' scores lie in (-1, 0), every
// other anchor carries the sentinel -1), so a wave's 64 atomics hit one or two bins and serialised - 66 us per step at ANY batch size,
// between the forward and the backward pass where nothing else runs (76 us before the keys moved into registers).
// KREG > 0: the row's keys live in KREG registers per thread (A <= KREG * 1024); KREG == 0: the streaming form for any A (re-reads the
// row for every bit: only rows beyond 88 * 1024 anchors take it).
template <int KREG>
__global__ __launch_bounds__(1024) void kth_largest_rows_kernel(const float* __restrict__ score, const int* __restrict__ counts, float* __restrict__ thr,
                                                                int* __restrict__ kout, int A, float ratio, int at_least_one) {
  __shared__ unsigned part[2][16];
  const int b = blockIdx.x;
  const float* row = score + (long)b * A;
  int k = (int)(ratio * (float)counts[b * 2]);
  k = min(k, counts[b * 2 + 1]);
  if (at_least_one) k = max(k, 1);
  if (threadIdx.x == 0) kout[b] = k;
  // (k derives from counts[b] alone: it is BLOCK-UNIFORM, so this return in front of the loops' __syncthreads() is taken by all 1024 threads or none)
  if (k <= 0 || k > A) { if (threadIdx.x == 0) thr[b] = k <= 0 ? INFINITY : -INFINITY; return; }
  [[maybe_unused]] unsigned keys[KREG > 0 ? KREG : 1];
  if constexpr (KREG > 0) {
    // (unconditional loads from clamped addresses: a load under `if (a < A)` became a branch + s_waitcnt vmcnt(0) per slab - 36 dependent
    // L2 round trips, most of the kernel's 53 us)
    float raw[KREG];
#pragma unroll
    for (int j = 0; j < KREG; ++j) raw[j] = row[min(j * 1024 + (int)threadIdx.x, A - 1)];
#pragma unroll
    for (int j = 0; j < KREG; ++j)
      keys[j] = (j * 1024 + (int)threadIdx.x < A) ? f2key(raw[j]) : 0u;      // (a key of 0 never has the tested bit set: slots beyond the row never count)
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  unsigned prefix = 0, mask = 0;                       // the k-th largest key restricted to the bits decided so far
  unsigned kk = (unsigned)k;
  for (int bit = 31; bit >= 0; --bit) {
    const unsigned bm = 1u << bit, m2 = mask | bm, want = prefix | bm;
    unsigned cnt = 0;                                  // wave-uniform: matching keys of this wave whose bit is set
    if constexpr (KREG > 0) {
      // per-lane count in the vector ALU (and + compare + add-with-carry per key; slabs beyond the row hold zeros), then ONE wave sum of the
      // (at most 7-bit) lane counts by a ballot per bit - a ballot + scalar population count per KEY made every compare wait for its
      // round trip through the scalar unit (35 us)
      unsigned cl = 0;
#pragma unroll
      for (int j = 0; j < KREG; ++j) cl += ((keys[j] & m2) == want) ? 1u : 0u;
#pragma unroll
      for (int t = 0; (1 << t) <= KREG; ++t) cnt += (unsigned)__popcll(__ballot((cl >> t) & 1u)) << t;
    } else {
      for (int a0 = 0; a0 < A; a0 += 1024) {
        const int a = a0 + (int)threadIdx.x;
        const unsigned key = a < A ? f2key(row[a]) : 0u;
        cnt += (unsigned)__popcll(__ballot((key & m2) == want));
      }
    }
    unsigned* slot = part[bit & 1];                    // (two buffers: the one written now was last read two bits ago, behind a barrier)
    if (lane == 0) slot[wave] = cnt;
    __syncthreads();
    unsigned tot = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) tot += slot[w];
    if (tot >= kk) prefix = want;                      // the k-th largest has this bit set
    else kk -= tot;                                    // ... or lies among the keys without it
    mask = m2;
  }
  if (threadIdx.x == 0) thr[b] = key2f(prefix);
}

// acc[0]=ce_sum acc[1]=n_selected acc[2]=loc_sum acc[3]=n_pos
__global__ void detection_loss_fwd_kernel(const float* __restrict__ cls, const float* __restrict__ loc, const int* __restrict__ labels,
                                          const float* __restrict__ loc_t, const float* __restrict__ score, const float* __restrict__ thr,
                                          unsigned char* __restrict__ sel, float* __restrict__ acc, int A, long total) {
  __shared__ float sh[16];
  float ce = 0.f, ns = 0.f, ll = 0.f, np = 0.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int b = (int)(i / A);
    const int lab = labels[i];
    const bool pos = lab > 0;
    const bool s = pos || (lab == 0 && score[i] >= thr[b]);
    sel[i] = (unsigned char)(s ? (pos ? 2 : 1) : 0);
    if (s) {
      const float2 l = *reinterpret_cast<const float2*>(cls + i * 2);
      const float mx = fmaxf(l.x, l.y);
      const float lse = mx + logf(expf(l.x - mx) + expf(l.y - mx));
      ce += lse - (pos ? l.y : l.x);
      ns += 1.f;
    }
    if (pos) {
      const float4 p = *reinterpret_cast<const float4*>(loc + i * 4);
      const float4 t = *reinterpret_cast<const float4*>(loc_t + i * 4);
      const float d[4] = {p.x - t.x, p.y - t.y, p.z - t.z, p.w - t.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) { const float ad = fabsf(d[j]); ll += ad < 1.f ? 0.5f * d[j] * d[j] : ad - 0.5f; }
      np += 1.f;
    }
  }
  ce = block_sum(ce, sh); ns = block_sum(ns, sh); ll = block_sum(ll, sh); np = block_sum(np, sh);
  if (threadIdx.x == 0) { atomicAdd(acc, ce); atomicAdd(acc + 1, ns); atomicAdd(acc + 2, ll); atomicAdd(acc + 3, np); }
}

// dcls = sel ? (softmax - onehot) * ce_scale / n_sel : 0 ;  dloc = pos ? d smoothL1 / n_pos : 0   (both x gscale)
__global__ void detection_loss_bwd_kernel(const float* __restrict__ cls, const float* __restrict__ loc, const float* __restrict__ loc_t,
                                          const unsigned char* __restrict__ sel, const float* __restrict__ acc, float* __restrict__ dcls,
                                          float* __restrict__ dloc, float ce_scale, float loc_scale, long total) {
  const float kc = acc[1] > 0.f ? ce_scale / acc[1] : 0.f;
  const float kl = acc[3] > 0.f ? loc_scale / acc[3] : 0.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const unsigned char s = sel[i];
    float2 g = make_float2(0.f, 0.f);
    if (s) {
      const float2 l = *reinterpret_cast<const float2*>(cls + i * 2);
      const float mx = fmaxf(l.x, l.y);
      const float e0 = expf(l.x - mx), e1 = expf(l.y - mx), inv = 1.f / (e0 + e1);
      g.x = (e0 * inv - (s == 1 ? 1.f : 0.f)) * kc;
      g.y = (e1 * inv - (s == 2 ? 1.f : 0.f)) * kc;
    }
    *reinterpret_cast<float2*>(dcls + i * 2) = g;
    float4 d = make_float4(0.f, 0.f, 0.f, 0.f);
    if (s == 2) {
      const float4 p = *reinterpret_cast<const float4*>(loc + i * 4);
      const float4 t = *reinterpret_cast<const float4*>(loc_t + i * 4);
      const float df[4] = {p.x - t.x, p.y - t.y, p.z - t.z, p.w - t.w};
      float o[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = (fabsf(df[j]) < 1.f ? df[j] : (df[j] > 0.f ? 1.f : -1.f)) * kl;
      d = make_float4(o[0], o[1], o[2], o[3]);
    }
    *reinterpret_cast<float4*>(dloc + i * 4) = d;
  }
}

// ---------------------------------------------------------------- optimizer
// flat fp32 buffers; segment s covers [seg[s], seg[s+1]); gmult = gradient multiplier, wdc = weight_decay * coefficient
// Vector form: every segment starts on a 64-element boundary and total is padded to one (danhip_sgd_momentum_flat checks), so a
// float4 never straddles two variables.  A thread walks float4s in ascending order: its segment index only moves forward
// (one binary search at its first element, then a short scan), instead of a 7-step search per element.
// `dyn` (optional): the device-resident dynamic loss-scale state {scale, good_steps, growth_interval, found_nonfinite}: the gradient is
// divided by dyn[0] and a step whose gradients hold an inf / NaN (dyn[3] != 0, set by grad_nonfinite_kernel) leaves w and v untouched.
__global__ void sgd_momentum_flat_kernel(float* __restrict__ w, const float* __restrict__ g, float* __restrict__ v, const long* __restrict__ seg,
                                         const float* __restrict__ gmult, const float* __restrict__ wdc, int nseg, long total, float lr,
                                         float momentum, float gscale, float* __restrict__ l2_out, const float* __restrict__ dyn) {
  __shared__ float sh[8];
  float l2 = 0.f;
  bool skip = false;
  if (dyn) { gscale = 1.f / dyn[0]; skip = dyn[3] != 0.f; }
  const long n4 = total >> 2;
  const long stride = (long)gridDim.x * blockDim.x;
  long q = (long)blockIdx.x * blockDim.x + threadIdx.x;
  int lo = 0;
  if (q < n4) {
    int hi = nseg;                                      // seg[lo] <= 4q < seg[lo+1]
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (seg[mid] <= q * 4) lo = mid; else hi = mid; }
  }
  auto update = [&](long qq, int sg, float4 wi, const float4 gi, float4 vi) __attribute__((always_inline)) {
    const float c = wdc[sg], gm = gmult[sg];
    l2 += 0.5f * c * (wi.x * wi.x + wi.y * wi.y + wi.z * wi.z + wi.w * wi.w);
    vi.x = momentum * vi.x + (gi.x * gscale + c * wi.x) * gm;
    vi.y = momentum * vi.y + (gi.y * gscale + c * wi.y) * gm;
    vi.z = momentum * vi.z + (gi.z * gscale + c * wi.z) * gm;
    vi.w = momentum * vi.w + (gi.w * gscale + c * wi.w) * gm;
    wi.x -= lr * vi.x; wi.y -= lr * vi.y; wi.z -= lr * vi.z; wi.w -= lr * vi.w;
    if (!skip) {
      reinterpret_cast<float4*>(v)[qq] = vi;
      reinterpret_cast<float4*>(w)[qq] = wi;
    }
  };
  // two float4s per trip (round 4): all six loads are issued before the first use - with 65 M (DAN) .. 205 M (PyramidBox) parameters a
  // thread makes 16 - 50 trips, and one float4 triple in flight per thread left the kernel at 3.3 TB/s
  for (; q + stride < n4; q += 2 * stride) {
    const long q1 = q + stride;
    const float4 w0 = reinterpret_cast<const float4*>(w)[q], w1 = reinterpret_cast<const float4*>(w)[q1];
    const float4 g0 = reinterpret_cast<const float4*>(g)[q], g1 = reinterpret_cast<const float4*>(g)[q1];
    const float4 v0 = reinterpret_cast<const float4*>(v)[q], v1 = reinterpret_cast<const float4*>(v)[q1];
    while (lo + 1 < nseg && seg[lo + 1] <= q * 4) ++lo;
    int lo1 = lo;
    while (lo1 + 1 < nseg && seg[lo1 + 1] <= q1 * 4) ++lo1;
    update(q, lo, w0, g0, v0);
    update(q1, lo1, w1, g1, v1);
    lo = lo1;
  }
  for (; q < n4; q += stride) {
    while (lo + 1 < nseg && seg[lo + 1] <= q * 4) ++lo;
    update(q, lo, reinterpret_cast<const float4*>(w)[q], reinterpret_cast<const float4*>(g)[q], reinterpret_cast<const float4*>(v)[q]);
  }
  if (l2_out) {
    l2 = block_sum(l2, sh);
    if (threadIdx.x == 0) atomicAdd(l2_out, l2);
  }
}

__global__ void grad_nonfinite_kernel(const float* __restrict__ g, long total, float* __restrict__ dyn) {
  bool bad = false;
  const long n4 = total >> 2;
  for (long q = (long)blockIdx.x * blockDim.x + threadIdx.x; q < n4; q += (long)gridDim.x * blockDim.x) {
    const float4 gi = reinterpret_cast<const float4*>(g)[q];
    const float t = gi.x + gi.y + gi.z + gi.w;            // inf - inf and NaN both end up non-finite
    bad = bad || !(fabsf(t) <= 3.4028234663852886e+38f) || !(fabsf(gi.x) <= 3.4028234663852886e+38f) || !(fabsf(gi.y) <= 3.4028234663852886e+38f) ||
          !(fabsf(gi.z) <= 3.4028234663852886e+38f) || !(fabsf(gi.w) <= 3.4028234663852886e+38f);
  }
  if (__any(bad) && (threadIdx.x & 63) == 0) dyn[3] = 1.f;  // benign race: every writer stores the same value
}
// torch.cuda.amp.GradScaler's rule: halve after a non-finite step, double after `interval` clean ones
__global__ void loss_scale_update_kernel(float* __restrict__ dyn) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  if (dyn[3] != 0.f) { dyn[0] = fmaxf(dyn[0] * 0.5f, 1.f); dyn[1] = 0.f; }
  else {
    dyn[1] += 1.f;
    if (dyn[1] >= dyn[2]) { dyn[0] = fminf(dyn[0] * 2.f, 16777216.f); dyn[1] = 0.f; }
  }
  dyn[3] = 0.f;
}

inline int grid_for(long total, int block, int cap = 4096) {
  long b = (total + block - 1) / block;
  if (b > cap) b = cap;
  if (b < 1) b = 1;
  return (int)b;
}

}  // namespace

extern "C" int danhip_head_split_fwd(const float* h, float* loc, float* cls, int32_t B, int32_t HW, int32_t Ch, int32_t nneg, int32_t npos,
                                     int32_t A, int32_t anchor_offset, void* stream) {
  DH_REQUIRE(h && loc && cls && B > 0 && HW > 0 && nneg >= 1 && npos >= 1 && Ch == 4 + nneg + npos, DANHIP_EINVAL,
             "head_split_fwd: bad arguments (Ch must be 4+nneg+npos, depth 1)");
  DH_REQUIRE(anchor_offset >= 0 && anchor_offset + HW <= A, DANHIP_EINVAL, "head_split_fwd: anchor range out of bounds");
  hipLaunchKernelGGL(head_split_fwd_kernel, dim3(grid_for((long)B * HW, 256)), dim3(256), 0, (hipStream_t)stream, h, loc, cls, B, HW, Ch, nneg,
                     npos, A, anchor_offset);
  DH_LAUNCH_CHECK();
  return DANHIP_OK;
}

extern "C" int danhip_head_split_bwd(const float* h, const float* dloc, const float* dcls, float* dy, int32_t B, int32_t HW, int32_t Ch,
                                     int32_t nneg, int32_t npos, int32_t A, int32_t anchor_offset, void* stream) {
  DH_REQUIRE(h && dloc && dcls && dy && B > 0 && HW > 0 && nneg >= 1 && npos >= 1 && Ch == 4 + nneg + npos, DANHIP_EINVAL,
             "head_split_bwd: bad arguments");
  DH_REQUIRE(anchor_offset >= 0 && anchor_offset + HW <= A, DANHIP_EINVAL, "head_split_bwd: anchor range out of bounds");
  hipLaunchKernelGGL(head_split_bwd_kernel, dim3(grid_for((long)B * HW, 256)), dim3(256), 0, (hipStream_t)stream, h, dloc, dcls, dy, B, HW, Ch,
                     nneg, npos, A, anchor_offset);
  DH_LAUNCH_CHECK();
  return DANHIP_OK;
}

extern "C" int danhip_hard_neg_select(const float* cls, const int32_t* labels, float* score, int32_t* counts, float* thr, int32_t* k_out,
                                      int32_t B, int32_t A, float negative_ratio, int at_least_one, void* stream) {
  DH_REQUIRE(cls && labels && score && counts && thr && k_out && B > 0 && A > 0, DANHIP_EINVAL, "hard_neg_select: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  if (hipMemsetAsync(counts, 0, sizeof(int32_t) * 2 * B, s) != hipSuccess) { danhip_set_error("hard_neg_select: memset failed"); return DANHIP_ELAUNCH; }
  // (few, fat blocks: the block totals end in atomics on two words per row - 64 blocks per row spent most of the kernel queueing on them)
  hipLaunchKernelGGL(hard_neg_scores_kernel, dim3((unsigned)grid_for(A, 1024, 8), (unsigned)B), dim3(1024), 0, s, cls, labels, score, counts, A);
  DH_LAUNCH_CHECK();
  if (A <= 36 * 1024)
    hipLaunchKernelGGL(kth_largest_rows_kernel<36>, dim3((unsigned)B), dim3(1024), 0, s, score, counts, thr, k_out, A, negative_ratio, at_least_one);
  else if (A <= 88 * 1024)          // 1024 x 1024 inputs: 87 360 anchors
    hipLaunchKernelGGL(kth_largest_rows_kernel<88>, dim3((unsigned)B), dim3(1024), 0, s, score, counts, thr, k_out, A, negative_ratio, at_least_one);
  else
    hipLaunchKernelGGL(kth_largest_rows_kernel<0>, dim3((unsigned)B), dim3(1024), 0, s, score, counts, thr, k_out, A, negative_ratio, at_least_one);
  DH_LAUNCH_CHECK();
  return DANHIP_OK;
}

extern "C" int danhip_detection_loss_fwd(const float* cls, const float* loc, const int32_t* labels, const float* loc_targets, const float* score,
                                         const float* thr, uint8_t* sel, float* acc4, int32_t B, int32_t A, void* stream) {
  DH_REQUIRE(cls && loc && labels && loc_targets && score && thr && sel && acc4 && B > 0 && A > 0, DANHIP_EINVAL, "detection_loss_fwd: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  if (hipMemsetAsync(acc4, 0, sizeof(float) * 4, s) != hipSuccess) { danhip_set_error("detection_loss_fwd: memset failed"); return DANHIP_ELAUNCH; }
  const long total = (long)B * A;
  // (the four sums end in atomics on ONE 16-byte word: 1024 blocks spent ~40 of the kernel's 57 us queueing there; 128 fat blocks do not)
  hipLaunchKernelGGL(detection_loss_fwd_kernel, dim3(grid_for(total, 1024, 128)), dim3(1024), 0, s, cls, loc, labels, loc_targets, score, thr, sel,
                     acc4, A, total);
  DH_LAUNCH_CHECK();
  return DANHIP_OK;
}

extern "C" int danhip_detection_loss_bwd(const float* cls, const float* loc, const float* loc_targets, const uint8_t* sel, const float* acc4,
                                         float* dcls, float* dloc, float ce_scale, float loc_scale, int32_t B, int32_t A, void* stream) {
  DH_REQUIRE(cls && loc && loc_targets && sel && acc4 && dcls && dloc && B > 0 && A > 0, DANHIP_EINVAL, "detection_loss_bwd: bad arguments");
  const long total = (long)B * A;
  hipLaunchKernelGGL(detection_loss_bwd_kernel, dim3(grid_for(total, 256, 2048)), dim3(256), 0, (hipStream_t)stream, cls, loc, loc_targets, sel,
                     acc4, dcls, dloc, ce_scale, loc_scale, total);
  DH_LAUNCH_CHECK();
  return DANHIP_OK;
}

extern "C" int danhip_sgd_momentum_flat(float* w, const float* g, float* v, const int64_t* seg_starts, const float* gmult, const float* wd_coef,
                                        int32_t nseg, int64_t total, float lr, float momentum, float grad_scale, float* l2_out, void* stream) {
  DH_REQUIRE(w && g && v && seg_starts && gmult && wd_coef && nseg > 0 && total > 0, DANHIP_EINVAL, "sgd_momentum_flat: bad arguments");
  DH_REQUIRE(total % 4 == 0 && (((uintptr_t)w | (uintptr_t)g | (uintptr_t)v) & 15) == 0, DANHIP_EINVAL,
             "sgd_momentum_flat: buffers must be 16-byte aligned, total a multiple of 4 (segments start on 64-element boundaries)");
  hipLaunchKernelGGL(sgd_momentum_flat_kernel, dim3(grid_for(total / 4, 256, 4096)), dim3(256), 0, (hipStream_t)stream, w, g, v,
                     reinterpret_cast<const long*>(seg_starts), gmult, wd_coef, nseg, (long)total, lr, momentum, grad_scale, l2_out,
                     (const float*)nullptr);
  DH_LAUNCH_CHECK();
  return DANHIP_OK;
}

// The same update under a dynamic loss scale kept on the device (no host round trip, capturable in the step's hipGraph):
// loss_scale_state = {scale, clean steps so far, growth interval, scratch flag}.  One call = non-finite check of g, the update with
// g / scale (skipped entirely when the check fired), then the scale rule (x0.5 after a skipped step, x2 after `interval` clean ones).
extern "C" int danhip_sgd_momentum_flat_dynamic(float* w, const float* g, float* v, const int64_t* seg_starts, const float* gmult, const float* wd_coef,
                                                int32_t nseg, int64_t total, float lr, float momentum, float* loss_scale_state, float* l2_out,
                                                void* stream) {
  DH_REQUIRE(w && g && v && seg_starts && gmult && wd_coef && loss_scale_state && nseg > 0 && total > 0, DANHIP_EINVAL,
             "sgd_momentum_flat_dynamic: bad arguments");
  DH_REQUIRE(total % 4 == 0 && (((uintptr_t)w | (uintptr_t)g | (uintptr_t)v) & 15) == 0, DANHIP_EINVAL,
             "sgd_momentum_flat_dynamic: buffers must be 16-byte aligned, total a multiple of 4");
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(grad_nonfinite_kernel, dim3(grid_for(total / 4, 256, 2048)), dim3(256), 0, s, g, (long)total, loss_scale_state);
  hipLaunchKernelGGL(sgd_momentum_flat_kernel, dim3(grid_for(total / 4, 256, 4096)), dim3(256), 0, s, w, g, v, reinterpret_cast<const long*>(seg_starts),
                     gmult, wd_coef, nseg, (long)total, lr, momentum, 1.f, l2_out, (const float*)loss_scale_state);
  hipLaunchKernelGGL(loss_scale_update_kernel, dim3(1), dim3(64), 0, s, loss_scale_state);
  DH_LAUNCH_CHECK();
  return DANHIP_OK;
}

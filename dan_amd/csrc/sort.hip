// Stable descending arg-sort of one fp32 score vector: the ordering step of utility/bbox_util.py:61-73 (tf.nn.top_k: ties -> lower index
// first) and :75-91 (tf.image.non_max_suppression consumes candidates by descending score) — round 3 replaces torch.sort on the
// product path.
//
// Every (score, index) pair becomes ONE 64-bit key whose ascending order is the wanted order:
//     key = (~monotone(score)) << 32 | index        monotone(x) = bits ^ (sign ? 0xFFFFFFFF : 0x80000000)
// (descending score; equal scores — bit-identical, or +0 / -0, which compare equal and are mapped to the same key — by ascending index,
// i.e. a STABLE sort; `ties_high_index_first` is numpy's argsort()[::-1] instead: eval_dan.py:255; NaNs of either sign sort first, like
// torch.sort(descending=True); the scores here are softmax outputs).  Keys are unique, so a plain bitonic network is exact.
//
// n <= 34 125 anchors at 640 x 640 and 87 360 at 1024 x 1024: the network runs on chunks of 8192 keys in LDS (64 KB, 1024 threads x 8
// keys) and only the compare-exchange distances >= 8192 touch global memory: 1 + 3 + 6 launches for 65 536 padded keys.
#include "common.h"

namespace {

constexpr int CHUNK = 8192, THREADS = 1024;

__device__ __forceinline__ unsigned long long sort_key(float score, unsigned idx) {
  unsigned b = __builtin_bit_cast(unsigned, score);
  if ((b << 1) == 0u) b = 0u;                                      // -0 == +0
  if ((b & 0x7FFFFFFFu) > 0x7F800000u) b = 0x7FC00000u;            // every NaN (either sign, any payload) = one value above +inf: first, by index,
                                                                   // as torch.sort(descending=True); also keeps real keys below the pad key ~0
  const unsigned mono = b ^ ((b >> 31) ? 0xFFFFFFFFu : 0x80000000u);
  return ((unsigned long long)(~mono) << 32) | idx;
}

__device__ __forceinline__ void cmpx(unsigned long long& a, unsigned long long& b, bool up) {
  const bool sw = (a > b) == up;
  const unsigned long long lo = sw ? b : a, hi = sw ? a : b;
  a = lo; b = hi;
}

// stages j = jstart .. 1 of merge size k on the chunk in LDS; element e of the chunk is global element base + e
__device__ __forceinline__ void lds_stages(unsigned long long* s, int base, int k, int jstart) {
  for (int j = jstart; j >= 1; j >>= 1) {
    for (int t = threadIdx.x; t < CHUNK / 2; t += THREADS) {
      const int lo = ((t & ~(j - 1)) << 1) | (t & (j - 1)), hi = lo | j;
      const bool up = ((base + lo) & k) == 0;
      unsigned long long a = s[lo], b = s[hi];
      cmpx(a, b, up);
      s[lo] = a; s[hi] = b;
    }
    __syncthreads();
  }
}

// builds the keys of chunk blockIdx.x (padding = largest key) and sorts the chunk: merge sizes 2 .. CHUNK, alternating direction per chunk
__global__ __launch_bounds__(THREADS) void sort_chunks_kernel(const float* __restrict__ scores, int n, unsigned flip, unsigned long long* __restrict__ keys) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long s[];
  const int base = blockIdx.x * CHUNK;
  for (int e = threadIdx.x; e < CHUNK; e += THREADS) {
    const int i = base + e;
    s[e] = i < n ? sort_key(scores[i], (unsigned)i ^ flip) : ~0ull;      // flip = 0xFFFFFFFF: equal scores by DESCENDING index
  }
  __syncthreads();
  for (int k = 2; k <= CHUNK; k <<= 1) lds_stages(s, base, k, k >> 1);
  for (int e = threadIdx.x; e < CHUNK; e += THREADS) keys[base + e] = s[e];
}

// one compare-exchange stage at distance j >= CHUNK of merge size k, in global memory
__global__ __launch_bounds__(256) void sort_global_stage_kernel(unsigned long long* __restrict__ keys, int npad, int k, int j) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= npad / 2) return;
  const int lo = ((t & ~(j - 1)) << 1) | (t & (j - 1)), hi = lo | j;
  unsigned long long a = keys[lo], b = keys[hi];
  cmpx(a, b, (lo & k) == 0);
  keys[lo] = a; keys[hi] = b;
}

// the remaining stages (distances CHUNK / 2 .. 1) of merge size k > CHUNK; the last merge writes the indices
__global__ __launch_bounds__(THREADS) void sort_finish_merge_kernel(unsigned long long* __restrict__ keys, int k, int n, unsigned flip, int* __restrict__ idx_out) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long s[];
  const int base = blockIdx.x * CHUNK;
  for (int e = threadIdx.x; e < CHUNK; e += THREADS) s[e] = keys[base + e];
  __syncthreads();
  lds_stages(s, base, k, CHUNK >> 1);
  if (idx_out) {
    for (int e = threadIdx.x; e < CHUNK; e += THREADS)
      if (base + e < n) idx_out[base + e] = (int)((unsigned)(s[e] & 0xFFFFFFFFull) ^ flip);
  } else {
    for (int e = threadIdx.x; e < CHUNK; e += THREADS) keys[base + e] = s[e];
  }
}

__global__ __launch_bounds__(THREADS) void sort_emit_kernel(const unsigned long long* __restrict__ keys, int n, unsigned flip, int* __restrict__ idx_out) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) idx_out[i] = (int)((unsigned)(keys[i] & 0xFFFFFFFFull) ^ flip);
}

long padded(long n) {
  long p = CHUNK;
  while (p < n) p <<= 1;
  return p;
}

}  // namespace

extern "C" size_t danhip_argsort_workspace_bytes(int64_t n) { return n <= 0 ? 0 : (size_t)padded(n) * sizeof(unsigned long long); }

extern "C" int danhip_argsort_desc_f32(const float* scores, int64_t n, int32_t ties_high_index_first, int32_t* idx_out, void* workspace,
                                       size_t workspace_bytes, void* stream) {
  DH_REQUIRE(n >= 0 && n <= (1ll << 24), DANHIP_EINVAL, "argsort_desc: n = %lld out of range (0 .. 2^24)", (long long)n);
  if (n == 0) return DANHIP_OK;
  DH_REQUIRE(scores && idx_out && workspace, DANHIP_EINVAL, "argsort_desc: null pointer");
  DH_REQUIRE(workspace_bytes >= danhip_argsort_workspace_bytes(n), DANHIP_EINVAL, "argsort_desc: workspace of %zu bytes, %zu needed", workspace_bytes,
             danhip_argsort_workspace_bytes(n));
  hipStream_t s = (hipStream_t)stream;
  const unsigned flip = ties_high_index_first ? 0xFFFFFFFFu : 0u;
  const int npad = (int)padded(n), chunks = npad / CHUNK;
  unsigned long long* keys = reinterpret_cast<unsigned long long*>(workspace);
  constexpr int LDS = CHUNK * 8;
  static const bool attr_ok = hipFuncSetAttribute(reinterpret_cast<const void*>(&sort_chunks_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LDS) == hipSuccess &&
                              hipFuncSetAttribute(reinterpret_cast<const void*>(&sort_finish_merge_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LDS) == hipSuccess;
  (void)attr_ok;
  hipLaunchKernelGGL(sort_chunks_kernel, dim3(chunks), dim3(THREADS), LDS, s, scores, (int)n, flip, keys);
  DH_LAUNCH_CHECK();
  if (chunks == 1) {
    hipLaunchKernelGGL(sort_emit_kernel, dim3(8), dim3(THREADS), 0, s, keys, (int)n, flip, idx_out);
    DH_LAUNCH_CHECK();
    return DANHIP_OK;
  }
  for (int k = 2 * CHUNK; k <= npad; k <<= 1) {
    for (int j = k >> 1; j >= CHUNK; j >>= 1) {
      hipLaunchKernelGGL(sort_global_stage_kernel, dim3((npad / 2 + 255) / 256), dim3(256), 0, s, keys, npad, k, j);
      DH_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(sort_finish_merge_kernel, dim3(chunks), dim3(THREADS), LDS, s, keys, k, (int)n, flip, k == npad ? idx_out : nullptr);
    DH_LAUNCH_CHECK();
  }
  return DANHIP_OK;
}

// Phase-length probe for the two-wave-group MFMA skeleton of conv_halo.hip / conv_wgrad_rows.hip (gfx950):
// how much of the MFMA rate does a persistent 512-thread workgroup keep when its two wave groups alternate a memory phase
// (LDS-DMA issue + ds_read_b128 fragment reads, counted vmcnt) and an MFMA phase (N x v_mfma_f32_16x16x32_bf16) between raw s_barriers,
// as a function of the phase length N?  The convolution kernels run N = 32 (halo forward / data gradient) and N = 36 (row-streaming
// weight gradient) at 55 % / 64 % MFMA busy; this probe measures the same structure stripped of everything else, for N = 32 .. 96,
// with the reads-per-MFMA and DMAs-per-MFMA ratios of those kernels — to decide whether a longer-phase kernel is worth building.
// Build: hipcc --offload-arch=gfx950 -O3 tools/phase_probe.hip -o tools/phase_probe ; run on the GPU box.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
#define LDS_AS __attribute__((address_space(3)))
static unsigned g_window = 2u << 20;

__device__ __forceinline__ void dma16(u32x4 rsrc, unsigned voff, unsigned lds_addr) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(rsrc), "s"(lds_addr) : "memory");
}
template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// MODE 0: two groups alternate (A: mem | b1 | mma | b2 ; B: mma | b1 | mem | b2), all fragment reads in the mem phase
// MODE 1: the same, but only the first third of the reads in the mem phase, the rest issued inside the MFMA phase (just-in-time)
// MODE 2: MFMA only (no reads, no DMA, no barriers): the clock-limited ceiling of the box
// MODE 3: all eight waves run the same stream (reads of the next step interleaved with the MFMAs, one barrier per phase)
// MODE 4: MODE 0 without the second barrier (the round-3 form of the kernels)
template <int NM, int NR, int ND, int MODE>
__global__ __launch_bounds__(512, 2) __attribute__((amdgpu_waves_per_eu(2, 2))) void probe(const unsigned short* __restrict__ src, float* __restrict__ out,
                                                                                          int phases, unsigned src_bytes, unsigned window) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2;
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(LDS_AS char*)smem);
  const unsigned long long a = (unsigned long long)src;
  const u32x4 rsrc = {(unsigned)a, (unsigned)(a >> 32) & 0xFFFFu, src_bytes, 0x00020000u};
  // fill the first 64 KiB of LDS once (so that the fragments are not all zeros: DVFS)
  for (int i = 0; i < 8; ++i) dma16(rsrc, (unsigned)((blockIdx.x * 64 + wave * 8 + i) * 1024 + lane * 16) & (src_bytes - 1u), lds0 + (wave * 8 + i) * 1024);
  wait_vmcnt<0>();
  __builtin_amdgcn_s_barrier();

  f32x4 acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8 fa[8], fb[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    fa[i] = *reinterpret_cast<const bf16x8*>(smem + (i * 1024 + lane * 16));
    fb[i] = *reinterpret_cast<const bf16x8*>(smem + ((8 + i) * 1024 + lane * 16));
  }
  unsigned doff = (unsigned)(wave) * 65536u + (unsigned)lane * 16u;      // every workgroup walks the same window (the packed weights of a layer)      // this wave's DMA cursor in the source buffer
  const unsigned ring = 65536;                                                              // DMA lands in the second 64 KiB of LDS (never read: pure traffic)

  auto reads = [&](int first, int count, int ph) __attribute__((always_inline)) {
#pragma unroll
    for (int r = 0; r < count; ++r) {
      const int k = first + r;
      const bf16x8 v = *reinterpret_cast<const bf16x8*>(smem + ((((k + ph) & 63) * 1024 + lane * 16)));
      if (k & 1) fa[(k >> 1) & 7] = v; else fb[(k >> 1) & 7] = v;
    }
  };
  auto dmas = [&](int ph) __attribute__((always_inline)) {
#pragma unroll
    for (int d = 0; d < ND; ++d) {
      dma16(rsrc, doff & (window - 1u), lds0 + ring + (unsigned)(((ph * ND + d) & 7) * 8 + wave) * 1024);      // (src_bytes is a power of two)
      doff += 1024;
    }
  };
  auto mma = [&](int first, int count) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < count; ++i) {
      const int k = first + i;
      acc[k & 15] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[k & 7], fb[(k >> 3) & 7], acc[k & 15], 0, 0, 0);
    }
  };
  constexpr int PRE = MODE == 1 ? (NR + 2) / 3 : NR;          // reads issued in the mem phase

  if (MODE == 2) {
    for (int ph = 0; ph < phases; ++ph) { mma(0, NM); }
  } else if (MODE == 3) {
    for (int ph = 0; ph < phases; ++ph) {
      dmas(ph);
      __builtin_amdgcn_sched_barrier(0);
      // interleave: a third of the reads, a third of the MFMAs, ...
      reads(0, NR / 3, ph); mma(0, NM / 3);
      reads(NR / 3, NR / 3, ph); mma(NM / 3, NM / 3);
      reads(2 * (NR / 3), NR - 2 * (NR / 3), ph); mma(2 * (NM / 3), NM - 2 * (NM / 3));
      wait_vmcnt<2 * ND>();
      __builtin_amdgcn_s_waitcnt(0xC07F);
      __builtin_amdgcn_s_barrier();
    }
  } else if (grp == 0) {
    for (int ph = 0; ph < phases; ++ph) {
      dmas(ph);
      __builtin_amdgcn_sched_barrier(0);
      reads(0, PRE, ph);
      __builtin_amdgcn_sched_barrier(0);
      wait_vmcnt<2 * ND>();
      __builtin_amdgcn_s_waitcnt(0xC07F);
      __builtin_amdgcn_s_barrier();                // b1
      __builtin_amdgcn_sched_barrier(0);
      if (MODE == 1) {
        mma(0, NM / 3); reads(PRE, (NR - PRE) / 2, ph); mma(NM / 3, NM / 3); reads(PRE + (NR - PRE) / 2, NR - PRE - (NR - PRE) / 2, ph);
        mma(2 * (NM / 3), NM - 2 * (NM / 3));
        __builtin_amdgcn_s_waitcnt(0xC07F);
      } else {
        mma(0, NM);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (MODE != 4) __builtin_amdgcn_s_barrier();                // b2
    }
  } else {
    for (int ph = 0; ph < phases; ++ph) {
      __builtin_amdgcn_s_waitcnt(0xC07F);
      __builtin_amdgcn_sched_barrier(0);
      if (MODE == 1) {
        mma(0, NM / 3); reads(PRE, (NR - PRE) / 2, ph); mma(NM / 3, NM / 3); reads(PRE + (NR - PRE) / 2, NR - PRE - (NR - PRE) / 2, ph);
        mma(2 * (NM / 3), NM - 2 * (NM / 3));
      } else {
        mma(0, NM);
      }
      __builtin_amdgcn_sched_barrier(0);
      wait_vmcnt<2 * ND>();
      __builtin_amdgcn_s_barrier();                // b1
      dmas(ph);
      __builtin_amdgcn_sched_barrier(0);
      reads(0, PRE, ph + 1);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_waitcnt(0xC07F);
      if (MODE != 4) __builtin_amdgcn_s_barrier();                // b2
    }
  }
  wait_vmcnt<0>();
  f32x4 s = acc[0];
#pragma unroll
  for (int i = 1; i < 16; ++i) s += acc[i];
  out[(size_t)blockIdx.x * 512 + tid] = s[0] + s[1] + s[2] + s[3];
}


template <int NM, int NR, int ND, int MODE>
static void run(const char* name, const unsigned short* src, float* out, unsigned src_bytes, double mfma_only_tf) {
  const int lds = 128 * 1024;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&probe<NM, NR, ND, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  const int phases = 400000 / NM;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((probe<NM, NR, ND, MODE>), dim3(256), dim3(512), lds, 0, src, out, phases, src_bytes, g_window);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  const int reps = 5;
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((probe<NM, NR, ND, MODE>), dim3(256), dim3(512), lds, 0, src, out, phases, src_bytes, g_window);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  ms /= reps;
  const double flop = 256.0 * 8 * (double)phases * NM * 16384.0;
  const double tf = flop / (ms * 1e-3) / 1e12;
  printf("%-44s N=%3d reads=%3d dma=%d  %8.3f ms  %8.1f TFLOP/s  %5.1f %% of the MFMA-only rate\n", name, NM, NR, ND, ms, tf,
         mfma_only_tf > 0 ? 100.0 * tf / mfma_only_tf : 100.0);
}


// ONE wave per SIMD (256-thread workgroup, the whole 512-register file per wave): every wave runs the software-pipelined stream -- the
// fragment reads of phase ph + 1 are issued before the MFMAs of phase ph and land under them (two fragment sets), DMA issue first, one
// barrier per phase.  No partner wave covers anything: what this reaches is what a one-wave design can reach.
template <int NM, int NR, int ND>
__global__ __launch_bounds__(256, 1) void probe_single(const unsigned short* __restrict__ src, float* __restrict__ out, int phases, unsigned src_bytes,
                                                       unsigned window) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(LDS_AS char*)smem);
  const unsigned long long a = (unsigned long long)src;
  const u32x4 rsrc = {(unsigned)a, (unsigned)(a >> 32) & 0xFFFFu, src_bytes, 0x00020000u};
  for (int i = 0; i < 16; ++i) dma16(rsrc, (unsigned)((blockIdx.x * 64 + wave * 16 + i) * 1024 + lane * 16) & (src_bytes - 1u), lds0 + (wave * 16 + i) * 1024);
  wait_vmcnt<0>();
  __builtin_amdgcn_s_barrier();
  f32x4 acc[32];
#pragma unroll
  for (int i = 0; i < 32; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8 fa[2][8], fb[2][8];
#pragma unroll
  for (int sset = 0; sset < 2; ++sset)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      fa[sset][i] = *reinterpret_cast<const bf16x8*>(smem + (i * 1024 + lane * 16));
      fb[sset][i] = *reinterpret_cast<const bf16x8*>(smem + ((8 + i) * 1024 + lane * 16));
    }
  unsigned doff = (unsigned)(wave) * 65536u + (unsigned)lane * 16u;
  const unsigned ring = 65536;
  auto step = [&](auto setc, int ph) __attribute__((always_inline)) {
    constexpr int S = decltype(setc)::value;
#pragma unroll
    for (int d = 0; d < 2 * ND; ++d) {               // four waves move what eight moved
      dma16(rsrc, doff & (window - 1u), lds0 + ring + (unsigned)(((ph * 2 * ND + d) & 15) * 4 + wave) * 1024);
      doff += 1024;
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int r = 0; r < NR; ++r) {                   // next phase's fragments into the other set
      const bf16x8 v = *reinterpret_cast<const bf16x8*>(smem + ((((r + ph) & 63) * 1024 + lane * 16)));
      if (r & 1) fa[1 - S][(r >> 1) & 7] = v; else fb[1 - S][(r >> 1) & 7] = v;
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < NM; ++i) acc[i & 31] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[S][i & 7], fb[S][(i >> 3) & 7], acc[i & 31], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    wait_vmcnt<4 * ND>();
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_s_barrier();
  };
  for (int ph = 0; ph < phases; ph += 2) {
    step(std::integral_constant<int, 0>{}, ph);
    step(std::integral_constant<int, 1>{}, ph + 1);
  }
  wait_vmcnt<0>();
  f32x4 s = acc[0];
#pragma unroll
  for (int i = 1; i < 32; ++i) s += acc[i];
  out[(size_t)blockIdx.x * 256 + tid] = s[0] + s[1] + s[2] + s[3];
}

template <int NM, int NR, int ND>
static void run_single(const char* name, const unsigned short* src, float* out, unsigned src_bytes, double mfma_only_tf) {
  const int lds = 128 * 1024;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&probe_single<NM, NR, ND>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  const int phases = (400000 / NM) & ~1;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((probe_single<NM, NR, ND>), dim3(256), dim3(256), lds, 0, src, out, phases, src_bytes, g_window);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  const int reps = 5;
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((probe_single<NM, NR, ND>), dim3(256), dim3(256), lds, 0, src, out, phases, src_bytes, g_window);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  ms /= reps;
  const double tf = 256.0 * 4 * (double)phases * NM * 16384.0 / (ms * 1e-3) / 1e12;
  printf("%-44s N=%3d reads=%3d dma=%d  %8.3f ms  %8.1f TFLOP/s  %5.1f %% of the MFMA-only rate\n", name, NM, NR, 2 * ND, ms, tf, 100.0 * tf / mfma_only_tf);
}

template <int NM, int NR, int ND, int MODE>
static double rate(const unsigned short* src, float* out, unsigned src_bytes) {
  const int lds = 128 * 1024;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&probe<NM, NR, ND, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  const int phases = 400000 / NM;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((probe<NM, NR, ND, MODE>), dim3(256), dim3(512), lds, 0, src, out, phases, src_bytes, g_window);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((probe<NM, NR, ND, MODE>), dim3(256), dim3(512), lds, 0, src, out, phases, src_bytes, g_window);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return 256.0 * 8 * (double)phases * NM * 16384.0 / (ms / 5 * 1e-3) / 1e12;
}

int main(int argc, char** argv) {
  if (argc > 1) g_window = (unsigned)atoi(argv[1]) << 20;       // DMA source window in MiB (power of two): 2 = L2-resident weights, 512 = an HBM stream
  printf("DMA source window: %u MiB\n", g_window >> 20);
  const unsigned src_bytes = 512u << 20;
  unsigned short* src;
  float* out;
  CK(hipMalloc(&src, src_bytes));
  CK(hipMalloc(&out, 256 * 512 * sizeof(float)));
  std::vector<unsigned short> h(src_bytes / 2);
  unsigned x = 12345u;
  for (size_t i = 0; i < h.size(); ++i) { x = x * 1664525u + 1013904223u; h[i] = (unsigned short)(0x3c00u | ((x >> 9) & 0x83ffu)); }   // bf16 in +-[0.5, 2): random, finite
  CK(hipMemcpy(src, h.data(), src_bytes, hipMemcpyHostToDevice));
  const double peak = rate<64, 0, 0, 2>(src, out, src_bytes);
  printf("MFMA only (no reads, no DMA, no barriers): %.1f TFLOP/s on this box (random operands)\n", peak);
  // halo forward today: 32 MFMAs per phase, 16 reads, ~3 DMA pieces per wave and phase
  run<32, 16, 3, 0>("halo fwd/dgrad today (1 tap x 64 ch)", src, out, src_bytes, peak);
  run<64, 32, 5, 0>("2 taps per phase", src, out, src_bytes, peak);
  run<64, 32, 5, 1>("2 taps per phase, reads just in time", src, out, src_bytes, peak);
  run<96, 36, 5, 0>("3 taps x 32 ch, 64px x 128co wave tile", src, out, src_bytes, peak);
  run<96, 36, 5, 1>("  ... reads just in time", src, out, src_bytes, peak);
  run<96, 36, 5, 3>("  ... all waves one stream, 1 barrier/phase", src, out, src_bytes, peak);
  run<32, 16, 3, 3>("1 tap, all waves one stream", src, out, src_bytes, peak);
  run<64, 24, 3, 3>("GEMM-template ratio (64 MFMA, 24 reads)", src, out, src_bytes, peak);
  // weight gradient today: 36 MFMAs, 14 transposing reads (8 bytes per lane each: counted here as 7 b128), 2 DMA
  run<36, 7, 2, 0>("row-streaming wgrad today", src, out, src_bytes, peak);
  run<72, 14, 4, 0>("wgrad, 2 K-steps per phase", src, out, src_bytes, peak);
  run<72, 14, 4, 1>("wgrad, 2 K-steps, reads just in time", src, out, src_bytes, peak);
  run<32, 0, 0, 0>("barriers only, 32 MFMAs per phase", src, out, src_bytes, peak);
  run<96, 0, 0, 0>("barriers only, 96 MFMAs per phase", src, out, src_bytes, peak);
  run<32, 16, 0, 0>("reads, no DMA, 32", src, out, src_bytes, peak);
  run<32, 0, 3, 0>("DMA, no reads, 32", src, out, src_bytes, peak);
  run<32, 16, 1, 0>("32 MFMA, 16 reads, 1 DMA", src, out, src_bytes, peak);
  run<32, 16, 2, 0>("32 MFMA, 16 reads, 2 DMA", src, out, src_bytes, peak);
  run<32, 12, 2, 0>("32 MFMA, 12 reads, 2 DMA (64px x 128co wave tile)", src, out, src_bytes, peak);
  run<64, 24, 3, 0>("64 MFMA, 24 reads, 3 DMA, two groups", src, out, src_bytes, peak);
  run<96, 36, 3, 0>("96 MFMA, 36 reads, 3 DMA", src, out, src_bytes, peak);
  run<36, 7, 1, 0>("wgrad ratio, 1 DMA", src, out, src_bytes, peak);
  // round 3: the one-barrier form of today's kernels, and the one-wave-per-SIMD alternative ("open after round 3")
  run<32, 16, 3, 4>("halo today, ONE barrier per phase", src, out, src_bytes, peak);
  run<32, 12, 2, 4>("64px x 128co wave tile, one barrier", src, out, src_bytes, peak);
  run<36, 7, 2, 4>("wgrad today, one barrier", src, out, src_bytes, peak);
  run_single<32, 12, 1>("ONE wave per SIMD: 32 MFMA, 12 reads", src, out, src_bytes, peak);
  run_single<64, 24, 2>("one wave per SIMD: 64 MFMA, 24 reads", src, out, src_bytes, peak);
  run_single<64, 16, 2>("one wave per SIMD: 64 MFMA, 16 reads (128x128 wave tile)", src, out, src_bytes, peak);
  run_single<128, 32, 3>("one wave per SIMD: 128 MFMA, 32 reads", src, out, src_bytes, peak);
  run_single<64, 0, 0>("one wave per SIMD: MFMA + barrier only", src, out, src_bytes, peak);
  return 0;
}

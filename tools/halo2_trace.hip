// Where does a step of the two-wave-group kernels go?  (named after the round-3 experiment conv_halo2.hip, removed in round 4: it ended 2-4 % /
// 15-20 % behind conv_halo.hip; profiles/r3 keeps its traces)  Shader-clock stamps of workgroup 0 (wave 0 of group A, wave 4 of group B) at four points of every
// step of the first 128 steps, kept in LDS (a global store would count in vmcnt and perturb the counted waits) and dumped at the end:
//   group A:  0 step start   1 memory phase done (fragments read, DMA issued, counted wait passed)   2 past barrier b1   3 MFMAs issued
//   group B:  0 step start   1 MFMAs issued   2 past barrier b1 (counted wait before it)             3 memory phase done
// Build (the kernel source is compiled INTO this program with -DH2_TRACE; the library build has no stamps):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DTRACE_HALO1 -I dan_amd/csrc -o tools/halo1_trace tools/halo2_trace.hip     (conv_halo.hip)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DTRACE_WGRAD -I dan_amd/csrc -o tools/wgrad_trace tools/halo2_trace.hip     (conv_wgrad_rows.hip)
// Run:  tools/halo2_trace [fwd|dgrad] [N H W C Co]       (default: conv3_2 of the benchmark, 16 x 160 x 160 x 256 -> 256)
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#ifdef TRACE_WGRAD      // the row-streaming weight gradient (conv_wgrad_rows.hip): "fwd" / "dgrad" on the command line are ignored
#define WR_TRACE 1
#include "../dan_amd/csrc/conv_wgrad_rows.hip"
#define TRACE_BUFFER wr_trace_buffer
#elif defined(TRACE_HALO1)      // the production kernel (conv_halo.hip, 8 x 32 pixel tiles, 64-channel chunks) instead of conv_halo2.hip
#define H_TRACE 1
#include "../dan_amd/csrc/conv_halo.hip"
#define TRACE_LAUNCH danhip_launch_conv_halo
#define TRACE_BUFFER h_trace_buffer
#else
#error "build with -DTRACE_HALO1 (conv_halo.hip) or -DTRACE_WGRAD (conv_wgrad_rows.hip): the 512-pixel-tile experiment conv_halo2.hip was removed in round 4"
#endif

#include <cstdarg>
void danhip_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vfprintf(stderr, fmt, ap);
  va_end(ap);
  fputc('\n', stderr);
}

int danhip_option(const char* name) {
  if (!strcmp(name, "halo2")) return 1;
  if (!strcmp(name, "halo_b2")) { const char* e = getenv("DANHIP_HALO_B2"); return e ? atoi(e) : 0; }
  if (!strcmp(name, "wgrad_b2")) { const char* e = getenv("DANHIP_WGRAD_B2"); return e ? atoi(e) : 0; }
  if (!strcmp(name, "halo2_ablate")) { const char* e = getenv("DANHIP_HALO2_ABLATE"); return e ? atoi(e) : 0; }
  return 0;
}

static unsigned short rnd_bf16(unsigned& st) {
  st = st * 1664525u + 1013904223u;
  const float v = ((st >> 8) & 0xffff) / 65536.0f - 0.5f;
  unsigned u;
  memcpy(&u, &v, 4);
  return (unsigned short)(u >> 16);
}

int main(int argc, char** argv) {
  const bool dgrad = argc > 1 && !strcmp(argv[1], "dgrad");
  int N = 16, H = 160, W = 160, C = 256, Co = 256;
  if (argc > 6) { N = atoi(argv[2]); H = atoi(argv[3]); W = atoi(argv[4]); C = atoi(argv[5]); Co = atoi(argv[6]); }
  const size_t nx = (size_t)N * H * W * C, ny = (size_t)N * H * W * Co, nw = (size_t)Co * 9 * C;
  std::vector<unsigned short> hx(nx), hw(nw);
  unsigned st = 12345u;
  for (auto& v : hx) v = rnd_bf16(st);
  for (auto& v : hw) v = rnd_bf16(st);
  bf16_t *dx, *dw, *dy;
  float* db;
  unsigned char* dbits;
  hipMalloc(&dx, nx * 2); hipMalloc(&dw, nw * 2); hipMalloc(&dy, ny * 2); hipMalloc(&db, Co * 4); hipMalloc(&dbits, ny / 8);
  hipMemcpy(dx, hx.data(), nx * 2, hipMemcpyHostToDevice);
  hipMemcpy(dw, hw.data(), nw * 2, hipMemcpyHostToDevice);
  {                                                  // (the weight-gradient mode reads y as its dY operand)
    std::vector<unsigned short> hy(ny);
    for (auto& v : hy) v = rnd_bf16(st);
    hipMemcpy(dy, hy.data(), ny * 2, hipMemcpyHostToDevice);
  }
  hipMemset(db, 0, Co * 4);
  hipMemset(dbits, 0xA5, ny / 8);
  ConvArgs a{};
  a.x = dx; a.w = dw; a.y = dy;
  a.N = N; a.H = H; a.W = W; a.C = C; a.Ho = H; a.Wo = W; a.Co = Co;
  a.kh = a.kw = 3; a.stride = 1; a.pad_t = a.pad_l = 1; a.dstride = 1;
  a.M = N * H * W; a.taps = 9; a.Kpad = 9 * C; a.ktiles = a.Kpad / 64; a.cpt = C / 64;
  if (dgrad) { a.mask_bits = dbits; } else { a.bias = db; a.relu = 1; a.bits_out = dbits; }
  hipStream_t s;
  hipStreamCreate(&s);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
#ifdef TRACE_WGRAD
  float* ddw;
  hipMalloc(&ddw, (size_t)9 * C * Co * 4);
  danhip_conv_desc dd{};
  dd.N = N; dd.H = H; dd.W = W; dd.Cin = C; dd.Ho = H; dd.Wo = W; dd.Cout = Co; dd.kh = dd.kw = 3; dd.stride = 1;
  auto launch = [&]() { hipMemsetAsync(ddw, 0, (size_t)9 * C * Co * 4, s); return danhip_launch_wgrad_rows(&dd, dx, dy, ddw, nullptr, C, s, nullptr, 0); };
#else
  auto launch = [&]() { return TRACE_LAUNCH(a, s); };
#endif
  for (int i = 0; i < 3; ++i)
    if (launch() != 0) { fprintf(stderr, "not eligible / launch failed\n"); return 1; }
  hipEventRecord(e0, s);
  const int reps = 20;
  for (int i = 0; i < reps; ++i) launch();
  hipEventRecord(e1, s);
  hipStreamSynchronize(s);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  ms /= reps;
  printf("%s %dx%dx%dx%d->%d  %.3f ms  %.1f TFLOP/s\n", dgrad ? "dgrad" : "fwd", N, H, W, C, Co, ms, 2.0 * N * H * W * 9.0 * C * Co / ms * 1e-9);
  std::vector<unsigned> tr(1024 + 16);
  hipMemcpy(tr.data(), TRACE_BUFFER(), 4096 + 64, hipMemcpyDeviceToHost);
  // steady state: steps 18 .. 125 (past the first item's prologue); a chunk is 9 steps
  const char* namesA[4] = {"memory phase (reads, DMA issue, counted wait)", "wait at b1", "MFMA phase", "wait at b2"};
  const char* namesB[4] = {"MFMA phase", "counted wait + b1", "memory phase (epilogue, reads, DMA issue)", "wait at b2"};
  for (int grp = 0; grp < 2; ++grp) {
    double sum[4] = {0, 0, 0, 0}, mx[4] = {0, 0, 0, 0};
    int cnt = 0;
    for (int i = 18; i < 126; ++i) {
      const unsigned* t = &tr[(grp * 128 + i) * 4];
      const unsigned* tn = &tr[(grp * 128 + i + 1) * 4];
      const unsigned d[4] = {t[1] - t[0], t[2] - t[1], t[3] - t[2], tn[0] - t[3]};
      for (int k = 0; k < 4; ++k) { sum[k] += d[k]; if (d[k] > mx[k]) mx[k] = d[k]; }
      ++cnt;
    }
    printf("group %c (wave %d), mean / max clocks per step over steps 18..125:\n", grp ? 'B' : 'A', grp * 4);
    double tot = 0;
    for (int k = 0; k < 4; ++k) { printf("  %-46s %8.1f  %8.0f\n", (grp ? namesB : namesA)[k], sum[k] / cnt, mx[k]); tot += sum[k] / cnt; }
    printf("  %-46s %8.1f\n", "step", tot);
  }
  for (int grp = 0; grp < 2; ++grp) {
    const unsigned* e = &tr[1024 + grp * 8];
    printf("group %c first epilogue (forward: start | setup | pair 0 | pair 1 | pair 2 | pair 3 | bits | end), clocks:", grp ? 'B' : 'A');
    for (int k = 1; k < 8; ++k) printf(" %u", e[k] - e[k - 1]);
    printf("   total %u\n", e[7] - e[0]);
  }
  if (getenv("H2_TRACE_DUMP")) {
    for (int i = 0; i < 128; ++i) {
      const unsigned* ta = &tr[i * 4];
      const unsigned* tb = &tr[(128 + i) * 4];
      printf("%3d tap %d  A %u %u %u %u   B %u %u %u %u\n", i, i % 9, ta[1] - ta[0], ta[2] - ta[1], ta[3] - ta[2], tr[(i + 1) * 4 < 512 ? (i + 1) * 4 : i * 4] - ta[3],
             tb[1] - tb[0], tb[2] - tb[1], tb[3] - tb[2], tr[(128 + (i + 1 < 128 ? i + 1 : i)) * 4] - tb[3]);
    }
  }
  return 0;
}

// In-kernel clock and loop share of the row-streaming weight gradient (conv_wgrad_rows.hip), the check MI355X_MICROARCH.md 'DVFS give-back'
// item 6 / cdna_hip_programming.md rule 28 prescribes before spending effort on a tighter issue stream:
//   clock      = d(s_memtime) / d(s_memrealtime) x 100 MHz around the K-step loop (median over workgroups),
//   loop share = (loop end - loop start) / (kernel end - kernel start) per workgroup, in constant-rate ticks,
//   launch     = (last workgroup end - first workgroup start) against the event-bracketed launch time.
// The kernel source is compiled INTO this program with -DWR_CLOCK (four scalar stamp pairs per workgroup, written to a buffer of their own
// outside the loop); the library build has no stamps.  Runs >= 2 s of back-to-back launches on random data before the measured launch.
// Build:  hipcc --offload-arch=gfx950 -O3 -std=c++17 -DWR_CLOCK -I dan_amd/csrc -o tools/clock_probe tools/clock_probe.hip
// Run:    tools/clock_probe [N H W C Co]        (default conv3_2 of the benchmark: 16 x 160 x 160 x 256 -> 256)
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../dan_amd/csrc/conv_wgrad_rows.hip"

void danhip_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vfprintf(stderr, fmt, ap);
  va_end(ap);
  fputc('\n', stderr);
}
int danhip_option(const char* name) {
  if (!strcmp(name, "wgrad_b2")) { const char* e = getenv("DANHIP_WGRAD_B2"); return e ? atoi(e) : 0; }
  if (!strcmp(name, "wgrad_slab")) { const char* e = getenv("DANHIP_WGRAD_SLAB"); return e ? atoi(e) : 1; }
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
  int N = 16, H = 160, W = 160, C = 256, Co = 256;
  if (argc > 5) { N = atoi(argv[1]); H = atoi(argv[2]); W = atoi(argv[3]); C = atoi(argv[4]); Co = atoi(argv[5]); }
  const size_t nx = (size_t)N * H * W * C, ny = (size_t)N * H * W * Co;
  std::vector<unsigned short> hx(nx), hy(ny);
  unsigned st = 12345u;
  for (auto& v : hx) v = rnd_bf16(st);
  for (auto& v : hy) v = rnd_bf16(st);
  bf16_t *dx, *dy;
  float* ddw;
  hipMalloc(&dx, nx * 2); hipMalloc(&dy, ny * 2); hipMalloc(&ddw, (size_t)9 * C * Co * 4);
  hipMemcpy(dx, hx.data(), nx * 2, hipMemcpyHostToDevice);
  hipMemcpy(dy, hy.data(), ny * 2, hipMemcpyHostToDevice);
  hipMemset(ddw, 0, (size_t)9 * C * Co * 4);
  danhip_conv_desc dd{};
  dd.N = N; dd.H = H; dd.W = W; dd.Cin = C; dd.Ho = H; dd.Wo = W; dd.Cout = Co; dd.kh = dd.kw = 3; dd.stride = 1;
  hipStream_t s;
  hipStreamCreate(&s);
  auto launch = [&]() { return danhip_launch_wgrad_rows(&dd, dx, dy, ddw, nullptr, C, s, nullptr, 0); };
  if (launch() != 0) { fprintf(stderr, "not eligible / launch failed\n"); return 1; }
  hipStreamSynchronize(s);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  // >= 2 s of back-to-back launches (the clock the chip settles at under this load), then the measured batch
  hipEventRecord(e0, s);
  launch();
  hipEventRecord(e1, s);
  hipStreamSynchronize(s);
  float one = 0;
  hipEventElapsedTime(&one, e0, e1);
  const int warm = (int)(2200.0f / (one > 0.01f ? one : 0.01f)) + 1;
  for (int i = 0; i < warm; ++i) launch();
  const int reps = 50;
  hipEventRecord(e0, s);
  for (int i = 0; i < reps; ++i) launch();
  hipEventRecord(e1, s);
  hipStreamSynchronize(s);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  ms /= reps;
  const double flop = 2.0 * N * H * W * 9.0 * C * Co;
  printf("conv_wgrad_rows %dx%dx%dx%d->%d: %.4f ms per launch (events over %d back-to-back launches after %d warm-up launches), %.1f TFLOP/s = %.3f of 2500\n",
         N, H, W, C, Co, ms, reps, warm, flop / ms * 1e-9, flop / ms * 1e-9 / 2500.0);
  const int blocks = wr_clock_blocks();
  std::vector<unsigned long long> t((size_t)blocks * 8);
  hipMemcpy(t.data(), wr_clock_buffer(), t.size() * 8, hipMemcpyDeviceToHost);
  // per workgroup: [0] memtime, [1] realtime at kernel entry; [2],[3] loop start; [4],[5] loop end; [6],[7] kernel end (after the epilogue's atomics were issued)
  std::vector<double> clk, share, pro, epi, loop_us;
  unsigned long long first = ~0ull, last = 0, first_loop_end = ~0ull, last_loop_end = 0;
  for (int b = 0; b < blocks; ++b) {
    const unsigned long long* q = &t[(size_t)b * 8];
    if (q[7] <= q[1]) continue;
    clk.push_back((double)(q[4] - q[2]) / (double)(q[5] - q[3]) * 100.0);          // MHz
    share.push_back((double)(q[5] - q[3]) / (double)(q[7] - q[1]));
    pro.push_back((double)(q[3] - q[1]) * 0.01);                                   // us
    loop_us.push_back((double)(q[5] - q[3]) * 0.01);
    epi.push_back((double)(q[7] - q[5]) * 0.01);
    first = std::min(first, q[1]); last = std::max(last, q[7]);
    first_loop_end = std::min(first_loop_end, q[5]); last_loop_end = std::max(last_loop_end, q[5]);
  }
  auto med = [](std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
  auto mn = [](const std::vector<double>& v) { return *std::min_element(v.begin(), v.end()); };
  auto mx = [](const std::vector<double>& v) { return *std::max_element(v.begin(), v.end()); };
  printf("workgroups stamped: %zu of %d\n", clk.size(), blocks);
  printf("in-kernel clock over the K-step loop: median %.0f MHz (min %.0f, max %.0f)\n", med(clk), mn(clk), mx(clk));
  printf("loop share of a workgroup's lifetime: median %.3f (min %.3f, max %.3f); before the loop %.1f us (max %.1f), after it %.1f us (max %.1f)\n", med(share), mn(share), mx(share),
         med(pro), mx(pro), med(epi), mx(epi));
  printf("launch: first workgroup start -> last workgroup end %.1f us (event-bracketed launch %.1f us); loop ends spread over %.1f us\n", (double)(last - first) * 0.01, ms * 1e3,
         (double)(last_loop_end - first_loop_end) * 0.01);
  const int steps = wr_clock_steps();
  const double per_step = med(clk) * med(loop_us) / steps;          // MHz x us = clocks
  const double ideal = 9.0 * (Co % 128 == 0 ? 4 : 2) * 2 * 16;        // 9 taps x NO co fragments x 2 waves per SIMD x 16 clocks per 16x16x32 MFMA
  printf("rows (K-steps with MFMAs) per workgroup %d, loop %.1f us: %.0f shader clocks per K-step (MFMA alone: %.0f) -> MFMA pipe %.1f %% busy inside the loop\n", steps,
         med(loop_us), per_step, ideal, 100.0 * ideal / per_step);
  return 0;
}

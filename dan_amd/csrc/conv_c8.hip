// 3x3 / stride-1 convolution of the 3-channel image (stored padded to 8 channels) to 64 channels with bias + ReLU: conv1_1 of
// every backbone (net/sfd_net.py:128, conv_block(inputs, 64, 2, ..) first layer).
//
// 0.7 GMAC per image against 52 MB of bf16 output: the layer is bound by the HBM WRITE of its output (839 MB at batch 16,
// 640 x 640 -> ~0.17 ms at 5 TB/s), so the kernel is built around full 64-byte store segments and enough loads in flight,
// not around MFMA occupancy:
//   * K = 9 taps x 8 channels = 72, padded to 96 = three 16x16x32 MFMA steps.  One tap of one pixel is exactly one lane's
//     B fragment (8 bf16 = 16 B), so the implicit-GEMM operand is loaded straight from the NHWC image with one 16-byte
//     load per lane and step — no LDS, no repacking: lane (px = lane & 15, kq = lane >> 4) loads tap 4*step + kq of pixel px.
//   * weights are the A operand (4 tiles x 3 steps = 12 fragments = 48 VGPRs, resident), with the rows of each pair of tiles
//     permuted so that a lane ends up owning 8 CONSECUTIVE output channels of one pixel: two 16-byte stores per lane and
//     unit; neighbouring pixels' lanes trade halves so that each store instruction writes 8 whole 128-byte lines (round 5).
//   * a wave walks units of 16 consecutive pixels of one image row, two units per iteration, the next pair's 6 loads
//     issued before the current pair is multiplied and stored.
#include "conv_common.h"

namespace {

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

// bit r = packed 16-bit ReLU output r > 0 (conv_common.h)
__device__ __forceinline__ unsigned c8_pos_bits8(const u32x4& t) { return dh_pos_bits8_acc<0>(t, 0u); }

struct C8Geom {
  int units_x;            // ceil(W / 16)
  long units;             // N * H * units_x
  FastDiv div_ux, div_uxh;
};

template <bool NT>
__global__ __launch_bounds__(256) void conv3x3_c8_kernel(const ConvArgs a, const C8Geom g) {
  const int lane = threadIdx.x & 63;
  const int px = lane & 15, q = lane >> 4;
  const long wave_id = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const long nwaves = (long)gridDim.x * (blockDim.x >> 6);

  // A fragments: MFMA tile m (pair P = m >> 1, half h = m & 1), row r = lane & 15 -> co = P*32 + (r >> 2)*8 + h*4 + (r & 3)
  bf16x8 wr[4][3];
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    const int co = (m >> 1) * 32 + (px >> 2) * 8 + (m & 1) * 4 + (px & 3);
#pragma unroll
    for (int s = 0; s < 3; ++s) wr[m][s] = *reinterpret_cast<const bf16x8*>(a.w + (size_t)co * a.Kpad + s * 32 + q * 8);   // k >= 72 is zero padding
  }
  float bias[2][8];
#pragma unroll
  for (int P = 0; P < 2; ++P)
#pragma unroll
    for (int j = 0; j < 8; ++j) bias[P][j] = a.bias ? a.bias[P * 32 + q * 8 + j] : 0.f;

  // taps of this lane in the three steps: t = 4*s + q  (s = 2: only q == 0 is a real tap).  A unit's (n, y, x0) are wave-uniform
  // (scalar registers); per lane only a constant pixel offset and two range compares remain per load.
  const int dy0 = q / 3 - 1, dx0 = q % 3 - 1;
  const int dy1 = (4 + q) / 3 - 1, dx1 = (4 + q) % 3 - 1;
  const bool tap2_ok = (q == 0);                                                  // tap 8 = (+1, +1)
  const int off0 = dy0 * a.W + dx0 + px, off1 = dy1 * a.W + dx1 + px, off2 = a.W + 1 + px;
  const uint4* __restrict__ xin = reinterpret_cast<const uint4*>(a.x);           // one uint4 per pixel (8 bf16 channels)

  // pixel index fits 31 bits (check_desc); padding lanes read pixel 0 and are masked to zero
  auto load_tap = [&](int base, int yy, int xx, int off, bool en) __attribute__((always_inline)) -> uint4 {
    const bool ok = en && (unsigned)yy < (unsigned)a.H && (unsigned)xx < (unsigned)a.W;
    uint4 v = xin[ok ? (unsigned)(base + off) : 0u];
    const unsigned keep = ok ? 0xFFFFFFFFu : 0u;
    v.x &= keep; v.y &= keep; v.z &= keep; v.w &= keep;
    return v;
  };
  struct Unit { int y, x0, base; };                                               // base = (n*H + y)*W + x0
  auto unit_coords = [&](long u) __attribute__((always_inline)) -> Unit {
    const unsigned uu = (unsigned)__builtin_amdgcn_readfirstlane((int)u);
    const unsigned n = fdiv(uu, g.div_uxh);
    const unsigned rem = uu - n * (unsigned)(g.units_x * a.H);
    const unsigned y = fdiv(rem, g.div_ux);
    const unsigned x0 = (rem - y * (unsigned)g.units_x) * 16u;
    Unit r;
    r.y = (int)y; r.x0 = (int)x0; r.base = (int)((n * (unsigned)a.H + y) * (unsigned)a.W + x0);
    return r;
  };
  auto load_unit = [&](const Unit& t, bool en, uint4& b0, uint4& b1, uint4& b2) __attribute__((always_inline)) {
    b0 = load_tap(t.base, t.y + dy0, t.x0 + px + dx0, off0, en);
    b1 = load_tap(t.base, t.y + dy1, t.x0 + px + dx1, off1, en);
    b2 = load_tap(t.base, t.y + 1, t.x0 + px + 1, off2, en && tap2_ok);
  };

  auto compute_store = [&](uint4 b0, uint4 b1, uint4 b2, const Unit& t) __attribute__((always_inline)) {
    f32x4 acc[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) acc[m] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const bf16x8 f0 = __builtin_bit_cast(bf16x8, b0), f1 = __builtin_bit_cast(bf16x8, b1), f2 = __builtin_bit_cast(bf16x8, b2);
#pragma unroll
    for (int m = 0; m < 4; ++m) acc[m] = DH_MFMA_16x16x32(wr[m][0], f0, acc[m]);
#pragma unroll
    for (int m = 0; m < 4; ++m) acc[m] = DH_MFMA_16x16x32(wr[m][1], f1, acc[m]);
#pragma unroll
    for (int m = 0; m < 4; ++m) acc[m] = DH_MFMA_16x16x32(wr[m][2], f2, acc[m]);
    unsigned pbyte[2] = {0u, 0u};
    u32x4 o2[2];
#pragma unroll
    for (int P = 0; P < 2; ++P) {                   // this lane: output channels [P*32 + q*8, +8) of pixel x0 + px
      f32x4 lo = acc[2 * P], hi = acc[2 * P + 1];
      lo[0] += bias[P][0]; lo[1] += bias[P][1]; lo[2] += bias[P][2]; lo[3] += bias[P][3];
      hi[0] += bias[P][4]; hi[1] += bias[P][5]; hi[2] += bias[P][6]; hi[3] += bias[P][7];
      if (a.relu) {
        lo[0] = dh_relu(lo[0]); lo[1] = dh_relu(lo[1]); lo[2] = dh_relu(lo[2]); lo[3] = dh_relu(lo[3]);
        hi[0] = dh_relu(hi[0]); hi[1] = dh_relu(hi[1]); hi[2] = dh_relu(hi[2]); hi[3] = dh_relu(hi[3]);
      }
      o2[P] = u32x4{pack2bf(lo[0], lo[1]), pack2bf(lo[2], lo[3]), pack2bf(hi[0], hi[1]), pack2bf(hi[2], hi[3])};
      pbyte[P] = c8_pos_bits8(o2[P]);
    }
    // WHOLE 128-byte lines per store instruction (round 5): a lane's two pieces are the [0, 64) and [64, 128) halves of its pixel's line, so
    // storing piece P from every lane wrote 16 half lines per instruction and the layer's 839 MB went out as 1062 MB of write traffic
    // (rocprofv3 WRITE_SIZE).  Neighbouring pixels (lane ^ 1) trade pieces instead: the even pixel's lane ends up with both pixels' low
    // halves, the odd one with both high halves; instruction 1 then writes the even pixels' lines whole (low half from the even lane, high
    // half from the odd one), instruction 2 the odd pixels'.
    {
      const bool odd = px & 1;
      u32x4 give, got;
#pragma unroll
      for (int e = 0; e < 4; ++e) give[e] = odd ? o2[0][e] : o2[1][e];
#pragma unroll
      for (int e = 0; e < 4; ++e) got[e] = dh_lane_xor1(give[e]);
      // even lane: own low half (pixel px) + the odd neighbour's low half (pixel px + 1); odd lane: the even neighbour's high half (px - 1) + own high half
      const u32x4 first = odd ? got : o2[0], second = odd ? o2[1] : got;
      const int pe = px & ~1;                       // the even pixel of the pair
      bf16_t* line = reinterpret_cast<bf16_t*>(a.y) + ((size_t)(unsigned)(t.base + pe)) * 64 + (odd ? 32 : 0) + q * 8;
      if (t.x0 + pe < a.W) {
        if (NT) __builtin_nontemporal_store(first, reinterpret_cast<u32x4*>(line));
        else *reinterpret_cast<u32x4*>(line) = first;
      }
      if (t.x0 + pe + 1 < a.W) {
        if (NT) __builtin_nontemporal_store(second, reinterpret_cast<u32x4*>(line + 64));
        else *reinterpret_cast<u32x4*>(line + 64) = second;
      }
    }
    // ReLU bit mask of the output for the next layer's data gradient (danhip_relu_bits layout: 8 bytes per pixel): this lane's two bytes
    // are q and 4 + q of pixel px; the four lanes of a pixel (q = 0..3: lanes px, px + 16, px + 32, px + 48) are OR-ed by two row swaps
    if (a.bits_out) {                               // (uniform)
      const unsigned lo = dh_or_rows(pbyte[0] << (8 * q)), hi = dh_or_rows(pbyte[1] << (8 * q));
      if (q == 0 && t.x0 + px < a.W) *reinterpret_cast<uint2*>(a.bits_out + ((size_t)(unsigned)(t.base + px)) * 8) = uint2{lo, hi};
    }
  };

  // software pipeline: the loads of the next pair of units are in flight while the current pair is multiplied and stored
  long u = wave_id * 2;
  if (u >= g.units) return;
  bool two = u + 1 < g.units;
  Unit t0 = unit_coords(u), t1 = unit_coords(two ? u + 1 : u);
  uint4 a0, a1, a2, c0, c1, c2;
  load_unit(t0, true, a0, a1, a2);
  load_unit(t1, two, c0, c1, c2);
  while (true) {
    const long un = u + nwaves * 2;
    const bool more = un < g.units;
    const bool ntwo = more && un + 1 < g.units;
    Unit nt0 = t0, nt1 = t1;
    uint4 na0 = a0, na1 = a1, na2 = a2, nc0 = c0, nc1 = c1, nc2 = c2;
    if (more) {
      nt0 = unit_coords(un);
      nt1 = unit_coords(ntwo ? un + 1 : un);
      load_unit(nt0, true, na0, na1, na2);
      load_unit(nt1, ntwo, nc0, nc1, nc2);
    }
    compute_store(a0, a1, a2, t0);
    if (two) compute_store(c0, c1, c2, t1);
    if (!more) break;
    u = un; two = ntwo; t0 = nt0; t1 = nt1;
    a0 = na0; a1 = na1; a2 = na2; c0 = nc0; c1 = nc1; c2 = nc2;
  }
}

bool c8_eligible(const ConvArgs& a) {
  return a.kh == 3 && a.kw == 3 && a.stride == 1 && a.dstride == 1 && a.pad_t == 1 && a.pad_l == 1 && a.C == 8 && a.Co == 64 && !a.mask && !a.resid &&
         !a.accumulate && !a.out_f32 && a.H == a.Ho && a.W == a.Wo && a.Kpad >= 96 && (long)a.N * a.H * ((a.W + 15) / 16) < (1l << 31);
}

}  // namespace

const char* danhip_conv_c8_label(const ConvArgs& a) { return c8_eligible(a) ? "conv3x3_c8_kernel<true>" : nullptr; }

int danhip_launch_conv_c8(const ConvArgs& a, hipStream_t s) {
  if (!c8_eligible(a)) return 1;
  C8Geom g{};
  g.units_x = (a.W + 15) / 16;
  g.units = (long)a.N * a.H * g.units_x;
  g.div_ux = make_fastdiv((unsigned)g.units_x);
  g.div_uxh = make_fastdiv((unsigned)(g.units_x * a.H));
  long blocks = (g.units + 7) / 8;                       // 4 waves x 2 units per block iteration
  if (blocks > 256 * 16) blocks = 256 * 16;
  // non-temporal stores: the 839 MB output is not re-read before it has left the L2 (measured 0.27 vs 0.32 ms)
  hipLaunchKernelGGL((conv3x3_c8_kernel<true>), dim3((unsigned)blocks), dim3(256), 0, s, a, g);
  DH_LAUNCH_CHECK();
  return DANHIP_OK;
}

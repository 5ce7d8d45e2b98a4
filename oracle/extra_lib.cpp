// CPU oracle for the reference's two ExtraLib custom ops — TEST INFRASTRUCTURE ONLY
// (see oracle/__init__.py; nothing under dan_amd/ links or calls this).
//
// Restates, as plain functions over raw arrays (no TensorFlow):
//   small_mining_match       <- cpp/ExtraLib/small_mining_match.cc:68-222  (SmallMiningMatchFunctor<CPU>)
//   dynamic_anchor_routing   <- cpp/ExtraLib/dynamic_anchor_routing.cc:188-408 (DynamicAnchorRoutingFunctor<CPU>)
//
// The reference sources include TensorFlow headers (op_kernel.h, Eigen tensors, work_sharder) and cannot be
// compiled in this image, so there is no oracle/_ref build of them; this restatement keeps the reference's
// arithmetic types (float vs double promotions), loop order, comparison operators and the libstdc++
// std::priority_queue it pops from.  Pinned by tests/golden/kats.json (hand-traced KATs).
//
// Build: see oracle/Makefile (g++ -O2 -shared -fPIC).
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <queue>
#include <vector>

namespace {
struct Cand {                      // small_mining_match.cc:56-63 (DistancePair): heap ordered by dist only
  int64_t anchor; int64_t gt; float dist;
  bool operator<(const Cand& o) const { return o.dist > dist; }
};
}  // namespace

extern "C" {

// overlaps [A,G] row-major.  match_indices int32 [A], match_scores float [A].
void oracle_small_mining_match(const float* ov, int32_t A, int32_t G, float neg_low, float neg_high, float pos_thres,
                               int32_t min_match, float stop_pos_thres, int32_t* match_idx, float* match_score) {
  std::vector<int32_t> cnt(G, 0);
  // phase 1: per-anchor arg-max + thresholds  (small_mining_match.cc:72-94)
  for (int64_t a = 0; a < A; ++a) {
    const float* row = ov + a * G;
    int32_t best = 0;
    float bs = std::numeric_limits<float>::lowest();
    for (int32_t g = 0; g < G; ++g)
      if (row[g] > bs) { best = g; bs = row[g]; }
    match_score[a] = bs;
    if (bs >= neg_low && bs < neg_high) match_idx[a] = -1;
    else if (bs >= pos_thres) { match_idx[a] = best; cnt[best] += 1; }
    else match_idx[a] = -2;
  }
  // phase 2: every gt grabs its best anchor(s), fp-epsilon ties included  (:160-187)
  const float eps = std::numeric_limits<float>::epsilon();
  for (int64_t g = 0; g < G; ++g) {
    float bs = std::numeric_limits<float>::lowest();
    std::vector<int32_t> maybe;
    for (int32_t a = 0; a < A; ++a) {
      float s = ov[(int64_t)a * G + g];
      if (s > bs) bs = s;
      if (std::abs(s - bs) < eps) maybe.push_back(a);
    }
    for (int32_t a : maybe) {
      float s = ov[(int64_t)a * G + g];
      if (std::abs(s - bs) < eps) {
        match_score[a] = s;
        if (match_idx[a] > -1) cnt[match_idx[a]] -= 1;
        match_idx[a] = (int32_t)g;
        cnt[g] += 1;
      }
    }
  }
  // phase 3: hard-face compensation  (:199-222)
  for (int64_t g = 0; g < G; ++g) {
    if (cnt[g] >= min_match) continue;
    std::priority_queue<Cand> q;
    for (int32_t a = 0; a < A; ++a) {
      float s = ov[(int64_t)a * G + g];
      if (match_idx[a] < 0 && s > stop_pos_thres) q.push(Cand{a, g, s});
    }
    while (!q.empty()) {
      Cand p = q.top();
      if (cnt[g] >= min_match) break;
      cnt[g] += 1;
      match_score[p.anchor] = p.dist;
      match_idx[p.anchor] = (int32_t)g;
      q.pop();
    }
  }
}

// Shared cell computation: returns false when the source must be skipped.
static inline bool route_cell(float ymin, float xmin, float ymax, float xmax, int32_t fh, int32_t fw, int32_t depth,
                              int32_t stride, int64_t index, int64_t* cell) {
  // dynamic_anchor_routing.cc:262-279 (train pass 2) == :351-371 (eval): note the double-precision 2.*stride
  int64_t cx = static_cast<int64_t>(std::round((xmin + xmax) / (2. * stride)));
  if (xmin / stride < -1 || xmax / stride > fw + 1 - 1.) return false;
  cx = std::min(cx, static_cast<int64_t>(fw - 1));
  cx = std::max(cx, static_cast<int64_t>(0));
  int64_t cy = static_cast<int64_t>(std::round((ymin + ymax) / (2. * stride)));
  if (ymin / stride < -1 || ymax / stride > fh + 1 - 1.) return false;
  cy = std::min(cy, static_cast<int64_t>(fh - 1));
  cy = std::max(cy, static_cast<int64_t>(0));
  *cell = (cy * fw + cx) * depth + index % depth;
  return true;
}

// Eval mode (trainging=false): dynamic_anchor_routing.cc:328-408.
// anchors [N,4] decoded stage-1 boxes, gt_targets [N,4] stage-2 offsets (already / [20,20,10,10]),
// labels [N] stage-2 scores, mask_in [N].  Outputs mask_out int32 [N], decode_out float [N,4].
void oracle_dynamic_anchor_routing_eval(const float* anchors, const float* gt_targets, const float* labels,
                                        const int32_t* mask_in, int64_t N, int32_t fh, int32_t fw, int32_t depth,
                                        int32_t stride, int32_t* mask_out, float* decode_out) {
  std::vector<float> prior(N, 0.f);
  std::memset(mask_out, 0, sizeof(int32_t) * N);
  std::memset(decode_out, 0, sizeof(float) * 4 * N);
  for (int64_t i = 0; i < N; ++i) {
    if (mask_in[i] < 1) { mask_out[i] = -1; continue; }
    float ymin = anchors[i * 4], xmin = anchors[i * 4 + 1], ymax = anchors[i * 4 + 2], xmax = anchors[i * 4 + 3];
    if (xmax - xmin < 1 || ymax - ymin < 1) continue;
    int64_t n;
    if (!route_cell(ymin, xmin, ymax, xmax, fh, fw, depth, stride, i, &n)) continue;
    if (labels[i] > prior[n]) {
      if (mask_out[n] < 0) continue;
      prior[n] = labels[i];
      mask_out[n] = 1;
      decode_out[n * 4] = ymin; decode_out[n * 4 + 1] = xmin; decode_out[n * 4 + 2] = ymax; decode_out[n * 4 + 3] = xmax;
    }
  }
  for (int64_t i = 0; i < N; ++i) {
    mask_out[i] = std::max(0, mask_out[i]);
    float ymin = decode_out[i * 4], xmin = decode_out[i * 4 + 1], ymax = decode_out[i * 4 + 2], xmax = decode_out[i * 4 + 3];
    float pcy = (ymin + ymax) / 2.;
    float pcx = (xmin + xmax) / 2.;
    float ph = (ymax - ymin + 1.);
    float pw = (xmax - xmin + 1.);
    float ty = gt_targets[i * 4], tx = gt_targets[i * 4 + 1], th = gt_targets[i * 4 + 2], tw = gt_targets[i * 4 + 3];
    th = std::exp(th) * ph;
    tw = std::exp(tw) * pw;
    ty = ty * ph + pcy;
    tx = tx * pw + pcx;
    decode_out[i * 4] = ty - (th - 1.) / 2.;
    decode_out[i * 4 + 1] = tx - (tw - 1.) / 2.;
    decode_out[i * 4 + 2] = ty + (th - 1.) / 2.;
    decode_out[i * 4 + 3] = tx + (tw - 1.) / 2.;
  }
}

// Train mode (trainging=true): dynamic_anchor_routing.cc:203-327.
// The reference draws dis(gen) from std::mt19937(std::random_device) at the reservoir test (:306), which is
// not reproducible; here the caller supplies one uniform u[i] in [0,1) per source index i (the build's
// counter-based stream, see dan_amd/csrc/routing.hip), consumed only when source i reaches the test.  Draws are
// i.i.d. so the distribution equals the reference's.
void oracle_dynamic_anchor_routing_train(const float* anchors, const float* gt_targets, const float* labels,
                                         const int32_t* mask_in, const double* u, int64_t N, int32_t fh, int32_t fw,
                                         int32_t depth, int32_t stride, float thres, float ignore_thres,
                                         int32_t* mask_out, float* decode_out) {
  std::vector<int32_t> matched(N, 0);
  std::memset(mask_out, 0, sizeof(int32_t) * N);
  std::memset(decode_out, 0, sizeof(float) * 4 * N);
  // pass 1: cells that contain a gt centre  (:205-243)
  for (int64_t i = 0; i < N; ++i) {
    if (labels[i] > 0.) {
      float gy0 = gt_targets[i * 4], gx0 = gt_targets[i * 4 + 1], gy1 = gt_targets[i * 4 + 2], gx1 = gt_targets[i * 4 + 3];
      if (gx1 - gx0 < 1 || gy1 - gy0 < 1) continue;
      int64_t cx = static_cast<int64_t>(std::round((gx0 + gx1) / (2. * stride)));
      if (cx < -stride || cx > fw + stride - 1.) continue;
      cx = std::min(cx, static_cast<int64_t>(fw - 1));
      cx = std::max(cx, static_cast<int64_t>(0));
      int64_t cy = static_cast<int64_t>(std::round((gy0 + gy1) / (2. * stride)));
      if (cy < -stride || cy > fh + stride - 1.) continue;
      cy = std::min(cy, static_cast<int64_t>(fh - 1));
      cy = std::max(cy, static_cast<int64_t>(0));
      int64_t n = (cy * fw + cx) * depth + i % depth;
      matched[n] = 1;
      mask_out[n] = 1;
    }
  }
  // pass 2: route stage-1 boxes  (:245-326)
  for (int64_t i = 0; i < N; ++i) {
    if (mask_in[i] < 1) { if (mask_out[i] < 1) mask_out[i] = -1; continue; }
    float ymin = anchors[i * 4], xmin = anchors[i * 4 + 1], ymax = anchors[i * 4 + 2], xmax = anchors[i * 4 + 3];
    if (xmax - xmin < 1 || ymax - ymin < 1) { if (mask_out[i] < 1) mask_out[i] = -1; continue; }
    int64_t n;
    if (!route_cell(ymin, xmin, ymax, xmax, fh, fw, depth, stride, i, &n)) continue;
    float gy0 = gt_targets[i * 4], gx0 = gt_targets[i * 4 + 1], gy1 = gt_targets[i * 4 + 2], gx1 = gt_targets[i * 4 + 3];
    float iy0 = std::max(ymin, gy0), ix0 = std::max(xmin, gx0), iy1 = std::min(ymax, gy1), ix1 = std::min(xmax, gx1);
    float h = std::max(iy1 - iy0 + 1., 0.);
    float w = std::max(ix1 - ix0 + 1., 0.);
    float inter = h * w;
    float area_a = (gy1 - gy0 + 1.) * (gx1 - gx0 + 1.);
    float area_b = (ymax - ymin + 1.) * (xmax - xmin + 1.);
    float uni = area_a + area_b - inter;
    if (labels[i] > 0.) {
      if (std::abs(uni) <= 1.) continue;
      else if (inter / uni <= ignore_thres) continue;
      if (inter / uni < thres) { if (mask_out[i] < 1) mask_out[i] = -1; }
      if (u[i] <= 1. / (matched[n] + 1)) {
        matched[n] += 1;
        mask_out[n] = 1;
        float pcy = (ymin + ymax) / 2.;
        float pcx = (xmin + xmax) / 2.;
        float ph = (ymax - ymin + 1.);
        float pw = (xmax - xmin + 1.);
        float gcy = (gy0 + gy1) / 2.;
        float gcx = (gx0 + gx1) / 2.;
        float gh = (gy1 - gy0 + 1.);
        float gw = (gx1 - gx0 + 1.);
        decode_out[n * 4] = (gcy - pcy) / ph;
        decode_out[n * 4 + 1] = (gcx - pcx) / pw;
        decode_out[n * 4 + 2] = std::log(std::max(gh / ph, std::numeric_limits<float>::epsilon()));
        decode_out[n * 4 + 3] = std::log(std::max(gw / pw, std::numeric_limits<float>::epsilon()));
      }
    }
  }
}

// Counter-based uniform stream shared (bit-exactly) with the HIP kernel: u in [0,1) from a 64-bit mix of
// (seed, counter).  splitmix64 finaliser; 53 mantissa bits.
double oracle_uniform(uint64_t seed, uint64_t counter) {
  uint64_t z = seed + 0x9E3779B97F4A7C15ull * (counter + 1);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z = z ^ (z >> 31);
  return (double)(z >> 11) * (1.0 / 9007199254740992.0);
}

}  // extern "C"

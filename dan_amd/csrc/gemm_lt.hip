// Pointwise (1x1, stride 1) convolutions are plain GEMMs  Y[M, Cout] = X[M, Cin] * W  (+ bias, ReLU)  and
// dX[M, Cin] (+)= dY[M, Cout8] * W^T: no window, no gather, nothing for a convolution kernel to fuse.  They go to hipBLASLt
// (the ROCm GEMM library) — measured on MI355X at 0.6-1.0 PFLOP/s on the deformable-conv / LFPN / context-module shapes
// where the implicit-GEMM kernel, built around the 3x3 gather, reaches 0.23-0.49 (tools/bench_conv.py --set pb, tools/gemm_probe.py).
// The packed weights need no second layout: wf is W^T [Cout, Kpad] and wb is W [Cin, Npad], both row-major bf16.
// Weight gradients stay on the hand-written kernels (the library's huge-K reduction shapes are 2-5x slower than conv_wgrad).
//
// Column-major view used below (hipBLASLt's convention): a row-major [R, C] matrix with leading dimension ld is the
// column-major [C, R] matrix with the same ld; D^T = op(A) * B with A = the packed weight (transposed), B = the activations.
#include <hipblaslt/hipblaslt.h>

#include <cstdlib>
#include <map>
#include <mutex>
#include <tuple>

#include "common.h"

namespace {

struct Plan {
  hipblasLtMatmulDesc_t desc = nullptr;
  hipblasLtMatrixLayout_t a = nullptr, b = nullptr, d = nullptr;
  hipblasLtMatmulAlgo_t algo;
  size_t ws = 0;
  bool ok = false;
};

std::mutex g_mu;
hipblasLtHandle_t g_handle = nullptr;
void* g_ws = nullptr;
constexpr size_t kWorkspace = 64u << 20;
using Key = std::tuple<int, long, int, int, int, int, int>;      // m, n, k, lda, ldb, ldd, epilogue
std::map<Key, Plan> g_plans;
int g_state = 0;                                                 // 0 untried, 1 usable, -1 unavailable

bool init_locked() {
  if (g_state) return g_state > 0;
  g_state = -1;
  const char* off = getenv("DANHIP_NO_BLASLT");
  if (off && off[0] == '1') return false;
  if (hipblasLtCreate(&g_handle) != HIPBLAS_STATUS_SUCCESS) return false;
  if (hipMalloc(&g_ws, kWorkspace) != hipSuccess) return false;
  g_state = 1;
  return true;
}

#if defined(DANHIP_FP16)
constexpr hipDataType kAct = HIP_R_16F;
#else
constexpr hipDataType kAct = HIP_R_16BF;
#endif

Plan& plan_locked(int m, long n, int k, int lda, int ldb, int ldd, int epi, const float* bias) {
  const Key key{m, n, k, lda, ldb, ldd, epi};
  Plan& p = g_plans[key];
  if (p.desc) return p;
  hipblasLtMatmulDescCreate(&p.desc, HIPBLAS_COMPUTE_32F, HIP_R_32F);
  const hipblasOperation_t ta = HIPBLAS_OP_T, tb = HIPBLAS_OP_N;
  hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_TRANSA, &ta, sizeof(ta));
  hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_TRANSB, &tb, sizeof(tb));
  const hipblasLtEpilogue_t e = (hipblasLtEpilogue_t)epi;
  hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_EPILOGUE, &e, sizeof(e));
  if (epi == HIPBLASLT_EPILOGUE_BIAS || epi == HIPBLASLT_EPILOGUE_RELU_BIAS) {
    const hipDataType bt = HIP_R_32F;
    hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_BIAS_DATA_TYPE, &bt, sizeof(bt));
    hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_BIAS_POINTER, &bias, sizeof(bias));
  }
  hipblasLtMatrixLayoutCreate(&p.a, kAct, k, m, lda);            // stored [k x m] column-major, used transposed
  hipblasLtMatrixLayoutCreate(&p.b, kAct, k, n, ldb);
  hipblasLtMatrixLayoutCreate(&p.d, kAct, m, n, ldd);
  hipblasLtMatmulPreference_t pref;
  hipblasLtMatmulPreferenceCreate(&pref);
  const uint64_t wsz = kWorkspace;
  hipblasLtMatmulPreferenceSetAttribute(pref, HIPBLASLT_MATMUL_PREF_MAX_WORKSPACE_BYTES, &wsz, sizeof(wsz));
  hipblasLtMatmulHeuristicResult_t res[1];
  int found = 0;
  const hipblasStatus_t st = hipblasLtMatmulAlgoGetHeuristic(g_handle, p.desc, p.a, p.b, p.d, p.d, pref, 1, res, &found);
  hipblasLtMatmulPreferenceDestroy(pref);
  if (st == HIPBLAS_STATUS_SUCCESS && found > 0 && res[0].workspaceSize <= kWorkspace) {
    p.algo = res[0].algo;
    p.ws = res[0].workspaceSize;
    p.ok = true;
  }
  return p;
}

}  // namespace

// D^T[m x n] (row-major [n, ldd]) = (accumulate ? D^T : 0) + Wp^T-view * Act, + bias[m], ReLU.  Returns DANHIP_OK when the library ran
// the product, 1 when the caller has to use its own kernel (library unavailable / no algorithm for the shape), < 0 on a failed launch.
int danhip_gemm_lt(int m, long n, int k, const void* wp, int lda, const void* act, int ldb, void* out, int ldd, const float* bias, int relu,
                   int accumulate, hipStream_t stream) {
  std::lock_guard<std::mutex> lock(g_mu);
  if (!init_locked()) return 1;
  const int epi = bias ? (relu ? HIPBLASLT_EPILOGUE_RELU_BIAS : HIPBLASLT_EPILOGUE_BIAS) : (relu ? HIPBLASLT_EPILOGUE_RELU : HIPBLASLT_EPILOGUE_DEFAULT);
  Plan& p = plan_locked(m, n, k, lda, ldb, ldd, epi, bias);
  if (!p.ok) return 1;
  if (bias) hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_BIAS_POINTER, &bias, sizeof(bias));
  const float alpha = 1.f, beta = accumulate ? 1.f : 0.f;
  const hipblasStatus_t st = hipblasLtMatmul(g_handle, p.desc, &alpha, wp, p.a, act, p.b, &beta, out, p.d, out, p.d, &p.algo, g_ws, p.ws, stream);
  DH_REQUIRE(st == HIPBLAS_STATUS_SUCCESS, DANHIP_ELAUNCH, "hipblasLtMatmul failed with status %d (m=%d n=%ld k=%d)", (int)st, m, n, k);
  return DANHIP_OK;
}

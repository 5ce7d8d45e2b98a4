// Error plumbing of the C ABI: thread-local message, never throws.
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <mutex>

#include "../../include/danhip.h"

static thread_local char g_err[512] = "";

void danhip_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* danhip_last_error(void) { return g_err; }
extern "C" int danhip_version(void) { return 4; }   // 2: danhip_deform_sample_bwd takes workspace_bytes; 3: danhip_comm_* (RCCL called directly); 4: danhip_comm_async_error / danhip_comm_abort
extern "C" int danhip_act_dtype(void) {
#ifdef DANHIP_FP16
  return DANHIP_F16;
#else
  return DANHIP_BF16;
#endif
}

// ---- kernel-selection switches (process-wide; initial values from the environment): danhip_set_option / danhip_get_option.
//   "splitk"     DANHIP_SPLITK      1 (default) / 0: split-K for maps with too few output tiles (when the caller passes scratch)
//   "wgrad_slab" DANHIP_WGRAD_SLAB  1 (default: short launches) / 0 (never) / 2 (always): weight-gradient partials as stores + combine pass
//   "halo_b2"    DANHIP_HALO_B2     0 (default) / 1: second workgroup barrier per step in conv_halo.hip (the round-2 form; A/B switch)
//   "wgrad_b2"   DANHIP_WGRAD_B2    0 (default) / 1: the same for conv_wgrad_rows.hip / conv_wgrad_pw.hip
//   "halo_general_epilogue" DANHIP_HALO_GENERAL_EPILOGUE  0 (default) / 1: conv_halo.hip always takes its general epilogue (A/B of the lean one)
//   "wgrad_c8"   DANHIP_WGRAD_C8    1 (default) / 0: the first layer's weight gradient on conv_wgrad_c8.hip (0: the general kernel; A/B and tests)
//   "deform_dx_untiled" DANHIP_DEFORM_DX_UNTILED  0 (default) / 1: the deformable backward's +-1 px gather as the wave-per-pixel kernel (A/B, tests)
//   "pw_dgrad_ld_bn" DANHIP_PW_DGRAD_LD_BN  128 (default) / 256: widest tile of conv_pointwise.hip's data gradient WITH epilogue inputs (A/B:
//                the 256-wide form reads dY once but measured no faster - S3FD 1198.6 vs 1197.6 img/s, DAN 433.9 vs 436.2)
namespace {
// Thread safety ("no mutable globals" on the data path): the table is filled from the environment exactly once
// (std::call_once, before the first read or write), every value is a relaxed atomic, and the kernels' host launchers read an option
// once per call into their argument struct - a concurrent danhip_set_option changes which FORM later launches take (all forms are
// result-equivalent up to fp32 summation order), never a launch already being assembled.
struct Opt { const char* name; const char* env; int def; std::atomic<int> value; };
Opt g_opts[] = {{"splitk", "DANHIP_SPLITK", 1, {0}}, {"wgrad_slab", "DANHIP_WGRAD_SLAB", 1, {0}}, {"halo_b2", "DANHIP_HALO_B2", 0, {0}},
                {"wgrad_b2", "DANHIP_WGRAD_B2", 0, {0}}, {"halo_general_epilogue", "DANHIP_HALO_GENERAL_EPILOGUE", 0, {0}},
                {"deform_bwd_form", "DANHIP_DEFORM_BWD_FORM", 0, {0}}, {"pw_dgrad_ld_bn", "DANHIP_PW_DGRAD_LD_BN", 128, {0}},
                {"wgrad_c8", "DANHIP_WGRAD_C8", 1, {0}}, {"deform_dx_untiled", "DANHIP_DEFORM_DX_UNTILED", 0, {0}}};
std::once_flag g_opts_once;
Opt* find_opt(const char* name) {
  std::call_once(g_opts_once, [] {
    for (Opt& o : g_opts) { const char* e = getenv(o.env); o.value.store(e ? atoi(e) : o.def, std::memory_order_relaxed); }
  });
  if (!name) return nullptr;
  for (Opt& o : g_opts)
    if (strcmp(o.name, name) == 0) return &o;
  return nullptr;
}
}  // namespace

int danhip_option(const char* name) {
  Opt* o = find_opt(name);
  return o ? o->value.load(std::memory_order_relaxed) : 0;
}
extern "C" int danhip_get_option(const char* name) { return danhip_option(name); }
extern "C" int danhip_set_option(const char* name, int value) {
  Opt* o = find_opt(name);
  if (!o) { danhip_set_error("set_option: unknown option '%s'", name ? name : "(null)"); return DANHIP_EINVAL; }
  o->value.store(value, std::memory_order_relaxed);
  return DANHIP_OK;
}

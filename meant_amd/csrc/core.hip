// Library-wide plumbing: thread-local error message, version, per-device properties, runtime options and
// launch-route counters.
//
// Contract: the library is re-entrant and holds no mutable globals beyond (i) a per-DEVICE table filled
// under a mutex (CU count, which kernels already had their dynamic-LDS limit raised on that device), (ii) the option
// table (atomics, initialised once from the environment), (iii) the route counters (atomics, diagnostics only).
// Everything keyed by device is looked up with hipGetDevice() at every call: a process that drives several GPUs
// (nn.DataParallel in the reference's pretrain_mlm.py:329) gets the right state on each.
#include "common.h"
#include <stdarg.h>
#include <string.h>
#include <stdlib.h>
#include <atomic>
#include <mutex>
#include <unordered_map>

static thread_local char g_err[512] = "";

void meant_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* meant_last_error(void) { return g_err; }

extern "C" int meant_version(void) { return 200; /* 0.2.0 */ }

// ---- per-device table ---------------------------------------------------------------------------
namespace {
struct DeviceState {
  int cus = -1;                                   // -1: not queried yet
  std::unordered_map<const void*, int> lds_raised; // kernel -> the MaxDynamicSharedMemorySize it was raised to on this device
};
std::mutex g_dev_mu;
DeviceState g_dev[MEANT_MAX_DEVICES];
}  // namespace

int meant_current_device(void) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MEANT_MAX_DEVICES) return -1;
  return dev;
}

extern "C" int meant_num_cus(void) {
  const int dev = meant_current_device();
  if (dev < 0) return 0;
  std::lock_guard<std::mutex> lock(g_dev_mu);
  DeviceState& st = g_dev[dev];
  if (st.cus < 0) {
    hipDeviceProp_t p;
    st.cus = hipGetDeviceProperties(&p, dev) == hipSuccess ? p.multiProcessorCount : 0;
  }
  return st.cus;
}

int meant_raise_dyn_lds(const void* kernel, int bytes) {
  const int dev = meant_current_device();
  MEANT_REQUIRE(dev >= 0, MEANT_ERR_LAUNCH, "no current HIP device (or device index >= %d)", MEANT_MAX_DEVICES);
  std::lock_guard<std::mutex> lock(g_dev_mu);
  DeviceState& st = g_dev[dev];
  const auto it = st.lds_raised.find(kernel);
  if (it != st.lds_raised.end() && it->second >= bytes) return MEANT_OK;    // a later, larger request raises it again
  const hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  MEANT_REQUIRE(e == hipSuccess, MEANT_ERR_LAUNCH, "hipFuncSetAttribute(%d bytes of LDS) failed on device %d: %s", bytes, dev,
                hipGetErrorString(e));
  st.lds_raised[kernel] = bytes;
  return MEANT_OK;
}

// ---- runtime options ------------------------------------------------------------------------------
namespace {
struct OptionDef { const char* name; const char* env; int dflt; };
// order == enum meant_option_id (common.h)
const OptionDef k_options[MEANT_OPT_COUNT] = {
    {"nt_stream", "MEANT_NT_STREAM", 1},           // 0: one-tile-per-workgroup 256x256 NT kernel instead of the streaming one
    {"nt_dynamic", "MEANT_NT_DYNAMIC", 1},         // 0: fixed persistent tile walk; 1: per-XCD counters; 2: draw but ignore (lab); 3: steal-only (tests); 4: fixed walk in runs of one A row panel (lab)
    {"deterministic", "MEANT_DETERMINISTIC", 0},   // 1: dW / dbias, the embedding gradient (d % 8 == 0, d <= 1024) and the norm gains are bit-reproducible (ordered reductions, no float atomics)
    {"nt_grid_cap", "MEANT_NT_GRID_CAP", 0},       // tests: cap the streaming GEMM's grid (0 = one workgroup per CU)
    {"attn_short", "MEANT_ATTN_SHORT", 1},         // 0: sequences of <= 16 tokens take the tiled flash kernels instead of attn_short.hip
    {"nt_ragged", "MEANT_NT_RAGGED", 1},           // 0: ragged M as streaming head + 128 x 128 tail launch instead of the overlapped last row tile
    {"nt_split", "MEANT_NT_SPLIT", 0},             // 1: streaming GEMM: waves 0-3 issue the B tiles at the top of a K-step, waves 4-7 the A tiles after their MFMAs
    {"attn_bwd1", "MEANT_ATTN_BWD1", 1},           // 0: attention backward always as two passes (dQ, then dK / dV) instead of the single-pass kernel where it applies
    {"nt_pp", "MEANT_NT_PP", 1},                   // 0: streaming GEMM in its lock-step form (gemm_bf16_nt256s_kernel) instead of the ping-pong kernel; bit 3 (lab): staggered start
    {"tn_pp", "MEANT_TN_PP", 1},                   // 0: 256 x 256 dW kernel in its lock-step form (gemm_bf16_tn256_kernel) instead of the ping-pong kernel
};
std::atomic<int> g_opt[MEANT_OPT_COUNT];
std::once_flag g_opt_once;
void options_init() {
  std::call_once(g_opt_once, [] {
    for (int i = 0; i < MEANT_OPT_COUNT; ++i) {
      const char* v = getenv(k_options[i].env);
      g_opt[i].store(v && *v ? atoi(v) : k_options[i].dflt, std::memory_order_relaxed);
    }
  });
}
int option_index(const char* name) {
  if (!name) return -1;
  for (int i = 0; i < MEANT_OPT_COUNT; ++i)
    if (!strcmp(name, k_options[i].name)) return i;
  return -1;
}
}  // namespace

int meant_opt(int id) {
  options_init();
  return g_opt[id].load(std::memory_order_relaxed);
}

extern "C" int meant_set_option(const char* name, int value) {
  options_init();
  const int i = option_index(name);
  MEANT_REQUIRE(i >= 0, MEANT_ERR_ARG, "meant_set_option: unknown option '%s'", name ? name : "(null)");
  g_opt[i].store(value, std::memory_order_relaxed);
  return MEANT_OK;
}

extern "C" int meant_get_option(const char* name, int* value) {
  options_init();
  const int i = option_index(name);
  MEANT_REQUIRE(i >= 0 && value, MEANT_ERR_ARG, "meant_get_option: unknown option '%s'", name ? name : "(null)");
  *value = g_opt[i].load(std::memory_order_relaxed);
  return MEANT_OK;
}

// ---- launch-route counters (diagnostics: tests assert that a shape reaches the kernel it is meant to test) ----------
namespace {
const char* const k_routes[MEANT_ROUTE_COUNT] = {
    "nt128", "nt256", "nt256s", "nt256s_rot", "nt_split", "tn128", "tn256", "tn256_det", "tn_tail", "gemm_f32",
    "attn_fwd", "attn_fwd_d128", "attn_fwd_d96", "attn_bwd", "attn_bwd_d128", "attn_bwd_d96",
    "attn_generic", "attn_cls", "attn_short", "nt_overlap", "attn_bwd1",
};
std::atomic<long long> g_route[MEANT_ROUTE_COUNT];
}  // namespace

void meant_route_hit(int route) { g_route[route].fetch_add(1, std::memory_order_relaxed); }

extern "C" int64_t meant_route_count(const char* route) {
  if (!route) return -1;
  for (int i = 0; i < MEANT_ROUTE_COUNT; ++i)
    if (!strcmp(route, k_routes[i])) return (int64_t)g_route[i].load(std::memory_order_relaxed);
  return -1;
}

extern "C" void meant_route_reset(void) {
  for (int i = 0; i < MEANT_ROUTE_COUNT; ++i) g_route[i].store(0, std::memory_order_relaxed);
}

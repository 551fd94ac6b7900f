// Error plumbing and device queries of the C ABI (see include/dpi_hip.h).
#include <stdarg.h>
#include <string.h>
#include "common.h"

static thread_local char g_err[512] = "";

void dpi_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int dpi_check_launch(const char* what) {
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    dpi_set_error("%s: %s", what, hipGetErrorString(e));
    return DPI_E_LAUNCH;
  }
  return DPI_OK;
}

extern "C" const char* dpi_last_error(void) { return g_err; }
extern "C" int dpi_version(void) { return 404; }
extern "C" int dpi_conv_desc_size(void) { return (int)sizeof(dpi_conv_desc); }

extern "C" int dpi_device_info(int device, int* cus, int* lds_bytes, size_t* hbm_bytes, char* name, int name_len) {
  hipDeviceProp_t p;
  const hipError_t e = hipGetDeviceProperties(&p, device);
  if (e != hipSuccess) {
    dpi_set_error("device_info: %s", hipGetErrorString(e));
    return DPI_E_LAUNCH;
  }
  if (cus) *cus = p.multiProcessorCount;
  if (lds_bytes) *lds_bytes = (int)p.sharedMemPerBlock;
  if (hbm_bytes) *hbm_bytes = p.totalGlobalMem;
  if (name && name_len > 0) {
    strncpy(name, p.gcnArchName, name_len - 1);
    name[name_len - 1] = 0;
  }
  return DPI_OK;
}

__global__ void dpi_marker_kernel() {}

extern "C" int dpi_profile_marker(int id, void* stream) {
  DPI_REQUIRE(id >= 1 && id < 65536, "profile_marker: id %d out of range", id);
  dpi_marker_kernel<<<(unsigned)id, 64, 0, (hipStream_t)stream>>>();
  return dpi_check_launch("profile_marker");
}

// The one exported switchboard over the hidden per-knob functions (dpi_hip_internal.h): tests / tools / A-B experiments only.
extern "C" int dpi_set_option(const char* key, int value) {
  DPI_REQUIRE(key != nullptr, "set_option: key is NULL");
  struct Entry { const char* key; int lo, hi; void (*set)(int); };
  static const Entry table[] = {
      {"splitk", 0, 1, dpi_set_splitk},
      {"dual_bwd_data", 0, 1, dpi_set_dual_bwd_data},
      {"bw_pair", 0, 2, dpi_set_bw_pair},
      {"bw_workgroups", 1, 1 << 24, [](int v) { dpi_set_bw_tuning(v, -1); }},
      {"bw_xcd_order", 0, 1, [](int v) { dpi_set_bw_tuning(0, v); }},
      {"mfma_min_cout", 0, 1 << 20, dpi_set_mfma_min_cout},
      {"bwd_weight_mfma_min_cout", 0, 1 << 20, dpi_set_bwd_weight_mfma_min_cout},
      {"fewco_mfma", 0, 1, dpi_set_fewco_mfma},
      {"q4", 0, 2, [](int v) { dpi_set_q4(v, -1); }},
      {"q4_ck", 0, 4, [](int v) { dpi_set_q4(-1, v); }},
      {"q4_debug", 0, 0xFFFF, dpi_set_q4_debug},
      {"bf16_debug", 0, 0x7F, dpi_set_bf16_debug},
  };
  for (const Entry& e : table)
    if (strcmp(key, e.key) == 0) {
      DPI_REQUIRE(value >= e.lo && value <= e.hi && !(strcmp(key, "q4_ck") == 0 && value != 0 && value != 2 && value != 4),
                  "set_option: %s = %d out of range [%d, %d]", key, value, e.lo, e.hi);
      e.set(value);
      return DPI_OK;
    }
  dpi_set_error("set_option: unknown key '%s'", key);
  return DPI_E_ARG;
}

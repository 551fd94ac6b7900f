// Library-internal tuning / test hooks behind dpi_set_option() (include/dpi_hip.h).  NOT part of the C ABI: hidden visibility, so libdpi_hip.so
// does not export them; tests/host_asan/driver.cpp links the objects directly and may call them.
#pragma once
#define DPI_INTERNAL __attribute__((visibility("hidden")))
extern "C" {
DPI_INTERNAL void dpi_set_bw_tuning(int want_workgroups, int xcd_order);   // <= 0 / < 0 keeps the current value
DPI_INTERNAL void dpi_set_bf16_debug(int flags);
DPI_INTERNAL void dpi_set_splitk(int on);
DPI_INTERNAL void dpi_set_dual_bwd_data(int on);
DPI_INTERNAL void dpi_set_bw_pair(int mode);
DPI_INTERNAL void dpi_set_mfma_min_cout(int n);
DPI_INTERNAL void dpi_set_bwd_weight_mfma_min_cout(int n);
DPI_INTERNAL void dpi_set_fewco_mfma(int on);
DPI_INTERNAL void dpi_set_q4(int on, int ck);                               // on < 0 keeps; ck not in {0, 2, 4} keeps
DPI_INTERNAL void dpi_set_q4_debug(int flags);
}

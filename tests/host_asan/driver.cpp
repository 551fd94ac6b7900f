// Host-side sanitizer run ("-fsanitize=address host build").  Every .cpp / .hip of csrc/ is compiled HOST-ONLY with
// -fsanitize=address,undefined (no device code, no GPU needed) and linked with this driver, which walks the host logic of the C ABI —
// descriptor validation, launch planning, workspace sizing, the 32-bit narrowing guards — over edge-case descriptors: the shapes of
// tests/test_gpu_bench_size.py (256x128x128 bench patch, the 512x256x256 field-scale patch), volumes at and beyond the 2^29-voxel
// limit of the kernels' 32-bit byte offsets, degenerate sizes, every (k, kd, stride, precision) combination, stale descriptor layouts.
// With no device present a planned launch comes back as DPI_E_LAUNCH from the HIP runtime; everything before it (the code under
// test) has run under the sanitizers.  Exit code 0 = no sanitizer report and every expectation met.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../../include/dpi_hip.h"

static int failures = 0;
#define EXPECT(cond, ...) do { if (!(cond)) { ++failures; std::printf("FAIL %s:%d: ", __FILE__, __LINE__); std::printf(__VA_ARGS__); std::printf("\n"); } } while (0)

static dpi_conv_desc desc(int cin, int cout, int D, int H, int W, int k, int kd, int stride, int precision = 0, int io = 0) {
  dpi_conv_desc d;
  d.size = (int)sizeof(dpi_conv_desc);
  d.Cin = cin; d.Cout = cout; d.D = D; d.H = H; d.W = W; d.k = k; d.kd = kd; d.stride = stride; d.precision = precision; d.io = io;
  return d;
}

int main() {
  EXPECT(dpi_version() >= 400, "version %d", dpi_version());
  EXPECT(dpi_conv_desc_size() == (int)sizeof(dpi_conv_desc), "desc size");
  // a buffer that is big enough for the few bytes host code may legitimately read from "device" pointers: none — host code must never
  // dereference them, ASAN would flag reads of this 16-byte allocation past its end
  std::vector<float> tiny(4, 0.f);
  float* p = tiny.data();
  long n_desc = 0, n_planned = 0, n_split = 0;
  const int channels[][2] = {{64, 4}, {4, 8}, {8, 13}, {25, 16}, {67, 4}, {25, 1}, {25, 25}, {51, 32}, {137, 8}, {105, 64}, {212, 128}, {554, 35},
                             {142, 213}, {1, 1}, {3, 5}, {426, 554}, {64, 25}, {67, 25}, {17, 26}};
  const int shapes[][3] = {{256, 128, 128}, {512, 256, 256}, {128, 64, 64}, {64, 64, 64}, {16, 8, 8}, {4, 4, 4}, {1, 1, 1}, {2, 3, 5}, {9, 17, 33},
                           {1, 170, 100}, {1, 1, 7}, {33, 31, 50}, {1024, 1024, 256}, {8, 8, 68}, {7, 25, 60}};
  for (auto& ch : channels)
    for (auto& sh : shapes)
      for (int k : {1, 3})
        for (int stride : {1, 2})
          for (int prec : {0, 1, 2}) {
            const int kd = sh[0] == 1 ? 1 : k;
            // storage types (ABI 400): fp32 everywhere, and for the fp32 / bf16 arithmetic modes bf16 activations + gradients
            // (the planners pick other kernels then: no 4x4x1 / pair kernels, the bf16 kernel on every big-tile shape)
            const int io = (prec < 2 && ((n_desc / 3) & 1)) ? 15 : 0;
            dpi_conv_desc d = desc(ch[0], ch[1], sh[0], sh[1], sh[2], k, kd, stride, prec, io);
            ++n_desc;
            const bool valid = !(k == 1 && stride == 2);
            const int nblk = dpi_conv_fwd_stat_blocks(&d);
            const size_t ws = dpi_conv_bwd_weight_ws_floats(&d);
            if (!valid) { EXPECT(nblk == 0 && ws == 0, "invalid desc must plan nothing (k1 s2)"); continue; }
            EXPECT(nblk > 0, "stat blocks %d for %d->%d %dx%dx%d k%d s%d p%d", nblk, ch[0], ch[1], sh[0], sh[1], sh[2], k, stride, prec);
            EXPECT(ws > 0 && ws < ((size_t)1 << 34), "bwd-weight workspace %zu floats for %d->%d %dx%dx%d k%d s%d p%d", ws, ch[0], ch[1], sh[0], sh[1], sh[2], k, stride, prec);
            // the launchers: planning runs, the launch itself fails without a device (or succeeds on a GPU box: both fine here)
            int rc = dpi_conv_fwd(&d, p, nullptr, p, nullptr, p, nullptr, nullptr);
            EXPECT(rc == DPI_OK || rc == DPI_E_LAUNCH, "conv_fwd rc %d: %s", rc, dpi_last_error());
            rc = dpi_conv_bwd_data(&d, p, p, p, 1, nullptr);
            EXPECT(rc == DPI_OK || rc == DPI_E_LAUNCH, "conv_bwd_data rc %d: %s", rc, dpi_last_error());
            // input-channel split (ABI 301): sizing, the split launch planning, a short workspace refused before any launch
            const size_t fws = dpi_conv_fwd_ws_floats(&d), bws = dpi_conv_bwd_data_ws_floats(&d);
            EXPECT(fws < ((size_t)1 << 32) && bws < ((size_t)1 << 32), "split workspace %zu / %zu floats", fws, bws);
            if (fws) {
              ++n_split;
              rc = dpi_conv_fwd_ws(&d, p, nullptr, p, p, p, nullptr, p, fws, nullptr);
              EXPECT(rc == DPI_OK || rc == DPI_E_LAUNCH, "conv_fwd_ws rc %d: %s", rc, dpi_last_error());
              rc = dpi_conv_fwd_ws(&d, p, nullptr, p, p, p, nullptr, p, fws / 2, nullptr);
              EXPECT(rc == DPI_E_ARG || rc == DPI_E_WORKSPACE, "short split workspace accepted (rc %d)", rc);
            }
            if (bws) {
              rc = dpi_conv_bwd_data_ws(&d, p, p, p, 1, p, bws, nullptr);
              EXPECT(rc == DPI_OK || rc == DPI_E_LAUNCH, "conv_bwd_data_ws rc %d: %s", rc, dpi_last_error());
            }
            EXPECT(dpi_conv_fwd_ws(&d, p, nullptr, p, p, p, nullptr, nullptr, 16, nullptr) == DPI_E_ARG, "workspace size without a workspace accepted");
            if (k == 3 && stride == 1) {      // the 3x3(x3) + 1x1(x1) pair that reads one tensor: fused or two launches, both plan here
              dpi_conv_desc d1 = desc(ch[0], ch[1] + 9, sh[0], sh[1], sh[2], 1, 1, 1, prec);
              rc = dpi_conv_bwd_data_dual(&d, p, p, &d1, p, p, p, 0, bws ? p : nullptr, bws, nullptr);
              EXPECT(rc == DPI_OK || rc == DPI_E_LAUNCH, "conv_bwd_data_dual rc %d: %s", rc, dpi_last_error());
              dpi_conv_desc dbad = desc(ch[0] + 1, ch[1], sh[0], sh[1], sh[2], 1, 1, 1, prec);
              EXPECT(dpi_conv_bwd_data_dual(&d, p, p, &dbad, p, p, p, 0, nullptr, 0, nullptr) == DPI_E_ARG, "dual: mismatched layers accepted");
            }
            rc = dpi_conv_bwd_weight(&d, p, nullptr, p, p, p, ws, nullptr);
            EXPECT(rc == DPI_OK || rc == DPI_E_LAUNCH, "conv_bwd_weight rc %d: %s", rc, dpi_last_error());
            rc = dpi_conv_bwd_weight(&d, p, nullptr, p, p, p, ws / 2, nullptr);       // short workspace must be refused before any launch
            EXPECT(ws < 2 || rc == DPI_E_WORKSPACE, "short workspace accepted (rc %d)", rc);
            ++n_planned;
          }
  // the 32-bit byte-offset limit of the stencil kernels: 2^29 voxels per channel
  {
    dpi_conv_desc d = desc(4, 4, 1024, 1024, 512, 3, 3, 1);
    EXPECT(dpi_conv_fwd(&d, p, nullptr, p, nullptr, p, nullptr, nullptr) == DPI_E_ARG, "2^29-voxel patch must be refused");
    d = desc(4, 4, 2047, 2047, 2047, 3, 3, 1);
    EXPECT(dpi_conv_fwd_stat_blocks(&d) == 0 && dpi_conv_bwd_weight_ws_floats(&d) == 0, "8.6e9-voxel patch must be refused");
    d = desc(4, 4, 1023, 1024, 512, 3, 3, 1);
    EXPECT(dpi_conv_fwd_stat_blocks(&d) > 0, "just below the limit must plan");
  }
  // malformed descriptors
  {
    dpi_conv_desc d = desc(4, 4, 8, 8, 8, 3, 3, 1);
    d.size = 36;
    EXPECT(dpi_conv_fwd(&d, p, nullptr, p, nullptr, p, nullptr, nullptr) == DPI_E_ARG && std::strstr(dpi_last_error(), "size"), "stale size field");
    EXPECT(dpi_conv_bwd_weight_ws_floats(&d) == 0, "stale size field must size nothing");
    for (int bad = 0; bad < 8; ++bad) {
      d = desc(4, 4, 8, 8, 8, 3, 3, 1);
      switch (bad) {
        case 0: d.Cin = 0; break; case 1: d.Cout = -1; break; case 2: d.k = 5; break; case 3: d.kd = 2; break; case 4: d.stride = 3; break;
        case 5: d.precision = 7; break; case 6: d.W = 0; break; case 7: d.kd = 1; break;      /* 2-D kernel on D = 8 */
      }
      EXPECT(dpi_conv_fwd(&d, p, nullptr, p, nullptr, p, nullptr, nullptr) == DPI_E_ARG, "bad descriptor %d accepted", bad);
      EXPECT(dpi_conv_bwd_data(&d, p, p, p, 0, nullptr) == DPI_E_ARG, "bad descriptor %d accepted by bwd_data", bad);
      EXPECT(dpi_conv_bwd_weight(&d, p, nullptr, p, p, p, 1 << 20, nullptr) == DPI_E_ARG, "bad descriptor %d accepted by bwd_weight", bad);
    }
    EXPECT(dpi_conv_fwd(nullptr, p, nullptr, p, nullptr, p, nullptr, nullptr) == DPI_E_ARG, "null descriptor");
    d = desc(4, 4, 8, 8, 8, 3, 3, 1);
    EXPECT(dpi_conv_fwd(&d, nullptr, nullptr, p, nullptr, p, nullptr, nullptr) == DPI_E_ARG, "null tensor");
  }
  // the other sizing functions: monotone, non-zero, no overflow at field scale
  {
    const size_t Vs[] = {1, 63, 64, 65, 1u << 20, (size_t)256 * 128 * 128, (size_t)512 * 256 * 256, (size_t)1 << 31, ((size_t)1 << 33) + 5};
    for (size_t V : Vs) {
      for (int C : {1, 4, 25, 67, 554}) EXPECT(dpi_stat_blocks(C, V) >= 1, "stat_blocks(%d, %zu)", C, V);
      EXPECT(dpi_loss_ws_doubles(V) >= 8, "loss ws %zu", V);
      EXPECT(dpi_max_ws_floats(V) >= 1, "max ws %zu", V);
    }
    EXPECT(dpi_upsample2x_bwd_ws_floats(426, 16, 8, 8, 32, 16, 16, 1) > 0, "upsample ws");
    EXPECT(dpi_upsample2x_bwd_ws_floats(51, 128, 64, 64, 256, 128, 128, 1) > 0, "upsample ws full size");
    EXPECT(dpi_upsample2x_bwd_ws_floats(51, 128, 64, 64, 255, 127, 127, 1) > 0, "upsample ws cropped");
  }
  EXPECT(n_split > 0, "no descriptor exercised the input-channel split");
  std::printf("host sanitizer driver: %ld descriptors, %ld planned launch triples (%ld with an input-channel split), %d failures\n", n_desc, n_planned, n_split, failures);
  return failures ? 1 : 0;
}

#!/usr/bin/env python3
"""Micro-benchmark of single convolution launches through the C ABI (HIP events on the launch stream).

    python tools/bench_conv.py [--cases name ...] [--reps 10] [--mfma-min-cout 8] [--shape 256 128 128]
"""
import argparse
import ctypes as C
import sys
import os

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from deep_prior_interpolation_amd import _lib, ops  # noqa: E402

CASES = {  # name: (Cin, Cout, k, stride, level)  level = spatial down-sampling exponent
    "enc0_64_4": (64, 4, 3, 1, 0), "enc0_4_8": (4, 8, 3, 1, 0), "enc0_8_13": (8, 13, 3, 1, 0), "res0_25_16": (25, 16, 3, 1, 0),
    "dec0_67_4": (67, 4, 3, 1, 0), "out_25_1": (25, 1, 3, 1, 0), "down0_25_25": (25, 25, 3, 2, 0),
    "enc1_25_8": (25, 8, 3, 1, 1), "enc1_8_17": (8, 17, 3, 1, 1), "enc1_17_26": (17, 26, 3, 1, 1), "res1_51_32": (51, 32, 3, 1, 1),
    "dec1_137_8": (137, 8, 3, 1, 1), "down1_51_51": (51, 51, 3, 2, 1),
    "enc2_51_17": (51, 17, 3, 1, 2), "enc2_35_53": (35, 53, 3, 1, 2), "res2_105_64": (105, 64, 3, 1, 2), "dec2_276_17": (276, 17, 3, 1, 2),
    "enc3_71_106": (71, 106, 3, 1, 3), "res3_212_128": (212, 128, 3, 1, 3), "enc4_142_213": (142, 213, 3, 1, 4),
    "dec3_554_35": (554, 35, 3, 1, 3), "dec4_212_71": (212, 71, 3, 1, 4), "enc4_71_142": (71, 142, 3, 1, 4), "enc3_35_71": (35, 71, 3, 1, 3),
    "dec3_105_35": (105, 35, 3, 1, 3), "enc2_17_35": (17, 35, 3, 1, 2), "dec2_51_17": (51, 17, 3, 1, 2),
    "down2_105_105": (105, 105, 3, 2, 2), "down3_212_212": (212, 212, 3, 2, 3),
    "sc0_64_25": (64, 25, 1, 1, 0), "sc0_67_25": (67, 25, 1, 1, 0), "res0_25_16_k1": (25, 16, 1, 1, 0), "sc1_137_51": (137, 51, 1, 1, 1),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", nargs="*", default=["res0_25_16", "enc0_8_13", "enc0_64_4", "dec0_67_4", "res1_51_32", "down0_25_25", "sc0_64_25"])
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--shape", type=int, nargs=3, default=[256, 128, 128])
    ap.add_argument("--mfma-min-cout", type=int, default=None)
    ap.add_argument("--bw-mfma-min-cout", type=int, default=None)
    ap.add_argument("--which", nargs="*", default=["fwd", "bwd_data", "bwd_weight"])
    ap.add_argument("--precision", default="fp32", choices=["fp32", "bf16", "bf16mm", "split"])
    ap.add_argument("--storage", default="fp32", choices=["fp32", "bf16"], help="storage type of the activation / gradient tensors (dpi_conv_desc.io)")
    ap.add_argument("--bf16-debug", type=int, default=0)
    ap.add_argument("--bw-want", type=int, default=0, help="backward-weight plan: workgroups aimed at (dpi_set_bw_tuning)")
    ap.add_argument("--bw-xcd", type=int, default=-1, help="backward-weight plan: XCD-aware workgroup order 0/1")
    ap.add_argument("--accumulate", action="store_true", help="backward-data adds into dx (gradient fan-in)")
    ap.add_argument("--q4", type=int, default=-1, help="4x4x1-MFMA few-output-channel kernel: 0 off, 1 where it pays, 2 forced (dpi_set_q4)")
    ap.add_argument("--q4-debug", type=int, default=0, help="phase-skipping bits of the q4 kernel (timing experiments, wrong results)")
    ap.add_argument("--q4-ck", type=int, default=-1, help="... its input channels per chunk (2 or 4)")
    a = ap.parse_args()
    L = _lib.load()
    if a.mfma_min_cout is not None:
        L.set_option("mfma_min_cout", a.mfma_min_cout)
    if a.bw_mfma_min_cout is not None:
        L.set_option("bwd_weight_mfma_min_cout", a.bw_mfma_min_cout)
    if a.bw_want > 0:
        L.set_option("bw_workgroups", a.bw_want)
    if a.bw_xcd >= 0:
        L.set_option("bw_xcd_order", a.bw_xcd)
    if a.q4 >= 0:
        L.set_option("q4", a.q4)
    if a.q4_ck in (0, 2, 4):
        L.set_option("q4_ck", a.q4_ck)
    if a.q4_debug:
        L.set_option("q4_debug", a.q4_debug)
    ops.set_precision(a.precision)
    if a.bf16_debug:
        L.set_option("bf16_debug", a.bf16_debug)
    dev = "cuda"
    print("%-16s %-10s %10s %9s %8s" % ("case", "kernel", "ms", "TFLOP/s", "GB/s(alg)"))
    for name in a.cases:
        cin, cout, k, s, lvl = CASES[name]
        shp = tuple(max(1, n >> lvl) for n in a.shape)
        adt = torch.bfloat16 if a.storage == "bf16" else torch.float32
        x = torch.randn((1, cin) + shp, device=dev).to(adt)
        w = torch.randn((cout, cin, k, k, k), device=dev) * 0.05
        b = torch.randn(cout, device=dev)
        d = ops.make_desc(x, w, s, adt)
        osh = ops.desc_out_dims(d)
        y = torch.empty((1, cout) + osh, device=dev, dtype=adt)
        dy = torch.randn((1, cout) + osh, device=dev).to(adt)
        dx = torch.empty_like(x)
        dw = torch.empty_like(w)
        vo = osh[0] * osh[1] * osh[2]
        flop = 2.0 * cin * k ** 3 * cout * vo
        byt = x.element_size() * float(x.numel() + y.numel())
        fns = {"fwd": lambda: ops.raw_conv_fwd(d, x, None, w, b, y), "bwd_data": lambda: ops.raw_conv_bwd_data(d, dy, w, dx, accumulate=a.accumulate),
               "bwd_weight": lambda: ops.raw_conv_bwd_weight(d, x, None, dy, dw)}
        for which in a.which:
            fn = fns[which]
            fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.reps):
                fn()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / a.reps
            print("%-16s %-10s %10.4f %9.2f %8.0f" % (name, which, ms, flop / ms / 1e9, byt / ms / 1e6), flush=True)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""gpurun_out/<prefix>_hbm_iteration_{FETCH,WRITE}_SIZE.txt (+ _family_*) -> the JSON bench.py reads for `roofline.traffic` /
`roofline.measured_over_algorithmic` (profiles/r03_traffic.json).  Units: rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KB (1024 B);
on gfx950 FETCH_SIZE counts 64-byte units as 32 (MI355X_MICROARCH.md, HBM / rocprofv3 section; confirmed by the round-1 calibration
on kernels with known byte counts, profiles/r01_traffic_dominant_conv.json): reads = FETCH_SIZE x 2, writes = WRITE_SIZE x 1."""
import hashlib
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import bmin_bytes, kernel_source_sha256  # noqa: E402  (digest of the kernel sources: bench.py refuses a counter file of another build)

pre, iters = sys.argv[1], int(sys.argv[2])
precision = sys.argv[3] if len(sys.argv) > 3 else "fp32"


def totals(path):
    per_iter, kernels = None, {}
    for line in open(path):
        m = re.match(r"\w+: total ([\d.e+]+) over all dispatches = ([\d.e+]+) per iteration", line)
        if m:
            per_iter = float(m.group(2))
            continue
        m = re.match(r"\s+([\d.e+]+) per iteration\s+[\d.]+ %\s+(.*)", line)
        if m:
            kernels[m.group(2).strip()] = float(m.group(1))
    return per_iter, kernels


def family(path):
    """per-launch averages of the 3x3x3 stride-1 backward-weight kernels (conv_bwd_weight_mfma_kernel<3, 1, 8, 2, *>, smallco)."""
    rows, key = [], None
    for line in open(path):
        if not line.startswith(" "):
            key = line.strip()
        else:
            m = re.match(r"\s+(\w+)\s+n=(\d+) avg=([\d.e+]+)", line)
            if m and m.group(1) in ("FETCH_SIZE", "WRITE_SIZE") and key and (
                    re.search(r"conv_bwd_weight_mfma(_merged|_pair)?_kernel<3, 1, 8, 2", key) or "smallco" in key
                    or "conv_bf16_bwd_weight_kernel" in key):
                rows.append((key, int(m.group(2)), float(m.group(3))))
    return rows


f_it, f_k = totals(pre + "_hbm_iteration_FETCH_SIZE.txt")
w_it, w_k = totals(pre + "_hbm_iteration_WRITE_SIZE.txt")
fam_f, fam_w = family(pre + "_family_FETCH_SIZE.txt"), family(pre + "_family_WRITE_SIZE.txt")
if not fam_f or not fam_w or not f_k or not w_k:
    # (the bf16 family files of round 5 were empty — the --match filter of the collecting step named the fp32 kernels — and the
    #  JSON was written all the same)
    sys.exit("make_traffic_json: empty counter input (%d / %d family rows, %d / %d kernels in the whole-iteration breakdown) under prefix %s — "
             "re-run the pmc_iter step (tools/profile_r06.sh; FAMILY_MATCH selects the backward-weight kernels by name)" % (len(fam_f), len(fam_w), len(f_k), len(w_k), pre))
n_launch = sum(n for _, n, _ in fam_f)
fam_read = sum(n * v for _, n, v in fam_f) * 1024.0 * 2.0
fam_write = sum(n * v for _, n, v in fam_w) * 1024.0
read_b, write_b = f_it * 1024.0 * 2.0, w_it * 1024.0
top = sorted(f_k.items(), key=lambda kv: -kv[1])[:8]
out = {
    "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes over `python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline "
              "--no-other-modes --no-c3-extra` (tools/profile_r03.sh pmc_iter), weight-gradient side stream off, %d iterations per pass; "
              "units KB = 1024 B; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950" % iters,
    "precision": precision,
    "csrc_sha256": kernel_source_sha256(),
    "libdpi_hip_so_sha256": hashlib.sha256(open(os.path.join(ROOT, "deep_prior_interpolation_amd", "libdpi_hip.so"), "rb").read()).hexdigest(),
    "whole_iteration": {"FETCH_SIZE_KB_per_iteration": f_it, "WRITE_SIZE_KB_per_iteration": w_it, "read_bytes": read_b, "write_bytes": write_b,
                        "hbm_bytes_per_iteration": read_b + write_b, "algorithmic_bytes_per_iteration": bmin_bytes(256 * 128 * 128, precision),
                        "measured_over_algorithmic": round((read_b + write_b) / bmin_bytes(256 * 128 * 128, precision), 3),
                        "top_readers_KB_raw_per_iteration": {k: v for k, v in top}},
    "dominant_family": {"family": "conv_bwd_weight k3 s1",
                        "kernels": "conv_bwd_weight_mfma_kernel<3, 1, 8, 2, *> / conv_bwd_weight_mfma_pair_kernel<3, 1, 8, 2> / ..._merged_kernel (both orientations) and conv_bwd_weight_smallco_kernel",
                        "launches_counted": n_launch, "launches_per_iteration": n_launch / float(iters),
                        "read_bytes_per_iteration": fam_read / iters, "write_bytes_per_iteration": fam_write / iters,
                        "hbm_bytes_per_launch_mean": (fam_read + fam_write) / max(n_launch, 1),
                        "per_grid_KB_raw": {"FETCH_SIZE": [[k, n, v] for k, n, v in fam_f], "WRITE_SIZE": [[k, n, v] for k, n, v in fam_w]}},
}


def kernel_family(name):
    """3x3x3 stride-1 family of a kernel name of the whole-iteration breakdown (template arguments as rocprofv3 prints them)."""
    if re.search(r"conv_bwd_weight_mfma(_merged|_pair)?_kernel<3, 1, 8, 2", name) or "smallco" in name:
        return "conv_bwd_weight k3 s1"
    m = re.match(r"conv_mfma_kernel<3, \d+, \d+, (true|false), (\d)", name)          # <KD, NR, NH, FLIP, S, ...>
    if m and m.group(2) == "1":
        return "conv_bwd_data k3 s1" if m.group(1) == "true" else "conv_fwd k3 s1"
    m = re.match(r"conv_q4_mfma_kernel<\d+, \d+, \d+, (true|false)", name)          # <R, NB, CK, FLIP, AL>
    if m:
        return "conv_bwd_data k3 s1" if m.group(1) == "true" else "conv_fwd k3 s1"
    m = re.match(r"conv_q4i_mfma_kernel<\d+, \d+, (true|false)", name)               # <R, NB, FLIP, AL, KHP>
    if m:
        return "conv_bwd_data k3 s1" if m.group(1) == "true" else "conv_fwd k3 s1"
    m = re.match(r"conv_bf16_kernel<3, \d+, \d+, (true|false)", name)                 # <KD, NR, NH, FLIP, NS, XB, YB> (bf16 / split modes)
    if m:
        return "conv_bwd_data k3 s1" if m.group(1) == "true" else "conv_fwd k3 s1"
    if name.startswith("conv_bf16_bwd_weight_kernel"):
        return "conv_bwd_weight k3 s1"
    if name.startswith("splitk_reduce_kernel"):
        return None                                                                   # serves forward and backward-data launches alike: left out
    return None


# every 3x3x3 stride-1 family from the whole-iteration per-kernel breakdown, so that bench.py finds the traffic of whichever family
# dominates the run (backward-weight and backward-data are within 1 % of each other)
fams = {}
for k in set(f_k) | set(w_k):
    fam = kernel_family(k)
    if fam:
        e = fams.setdefault(fam, {"read_bytes_per_iteration": 0.0, "write_bytes_per_iteration": 0.0, "kernels": []})
        e["read_bytes_per_iteration"] += f_k.get(k, 0.0) * 1024.0 * 2.0
        e["write_bytes_per_iteration"] += w_k.get(k, 0.0) * 1024.0
        e["kernels"].append(k)
for e in fams.values():
    e["hbm_bytes_per_iteration"] = e["read_bytes_per_iteration"] + e["write_bytes_per_iteration"]
    e["kernels"].sort()
out["families"] = fams
print(json.dumps(out, indent=1))

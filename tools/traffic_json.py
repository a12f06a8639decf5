#!/usr/bin/env python3
"""HBM bytes per launch of the k=7 convolution class from the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE).

    python tools/traffic_json.py gpurun_out/prof_r01/pmc_fetch/**/p_counter_collection.csv gpurun_out/prof_r01/pmc_write/**/p_counter_collection.csv \
        > profiles/traffic_conv_k7.json
FETCH_SIZE / WRITE_SIZE are reported in KiB; FETCH is doubled (gfx950 counts 128-byte requests as 64 B, MI355X_MICROARCH.md).
"""
import csv, json, re, sys

def mean_kib(path):
    tot = n = 0
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"]
        m = re.search(r"conv_mfma_kernel<\d+, \d+, 7,", name)
        if m:
            tot += float(r["Counter_Value"]); n += 1
    return tot / max(n, 1), n

f, nf = mean_kib(sys.argv[1])
w, nw = mean_kib(sys.argv[2])
json.dump({"kernel_class": "conv_k7 (every k=7 convolution launch of the step: residual-unit convs fused and unfused, decoder in-conv, stem, head)",
           "launches_profiled": nf, "fetch_KiB_per_launch": f, "write_KiB_per_launch": w,
           "hbm_bytes_per_launch": (2 * f + w) * 1024,
           "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes of `bench.py --steps 2 --warmup 1`; FETCH doubled "
                     "(gfx950 counts 128-B requests as 64 B, MI355X_MICROARCH.md); per-kernel table in profiles/r01_dac_b32.hbm_traffic_pmc.txt"},
          sys.stdout, indent=1)

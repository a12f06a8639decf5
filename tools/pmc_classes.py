#!/usr/bin/env python3
"""Per kernel-class HBM traffic and matrix-core busy fraction from rocprofv3 --pmc passes of a bench command.

    NC_LAUNCH_LOG=gpurun_out/X/launch_<pass>.log rocprofv3 --pmc FETCH_SIZE --kernel-trace ... -- python3 bench.py ...   (one pass per counter set)
    python tools/pmc_classes.py --key dac44k --out profiles/traffic.json \
        fetch=<dir>/p_counter_collection.csv:<launch log>  write=<dir>/...:<log>  sq=<dir>/...:<log>

Every dispatch is assigned a kernel class: by kernel name (dwconv_kernel -> dwconv, conv_thin_kernel -> head, ...) and, for the
implicit-GEMM template that serves several classes under one name, by zipping the engine's launch log (NC_LAUNCH_LOG, one line
per template launch in launch order) with the template's counter rows in dispatch order (thread counts must agree).
FETCH_SIZE / WRITE_SIZE are reported in KiB; FETCH is doubled (gfx950 tallies 128-B read requests as 64 B,
MI355X_MICROARCH.md "HBM").  mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x SQ_BUSY_CU_CYCLES); valu_busy = SQ_ACTIVE_INST_VALU (quad-cycles) / SQ_BUSY_CU_CYCLES; pipe_busy = their sum: SQ_BUSY_CU_CYCLES sums, over
the CUs, the cycles a CU had a wave resident; the MFMA counter sums busy cycles over the SIMDs.
"""
import argparse
import csv
import json
import os
import re
import sys
from collections import OrderedDict, defaultdict

KC = ("conv_k7", "conv_k1", "conv_down", "conv_up", "conv_misc", "rvq", "elem", "dwconv", "norm", "attn", "lstm", "stem", "head")
BY_NAME = [(r"conv1x1_kernel|conv1x1_stream_kernel|skinny_proj|snac_unit_kernel|res_a_kernel", "conv_k1"), (r"conv3_stream_kernel", "conv_misc"), (r"conv_small|down2_kernel|down4_kernel|down5_kernel", "conv_down"), (r"up2_kernel", "conv_up"),   # (the short-row kernel also serves two k=7 layers of the conv_misc class)
           (r"dwconv_kernel|dwconv_vec_kernel", "dwconv"), (r"layernorm_ct|layernorm_tile|gn_block|gn_final|gn_", "norm"),
           (r"local_attn", "attn"), (r"lstm_|lstm2_", "lstm"), (r"stem_", "stem"), (r"conv_thin(_inm)?_kernel", "head"),
           (r"vq_argmin|vq_gather|euclid_vq|euclid_rvq|dac_rvq|emb_sum", "rvq"),
           (r"avg_pool|rvq_update|pad_act|scale_kernel|overlap_add|rms_|randn", "elem")]   # (rms_ covers rms_scale_kernel)


def load(path_csv, path_log):
    rows = OrderedDict()      # dispatch id -> (name, threads, {counter: value}, start, end)
    for r in csv.DictReader(open(path_csv)):
        d = int(r["Dispatch_Id"])
        e = rows.setdefault(d, [r["Kernel_Name"], int(r["Grid_Size"]), {}, int(r["Start_Timestamp"]), int(r["End_Timestamp"])])
        e[2][r["Counter_Name"]] = e[2].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    log = [l.split() for l in open(path_log)] if path_log and os.path.exists(path_log) else []
    li = 0
    out = []
    for d in sorted(rows):
        name, threads, ctr, t0, t1 = rows[d]
        cls = None
        if "conv_mfma_kernel" in name:
            if li < len(log):
                rec = log[li]
                li += 1
                if int(rec[2]) != threads:
                    raise SystemExit(f"launch log out of step at dispatch {d}: log says {rec[2]} threads, trace {threads}")
                cls = KC[int(rec[1])]
            else:
                cls = "conv_misc"
        else:
            for pat, c in BY_NAME:
                if re.search(pat, name):
                    cls = c
                    break
        if cls:
            out.append((cls, name, threads, ctr, t1 - t0))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--key", required=True)
    ap.add_argument("--out", required=True)
    ap.add_argument("--skip-frac", type=float, default=0.0, help="drop this leading fraction of every class's launches (warm-up)")
    ap.add_argument("--lib", default=None, help="the engine library the passes ran on: its SHA-256 is stored beside the counters (bench.py "
                    "reports traffic only when the library it loads has the same hash)")
    ap.add_argument("passes", nargs="+")
    a = ap.parse_args()
    acc = defaultdict(lambda: defaultdict(float))
    cnt = defaultdict(lambda: defaultdict(int))
    for p in a.passes:
        tag, rest = p.split("=", 1)
        path_csv, _, path_log = rest.partition(":")
        for cls, name, threads, ctr, dur in load(path_csv, path_log):
            for k, v in ctr.items():
                acc[cls][k] += v
                cnt[cls][k] += 1
            acc[cls]["dur_ns:" + tag] += dur
            cnt[cls]["dur_ns:" + tag] += 1
    res = {}
    for cls in acc:
        e = {}
        n = max(cnt[cls].values())
        e["launches_profiled"] = n
        f = acc[cls].get("FETCH_SIZE")
        w = acc[cls].get("WRITE_SIZE")
        if f is not None:
            e["fetch_KiB_per_launch"] = f / cnt[cls]["FETCH_SIZE"]
        if w is not None:
            e["write_KiB_per_launch"] = w / cnt[cls]["WRITE_SIZE"]
        if f is not None and w is not None:
            e["hbm_bytes_per_launch"] = (2 * e["fetch_KiB_per_launch"] + e["write_KiB_per_launch"]) * 1024
        mb, bc = acc[cls].get("SQ_VALU_MFMA_BUSY_CYCLES"), acc[cls].get("SQ_BUSY_CU_CYCLES")
        if mb is not None and bc:
            e["mfma_busy"] = round(mb / (4.0 * bc), 4)
            e["SQ_VALU_MFMA_BUSY_CYCLES_per_launch"] = mb / cnt[cls]["SQ_VALU_MFMA_BUSY_CYCLES"]
            e["SQ_BUSY_CU_CYCLES_per_launch"] = bc / cnt[cls]["SQ_BUSY_CU_CYCLES"]
        va = acc[cls].get("SQ_ACTIVE_INST_VALU")
        if va is not None and bc:
            # round 6: on gfx950 the f32 matrix-core instructions and the vector ALU share one pipe per SIMD (profiles/r06_pipe_counters_encodec.txt):
            # SQ_ACTIVE_INST_VALU counts quad-cycles of vector instructions, so valu_busy = 4 * it / (4 SIMDs * busy CU cycles), and
            # pipe_busy = mfma_busy + valu_busy is the fraction of the SHARED pipe's cycles the class keeps occupied
            e["valu_busy"] = round(va / bc, 4)
            if "mfma_busy" in e:
                e["pipe_busy"] = round(e["mfma_busy"] + e["valu_busy"], 4)
        res[cls] = e
    res["_method"] = ("rocprofv3 --pmc passes (FETCH_SIZE | WRITE_SIZE | SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES, kernel trace only) of the bench "
                      "command; classes by kernel name + engine launch log (tools/pmc_classes.py); FETCH doubled per MI355X_MICROARCH.md")
    if a.lib:
        import hashlib
        res["_lib_sha256"] = hashlib.sha256(open(a.lib, "rb").read()).hexdigest()
    allj = {}
    if os.path.exists(a.out):
        allj = json.load(open(a.out))
    allj[a.key] = res
    json.dump(allj, open(a.out, "w"), indent=1, sort_keys=True)
    for cls, e in sorted(res.items()):
        if cls.startswith("_"):
            continue
        print(cls, json.dumps(e))


if __name__ == "__main__":
    main()

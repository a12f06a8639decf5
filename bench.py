#!/usr/bin/env python3
"""Headline benchmark: audio-seconds/sec of DAC-44.1 kHz encode+decode (x real-time), B=32 x 1 s clips per GPU.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One "step" = DAC.Encode (pad -> encoder -> 9-stage RVQ -> int64 codes + zQ) followed by DAC.Decode(zQ) on one
batch of 32 synthetic 1 s clips that is already resident in HBM.  With N>1 every rank (one process per GPU)
runs its own 32-clip shard (weak scaling, BASELINE config C4 = 256 clips over 8 GPUs) and the emitted code
tensors are all-gathered over RCCL on a side stream while the local decode runs.

The JSON line carries `roofline` for the dominant kernel class (the dilated k=7 residual-unit convolutions,
fp32 matrix-core implicit GEMM: bound "mfma", peak = 157.3 TFLOP/s dense fp32 MFMA on MI355X) measured with
HIP events on the launch stream, and `cpu_baseline`: the C oracle (kind "port") timed on the host cores on a
bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

FP32_MFMA_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
HBM_PEAK_GBS = 8000.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=32, help="clips per GPU")
    ap.add_argument("--seconds", type=float, default=1.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-clips", type=int, default=8, help="clips in the bounded CPU-baseline sample")
    ap.add_argument("--check", action="store_true", help="verify clip 0 of the last step against the oracle")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from neuralcodecs_amd import DAC, DACConfig
    from neuralcodecs_amd.weights import dac_synthetic_state_dict, save_blob, synthetic_pcm

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch multi-GPU runs with torch.distributed.run (one process per GPU)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # NC_BENCH_FORCE_DIST=1 exercises the RCCL path (process group, side-stream all-gather, barrier) with a single rank
    use_dist = world > 1 or os.environ.get("NC_BENCH_FORCE_DIST") == "1"
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    cfg = DACConfig.dac_44khz()
    T = int(round(args.seconds * cfg.sample_rate))
    B = args.batch
    blob = save_blob(dac_synthetic_state_dict(cfg, seed=42))
    model = DAC(cfg, device_index=local_rank)
    model.load_blob(blob)

    # synthetic clips (seed 1234 + global clip index), resident in HBM before the timed region
    pcm_h = synthetic_pcm(B, 1, T, cfg.sample_rate, seed=1234 + rank * B)
    pcm = torch.from_numpy(pcm_h).to(dev)
    Tz = model.frames(T)
    gathered = torch.empty((world * B, cfg.n_codebooks, Tz), dtype=torch.int64, device=dev) if use_dist else None
    side = torch.cuda.Stream(device=dev) if use_dist else None

    from neuralcodecs_amd import parallel

    def step():
        z, codes, lat, _, _ = model.encode(pcm)
        if use_dist:
            # all-gather the emitted codes on a side stream; the local decode only needs local z
            ev = torch.cuda.Event()
            ev.record()
            with torch.cuda.stream(side):
                side.wait_event(ev)
                parallel.all_gather_codes(codes, world * B, out=gathered)
        audio = model.decode(z)
        if use_dist:
            torch.cuda.current_stream().wait_stream(side)
        return codes, z, audio

    def sync():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync()
    model.profile_enable(True)
    model.profile_reset()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        codes, z, audio = step()
    sync()
    dt = time.perf_counter() - t0
    prof = model.profile_read()
    model.profile_enable(False)
    if use_dist:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    audio_seconds = world * B * args.seconds * args.steps
    value = audio_seconds / dt
    ms_per_step = dt / args.steps * 1e3

    out = None
    if rank == 0:
        k7 = prof["conv_k7"]
        ach_tflops = (k7["flops"] / (k7["ms"] * 1e-3)) / 1e12 if k7["ms"] > 0 else 0.0
        total_kernel_ms = sum(v["ms"] for v in prof.values())
        total_flops = sum(v["flops"] for v in prof.values())
        # HBM bytes per launch of the same kernel class from the PMC counters (separate rocprofv3 --pmc passes of this command,
        # FETCH_SIZE doubled per MI355X_MICROARCH.md; summary committed under profiles/): a measured constant of the build
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic_conv_k7.json")
        if os.path.exists(tpath) and B == 32 and abs(args.seconds - 1.0) < 1e-9:
            traffic = round(json.load(open(tpath))["hbm_bytes_per_launch"])
        roofline = {
            "kernel": "conv_mfma_kernel<K=7> (dilated k=7 residual-unit conv, fp32 MFMA implicit GEMM)",
            "bound": "mfma", "achieved": round(ach_tflops, 3), "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": round(ach_tflops / FP32_MFMA_PEAK_TFLOPS, 4), "traffic": traffic,
            "traffic_unit": "HBM bytes per launch (PMC FETCH_SIZE x2 + WRITE_SIZE, profiles/r01_dac_b32.hbm_traffic_pmc.txt)",
            "algorithmic_bytes_per_launch": round(k7["bytes"] / max(k7["launches"], 1)),
            "launches_per_step": k7["launches"] / max(args.steps, 1),
            "avg_launch_ms": k7["ms"] / max(k7["launches"], 1),
            "flops_per_launch": k7["flops"] / max(k7["launches"], 1),
            "share_of_kernel_time": round(k7["ms"] / total_kernel_ms, 4) if total_kernel_ms else None,
            "all_classes": {n: {"ms_per_step": round(v["ms"] / args.steps, 4),
                                "tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 3) if v["ms"] > 0 else 0.0,
                                "algo_GBps": round(v["bytes"] / (v["ms"] * 1e-3) / 1e9, 1) if v["ms"] > 0 else 0.0}
                            for n, v in prof.items()},
            "whole_step_tflops": round(total_flops / args.steps / (ms_per_step * 1e-3) / 1e12, 3),
        }
        cpu = None
        if not args.no_cpu_baseline and world == 1:
            from oracle import c_oracle
            ref = c_oracle.RefDAC(cfg, blob)
            n = max(1, min(args.cpu_clips, B))
            tc = time.perf_counter()
            rz, rcodes, _, _ = ref.encode(pcm_h[:n])
            raudio = ref.decode(rz)
            cdt = time.perf_counter() - tc
            cpu = {"value": round(n * args.seconds / cdt, 4), "unit": "audio-seconds/sec", "cores": int(c_oracle.lib().ref_num_threads()),
                   "kind": "port", "sample": f"{n} of the {B} clips of one step (encode+decode, C oracle with OpenMP), {cdt:.2f} s"}
            if args.check:
                same_codes = bool(np.array_equal(codes[:n].cpu().numpy(), rcodes))
                cpu["gpu_equals_oracle"] = {"codes_bit_exact": same_codes,
                                            "pcm_max_abs_diff": float(np.abs(audio[:n].cpu().numpy() - raudio).max())}
        out = {
            "metric": "audio-seconds/sec encode+decode (x real-time), DAC-44.1kHz B=32",
            "value": round(value, 2), "unit": "audio-seconds/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "DAC 44.1kHz 8kbps encode+decode, batch=%d x %.0f s clips per GPU (BASELINE configs[1])" % (B, args.seconds),
                       "clips_per_gpu": B, "clip_seconds": args.seconds, "global_batch": world * B,
                       "collective": "RCCL all_gather of int64 codes [B,9,87] per rank" if use_dist else "none"},
            "roofline": roofline, "cpu_baseline": cpu,
        }
        print(json.dumps(out), flush=True)
    if use_dist and rank == 0 and args.check:
        assert torch.equal(gathered[:B], codes), "gathered codes differ from the local codes"
    model.dispose()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

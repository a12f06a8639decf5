#!/usr/bin/env python3
"""Headline benchmark: audio-seconds/sec of DAC-44.1 kHz encode+decode (x real-time), B=32 x 1 s clips per GPU.

    python bench.py --gpus N --steps K --warmup W          # N > 1 without WORLD_SIZE: starts N fresh rank processes itself (self_launch)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
    python bench.py --gpus N --local-group ...              # ONE process drives the N devices through nc_group_create_local

One "step" = DAC.Encode (pad -> encoder -> 9-stage RVQ -> int64 codes + zQ) followed by DAC.Decode(zQ) on one
batch of 32 synthetic 1 s clips that is already resident in HBM.  With N>1 every rank (one process per GPU)
runs its own 32-clip shard (weak scaling, BASELINE config C4 = 256 clips over 8 GPUs) and the emitted code
tensors are all-gathered over RCCL on a side stream while the local decode runs.  `--config snac44k` runs the
same loop on BASELINE config C5's per-GPU share (SNAC 44.1 kHz + LocalMHA, 8 x 5 s clips per GPU, the four code
levels of a clip gathered in one collective).

The JSON line carries
  * `ms_per_step` / `value` from the wall clock around EXACTLY --steps steps (profiler off; barrier + synchronise both sides; max over
    ranks), `ms_per_step_median` from HIP events recorded between the steps, `encode_only` / `decode_only`;
  * `roofline` for the dominant kernel class (the dilated k=7 residual-unit convolutions, fp32 matrix-core implicit
    GEMM: bound "mfma", peak = 157.3 TFLOP/s dense fp32 MFMA on MI355X) measured with HIP events on the launch stream in a
    second pass of the same --steps steps (the per-launch event pairs stay out of the timed region);
  * `cpu_baseline`: the C oracle (kind "port") timed on the host cores on the step's WHOLE batch, one warm-up + 3 timed passes, median
    (~35 s per pass), the
    GPU == oracle check on those clips (`gpu_equals_oracle`, outside the timed region) and `aten_proxy`: the same graph
    as a sequence of ATen CPU operators (tools/aten_proxy.py, the closest stand-in for the reference's TorchSharp-CPU path):
    one warm-up + 3 timed passes over 8 clips, median;
  * `extra_configs` (N=1 only): BASELINE configs C3 (Encodec 48 kHz stereo, 16 x 2 s), C5's per-GPU share and C1
    (SNAC 24 kHz, 1 x 1 s) on the same GPU: ms, x real-time, compact per-class HIP-event table, dominant kernel class with its
    roofline fraction and a GPU == oracle check on one clip; their step times also as flat top-level keys (c3_... / c5_share_... / c1_...).
  The line is kept small (about 4 KB); the full per-class tables are written to gpurun_out/bench_detail.json (NC_BENCH_DETAIL=<path>).
  `roofline.traffic` comes from profiles/traffic.json only when the loaded library's SHA-256 equals the one recorded there
  (else null + "traffic_stale": true).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# the CPU-baseline legs run two OpenMP runtimes in this process (the C oracle's and PyTorch's): keep their idle workers from spinning
# against each other on many-core hosts (must be set before either starts)
os.environ.setdefault("OMP_WAIT_POLICY", "passive")
os.environ.setdefault("GOMP_SPINCOUNT", "0")

import numpy as np  # noqa: E402

FP32_MFMA_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
HBM_PEAK_GBS = 8000.0
MFMA_CLASSES = ("conv_k7", "conv_k1", "conv_down", "conv_up", "conv_misc", "lstm")   # dense contractions: priced against the fp32 matrix peak
# algorithmic work per audio-second of encode+decode (layer-fused byte model)
# (Encodec 48 kHz: a 2 s clip is cut into segments of 48000 + 48000 + 960 samples = 2.02 one-second segments of 12.28 GFLOP / 0.47 GB each,
#  i.e. 1.01 segment-equivalents per audio-second: 397 GFLOP per C3 step of 16 clips)
ALGO = {"dac44k": (201.8e9, 1.057e9), "encodec48k": (12.28e9 * 1.01, 0.47e9 * 1.01), "snac24k": (14.8e9, 0.44e9), "snac44k": (67.9e9, 1.23e9)}


def class_table(prof, steps, traffic=None):
    """Per kernel class: HIP-event time, launches, algorithmic TFLOP/s and GB/s.  With `traffic` (profiles/traffic.json of this
    workload: PMC FETCH_SIZE x2 + WRITE_SIZE per launch, a committed constant of the build) also the rate on the HBM side of the L2,
    `pmc_wire_GBps` = counted bytes per launch x launches / measured time, and its fraction of the 8 TB/s peak."""
    out = {}
    for n, v in prof.items():
        if v["launches"] == 0:
            continue
        ms = v["ms"]
        out[n] = {"ms_per_step": round(ms / steps, 4), "launches_per_step": round(v["launches"] / steps, 1),
                  "tflops": round(v["flops"] / (ms * 1e-3) / 1e12, 3) if ms > 0 else 0.0,
                  "algo_GBps": round(v["bytes"] / (ms * 1e-3) / 1e9, 1) if ms > 0 else 0.0}
        t = (traffic or {}).get(n)
        if t and ms > 0 and "hbm_bytes_per_launch" in t:
            wire = t["hbm_bytes_per_launch"] * v["launches"] / (ms * 1e-3) / 1e9
            out[n]["pmc_wire_GBps"] = round(wire, 1)
            out[n]["pmc_wire_frac_of_hbm_peak"] = round(wire / HBM_PEAK_GBS, 4)
    return out


def dominant(classes):
    """Class with the largest HIP-event time and its fraction of the roofline that bounds it."""
    if not classes:
        return None
    name = max(classes, key=lambda n: classes[n]["ms_per_step"])
    c = classes[name]
    if name in MFMA_CLASSES:
        return {"class": name, "bound": "mfma", "achieved": c["tflops"], "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": round(c["tflops"] / FP32_MFMA_PEAK_TFLOPS, 4), "ms_per_step": c["ms_per_step"]}
    return {"class": name, "bound": "hbm", "achieved": c["algo_GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(c["algo_GBps"] / HBM_PEAK_GBS, 4), "ms_per_step": c["ms_per_step"]}


_TRAFFIC_STALE = {}


def load_traffic(key):
    """HBM bytes from the PMC passes (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs, FETCH doubled per
    MI355X_MICROARCH.md), committed under profiles/traffic.json by tools/pmc_classes.py together with the SHA-256 of the engine library
    they were measured on.  Counters of ANOTHER build are not this build's traffic: when the loaded library's hash differs the entry is
    withheld (None) and traffic_stale(key) says so -- re-run tools/profile_round.sh after a kernel change."""
    p = os.path.join(ROOT, "profiles", "traffic.json")
    if not os.path.exists(p):
        return None
    t = json.load(open(p)).get(key)
    if not t:
        return None
    from neuralcodecs_amd import _lib
    if t.get("_lib_sha256") != _lib.lib_sha256():
        _TRAFFIC_STALE[key] = True
        return None
    _TRAFFIC_STALE[key] = False
    return t


def traffic_stale(key):
    return bool(_TRAFFIC_STALE.get(key, False))


def compact_classes(classes):
    """name -> [ms per step, launches per step, achieved (TFLOP/s for the matrix-core classes, algorithmic GB/s otherwise), fraction of
    the bound's peak]: the bench line stays small; the full tables go to the detail file."""
    out = {}
    for n, c in classes.items():
        if n in MFMA_CLASSES:
            out[n] = [round(c["ms_per_step"], 3), c["launches_per_step"], c["tflops"], round(c["tflops"] / FP32_MFMA_PEAK_TFLOPS, 3)]
        else:
            out[n] = [round(c["ms_per_step"], 3), c["launches_per_step"], c["algo_GBps"], round(c["algo_GBps"] / HBM_PEAK_GBS, 3)]
    return out


def timed(fn, steps, warmup, sync):
    for _ in range(warmup):
        fn()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        r = fn()
    sync()
    return (time.perf_counter() - t0) / steps, r


def extra_configs(dev, steps, warmup, check):
    """BASELINE configs C3 / C5-share / C1 on this GPU (inputs resident in HBM; HIP-event class times from the engine's profiler)."""
    import torch
    from neuralcodecs_amd import SNAC, Encodec
    from neuralcodecs_amd.config import EncodecConfig, SNACConfig
    from neuralcodecs_amd.weights import encodec_synthetic_state_dict, save_blob, snac_noise, snac_synthetic_state_dict, synthetic_pcm
    out = {}
    # All timings first, the GPU == oracle comparisons afterwards: the oracle's OpenMP team (and the ATen proxy's) leaves host threads
    # spinning for a while, which the launch-rate-sensitive one-clip configuration feels (C1 1.31 ms undisturbed, 1.50 ms measured
    # right behind an oracle pass).
    deferred = []

    def run(name, m, fn, B, secs, algo_key, oracle_check):
        dt, _ = timed(fn, steps, warmup, torch.cuda.synchronize)   # the step time: profiler off (no per-launch event pairs)
        m.profile_enable(True)
        m.profile_reset()
        timed(fn, steps, 0, torch.cuda.synchronize)               # the class table: HIP events around every launch
        prof = m.profile_read()
        m.profile_enable(False)
        tr = load_traffic(algo_key)
        classes = class_table(prof, steps, tr)
        fl, by = ALGO[algo_key]
        e = {"workload": name, "B": B, "clip_seconds": secs, "ms_per_step": round(dt * 1e3, 3), "x_realtime": round(B * secs / dt, 1),
             "whole_step_tflops": round(fl * B * secs / dt / 1e12, 3), "whole_step_algo_GBps": round(by * B * secs / dt / 1e9, 1),
             "kernel_ms_per_step": round(sum(c["ms_per_step"] for c in classes.values()), 3),
             "dominant": dominant(classes), "classes": classes,
             "pmc_traffic_ref": "profiles/traffic.json#" + algo_key, "traffic_stale": traffic_stale(algo_key)}
        if tr and e["dominant"]:      # matrix-core / vector busy of the dominant class on the pipe they share (profiles/traffic.json)
            t = tr.get(e["dominant"]["class"], {})
            for kk in ("mfma_busy", "valu_busy", "pipe_busy"):
                if kk in t:
                    e["dominant"][kk] = t[kk]
        if check:
            deferred.append((e, oracle_check))
        out[algo_key if algo_key != "snac44k" else "snac44k_c5_share"] = e

    # C3: Encodec 48 kHz stereo 12 kbps, 16 x 2 s
    cfg = EncodecConfig.encodec_48khz()
    blob = save_blob(encodec_synthetic_state_dict(cfg, seed=42))
    m = Encodec(cfg)
    m.load_blob(blob)
    B, secs = 16, 2.0
    T = int(secs * cfg.sampling_rate)
    xh = synthetic_pcm(B, cfg.channels, T, cfg.sampling_rate, seed=1234)
    x = torch.from_numpy(xh).to(dev)

    def enc_check(cfg=cfg, blob=blob, m=m, x=x, xh=xh, T=T):   # (bound now: the names are reused below and the check runs later)
        from oracle import c_oracle
        ref = c_oracle.RefEncodec(cfg, blob)
        fr = m.encode(x)
        au = m.decode(fr, T)
        rfr = ref.encode(xh[:1])
        rau = ref.decode(rfr)
        ok = all(np.array_equal(f.codes[:1].cpu().numpy(), r[0]) for f, r in zip(fr, rfr))
        return {"clips": 1, "codes_bit_exact": bool(ok), "pcm_max_abs_diff": float(np.abs(au[:1].cpu().numpy() - rau).max())}

    run("Encodec 48kHz stereo 12kbps encode+decode, batch=16 x 2 s (BASELINE configs[2])", m, lambda: m.decode(m.encode(x), T), B, secs,
        "encodec48k", enc_check)
    models = [m]

    # C5 per-GPU share (8 x 5 s) and C1 (1 x 1 s)
    for key, cfg, B, secs, label in (("snac44k", SNACConfig.snac_44khz(), 8, 5.0, "SNAC 44.1kHz + LocalMHA encode+decode, batch=8 x 5 s = one GPU's share of BASELINE configs[4]"),
                                     ("snac24k", SNACConfig.snac_24khz(), 1, 1.0, "SNAC 24kHz mono encode+decode, 1 x 1 s (BASELINE configs[0])")):
        blob = save_blob(snac_synthetic_state_dict(cfg, seed=42))
        m = SNAC(cfg)
        m.load_blob(blob)
        T = int(secs * cfg.sampling_rate)
        xh = synthetic_pcm(B, 1, T, cfg.sampling_rate, seed=1234)
        x = torch.from_numpy(xh).to(dev)
        frames = m.query(T)[1]
        nzh = snac_noise(cfg, B, frames, seed=3)
        nz = m.flat_noise(nzh, dev)   # (the ABI's layout: views of one device buffer -- input preparation, outside the timed region)

        def snac_check(cfg=cfg, blob=blob, m=m, x=x, xh=xh, nz=nz, nzh=nzh):
            from oracle import c_oracle
            ref = c_oracle.RefSNAC(cfg, blob)
            codes = m.encode(x)
            au = m.decode(codes, nz)
            _, _, rcodes = ref.encode(xh[:1])
            rau = ref.decode(rcodes, [n[:1] for n in nzh])
            ok = all(np.array_equal(c[:1].cpu().numpy(), r) for c, r in zip(codes, rcodes))
            return {"clips": 1, "codes_bit_exact": bool(ok), "pcm_max_abs_diff": float(np.abs(au[:1].cpu().numpy() - rau).max())}

        run(label, m, lambda m=m, x=x, nz=nz: m.decode(m.encode(x), nz), B, secs, key, snac_check)
        models.append(m)
    for e, chk in deferred:
        e["gpu_equals_oracle"] = chk()
    for mm in models:
        mm.dispose()
    return out


def aten_proxy(cfg, state_dict, pcm_h, seconds, iters=3):
    """The DAC graph as a sequence of ATen CPU operators -- what TorchSharp-CPU dispatches to (tools/aten_proxy.py: repo-owned, shares
    no code with oracle/): 1 warm-up + `iters` timed passes, median.  A reported baseline only."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import aten_proxy as ap
    return ap.run(cfg, state_dict, pcm_h, seconds, iters)


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def visible_gpus():
    """Devices this host exposes, counted in a FRESH child process: the launcher itself must never touch the GPU (a process that has
    initialised it may not start replacements of itself on this pool, and a parent that holds no GPU state cannot leak any)."""
    import subprocess
    try:
        r = subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"], capture_output=True, text=True, timeout=600)
        return int(r.stdout.strip().splitlines()[-1]) if r.returncode == 0 and r.stdout.strip() else 0
    except Exception:
        return 0


def self_launch(n, argv, timeout_s):
    """`python bench.py --gpus N` with no WORLD_SIZE in the environment (how a single host process -- the reference's callers batch in
    one process, Examples/Program.cs:228-322 -- or the driver's plain command starts it): spawn N fresh rank processes of this file
    (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set, a free rendezvous port), relay rank 0's single JSON line, exit
    non-zero as soon as any rank does (the others are stopped by PID).  The parent never imports torch or touches a GPU."""
    import socket
    import subprocess
    import threading
    have = visible_gpus()
    if have < n:
        sys.stderr.write(f"bench.py --gpus {n}: this host exposes {have} GPU(s); a multi-GPU run needs one device per rank "
                         f"(nothing was started)\n")
        return 2
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), NC_BENCH_FORCE_DIST="1", NC_BENCH_CHILD="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env.pop("NC_BENCH_SELF_LAUNCH", None)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env, text=True,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, stderr=None))
    lines = []

    def pump():
        for ln in procs[0].stdout:
            lines.append(ln.rstrip("\n"))
    th = threading.Thread(target=pump, daemon=True)
    th.start()
    t0, rc = time.time(), 0
    while True:
        codes = [p.poll() for p in procs]
        bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
        if bad:
            rc = bad[0][1] if bad[0][1] > 0 else 1
            sys.stderr.write(f"bench.py: rank {bad[0][0]} exited with code {bad[0][1]}; stopping the other ranks\n")
            break
        if all(c == 0 for c in codes):
            break
        if time.time() - t0 > timeout_s:
            sys.stderr.write(f"bench.py: ranks still running after {timeout_s} s; stopping them\n")
            rc = 124
            break
        time.sleep(0.1)
    if rc:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=10)
            except subprocess.TimeoutExpired:
                p.kill()
    th.join(timeout=10)
    js = [ln for ln in lines if ln.startswith("{")]
    for ln in lines:
        if not ln.startswith("{"):
            sys.stderr.write(ln + "\n")          # RCCL's banner etc.: not part of the one-line contract
    if rc == 0 and len(js) != 1:
        sys.stderr.write(f"bench.py: rank 0 printed {len(js)} JSON lines, expected one\n")
        rc = 1
    if rc == 0:
        print(js[0], flush=True)
    return rc


def median(v):
    v = sorted(v)
    return v[len(v) // 2] if len(v) % 2 else 0.5 * (v[len(v) // 2 - 1] + v[len(v) // 2])


def run_local_group(args):
    """`--local-group`: ONE process drives --gpus devices through nc_group_create_local (ncclCommInitAll + grouped in-place all-gathers on
    per-device side streams): the layout of a single C# host (Examples/Program.cs:228-322 batches in one process).  Every device holds
    its block of clips in HBM; a step = nc_group_*_encode_allgather_local_dev (all devices, asynchronous) + the local decodes + nc_group_wait.
    Timing: device synchronise of every device on both sides of EXACTLY --steps steps (one process: no barrier to take)."""
    import torch
    from neuralcodecs_amd import DAC, SNAC, DACConfig, parallel
    from neuralcodecs_amd.config import SNACConfig
    from neuralcodecs_amd.weights import dac_synthetic_state_dict, save_blob, snac_noise, snac_synthetic_state_dict, synthetic_pcm
    n = args.gpus
    have = torch.cuda.device_count()
    if have < n:
        raise SystemExit(f"bench.py --gpus {n} --local-group: this host exposes {have} GPU(s)")
    snac_mode = args.config == "snac44k"
    B = args.batch or (8 if snac_mode else 32)
    seconds = args.seconds or (5.0 if snac_mode else 1.0)
    cfg = SNACConfig.snac_44khz() if snac_mode else DACConfig.dac_44khz()
    sr = cfg.sampling_rate if snac_mode else cfg.sample_rate
    blob = save_blob(snac_synthetic_state_dict(cfg, seed=42) if snac_mode else dac_synthetic_state_dict(cfg, seed=42))
    T = int(round(seconds * sr))
    models, blocks, noises = [], [], []
    for d in range(n):
        m = (SNAC if snac_mode else DAC)(cfg, device_index=d)
        m.load_blob(blob)
        models.append(m)
        dev = torch.device("cuda", d)
        blocks.append(torch.from_numpy(synthetic_pcm(B, 1, T, sr, seed=1234 + d * B)).to(dev))
        if snac_mode:
            noises.append(m.flat_noise(snac_noise(cfg, B, m.query(T)[1], seed=3 + d), dev))
    g = parallel.Group.local(models)
    if args.pack_bits:
        g.set_code_bits(args.pack_bits)
    state = {}

    def step():
        if snac_mode:
            call, widths = g.snac_encode_allgather_local(blocks, codes_all=state.get("codes_all"))
            state["codes_all"] = call
            audio = []
            for d, m in enumerate(models):
                local = parallel.split_levels(call[d][d * B:(d + 1) * B], widths)    # (views of the device's own slot)
                with torch.cuda.device(d):
                    audio.append(m.decode([c.contiguous() for c in local], noises[d]))
        else:
            z, call, _ = g.dac_encode_allgather_local(blocks, codes_all=state.get("codes_all"))
            state["codes_all"] = call
            audio = []
            for d, m in enumerate(models):
                with torch.cuda.device(d):
                    audio.append(m.decode(z[d]))
        g.wait()
        return call, audio

    def sync():
        for d in range(n):
            torch.cuda.synchronize(d)

    for _ in range(args.warmup):
        step()
    sync()
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    with torch.cuda.device(0):
        marks[0].record()
    for i in range(args.steps):
        call, audio = step()
        with torch.cuda.device(0):
            marks[i + 1].record()
    sync()
    dt = time.perf_counter() - t0
    step_ms = [marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps)]
    for m in models:
        m.check_errors()
    ok = None
    if not args.no_check:
        # every device's copy of the gathered tensor equals every other's, and slot d equals a plain encode of block d on device d
        ok = all(torch.equal(call[0].cpu(), call[d].cpu()) for d in range(1, n))
        for d, m in enumerate(models):
            with torch.cuda.device(d):
                want = torch.cat([c.reshape(B, -1) for c in m.encode(blocks[d])], dim=1) if snac_mode else m.encode(blocks[d])[1]
                torch.cuda.synchronize(d)
            ok = ok and bool(torch.equal(call[d][d * B:(d + 1) * B].reshape(B, -1), want.reshape(B, -1)))
    models[0].profile_enable(True)
    models[0].profile_reset()
    for _ in range(args.steps):
        step()
    sync()
    prof = models[0].profile_read()
    models[0].profile_enable(False)
    classes = class_table(prof, args.steps, None)
    g.dispose()
    for m in models:
        m.dispose()
    ms = dt / args.steps * 1e3
    name = "SNAC-44.1kHz B=8 x 5 s per GPU" if snac_mode else "DAC-44.1kHz B=32"
    out = {"metric": "audio-seconds/sec encode+decode (x real-time), " + name, "value": round(n * B * seconds * args.steps / dt, 2),
           "unit": "audio-seconds/sec", "n_gpus": n, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 3),
           "ms_per_step_median": round(median(step_ms), 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
           "data": "synthetic",
           "config": {"workload": ("SNAC 44.1kHz + LocalMHA" if snac_mode else "DAC 44.1kHz 8kbps") + " encode+decode, batch=%d x %.0f s clips per GPU" % (B, seconds),
                      "clips_per_gpu": B, "clip_seconds": seconds, "global_batch": n * B, "launch": "one host process, nc_group_create_local",
                      "collective": "grouped RCCL all_gather of the codes, in place, one side stream per device" + (", %d-bit packed" % args.pack_bits if args.pack_bits else ""),
                      "gathered_equals_1gpu_every_slot_every_device": ok},
           "roofline": dict(dominant(classes) or {}, all_classes=classes, note="device 0's launches"), "cpu_baseline": None}
    assert ok is not False, "gathered codes differ between devices or from the plain encode"
    print(json.dumps(out), flush=True)
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="dac44k", choices=("dac44k", "snac44k"), help="dac44k = BASELINE C2/C4 (headline); snac44k = C5 share per GPU")
    ap.add_argument("--batch", type=int, default=0, help="clips per GPU (default: 32 for dac44k, 8 for snac44k)")
    ap.add_argument("--seconds", type=float, default=0.0, help="clip length (default: 1 s for dac44k, 5 s for snac44k)")
    ap.add_argument("--pack-bits", type=int, default=0, help="N > 1: all-gather the codes bit-packed (10 for DAC's 1024-entry codebooks, 12 for SNAC)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-clips", type=int, default=0, help="clips in the CPU-baseline sample (default: the whole batch of one step, BASELINE.md 3.1)")
    ap.add_argument("--cpu-iters", type=int, default=3, help="timed passes of the C-oracle baseline over the sample (BASELINE.md 3.1: >= 3, median reported; one pass of the C2 batch is ~35 s)")
    ap.add_argument("--proxy-iters", type=int, default=3, help="timed passes of the ATen operator-sequence proxy (median)")
    ap.add_argument("--proxy-clips", type=int, default=8, help="clips of the step the ATen proxy runs per pass")
    ap.add_argument("--no-extra", action="store_true", help="skip the extra_configs block (C3 / C5 share / C1)")
    ap.add_argument("--no-check", action="store_true", help="skip the GPU == oracle comparisons (outside the timed region)")
    ap.add_argument("--check", action="store_true", help="(default now) kept for compatibility")
    ap.add_argument("--local-group", action="store_true", help="ONE process drives --gpus devices through nc_group_create_local (single-host layout)")
    ap.add_argument("--launch-timeout", type=int, default=3300, help="self-launched ranks are stopped after this many seconds")
    args = ap.parse_args()

    if args.local_group:
        sys.exit(run_local_group(args))
    # `--gpus N` with no process group in the environment: be the launcher (fresh rank processes; this process never touches a GPU)
    if int(os.environ.get("WORLD_SIZE", "1")) == 1 and os.environ.get("NC_BENCH_CHILD") != "1" and \
            (args.gpus > 1 or os.environ.get("NC_BENCH_SELF_LAUNCH") == "1"):
        sys.exit(self_launch(args.gpus, sys.argv[1:], args.launch_timeout))

    import torch
    import torch.distributed as dist
    from neuralcodecs_amd import DAC, SNAC, DACConfig
    from neuralcodecs_amd.config import SNACConfig
    from neuralcodecs_amd.weights import dac_synthetic_state_dict, save_blob, snac_noise, snac_synthetic_state_dict, synthetic_pcm

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.stderr.write(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: running the {world}-rank job the launcher started\n")
    if torch.cuda.device_count() <= local_rank:
        raise SystemExit(f"bench.py rank {rank}: no GPU with index {local_rank} on this host ({torch.cuda.device_count()} visible)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # NC_BENCH_FORCE_DIST=1 exercises the RCCL path (process group, side-stream all-gather, barrier) with a single rank
    use_dist = world > 1 or os.environ.get("NC_BENCH_FORCE_DIST") == "1"
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from neuralcodecs_amd import parallel
    snac_mode = args.config == "snac44k"
    B = args.batch or (8 if snac_mode else 32)
    seconds = args.seconds or (5.0 if snac_mode else 1.0)
    side = torch.cuda.Stream(device=dev) if use_dist else None

    if snac_mode:
        cfg = SNACConfig.snac_44khz()
        sd = snac_synthetic_state_dict(cfg, seed=42)
        blob = save_blob(sd)
        model = SNAC(cfg, device_index=local_rank)
        model.load_blob(blob)
        T = int(round(seconds * cfg.sampling_rate))
        pcm_h = synthetic_pcm(B, 1, T, cfg.sampling_rate, seed=1234 + rank * B)
        pcm = torch.from_numpy(pcm_h).to(dev)
        Tz = model.query(T)[1]
        widths = model.query(T)[2]
        noise_h = snac_noise(cfg, B, Tz, seed=3 + rank)
        noise = model.flat_noise(noise_h, dev)   # (the ABI's layout: views of one device buffer, prepared before the timed region)
        gathered = torch.empty((world * B, sum(widths)), dtype=torch.int64, device=dev) if use_dist else None

        def step():
            codes = model.encode(pcm)
            if use_dist:
                # the four code levels of a clip travel as ONE collective (levels side by side, as the C ABI emits them)
                ev = torch.cuda.Event()
                ev.record()
                with torch.cuda.stream(side):
                    side.wait_event(ev)
                    parallel.all_gather_levels(codes, world * B, out=gathered, bits=args.pack_bits or None)
            audio = model.decode(codes, noise)
            if use_dist:
                torch.cuda.current_stream().wait_stream(side)
            return codes, None, audio
    else:
        cfg = DACConfig.dac_44khz()
        sd = dac_synthetic_state_dict(cfg, seed=42)
        blob = save_blob(sd)
        model = DAC(cfg, device_index=local_rank)
        model.load_blob(blob)
        T = int(round(seconds * cfg.sample_rate))
        # synthetic clips (seed 1234 + global clip index), resident in HBM before the timed region
        pcm_h = synthetic_pcm(B, 1, T, cfg.sample_rate, seed=1234 + rank * B)
        pcm = torch.from_numpy(pcm_h).to(dev)
        Tz = model.frames(T)
        gathered = torch.empty((world * B, cfg.n_codebooks, Tz), dtype=torch.int64, device=dev) if use_dist else None

        def step():
            z, codes, lat, _, _ = model.encode(pcm)
            if use_dist:
                # all-gather the emitted codes on a side stream; the local decode only needs local z
                ev = torch.cuda.Event()
                ev.record()
                with torch.cuda.stream(side):
                    side.wait_event(ev)
                    parallel.all_gather_codes(codes, world * B, out=gathered, bits=args.pack_bits or None)
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
    # ---- the timed region: EXACTLY --steps steps, profiler off, barrier + device synchronise on both sides.  A HIP event is recorded on
    # the launch stream between steps (asynchronous, no host wait) so that the per-step times -- and their median, SURVEY 8d -- are known.
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(args.steps):
        codes, z, audio = step()
        marks[i + 1].record()
    sync()
    dt = time.perf_counter() - t0
    step_ms = [marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps)]
    if use_dist:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    # ---- second pass (untimed): the per-class table, HIP events around every launch on the launch stream (roofline.achieved)
    model.profile_enable(True)
    model.profile_reset()
    for _ in range(args.steps):
        step()
    sync()
    prof = model.profile_read()
    model.profile_enable(False)

    # ---- encode-only / decode-only (Examples/Program.cs:252-291 calls them separately), same inputs, rank-local
    def half(fn):
        fn(); torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
        ev[0].record()
        for i in range(args.steps):
            fn()
            ev[i + 1].record()
        torch.cuda.synchronize()
        ms = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(args.steps))
        med = ms[len(ms) // 2] if len(ms) % 2 else 0.5 * (ms[len(ms) // 2 - 1] + ms[len(ms) // 2])
        return {"ms_median": round(med, 3), "ms_mean": round(sum(ms) / len(ms), 3), "x_realtime_median": round(B * seconds / (med * 1e-3), 1)}
    if snac_mode:
        enc_only = half(lambda: model.encode(pcm))
        dec_only = half(lambda: model.decode(codes, noise))
    else:
        enc_only = half(lambda: model.encode(pcm))
        dec_only = half(lambda: model.decode(z))

    audio_seconds = world * B * seconds * args.steps
    value = audio_seconds / dt
    ms_per_step = dt / args.steps * 1e3
    sms = sorted(step_ms)
    med_ms = sms[len(sms) // 2] if len(sms) % 2 else 0.5 * (sms[len(sms) // 2 - 1] + sms[len(sms) // 2])   # this rank's per-step GPU times

    if rank == 0:
        classes = class_table(prof, args.steps, load_traffic("snac44k" if snac_mode else "dac44k") if (B == (8 if snac_mode else 32)) else None)
        total_kernel_ms = sum(v["ms"] for v in prof.values())
        total_flops = sum(v["flops"] for v in prof.values())
        if snac_mode:
            dom = dominant(classes)
            roofline = dict(dom, kernel=dom["class"], traffic=None, all_classes=classes,
                            whole_step_tflops=round(total_flops / args.steps / (ms_per_step * 1e-3) / 1e12, 3))
            tr = load_traffic("snac44k")
            if tr:
                roofline["traffic"] = tr.get(dom["class"], {}).get("hbm_bytes_per_launch")
        else:
            k7 = prof["conv_k7"]
            ach_tflops = (k7["flops"] / (k7["ms"] * 1e-3)) / 1e12 if k7["ms"] > 0 else 0.0
            tr = load_traffic("dac44k") if (B == 32 and abs(seconds - 1.0) < 1e-9) else None
            tk7 = (tr or {}).get("conv_k7", {})
            algo_b = k7["bytes"] / max(k7["launches"], 1)
            roofline = {
                "kernel": "conv_mfma_kernel<K=7> (dilated k=7 residual-unit conv, fp32 MFMA implicit GEMM; fused units incl. their 1x1)",
                "bound": "mfma", "achieved": round(ach_tflops, 3), "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": round(ach_tflops / FP32_MFMA_PEAK_TFLOPS, 4),
                "traffic": round(tk7["hbm_bytes_per_launch"]) if tk7 else None,
                "traffic_stale": traffic_stale("dac44k"),
                "traffic_unit": "HBM bytes per launch of this class (PMC FETCH_SIZE x2 + WRITE_SIZE over exactly the launches the class counts)",
                "traffic_source": "profiles/traffic.json (rocprofv3 --pmc passes of this command, tools/profile_round.sh); reported only when the loaded library's SHA-256 equals the one recorded there",
                "traffic_over_algorithmic": round(tk7["hbm_bytes_per_launch"] / algo_b, 3) if tk7 and algo_b else None,
                "mfma_busy": tk7.get("mfma_busy"),
                "valu_busy": tk7.get("valu_busy"),       # f32 matrix-core and vector instructions share one pipe per SIMD on gfx950:
                "pipe_busy": tk7.get("pipe_busy"),       # mfma_busy + valu_busy = the fraction of that shared pipe the class occupies
                "algorithmic_bytes_per_launch": round(algo_b),
                "launches_per_step": k7["launches"] / max(args.steps, 1),
                "avg_launch_ms": round(k7["ms"] / max(k7["launches"], 1), 5),
                "flops_per_launch": k7["flops"] / max(k7["launches"], 1),
                "share_of_kernel_time": round(k7["ms"] / total_kernel_ms, 4) if total_kernel_ms else None,
                "all_classes": classes,
                "whole_step_tflops": round(total_flops / args.steps / (ms_per_step * 1e-3) / 1e12, 3),
            }
        extras = None   # (before the CPU legs: see extra_configs)
        if world == 1 and not use_dist and not args.no_extra and not snac_mode:
            try:
                extras = extra_configs(dev, 5, 2, not args.no_check)
            except Exception as e:
                extras = {"error": repr(e)}
        cpu = None
        if not args.no_cpu_baseline and world == 1:
            from oracle import c_oracle
            # the C restatement of the reference graph on this box's host cores, in this run: one warm-up pass, then
            # --cpu-iters timed passes over a bounded sample of the step's clips; median
            import statistics
            n = max(1, min((args.cpu_clips or B) if not snac_mode else 1, B))
            nw = min(2, n)                                       # warm-up: pages the weights in and spins the OpenMP team up
            if snac_mode:
                ref = c_oracle.RefSNAC(cfg, blob)

                def cpu_pass(n=n):
                    _, _, rc = ref.encode(pcm_h[:n])
                    return rc, ref.decode(rc, [x[:n] for x in noise_h])
            else:
                ref = c_oracle.RefDAC(cfg, blob)

                def cpu_pass(n=n):
                    rz, rc, _, _ = ref.encode(pcm_h[:n])
                    return rc, ref.decode(rz)
            cpu_pass(nw)
            ctimes = []
            for _ in range(max(1, args.cpu_iters)):
                tc = time.perf_counter()
                rcodes, raudio = cpu_pass()
                ctimes.append(time.perf_counter() - tc)
            cdt = statistics.median(ctimes)
            cpu = {"value": round(n * seconds / cdt, 4), "unit": "audio-seconds/sec", "cores": int(c_oracle.lib().ref_num_threads()),
                   "kind": "port", "cpu": cpu_model(), "host_cpus": os.cpu_count(), "iterations": len(ctimes), "median_s": round(cdt, 3),
                   "min_s": round(min(ctimes), 3), "max_s": round(max(ctimes), 3),
                   "sample": f"{n} of the {B} clips of one step (encode+decode, C oracle, OpenMP over clips x output channels): "
                             f"warm-up on {nw} clips + {len(ctimes)} timed pass(es) over the sample, median"}
            if not args.no_check:
                if snac_mode:
                    same_codes = all(np.array_equal(c[:n].cpu().numpy(), r) for c, r in zip(codes, rcodes))
                else:
                    same_codes = bool(np.array_equal(codes[:n].cpu().numpy(), rcodes))
                cpu["gpu_equals_oracle"] = {"clips": n, "codes_bit_exact": bool(same_codes),
                                            "pcm_max_abs_diff": float(np.abs(audio[:n].cpu().numpy() - raudio).max())}
            if not snac_mode:
                try:
                    cpu["aten_proxy"] = aten_proxy(cfg, sd, pcm_h[:max(1, min(args.proxy_clips, B))], seconds, args.proxy_iters)
                except Exception as e:   # the proxy is informational: never lose the bench line to it
                    cpu["aten_proxy"] = {"error": repr(e)}
        if snac_mode:
            metric = "audio-seconds/sec encode+decode (x real-time), SNAC-44.1kHz B=8 x 5 s per GPU"
            workload = "SNAC 44.1kHz + LocalMHA encode+decode, batch=%d x %.0f s clips per GPU (BASELINE configs[4] share)" % (B, seconds)
            coll = "RCCL all_gather of int64 codes [B,%d] (4 levels side by side) per rank" % sum(widths)
            if args.pack_bits:
                coll += ", %d-bit packed payload" % args.pack_bits
        else:
            metric = "audio-seconds/sec encode+decode (x real-time), DAC-44.1kHz B=32"
            workload = "DAC 44.1kHz 8kbps encode+decode, batch=%d x %.0f s clips per GPU (BASELINE configs[1])" % (B, seconds)
            coll = "RCCL all_gather of int64 codes [B,9,87] per rank" + (", %d-bit packed payload" % args.pack_bits if args.pack_bits else "")
        out = {
            "metric": metric,
            "value": round(value, 2), "unit": "audio-seconds/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "ms_per_step_median": round(med_ms, 3), "value_at_median_step": round(world * B * seconds / (med_ms * 1e-3), 2),
            "ms_per_step_min_max": [round(min(step_ms), 3), round(max(step_ms), 3)],
            "encode_only": enc_only, "decode_only": dec_only,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": workload, "clips_per_gpu": B, "clip_seconds": seconds, "global_batch": world * B,
                       "collective": coll if use_dist else "none"},
            "roofline": roofline, "cpu_baseline": cpu, "extra_configs": extras,
        }
        # The ONE line must be readable in a log tail: the full per-class tables go to a detail file, the line keeps a
        # compact form of them; the other BASELINE configs' step times are flat top-level keys as well.
        detail = json.loads(json.dumps(out))
        out["roofline"]["all_classes"] = compact_classes(roofline["all_classes"])
        out["roofline"]["all_classes_columns"] = ["ms_per_step", "launches_per_step", "TFLOP/s (matrix-core classes) or algorithmic GB/s", "frac_of_peak"]
        for k in ("traffic_unit", "traffic_source"):
            out["roofline"].pop(k, None)
        if isinstance(extras, dict) and "error" not in extras:
            slim = {}
            for k, e in extras.items():
                slim[k] = {kk: e[kk] for kk in ("B", "clip_seconds", "ms_per_step", "x_realtime", "whole_step_tflops", "kernel_ms_per_step",
                                                "traffic_stale", "gpu_equals_oracle") if kk in e}
                dm = e.get("dominant") or {}
                slim[k]["dominant"] = {kk: dm[kk] for kk in ("class", "bound", "achieved", "unit", "frac", "ms_per_step", "mfma_busy", "valu_busy", "pipe_busy") if kk in dm}
                slim[k]["launches_per_step"] = round(sum(c["launches_per_step"] for c in e["classes"].values()), 1)   # (class tables: the detail file)
            out["extra_configs"] = slim
            for k, flat in (("encodec48k", "c3_encodec48k_16x2s_ms_per_step"), ("snac44k_c5_share", "c5_share_snac44k_8x5s_ms_per_step"),
                            ("snac24k", "c1_snac24k_1x1s_ms_per_step")):
                if k in extras:
                    out[flat] = extras[k]["ms_per_step"]
        if cpu and isinstance(cpu.get("aten_proxy"), dict):
            out["cpu_baseline"] = dict(cpu, aten_proxy={k: v for k, v in cpu["aten_proxy"].items() if k in ("value", "unit", "threads", "iterations", "median_s", "error")})
        try:
            dpath = os.environ.get("NC_BENCH_DETAIL") or os.path.join(ROOT, "gpurun_out", "bench_detail.json")
            os.makedirs(os.path.dirname(dpath), exist_ok=True)
            json.dump(detail, open(dpath, "w"), indent=1)
            out["detail_file"] = os.path.relpath(dpath, ROOT)
        except OSError:
            pass
    gathered_ok = None
    if use_dist and not args.no_check:
        # the gathered codes of the N-GPU run must equal the 1-GPU run on the same inputs -- EVERY slot, on every rank:
        # each rank re-encodes every shard's clips (same seeds) on its own GPU, outside the timed region, and compares slot by slot
        ok = True
        for sh in range(world):
            xs = torch.from_numpy(synthetic_pcm(B, 1, T, cfg.sampling_rate if snac_mode else cfg.sample_rate, seed=1234 + sh * B)).to(dev)
            if snac_mode:
                want = torch.cat([c.reshape(B, -1) for c in model.encode(xs)], dim=1)
            else:
                want = model.encode(xs)[1]
            torch.cuda.synchronize()
            ok = ok and bool(torch.equal(gathered[sh * B:(sh + 1) * B], want))
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        gathered_ok = bool(flag.item() == 1)
        if rank == 0:
            out["config"]["gathered_equals_1gpu_every_slot_every_rank"] = gathered_ok
        assert gathered_ok, "gathered codes differ from the 1-GPU encode of the same clips"
    model.dispose()
    if use_dist:
        dist.destroy_process_group()
    if rank == 0:
        # the ONE JSON line goes out last: RCCL writes a version banner to the C stdio buffer of stdout, push that out first
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        sys.stdout.flush()
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()

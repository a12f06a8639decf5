"""GPU suite, child-process tests: (a) the persistent-LSTM timeout path (a faked timeout: the host-pointer entry points repeat the call
on the step-wise kernels, the device-pointer path reports it through nc_codec_check_errors and succeeds on the repeat); (b) nc_group
with MORE THAN ONE process / device -- rank mode with world = 2 (ncclCommInitRank, in-place all-gather, every slot checked against a
1-GPU encode) and local mode with 2 devices (ncclCommInitAll + grouped all-gathers, ragged batches).  (b) skips on boxes with fewer
than two GPUs (the build container's GPU box has one); the world = 1 / one-device forms of the same children run everywhere, so the
harness itself is exercised on every box."""
import os
import subprocess
import sys
import tempfile

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = os.path.join(ROOT, "tests", "_gpu_child.py")


def _n_gpus():
    import torch
    return torch.cuda.device_count()   # (counting devices does not initialise the GPU in this process)


def _run(args, env=None, timeout=600):
    e = dict(os.environ)
    e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if env:
        e.update(env)
    return subprocess.Popen([sys.executable, CHILD] + [str(a) for a in args], env=e, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)


def _finish(procs, timeout=600):
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise AssertionError("child timed out")
        outs.append(out)
    for p, out in zip(procs, outs):
        assert p.returncode == 0 and "CHILD_OK" in out, out[-3000:]


@pytest.mark.parametrize("mode", ["lstm_fallback", "lstm_fallback_dev"])
def test_lstm_timeout_falls_back_to_stepwise(mode):
    _finish([_run([mode], env={"NC_LSTM_FAKE_TIMEOUT": "1"})])


def test_two_threads_two_handles_one_device():
    """SURVEY 8b threading / include/nc_mi355x.h: distinct handles may run concurrently.  DAC || DAC, DAC || Encodec, Encodec || Encodec (C3
    shape), 20 iterations per thread, every result bit-equal to the serial run, no step-wise LSTM fallback (tests/_gpu_child.py threads)."""
    _finish([_run(["threads", 20])], timeout=1500)


def test_snac_fused_unit_guard_is_on_channels_times_steps():
    """SnacFusedUnit::usable bounds C * T (32-bit lane offsets), not T.  Three children at T = 36 864, B = 2 (73 728 columns: above
    the fused units' minimum): by default both the C = 64 encoder units (2.36 M elements per clip) and the C = 96 decoder units (3.54 M) are
    fused; with the bound lowered to 3 000 000 elements only the C = 64 units are (T alone is far below either bound); NC_SNAC_NO_FUSE fuses
    none.  The depthwise launches of the profile tell which path ran; all three stay bit-exact against the oracle."""
    import re
    counts = {}
    for name, env in (("default", {}), ("3e6", {"NC_SNAC_FUSE_MAX_ELEMS": "3000000"}), ("none", {"NC_SNAC_NO_FUSE": "1"})):
        p = _run(["snac_fuse_guard"], env=env)
        out, _ = p.communicate(timeout=900)
        assert p.returncode == 0 and "CHILD_OK" in out, out[-3000:]
        counts[name] = int(re.search(r"DWCONV_LAUNCHES (\d+)", out).group(1))
    assert counts["default"] < counts["3e6"] < counts["none"], counts        # fewer depthwise launches = more units fused


def test_group_children_single_device():
    with tempfile.TemporaryDirectory() as d:
        _finish([_run(["group_rank", 1, 0, os.path.join(d, "uid"), 4])])
    _finish([_run(["group_local", 1, 3])])


@pytest.mark.skipif(_n_gpus() < 2, reason="needs two GPUs (nc_group with world > 1)")
def test_group_rank_mode_world2():
    with tempfile.TemporaryDirectory() as d:
        uid = os.path.join(d, "uid")
        _finish([_run(["group_rank", 2, r, uid, 6]) for r in range(2)])


@pytest.mark.skipif(_n_gpus() < 2, reason="needs two GPUs (nc_group local mode with ndev > 1)")
def test_group_local_mode_two_devices_ragged():
    _finish([_run(["group_local", 2, 5])])


@pytest.mark.skipif(_n_gpus() < 4, reason="needs four GPUs")
def test_group_local_mode_more_devices_than_clips():
    _finish([_run(["group_local", 4, 3])])


# A four-row subset of tools/probe/envmatrix.sh: every fast path has a switch back to the path it replaced, and the parity tests must
# hold on those paths too (fresh interpreters: the switches are read once per process).
FALLBACK_ROWS = [
    {"NC_NO_GN_FUSE": "1", "NC_NO_IN2": "1", "NC_NO_CONV3S": "1"},                     # stand-alone GroupNorm sums, summed copies, windowed k=3
    {"NC_LSTM_STEPWISE": "1", "NC_NO_TINY_TILES": "1", "NC_NO_SUBPIXEL": "1"},         # step-wise LSTM, filled-grid tile rule, per-phase up-convs
    {"NC_NO_FUSE": "1", "NC_ENCODEC_NO_FUSE": "1", "NC_DAC_RVQ_STAGEWISE": "1"},       # two-launch residual units, padded copies, stage-wise RVQ
    {"NC_NO_FLAT_GN": "1", "NC_LSTM_NO_ELU": "1", "NC_NO_DIST_SMALL": "1", "NC_LSTM_UB": "2", "NC_NO_SUBPIXEL_ANY": "1"},   # one-clip GroupNorm tiles, ELU in the consumer, segmented staging, 8-wave LSTM
    {"NC_SNAC_NO_FUSE": "1", "NC_ATTN_NO_MFMA": "1", "NC_LN_TILE": "0"},                # SNAC units in two launches, vector attention, per-column LayerNorm
    {"NC_LSTM_FUSED": "1", "NC_RVQ_8WAVES": "1"},                                      # EXPERIMENTS=1 library: fused two-layer LSTM, 8-wave Euclidean RVQ
    {"NC_SYNC_ACQUIRE": "1"},                                                          # acquire fences in the LSTM exchange and the in-launch GroupNorm finish
    {"NC_LSTM_SPLIT": "1"},                                                            # EXPERIMENTS=1 library: role-split per-layer persistent LSTM (nc_lstm.hip lstm1_kernel)
    {"NC_SNAC_FUSE_MIN_COLS": "0", "NC_LN_TILE": "16"},                                # one-launch SNAC residual units on the small fixtures too
    {"NC_NO_XR": "1"},                                                                 # generic fragment addressing in the conv template
    {"NC_PW_STREAM": "1"},                                                             # EXPERIMENTS=1 library: streaming pointwise kernel
    {"NC_NO_XV": "1"},                                                                 # legacy instances for the two-tap up-convolutions (item-form staging)
    {"NC_NO_XV_K7": "1"},                                                              # legacy instances for the k = 7 convolutions on the long rows
    {"NC_DUO": "1"},                                                                   # EXPERIMENTS=1 library: two tiles per 8-wavefront workgroup (k = 7)
    {"NC_NO_RES_A": "1", "NC_NO_DOWN2": "1", "NC_NO_DOWN4": "1", "NC_NO_DOWN5": "1", "NC_NO_UP2": "1", "NC_NO_UP4": "1", "NC_NO_UP_PITCH": "1", "NC_RMS_TWO_PASS": "1", "NC_LSTM_NO_HTILE": "1"},                                                              # round-5 launches: two-launch first pass of the residual blocks, windowed stride-2 down-conv, two-pass RMS scale
]


# Switches of measured-and-rejected kernels: compiled by `make -C neuralcodecs_amd/csrc EXPERIMENTS=1` only (libnc_mi355x_exp.so, round 5:
# the shipped library carries shipped paths only).  Rows that name one run against that library when it has been built, and skip otherwise.
EXPERIMENT_SWITCHES = {"NC_LSTM_FUSED", "NC_LSTM_SPLIT", "NC_PW_STREAM", "NC_RVQ_8WAVES", "NC_LIGHT", "NC_WIDE", "NC_SPEC", "NC_DIST", "NC_DUO"}
EXP_LIB = os.path.join(ROOT, "neuralcodecs_amd", "libnc_mi355x_exp.so")


def _env_for(row):
    e = dict(os.environ, **row)
    if EXPERIMENT_SWITCHES & set(row):
        if not os.path.exists(EXP_LIB):
            pytest.skip("needs the EXPERIMENTS=1 library (make -C neuralcodecs_amd/csrc EXPERIMENTS=1)")
        e["NC_MI355X_LIB"] = EXP_LIB
    return e


@pytest.mark.parametrize("row", FALLBACK_ROWS, ids=lambda r: "+".join(sorted(r)))
def test_parity_under_fallback_switches(row):
    e = _env_for(row)
    cmd = [sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider",
           os.path.join(ROOT, "tests", "test_encodec_gpu.py") + "::test_encodec_small_vs_golden_and_oracle",
           os.path.join(ROOT, "tests", "test_encodec_gpu.py") + "::test_encodec48k_config_c3_shape",
           os.path.join(ROOT, "tests", "test_dac_gpu.py") + "::test_small_bit_exact_vs_c_oracle",
           os.path.join(ROOT, "tests", "test_snac_gpu.py")]
    r = subprocess.run(cmd, env=e, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-1500:])


def test_streaming_pointwise_variant_under_its_switch():
    """conv1x1_stream_kernel is no longer the default for the narrow long rows (round 4: the tile-per-workgroup kernel overtook it);
    NC_PW_STREAM=1 selects it, and it stays held to the oracle: its own bit-exactness cases and the SNAC suite (whose 44 kHz units it served)."""
    e = _env_for({"NC_PW_STREAM": "1"})
    cmd = [sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider",
           os.path.join(ROOT, "tests", "test_ops_gpu.py") + "::test_pointwise_streaming_variant_bit_exact",
           os.path.join(ROOT, "tests", "test_snac_gpu.py")]
    r = subprocess.run(cmd, env=e, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-1500:])


# The driver's first multi-GPU run launches bench.py under torch.distributed.run with no chance to debug it; the SAME code path
# (process group on RCCL, side-stream all-gather of the codes, barrier + max-over-ranks timing, every-slot verification, one JSON line)
# runs with a single rank under NC_BENCH_FORCE_DIST=1 -- in a fresh child created before this process touches the GPU.
@pytest.mark.parametrize("config,steps", [("dac44k", 2), ("snac44k", 1)])
def test_bench_distributed_path_one_rank(config, steps):
    import json
    import socket
    with socket.socket() as s:                                   # a free rendezvous port: several test processes may share a host
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    e = dict(os.environ, NC_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", LOCAL_RANK="0", WORLD_SIZE="1",
             HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", str(steps), "--warmup", "1", "--no-cpu-baseline", "--no-extra",
           "--config", config] + (["--pack-bits", "12"] if config == "snac44k" else [])   # (the SNAC run also moves its codes bit-packed)
    r = subprocess.run(cmd, env=e, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                      # exactly ONE JSON line, after RCCL's banner
    out = json.loads(lines[0])
    assert out["n_gpus"] == 1 and out["steps"] == steps and out["scaling"] == "weak" and out["value"] > 0
    assert out["config"]["collective"].startswith("RCCL all_gather")
    assert out["config"]["gathered_equals_1gpu_every_slot_every_rank"] is True
    assert out["encode_only"]["ms_median"] > 0 and out["decode_only"]["ms_median"] > 0 and out["ms_per_step_median"] > 0


# `python bench.py --gpus N` with NO process group in the environment: the bench is its own launcher -- fresh rank
# processes, a free rendezvous port, rank 0's one JSON line relayed, non-zero exit if any rank fails.  N = 1 of exactly that path runs on
# every box (NC_BENCH_SELF_LAUNCH=1: one child rank with the RCCL process group, side-stream all-gather and every-slot verification).
def _clean_env(**extra):
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "NC_BENCH_FORCE_DIST", "NC_BENCH_CHILD")}
    e["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    e.update(extra)
    return e


def _one_json_line(r):
    import json
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), r.stdout[-2000:]     # the launcher relays the JSON line and nothing else
    return json.loads(lines[0])


def test_bench_self_launch_one_rank():
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-extra"]
    r = subprocess.run(cmd, env=_clean_env(NC_BENCH_SELF_LAUNCH="1"), capture_output=True, text=True, timeout=900, cwd=ROOT)
    out = _one_json_line(r)
    assert out["n_gpus"] == 1 and out["steps"] == 2 and out["value"] > 0
    assert out["config"]["collective"].startswith("RCCL all_gather")
    assert out["config"]["gathered_equals_1gpu_every_slot_every_rank"] is True


def test_bench_self_launch_reports_a_failing_rank():
    """A rank that dies (here: an invalid clip length makes the engine refuse the encode) must end the launcher with a non-zero code."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-extra", "--seconds", "-1"]
    r = subprocess.run(cmd, env=_clean_env(NC_BENCH_SELF_LAUNCH="1"), capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode != 0 and "exited with code" in r.stderr, (r.returncode, r.stderr[-1500:])
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


@pytest.mark.parametrize("config,extra", [("dac44k", []), ("snac44k", ["--pack-bits", "12"])])
def test_bench_local_group_one_device(config, extra):
    """--local-group: ONE process drives the devices through nc_group_create_local + the device-resident grouped all-gather
    (nc_group_*_encode_allgather_local_dev), the layout of a single C# host (Examples/Program.cs:228-322); one device here."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--local-group", "--steps", "2", "--warmup", "1", "--config", config] + extra
    r = subprocess.run(cmd, env=_clean_env(), capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    import json
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["n_gpus"] == 1 and out["value"] > 0 and out["config"]["gathered_equals_1gpu_every_slot_every_device"] is True


@pytest.mark.skipif(_n_gpus() < 2, reason="needs two GPUs")
def test_bench_self_launch_and_local_group_two_gpus():
    base = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-extra"]
    out = _one_json_line(subprocess.run(base, env=_clean_env(), capture_output=True, text=True, timeout=1200, cwd=ROOT))
    assert out["n_gpus"] == 2 and out["config"]["gathered_equals_1gpu_every_slot_every_rank"] is True
    r = subprocess.run(base + ["--local-group"], env=_clean_env(), capture_output=True, text=True, timeout=1200, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]

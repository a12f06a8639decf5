"""Child processes of the GPU suite (fresh interpreters: created before any GPU call of their own, never via exec from a process that
touched the GPU).  python tests/_gpu_child.py <mode> [args...]; prints CHILD_OK on success, exits non-zero otherwise.

  lstm_fallback                       NC_LSTM_FAKE_TIMEOUT=1 in the environment: the first persistent LSTM launch is reported as timed out
  group_rank <world> <rank> <uidfile> <B_total>   rank mode of nc_group: one process per GPU (device = rank)
  group_local <ndev> <B_total>        local mode: this process drives ndev GPUs (ragged batches allowed)
  threads <iters>                     two host threads, two handles on device 0 (the ABI's threading contract)
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("OMP_WAIT_POLICY", "passive")
os.environ.setdefault("GOMP_SPINCOUNT", "0")

import numpy as np  # noqa: E402


def lstm_fallback():
    from conftest import encodec_cfg_from_meta, load_golden
    from neuralcodecs_amd import Encodec, _lib
    from neuralcodecs_amd.weights import encodec_synthetic_state_dict, save_blob
    from oracle import c_oracle
    g = load_golden("encodec_small48")
    cfg = encodec_cfg_from_meta(g["meta"])
    blob = save_blob(encodec_synthetic_state_dict(cfg, seed=g["meta"]["weight_seed"]))
    ref = c_oracle.RefEncodec(cfg, blob)
    rframes = ref.encode(g["pcm"])
    # host-pointer entry point: the engine notices the timeout after its own synchronisation and repeats the call step-wise
    m = Encodec(cfg)
    m.load_blob(blob)
    frames = m.encode(g["pcm"])
    for f, (rc, rs) in zip(frames, rframes):
        assert np.array_equal(f.codes, rc), "codes after the step-wise retry differ from the oracle"
    m.check_errors()                                   # nothing pending any more
    audio = m.decode(frames, g["pcm"].shape[-1])       # the handle stays on the step-wise kernels: still bit-exact
    assert np.array_equal(audio, ref.decode(rframes))
    m.dispose()
    print("CHILD_OK")


def lstm_fallback_dev():
    """device-pointer path: the failure surfaces at nc_codec_check_errors / the next call, the repeat succeeds"""
    import torch
    from conftest import encodec_cfg_from_meta, load_golden
    from neuralcodecs_amd import Encodec, _lib
    from neuralcodecs_amd.weights import encodec_synthetic_state_dict, save_blob
    from oracle import c_oracle
    g = load_golden("encodec_small48")
    cfg = encodec_cfg_from_meta(g["meta"])
    blob = save_blob(encodec_synthetic_state_dict(cfg, seed=g["meta"]["weight_seed"]))
    ref = c_oracle.RefEncodec(cfg, blob)
    rframes = ref.encode(g["pcm"])
    m = Encodec(cfg)
    m.load_blob(blob)
    x = torch.from_numpy(g["pcm"]).cuda()
    m.encode(x)                                        # flagged as timed out (asynchronously)
    torch.cuda.synchronize()
    try:
        m.check_errors()
        raise SystemExit("nc_codec_check_errors did not report the timed-out launch")
    except _lib.NcDeviceError as e:
        assert "step-wise" in str(e)
    frames = m.encode(x)                               # the repeat runs the step-wise kernels
    torch.cuda.synchronize()
    m.check_errors()
    for f, (rc, rs) in zip(frames, rframes):
        assert np.array_equal(f.codes.cpu().numpy(), rc)
    m.dispose()
    print("CHILD_OK")


def _dac_small():
    from conftest import dac_cfg_from_meta, load_golden
    from neuralcodecs_amd.weights import dac_synthetic_state_dict, save_blob
    g = load_golden("dac_small")
    cfg = dac_cfg_from_meta(g["meta"])
    return cfg, save_blob(dac_synthetic_state_dict(cfg, seed=g["meta"]["weight_seed"]))


def group_rank(world, rank, uidfile, b_total):
    """Every rank encodes its contiguous shard into its slot of the gathered buffer; after the all-gather EVERY slot must equal a plain
    encode of that shard's clips on this rank's own device (gathered codes == 1-GPU run)."""
    import torch
    from neuralcodecs_amd import DAC, parallel
    from neuralcodecs_amd.weights import synthetic_pcm
    cfg, blob = _dac_small()
    assert b_total % world == 0, "rank mode takes equal shards"
    torch.cuda.set_device(rank)
    m = DAC(cfg, device_index=rank)
    m.load_blob(blob)
    if rank == 0:
        uid = parallel.Group.unique_id()
        with open(uidfile + ".tmp", "wb") as f:
            f.write(uid)
        os.replace(uidfile + ".tmp", uidfile)
    else:
        t0 = time.time()
        while not os.path.exists(uidfile):
            if time.time() - t0 > 120:
                raise SystemExit("rank 0 never published the unique id")
            time.sleep(0.05)
        uid = open(uidfile, "rb").read()
    pcm = synthetic_pcm(b_total, 1, 3000, cfg.sample_rate, seed=5)
    lo, hi = parallel.shard_bounds(b_total, world, rank)
    grp = parallel.Group.rank(world, rank, uid, m)
    zd, call, latd = grp.dac_encode_allgather(torch.from_numpy(pcm[lo:hi]).cuda(rank))
    grp.wait()
    audio = m.decode(zd)                                # queued behind the gather on the codec's stream
    torch.cuda.synchronize(rank)
    z_all, codes_all, _, _, _ = m.encode(pcm)           # the whole batch on this device: the 1-GPU answer
    assert np.array_equal(call.cpu().numpy(), codes_all), f"rank {rank}: gathered codes differ from the 1-GPU encode"
    assert np.array_equal(zd.cpu().numpy(), z_all[lo:hi])
    assert np.array_equal(audio.cpu().numpy(), m.decode(z_all[lo:hi]))
    grp.dispose()
    m.dispose()
    print("CHILD_OK")


def group_local(ndev, b_total):
    from neuralcodecs_amd import DAC, SNAC, parallel
    from neuralcodecs_amd.weights import save_blob, snac_synthetic_state_dict, synthetic_pcm
    from conftest import load_golden, snac_cfg_from_meta
    cfg, blob = _dac_small()
    ms = []
    for d in range(ndev):
        m = DAC(cfg, device_index=d)
        m.load_blob(blob)
        ms.append(m)
    pcm = synthetic_pcm(b_total, 1, 3000, cfg.sample_rate, seed=5)
    z, codes, _, _, _ = ms[0].encode(pcm)
    loc = parallel.Group.local(ms)
    c2, z2 = loc.dac_encode_allgather_host(pcm, return_z=True)
    assert np.array_equal(c2, codes) and np.array_equal(z2, z), "local-mode gather differs from the 1-GPU encode"
    if b_total > 1:                                    # a ragged split (and, with ndev > b, devices without clips)
        c3 = loc.dac_encode_allgather_host(pcm[: b_total - 1])
        assert np.array_equal(c3, codes[: b_total - 1])
    c4 = loc.dac_encode_allgather_host(pcm[:1])
    assert np.array_equal(c4, codes[:1])
    # device-resident blocks (nc_group_dac_encode_allgather_local_dev): every device ends up with every block, slots sized for the largest
    import torch
    for bits in (0, 10):
        loc.set_code_bits(bits)
        sizes = parallel.shard_sizes(b_total, ndev)
        bmax, lo, blocks = max(sizes), 0, []
        for d, n in enumerate(sizes):
            blocks.append(torch.from_numpy(pcm[lo:lo + n]).to(torch.device("cuda", d)) if n else None)
            lo += n
        zs, call, _ = loc.dac_encode_allgather_local(blocks)
        loc.wait()
        for d in range(ndev):
            torch.cuda.synchronize(d)
        lo = 0
        for d, n in enumerate(sizes):
            for e in range(ndev):                      # slot d of EVERY device's copy
                assert np.array_equal(call[e][d * bmax: d * bmax + n].cpu().numpy(), codes[lo:lo + n]), ("local_dev slot", d, "on device", e, "bits", bits)
            if n:
                assert np.array_equal(zs[d].cpu().numpy(), z[lo:lo + n])
            lo += n
    loc.set_code_bits(0)
    loc.dispose()
    for m in ms:
        m.dispose()
    g = load_golden("snac_small")
    scfg = snac_cfg_from_meta(g["meta"])
    sblob = save_blob(snac_synthetic_state_dict(scfg, seed=g["meta"]["weight_seed"]))
    ss = []
    for d in range(ndev):
        m = SNAC(scfg, device_index=d)
        m.load_blob(sblob)
        ss.append(m)
    spcm = synthetic_pcm(b_total, 1, 3001, scfg.sampling_rate, seed=6)
    want = ss[0].encode(spcm)
    loc = parallel.Group.local(ss)
    for a, b in zip(loc.snac_encode_allgather_host(spcm), want):
        assert np.array_equal(a, b)
    sizes = parallel.shard_sizes(b_total, ndev)
    bmax, lo, blocks = max(sizes), 0, []
    for d, n in enumerate(sizes):
        blocks.append(torch.from_numpy(spcm[lo:lo + n]).to(torch.device("cuda", d)) if n else None)
        lo += n
    loc.set_code_bits(12)
    call, widths = loc.snac_encode_allgather_local(blocks)
    loc.wait()
    for d in range(ndev):
        torch.cuda.synchronize(d)
    flat_want = np.concatenate([w.reshape(b_total, -1) for w in want], axis=1)
    lo = 0
    for d, n in enumerate(sizes):
        for e in range(ndev):
            assert np.array_equal(call[e][d * bmax: d * bmax + n].cpu().numpy(), flat_want[lo:lo + n])
        lo += n
    loc.set_code_bits(0)
    # kind check before any cast: a SNAC group refuses the DAC entry point with NC_EINVAL, straight at the C ABI
    from neuralcodecs_amd import _lib
    sink = np.zeros(64, np.int64)
    st = _lib.lib().nc_group_dac_encode_allgather(loc._g, spcm.ctypes.data, spcm.shape[0], spcm.shape[2], 0, 0, sink.ctypes.data, None)
    assert st == _lib.NC_EINVAL, st
    loc.dispose()
    for m in ss:
        m.dispose()
    print("CHILD_OK")


def snac_fuse_guard():
    """the one-launch SNAC residual unit addresses a clip's [C][T] block with 32-bit byte offsets, so it may be taken
    only while C * T < 2^30 -- the bound had been on T alone.  NC_SNAC_FUSE_MAX_ELEMS (environment of this child) lowers the bound
    (tests/test_children_gpu.py: 3 000 000 -- at T = 36 864 the C = 96 units, 3.54 M elements, fall back while the C = 64 units stay fused);
    the depthwise launches of the profile tell which path ran, and the result stays bit-exact against the oracle either way."""
    import torch
    from neuralcodecs_amd import SNAC
    from neuralcodecs_amd.config import SNACConfig
    from neuralcodecs_amd.weights import save_blob, snac_noise, snac_synthetic_state_dict, synthetic_pcm
    from oracle import c_oracle
    cfg = SNACConfig.snac_44khz()
    blob = save_blob(snac_synthetic_state_dict(cfg, seed=42))
    m = SNAC(cfg)
    m.load_blob(blob)
    T = 36864
    pcm = synthetic_pcm(2, 1, T, cfg.sampling_rate, seed=4)
    x = torch.from_numpy(pcm).cuda()
    m.profile_enable(True)
    m.profile_reset()
    codes = m.encode(x)
    nz = snac_noise(cfg, 2, codes[-1].shape[1], seed=9)
    audio = m.decode(codes, m.flat_noise(nz, x.device))
    torch.cuda.synchronize()
    dw = m.profile_read()["dwconv"]["launches"]
    want = int(os.environ.get("NC_EXPECT_DW_LAUNCHES", "-1"))
    print("DWCONV_LAUNCHES", dw)
    assert want < 0 or dw == want, (dw, want)
    ref = c_oracle.RefSNAC(cfg, blob)
    _, _, rcodes = ref.encode(pcm[:1])
    for a, b in zip(codes, rcodes):
        assert np.array_equal(a[:1].cpu().numpy(), b)
    assert np.array_equal(audio[:1].cpu().numpy(), ref.decode(rcodes, [n[:1] for n in nz]))
    m.dispose()
    print("CHILD_OK")


def threads(iters):
    """include/nc_mi355x.h "threading": calls on one handle are serialised by the caller, DISTINCT handles may run concurrently (the
    reference: one model per thread, Modules/DAC/WNConv1d.cs:150 mutates module state in forward).  Two host threads, two handles on
    device 0, `iters` encode+decode iterations each, for the pairs DAC || DAC, DAC || Encodec and Encodec || Encodec (BASELINE C3 shape: the
    persistent LSTM needs its workgroups co-resident -- the per-device ticket orders the LSTM sections of the two handles).  Every
    iteration of every thread must equal the handle's own serial result bit for bit, through the host-pointer API (the handle's own
    stream) and through the device-pointer API on a per-thread torch stream; no NC_EDEVICE; no handle may have dropped to the step-wise LSTM."""
    import hashlib
    import threading
    import torch
    from neuralcodecs_amd import DAC, DACConfig, Encodec
    from neuralcodecs_amd.config import EncodecConfig
    from neuralcodecs_amd.weights import dac_synthetic_state_dict, encodec_synthetic_state_dict, save_blob, synthetic_pcm
    dcfg, ecfg = DACConfig.dac_44khz(), EncodecConfig.encodec_48khz()
    dblob = save_blob(dac_synthetic_state_dict(dcfg, seed=42))
    eblob = save_blob(encodec_synthetic_state_dict(ecfg, seed=42))
    dpcm = synthetic_pcm(4, 1, 44100, dcfg.sample_rate, seed=1234)
    epcm = synthetic_pcm(16, 2, 96000, ecfg.sampling_rate, seed=1234)       # C3: 16 x 2 s stereo
    dev = torch.device("cuda", 0)

    def digest(*arrs):
        h = hashlib.sha256()
        for a in arrs:
            h.update(np.ascontiguousarray(a).tobytes())
        return h.hexdigest()

    class DacJob:
        def __init__(self):
            self.m = DAC(dcfg)
            self.m.load_blob(dblob)
            self.x = torch.from_numpy(dpcm).to(dev)

        def host(self):
            z, codes, _, _, _ = self.m.encode(dpcm)
            return digest(codes, self.m.decode(z))

        def device(self):
            z, codes, _, _, _ = self.m.encode(self.x)
            au = self.m.decode(z)
            torch.cuda.current_stream().synchronize()
            self.m.check_errors()
            return digest(codes.cpu().numpy(), au.cpu().numpy())

    class EncJob:
        def __init__(self):
            self.m = Encodec(ecfg)
            self.m.load_blob(eblob)
            self.x = torch.from_numpy(epcm).to(dev)

        def host(self):
            fr = self.m.encode(epcm)
            return digest(*[f.codes for f in fr], self.m.decode(fr, epcm.shape[-1]))

        def device(self):
            fr = self.m.encode(self.x)
            au = self.m.decode(fr, epcm.shape[-1])
            torch.cuda.current_stream().synchronize()
            self.m.check_errors()
            return digest(*[f.codes.cpu().numpy() for f in fr], au.cpu().numpy())

    report = {}
    for name, kinds in (("dac||dac", (DacJob, DacJob)), ("dac||encodec", (DacJob, EncJob)), ("encodec||encodec", (EncJob, EncJob))):
        jobs = [k() for k in kinds]
        serial = []
        for j in jobs:                                  # the serial reference of each handle (host API == device API, too)
            h = j.host()
            with torch.cuda.stream(torch.cuda.Stream(dev)):
                assert j.device() == h, name + ": device-pointer result differs from the host-pointer result (serial)"
            serial.append(h)
        if kinds[0] is kinds[1]:
            assert serial[0] == serial[1], name + ": two handles of one model disagree"
        errs, t0 = [], time.time()

        def work(i):
            try:
                st = torch.cuda.Stream(dev)             # the device-pointer API runs on the caller's CURRENT stream: one per thread
                for it in range(iters):
                    if it % 2 == 0:
                        got = jobs[i].host()
                    else:
                        with torch.cuda.stream(st):
                            got = jobs[i].device()
                    if got != serial[i]:
                        errs.append("%s: thread %d iteration %d differs from its serial result" % (name, i, it))
                        return
            except BaseException as e:                  # noqa: BLE001 (reported below)
                errs.append("%s: thread %d: %r" % (name, i, e))
        th = [threading.Thread(target=work, args=(i,)) for i in range(2)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        assert not errs, errs
        stats = [j.m.lstm_stats() for j in jobs if isinstance(j, EncJob)]
        assert all(not sw and tmo == 0 for sw, tmo in stats), (name, "persistent LSTM fell back to step-wise", stats)
        report[name] = {"iterations_per_thread": iters, "seconds": round(time.time() - t0, 2), "lstm_stepwise_fallbacks": sum(t for _, t in stats)}
        for j in jobs:
            j.m.dispose()
    print("THREADS_REPORT", report)
    print("CHILD_OK")


if __name__ == "__main__":
    mode = sys.argv[1]
    if mode == "lstm_fallback":
        lstm_fallback()
    elif mode == "lstm_fallback_dev":
        lstm_fallback_dev()
    elif mode == "group_rank":
        group_rank(int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], int(sys.argv[5]))
    elif mode == "group_local":
        group_local(int(sys.argv[2]), int(sys.argv[3]))
    elif mode == "threads":
        threads(int(sys.argv[2]))
    elif mode == "snac_fuse_guard":
        snac_fuse_guard()
    else:
        raise SystemExit("unknown mode " + mode)

"""Child processes of the GPU suite (fresh interpreters: created before any GPU call of their own, never via exec from a process that
touched the GPU).  python tests/_gpu_child.py <mode> [args...]; prints CHILD_OK on success, exits non-zero otherwise.

  lstm_fallback                       NC_LSTM_FAKE_TIMEOUT=1 in the environment: the first persistent LSTM launch is reported as timed out
  group_rank <world> <rank> <uidfile> <B_total>   rank mode of nc_group: one process per GPU (device = rank)
  group_local <ndev> <B_total>        local mode: this process drives ndev GPUs (ragged batches allowed)
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
    # kind check before any cast: a SNAC group refuses the DAC entry point with NC_EINVAL, straight at the C ABI
    from neuralcodecs_amd import _lib
    sink = np.zeros(64, np.int64)
    st = _lib.lib().nc_group_dac_encode_allgather(loc._g, spcm.ctypes.data, spcm.shape[0], spcm.shape[2], 0, 0, sink.ctypes.data, None)
    assert st == _lib.NC_EINVAL, st
    loc.dispose()
    for m in ss:
        m.dispose()
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
    else:
        raise SystemExit("unknown mode " + mode)

"""GPU suite: the BASELINE.json batch sizes themselves, through the C ABI.

C2  DAC 44.1 kHz, B=32 x 1 s          C3  Encodec 48 kHz stereo 12 kbps, B=16 x 2 s          C5 share  SNAC 44.1 kHz + LocalMHA, B=8 x 5 s
A sampled subset of clips is compared with the C oracle bit for bit (codes, latents, PCM); every other clip is covered by batch
invariance: the full-batch result of a clip equals the engine's own result for that clip in a small batch (which contains an
oracle-checked clip).  Also: a > 16-segment Encodec clip (the reference's Decode has no frame-count limit) and handle
create/destroy cycles returning their HBM.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from conftest import load_golden, dac_cfg_from_meta, encodec_cfg_from_meta, snac_cfg_from_meta  # noqa: E402
from neuralcodecs_amd import DAC, SNAC, Encodec  # noqa: E402
from neuralcodecs_amd.weights import (dac_synthetic_state_dict, encodec_synthetic_state_dict, save_blob, snac_noise,  # noqa: E402
                                      snac_synthetic_state_dict, synthetic_pcm)
from oracle import c_oracle  # noqa: E402


def test_c2_dac44k_batch32_vs_oracle_and_batch_invariance():
    g = load_golden("dac44k_b1")
    cfg = dac_cfg_from_meta(g["meta"])
    blob = save_blob(dac_synthetic_state_dict(cfg, seed=42))
    m = DAC(cfg)
    m.load_blob(blob)
    ref = c_oracle.RefDAC(cfg, blob)
    B, T = 32, 44100
    pcm = synthetic_pcm(B, 1, T, cfg.sample_rate, seed=1234)
    z, codes, lat, _, _ = m.encode(pcm)
    audio = m.decode(z)
    assert codes.shape == (B, 9, 87) and audio.shape == (B, 1, 44544)
    pick = [0, 13, 31]
    rz, rcodes, rlat, _ = ref.encode(pcm[pick])
    assert np.array_equal(codes[pick], rcodes) and np.array_equal(z[pick], rz) and np.array_equal(lat[pick], rlat)
    assert np.array_equal(audio[pick], ref.decode(rz))
    for lo in range(0, B, 8):                                   # batch invariance: 8-clip groups reproduce the 32-clip run
        z8, c8, _, _, _ = m.encode(pcm[lo:lo + 8])
        assert np.array_equal(c8, codes[lo:lo + 8]) and np.array_equal(z8, z[lo:lo + 8])
        assert np.array_equal(m.decode(z8), audio[lo:lo + 8])
    m.dispose()


def test_c3_encodec48k_batch16x2s_vs_oracle_and_batch_invariance():
    g = load_golden("encodec48k_b1")
    cfg = encodec_cfg_from_meta(g["meta"])
    blob = save_blob(encodec_synthetic_state_dict(cfg, seed=42))
    m = Encodec(cfg)
    m.load_blob(blob)
    ref = c_oracle.RefEncodec(cfg, blob)
    B, T = 16, 96000
    pcm = synthetic_pcm(B, 2, T, cfg.sampling_rate, seed=1234)
    frames = m.encode(pcm)
    audio = m.decode(frames, T)
    assert [f.codes.shape for f in frames] == [(B, 8, 150), (B, 8, 150), (B, 8, 4)] and audio.shape == (B, 2, 96320)
    pick = [0, 9]
    rfr = ref.encode(pcm[pick])
    for f, r in zip(frames, rfr):
        assert np.array_equal(f.codes[pick], r[0]) and np.array_equal(f.scale[pick], r[1])
    assert np.array_equal(audio[pick], ref.decode(rfr))
    for lo in range(0, B, 4):
        f4 = m.encode(pcm[lo:lo + 4])
        assert all(np.array_equal(a.codes, b.codes[lo:lo + 4]) and np.array_equal(a.scale, b.scale[lo:lo + 4]) for a, b in zip(f4, frames))
        assert np.array_equal(m.decode(f4, T), audio[lo:lo + 4])
    m.dispose()


def test_c5_share_snac44k_batch8x5s_vs_oracle_and_batch_invariance():
    g = load_golden("snac44k_short")
    cfg = snac_cfg_from_meta(g["meta"])
    blob = save_blob(snac_synthetic_state_dict(cfg, seed=42))
    m = SNAC(cfg)
    m.load_blob(blob)
    ref = c_oracle.RefSNAC(cfg, blob)
    B, T = 8, 220500
    pcm = synthetic_pcm(B, 1, T, cfg.sampling_rate, seed=1234)
    nz = snac_noise(cfg, B, 576, seed=3)
    codes = m.encode(pcm)
    audio = m.decode(codes, nz)
    assert [c.shape for c in codes] == [(B, 72), (B, 144), (B, 288), (B, 576)] and audio.shape == (B, 1, 221184)
    pick = [5]
    _, _, rcodes = ref.encode(pcm[pick])
    for a, b in zip(codes, rcodes):
        assert np.array_equal(a[pick], b)
    assert np.array_equal(audio[pick], ref.decode(rcodes, [n[pick] for n in nz]))
    for lo in range(0, B, 2):
        c2 = m.encode(pcm[lo:lo + 2])
        assert all(np.array_equal(a, b[lo:lo + 2]) for a, b in zip(c2, codes))
        assert np.array_equal(m.decode(c2, [n[lo:lo + 2] for n in nz]), audio[lo:lo + 2])
    m.dispose()


def test_encodec_long_clip_more_than_16_segments():
    """A 48 kHz clip of 21 s = 22 segments: encode, decode and the overlap-add have no frame-count limit (Encodec.cs:213-235)."""
    g = load_golden("encodec_small48")
    cfg = encodec_cfg_from_meta(g["meta"])
    blob = save_blob(encodec_synthetic_state_dict(cfg, seed=g["meta"]["weight_seed"]))
    m = Encodec(cfg)
    m.load_blob(blob)
    ref = c_oracle.RefEncodec(cfg, blob)
    T = int(21.3 * cfg.sampling_rate * 0.25)                     # segment = 0.25 s in the reduced model -> 22 segments
    pcm = synthetic_pcm(2, 2, T, cfg.sampling_rate, seed=5)
    frames = m.encode(pcm)
    assert len(frames) > 16 and len(frames) == m.query(T)[0]
    rfr = ref.encode(pcm)
    assert all(np.array_equal(f.codes, r[0]) for f, r in zip(frames, rfr))
    assert np.array_equal(m.decode(frames, T), ref.decode(rfr))
    m.dispose()
    # the full-size 48 kHz model on a 20.5 s mono-batch clip: 21 segments (device API, shapes + determinism + overlap-add weights sum)
    g = load_golden("encodec48k_b1")
    cfg = encodec_cfg_from_meta(g["meta"])
    m = Encodec(cfg)
    m.load_blob(save_blob(encodec_synthetic_state_dict(cfg, seed=42)))
    T = int(20.5 * 48000)
    pcm = synthetic_pcm(1, 2, T, cfg.sampling_rate, seed=6)
    frames = m.encode(pcm)
    assert len(frames) == 21
    a1 = m.decode(frames, T)
    assert a1.shape[-1] >= T and np.isfinite(a1).all() and np.array_equal(a1, m.decode(frames, T))
    m.dispose()


def test_handles_return_their_hbm_on_dispose():
    """DAC.Dispose semantics: create -> load -> run -> destroy cycles must not accumulate device memory."""
    import torch
    g = load_golden("dac44k_b1")
    cfg = dac_cfg_from_meta(g["meta"])
    blob = save_blob(dac_synthetic_state_dict(cfg, seed=42))
    pcm = synthetic_pcm(4, 1, 44100, cfg.sample_rate, seed=1)
    used = []
    for _ in range(4):
        m = DAC(cfg)
        m.load_blob(blob)
        m.decode(m.encode(pcm)[0])
        m.dispose()
        torch.cuda.synchronize()
        free, total = torch.cuda.mem_get_info()
        used.append(total - free)
    assert used[-1] - used[0] < 64 << 20, f"device memory grows across create/destroy cycles: {used}"


def test_encodec_large_batch_lstm_tile_groups_batch_invariance():
    """B = 40 x 2 equal segments = 80 LSTM columns = 5 column tiles: the persistent LSTM layer runs as two launch groups (4 + 1
    tiles) next to the tail segment's group on a side stream; every clip must equal its result in a 4-clip batch."""
    g = load_golden("encodec48k_b1")
    cfg = encodec_cfg_from_meta(g["meta"])
    m = Encodec(cfg)
    m.load_blob(save_blob(encodec_synthetic_state_dict(cfg, seed=42)))
    B, T = 40, 96000
    pcm = synthetic_pcm(B, 2, T, cfg.sampling_rate, seed=77)
    frames = m.encode(pcm)
    audio = m.decode(frames, T)
    for lo in (0, 16, 36):
        f4 = m.encode(pcm[lo:lo + 4])
        assert all(np.array_equal(a.codes, b.codes[lo:lo + 4]) and np.array_equal(a.scale, b.scale[lo:lo + 4]) for a, b in zip(f4, frames))
        assert np.array_equal(m.decode(f4, T), audio[lo:lo + 4])
    m.dispose()


# ---- BASELINE configs C4 / C5 at their GLOBAL batch size on one GPU ---------------------------------------------------
# No multi-GPU node exists for this build, so the 8-way form is exercised as far as one GPU allows: (a) the whole batch in ONE call (B*C*T
# beyond 2^31 bytes per activation), (b) every one of the eight parallel.shard_bounds(., 8, r) blocks encoded alone == its rows of the
# whole-batch result, sampled clips == the C oracle, (c) the blocks pushed through one-member groups into their rows of the gathered tensor,
# (d) an EIGHT-member group on this one device (NC_GROUP_PEER_COPY: eight codec handles, eight streams, every member ends with all eight
# slots -- the slot arithmetic of csrc/nc_group.hip for W = 8, int64 and bit-packed, equal and ragged blocks).  What stays unmeasured: the
# RCCL collective itself across GPUs.

def _eight_blocks(n):
    from neuralcodecs_amd.parallel import shard_bounds
    return [shard_bounds(n, 8, r) for r in range(8)]


def test_c4_dac44k_global_batch256_eight_way_on_one_gpu():
    import torch
    from neuralcodecs_amd import parallel
    g = load_golden("dac44k_b1")
    cfg = dac_cfg_from_meta(g["meta"])
    blob = save_blob(dac_synthetic_state_dict(cfg, seed=42))
    m = DAC(cfg)
    m.load_blob(blob)
    B, T = 256, 44100
    pcm = synthetic_pcm(B, 1, T, cfg.sample_rate, seed=1234)
    x = torch.from_numpy(pcm).cuda()
    z, codes, lat, _, _ = m.encode(x)                                     # (a) ONE call over the global batch, device resident
    audio = m.decode(z)
    torch.cuda.synchronize()
    assert codes.shape == (B, 9, 87) and audio.shape == (B, 1, 44544)
    codes_h, z_h, audio_h = codes.cpu().numpy(), z.cpu().numpy(), audio.cpu().numpy()
    ref = c_oracle.RefDAC(cfg, blob)
    pick = [0, 100, 255]                                                  # (first / middle / last shard)
    rz, rcodes, rlat, _ = ref.encode(pcm[pick])
    assert np.array_equal(codes_h[pick], rcodes) and np.array_equal(z_h[pick], rz) and np.array_equal(lat.cpu().numpy()[pick], rlat)
    assert np.array_equal(audio_h[pick], ref.decode(rz))
    gathered = torch.full((B, 9, 87), -1, dtype=torch.int64, device="cuda")
    one = parallel.Group.local([m])
    for lo, hi in _eight_blocks(B):                                       # (b) + (c)
        assert hi - lo == 32
        zs, cs, _, _, _ = m.encode(x[lo:hi])
        assert torch.equal(cs, codes[lo:hi]) and torch.equal(zs, z[lo:hi])
        assert torch.equal(m.decode(zs), audio[lo:hi])
        one.dac_encode_allgather_local([x[lo:hi]], codes_all=[gathered[lo:hi]])
        one.wait()
    torch.cuda.synchronize()
    assert torch.equal(gathered, codes)
    one.dispose()
    # (d) eight members on this device
    ms = [m] + [DAC(cfg) for _ in range(7)]
    for k in ms[1:]:
        k.load_blob(blob)
    grp = parallel.Group.local(ms, peer_copy=True)
    blocks = [x[lo:hi] for lo, hi in _eight_blocks(B)]
    for bits in (0, 10):
        grp.set_code_bits(bits)
        zs, call, _ = grp.dac_encode_allgather_local(blocks)
        dec = [k.decode(zz) for k, zz in zip(ms, zs)]                      # the local decode, queued before the gather is awaited
        grp.wait()
        torch.cuda.synchronize()
        for d in range(8):
            assert torch.equal(call[d], codes), (bits, d)
            assert torch.equal(dec[d], audio[32 * d: 32 * d + 32])
    grp.set_code_bits(0)
    # ragged: 250 clips -> blocks of 32,32,31,... (slots sized for 32, zero padded), device form and host form
    rb = _eight_blocks(250)
    zs, call, _ = grp.dac_encode_allgather_local([x[lo:hi] for lo, hi in rb])
    grp.wait()
    torch.cuda.synchronize()
    for d in range(8):
        for r, (lo, hi) in enumerate(rb):
            assert torch.equal(call[d][32 * r: 32 * r + (hi - lo)], codes[lo:hi])
            assert int(call[d][32 * r + (hi - lo): 32 * r + 32].abs().sum()) == 0
    hc, hz = grp.dac_encode_allgather_host(pcm[:250], return_z=True)
    assert np.array_equal(hc, codes_h[:250]) and np.array_equal(hz, z_h[:250])
    grp.set_code_bits(10)
    assert np.array_equal(grp.dac_encode_allgather_host(pcm), codes_h)
    grp.dispose()
    for k in ms:
        k.dispose()


def test_c5_snac44k_global_batch64x5s_eight_way_on_one_gpu():
    import torch
    from neuralcodecs_amd import parallel
    g = load_golden("snac44k_short")
    cfg = snac_cfg_from_meta(g["meta"])
    blob = save_blob(snac_synthetic_state_dict(cfg, seed=42))
    m = SNAC(cfg)
    m.load_blob(blob)
    B, T = 64, 220500
    pcm = synthetic_pcm(B, 1, T, cfg.sampling_rate, seed=1234)
    nz = snac_noise(cfg, B, 576, seed=3)
    x = torch.from_numpy(pcm).cuda()
    nzd = [torch.from_numpy(n).cuda() for n in nz]
    codes = m.encode(x)                                                    # (a) ONE call over the global batch
    audio = m.decode(codes, nzd)
    torch.cuda.synchronize()
    assert [tuple(c.shape) for c in codes] == [(B, 72), (B, 144), (B, 288), (B, 576)] and tuple(audio.shape) == (B, 1, 221184)
    flat, widths = parallel.concat_levels(codes)
    assert widths == [72, 144, 288, 576]
    ref = c_oracle.RefSNAC(cfg, blob)
    pick = [63]
    _, _, rcodes = ref.encode(pcm[pick])
    for a, b in zip(codes, rcodes):
        assert np.array_equal(a.cpu().numpy()[pick], b)
    assert np.array_equal(audio.cpu().numpy()[pick], ref.decode(rcodes, [n[pick] for n in nz]))
    gathered = torch.full((B, 1080), -1, dtype=torch.int64, device="cuda")
    one = parallel.Group.local([m])
    for lo, hi in _eight_blocks(B):                                        # (b) + (c)
        assert hi - lo == 8
        cs = m.encode(x[lo:hi])
        assert all(torch.equal(a, b[lo:hi]) for a, b in zip(cs, codes))
        assert torch.equal(m.decode(cs, [n[lo:hi] for n in nzd]), audio[lo:hi])
        one.snac_encode_allgather_local([x[lo:hi]], codes_all=[gathered[lo:hi]])
        one.wait()
    torch.cuda.synchronize()
    assert torch.equal(gathered, flat)
    one.dispose()
    ms = [m] + [SNAC(cfg) for _ in range(7)]                               # (d) eight members on this device
    for k in ms[1:]:
        k.load_blob(blob)
    grp = parallel.Group.local(ms, peer_copy=True)
    blocks = [x[lo:hi] for lo, hi in _eight_blocks(B)]
    for bits in (0, 12):                                                   # 4096-entry codebooks: 12 bits
        grp.set_code_bits(bits)
        call, w = grp.snac_encode_allgather_local(blocks)
        grp.wait()
        torch.cuda.synchronize()
        assert w == widths
        for d in range(8):
            assert torch.equal(call[d], flat), (bits, d)
    grp.set_code_bits(12)
    host = grp.snac_encode_allgather_host(pcm[:61])                        # ragged: 8,8,8,8,8,7,7,7
    for a, b in zip(host, codes):
        assert np.array_equal(a, b.cpu().numpy()[:61])
    grp.dispose()
    for k in ms:
        k.dispose()

"""GPU suite: the nc_group entry points (multi-GPU behind the C ABI, RCCL opened by the engine itself) with ONE rank / ONE device --
what a 1-GPU box can exercise: the all-gather of a single rank must hand back exactly the codes of the plain encode, in both modes,
and the gathered buffer must be usable by a decode queued behind nc_group_wait.  The N > 1 logic (contiguous shards, slot offsets) is
covered on CPU by tests/test_sharding_cpu.py."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from conftest import dac_cfg_from_meta, load_golden, snac_cfg_from_meta  # noqa: E402
from neuralcodecs_amd import DAC, SNAC, parallel  # noqa: E402
from neuralcodecs_amd.weights import dac_synthetic_state_dict, save_blob, snac_synthetic_state_dict, synthetic_pcm  # noqa: E402


def test_dac_group_rank_and_local_modes_single_device():
    import torch
    g = load_golden("dac_small")
    cfg = dac_cfg_from_meta(g["meta"])
    m = DAC(cfg)
    m.load_blob(save_blob(dac_synthetic_state_dict(cfg, seed=g["meta"]["weight_seed"])))
    pcm = synthetic_pcm(4, 1, 3000, cfg.sample_rate, seed=5)
    z, codes, lat, _, _ = m.encode(pcm)
    audio = m.decode(z)
    grp = parallel.Group.rank(1, 0, parallel.Group.unique_id(), m)
    zd, call, latd = grp.dac_encode_allgather(torch.from_numpy(pcm).cuda())
    grp.wait()
    ad = m.decode(zd)
    torch.cuda.synchronize()
    assert np.array_equal(call.cpu().numpy(), codes) and np.array_equal(zd.cpu().numpy(), z) and np.array_equal(latd.cpu().numpy(), lat)
    assert np.array_equal(ad.cpu().numpy(), audio)
    grp.dispose()
    loc = parallel.Group.local([m])
    c2, z2 = loc.dac_encode_allgather_host(pcm, return_z=True)
    assert np.array_equal(c2, codes) and np.array_equal(z2, z)
    c3 = loc.dac_encode_allgather_host(pcm, n_quantizers=2)
    assert c3.shape[1] == 2 and np.array_equal(c3, codes[:, :2])
    with pytest.raises(ValueError):
        parallel.Group.local([m, m])                         # two handles on one device
    loc.dispose()
    m.dispose()


def test_snac_group_levels_in_one_collective():
    import torch
    g = load_golden("snac_small")
    cfg = snac_cfg_from_meta(g["meta"])
    m = SNAC(cfg)
    m.load_blob(save_blob(snac_synthetic_state_dict(cfg, seed=g["meta"]["weight_seed"])))
    pcm = synthetic_pcm(3, 1, 3001, cfg.sampling_rate, seed=6)
    codes = m.encode(pcm)
    grp = parallel.Group.rank(1, 0, parallel.Group.unique_id(), m)
    flat, widths = grp.snac_encode_allgather(torch.from_numpy(pcm).cuda())
    grp.wait()
    torch.cuda.synchronize()
    for a, b in zip(parallel.split_levels(flat.cpu().numpy(), widths), codes):
        assert np.array_equal(a, b)
    grp.dispose()
    loc = parallel.Group.local([m])
    for a, b in zip(loc.snac_encode_allgather_host(pcm), codes):
        assert np.array_equal(a, b)
    loc.dispose()
    m.dispose()


def test_bit_packed_all_gather_payload_equals_the_plain_one():
    """nc_group_set_code_bits / parallel.all_gather_codes(bits=): the collective moves the codes in the BitPacker wire layout
    (Modules/Encodec/BitPacker.cs; 10 bits for 1024-entry codebooks, 8 here for the fixtures' 64 / 256 entries) and every slot is
    unpacked back on the device -- the int64 tensors the caller sees are those of the unpacked path, in rank and in local mode."""
    import torch
    import torch.distributed as dist
    g = load_golden("dac_small")
    cfg = dac_cfg_from_meta(g["meta"])
    m = DAC(cfg)
    m.load_blob(save_blob(dac_synthetic_state_dict(cfg, seed=g["meta"]["weight_seed"])))
    pcm = synthetic_pcm(5, 1, 3000, cfg.sample_rate, seed=8)
    _, codes, _, _, _ = m.encode(pcm)
    grp = parallel.Group.rank(1, 0, parallel.Group.unique_id(), m)
    for bits in (6, 7, 24, 0):                                # 64-entry codebooks: 6 bits is exact; 0 switches back to int64
        grp.set_code_bits(bits)
        _, call, _ = grp.dac_encode_allgather(torch.from_numpy(pcm).cuda())
        grp.wait()
        torch.cuda.synchronize()
        assert np.array_equal(call.cpu().numpy(), codes), bits
    with pytest.raises(ValueError):
        grp.set_code_bits(25)                                 # BitPacker.cs MaxBits
    grp.dispose()
    loc = parallel.Group.local([m])
    loc.set_code_bits(6)
    assert np.array_equal(loc.dac_encode_allgather_host(pcm), codes)
    loc.dispose()
    # SNAC: the levels of a clip side by side, 256-entry codebooks
    gs = load_golden("snac_small")
    scfg = snac_cfg_from_meta(gs["meta"])
    sm = SNAC(scfg)
    sm.load_blob(save_blob(snac_synthetic_state_dict(scfg, seed=gs["meta"]["weight_seed"])))
    spcm = synthetic_pcm(3, 1, 3001, scfg.sampling_rate, seed=6)
    scodes = sm.encode(spcm)
    sg = parallel.Group.rank(1, 0, parallel.Group.unique_id(), sm)
    sg.set_code_bits(8)
    flat, widths = sg.snac_encode_allgather(torch.from_numpy(spcm).cuda())
    sg.wait()
    torch.cuda.synchronize()
    for a, b in zip(parallel.split_levels(flat.cpu().numpy(), widths), scodes):
        assert np.array_equal(a, b)
    sg.dispose()
    sm.dispose()
    # the torch.distributed form (bench.py --pack-bits), one rank on RCCL
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        ct = torch.from_numpy(codes).cuda()
        assert torch.equal(parallel.all_gather_codes(ct, ct.shape[0], bits=6), ct)
        assert torch.equal(parallel.all_gather_codes(ct, ct.shape[0], bits=10), ct)
        with pytest.raises(ValueError):
            parallel.all_gather_codes(torch.from_numpy(codes), ct.shape[0], bits=6)     # packed payloads are device work
    finally:
        dist.destroy_process_group()
    m.dispose()


def test_packed_rows_of_a_rank_without_clips_keep_the_row_width():
    """parallel._pack_rows sized its packed rows from codes[0] -- a rank holding NO clips (ragged shards, world > n_clips) packed
    [0, 0] while the others packed [b, nb], and the padded all_gather_into_tensor sizes then differed across ranks.  The row width must come
    from the tensor's shape alone."""
    import torch
    from neuralcodecs_amd import _lib
    empty = torch.empty((0, 9, 87), dtype=torch.int64, device="cuda")
    some = torch.randint(0, 1024, (3, 9, 87), dtype=torch.int64, device="cuda")
    nb = int(_lib.lib().nc_packed_bytes(9 * 87, 10))
    assert tuple(parallel._pack_rows(empty, 10).shape) == (0, nb)
    packed = parallel._pack_rows(some, 10)
    assert tuple(packed.shape) == (3, nb)
    assert torch.equal(parallel._unpack_rows(packed, (9, 87), 10), some)


def test_encodec_group_rank_and_local_modes_single_device():
    """nc_group_encodec_encode_allgather[_local]_dev (VERDICT r4 "missing" 5; Models/Encodec.cs:259-285): a rank's frames -- codes of every
    segment end to end + the per-segment scales -- land in its block of the gathered tensors, plain and 10-bit packed (BitsPerCodebook,
    Encodec.cs:87); world = 1 / one device here: the gathered block must be what a plain encode returns, and decode must accept its views."""
    import torch
    from conftest import encodec_cfg_from_meta
    from neuralcodecs_amd import Encodec
    from neuralcodecs_amd.weights import encodec_synthetic_state_dict
    g = load_golden("encodec_small48")
    cfg = encodec_cfg_from_meta(g["meta"])
    m = Encodec(cfg)
    m.load_blob(save_blob(encodec_synthetic_state_dict(cfg, seed=g["meta"]["weight_seed"])))
    pcm = g["pcm"]
    B, T = pcm.shape[0], pcm.shape[-1]
    want = m.encode(pcm)
    x = torch.from_numpy(pcm).cuda()
    for bits in (0, 10):
        grp = parallel.Group.rank(1, 0, parallel.Group.unique_id(), m)
        grp.set_code_bits(bits)
        call, sall, lens, nq = grp.encodec_encode_allgather(x)
        grp.wait()
        torch.cuda.synchronize()
        frames = parallel.Group.encodec_frames(call[0], None if sall is None else sall[0], B, nq, lens)
        assert len(frames) == len(want)
        for (c, s), w in zip(frames, want):
            assert np.array_equal(c.cpu().numpy(), w.codes)
            assert (s is None) == (w.scale is None) and (s is None or np.array_equal(s.cpu().numpy().reshape(-1), np.asarray(w.scale).reshape(-1)))
        grp.dispose()
        loc = parallel.Group.local([m])
        loc.set_code_bits(bits)
        lcall, lsall, lens2, nq2 = loc.encodec_encode_allgather_local([x])
        loc.wait()
        torch.cuda.synchronize()
        assert lens2 == lens and nq2 == nq and torch.equal(lcall[0], call) and (lsall is None or torch.equal(lsall[0], sall))
        loc.dispose()
    m.dispose()


def test_encodec_four_member_peer_copy_group_on_one_device():
    """Round 6: the peer-copy transport (NC_GROUP_PEER_COPY) with FOUR Encodec handles on the one device -- codes (int64 and 10-bit packed) and
    the per-segment scales of every member's block reach every member's copy of the gathered tensors; block d == a plain encode of its clips."""
    import torch
    from conftest import encodec_cfg_from_meta
    from neuralcodecs_amd import Encodec
    from neuralcodecs_amd.weights import encodec_synthetic_state_dict
    g = load_golden("encodec_small48")
    cfg = encodec_cfg_from_meta(g["meta"])
    blob = save_blob(encodec_synthetic_state_dict(cfg, seed=g["meta"]["weight_seed"]))
    ms = [Encodec(cfg) for _ in range(4)]
    for m in ms:
        m.load_blob(blob)
    B, T = 2, 8100
    pcm = synthetic_pcm(4 * B, cfg.channels, T, cfg.sampling_rate, seed=17)
    want = [ms[0].encode(pcm[d * B:(d + 1) * B]) for d in range(4)]
    blocks = [torch.from_numpy(pcm[d * B:(d + 1) * B]).cuda() for d in range(4)]
    grp = parallel.Group.local(ms, peer_copy=True)
    for bits in (0, 10):
        grp.set_code_bits(bits)
        call, sall, lens, nq = grp.encodec_encode_allgather_local(blocks)
        grp.wait()
        torch.cuda.synchronize()
        for member in range(4):
            for d in range(4):
                frames = parallel.Group.encodec_frames(call[member][d], None if sall is None else sall[member][d], B, nq, lens)
                assert len(frames) == len(want[d])
                for (c, s), w in zip(frames, want[d]):
                    assert np.array_equal(c.cpu().numpy(), w.codes), (bits, member, d)
                    assert s is None or np.array_equal(s.cpu().numpy().reshape(-1), np.asarray(w.scale).reshape(-1))
    with pytest.raises(ValueError):
        parallel.Group.local([ms[0], ms[0]], peer_copy=True)      # the same handle twice: one codec (stream, workspace) per member
    grp.dispose()
    for m in ms:
        m.dispose()

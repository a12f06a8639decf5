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

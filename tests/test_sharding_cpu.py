"""CPU suite: the multi-GPU sharding logic on gloo, world_size 2 (one process per "GPU").

Each rank encodes its contiguous block of clips with the C oracle (stand-in for the per-GPU engine: the engine itself needs a
device), the integer codes are all-gathered, and the result must equal the single-process encode of the whole batch byte for byte
(SURVEY 8e verification rule).  Ragged shards (3 clips over 2 ranks) and the SNAC level concat/split helpers are covered too.
"""
import os
import socket

import numpy as np
import pytest

from neuralcodecs_amd import parallel


def test_shard_bounds_cover_the_batch():
    for n in (0, 1, 7, 32, 256):
        for w in (1, 2, 3, 8):
            spans = [parallel.shard_bounds(n, w, r) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1 and sizes == parallel.shard_sizes(n, w)
    assert parallel.shard_bounds(256, 8, 3) == (96, 128)          # BASELINE config C4: 32 clips per GPU
    with pytest.raises(ValueError):
        parallel.shard_bounds(4, 2, 2)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_clips, out_dir):
    import torch
    import torch.distributed as dist
    from conftest import dac_cfg_from_meta, load_golden
    from neuralcodecs_amd.weights import dac_synthetic_state_dict, save_blob, synthetic_pcm
    from oracle import c_oracle
    torch.set_num_threads(1)
    os.environ["OMP_NUM_THREADS"] = "2"
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    g = load_golden("dac_small")
    cfg = dac_cfg_from_meta(g["meta"])
    ref = c_oracle.RefDAC(cfg, save_blob(dac_synthetic_state_dict(cfg, seed=g["meta"]["weight_seed"])))
    pcm = synthetic_pcm(n_clips, 1, 1500, cfg.sample_rate, seed=77)      # every rank can regenerate the whole batch
    lo, hi = parallel.shard_bounds(n_clips, world, rank)
    _, codes, _, _ = ref.encode(pcm[lo:hi])
    gathered = parallel.all_gather_codes(torch.from_numpy(codes), n_clips)
    flat, widths = parallel.concat_levels([gathered[:, 0, :], gathered[:, 1, :3]])
    lv = parallel.split_levels(flat, widths)
    assert torch.equal(lv[0], gathered[:, 0, :]) and torch.equal(lv[1], gathered[:, 1, :3])
    # SNAC-style level lists: this rank's levels -> one collective -> every rank holds all clips' levels in clip order
    mine = [torch.from_numpy(codes[:, 0, :]), torch.from_numpy(codes[:, 1, :3])]
    allv = parallel.split_levels(parallel.all_gather_levels(mine, n_clips), [codes.shape[-1], 3])
    assert torch.equal(allv[0], gathered[:, 0, :]) and torch.equal(allv[1], gathered[:, 1, :3])
    if rank == 0:
        _, full, _, _ = ref.encode(pcm)
        np.save(os.path.join(out_dir, "ok.npy"), np.array([int(np.array_equal(gathered.numpy(), full))]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_clips", [4, 3])
def test_two_rank_gather_equals_single_process(tmp_path, n_clips):
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_worker, args=(2, port, n_clips, str(tmp_path)), nprocs=2, join=True)
    assert np.load(tmp_path / "ok.npy")[0] == 1


def test_bench_gpus2_without_devices_fails_fast_with_a_clear_message():
    """`python bench.py --gpus 2` with no WORLD_SIZE is its own launcher (bench.self_launch).  On a host with fewer than two GPUs it must
    say so and exit non-zero -- before starting any rank, without a rendezvous to hang in."""
    import subprocess
    import sys
    import time
    import torch
    if torch.cuda.device_count() >= 2:
        import pytest
        pytest.skip("this host could really run two ranks")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "NC_BENCH_CHILD")}
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 2, (r.returncode, r.stderr[-1000:])
    assert "GPU(s)" in r.stderr and "nothing was started" in r.stderr
    assert r.stdout.strip() == ""
    assert time.time() - t0 < 240

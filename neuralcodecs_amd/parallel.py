"""Batch sharding over the GPUs of one node (one process per GPU, torch.distributed; backend "nccl" is RCCL on ROCm).

The Encode/Decode path has no cross-clip dependency (no BatchNorm; GroupNorm / LayerNorm / RMS scale are per
sample; RVQ is per frame), so clips are split into contiguous blocks, every rank runs the single-GPU engine on its block with
replicated weights, and the only collective is the all-gather of the emitted integer codes.  Decode needs only the local
latents, so the gather is issued on a side stream and overlaps the local decode.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple


def shard_bounds(n_clips: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous block [lo, hi) of rank `rank`; the first n_clips % world ranks hold one clip more."""
    if world <= 0 or not (0 <= rank < world) or n_clips < 0:
        raise ValueError("bad shard request")
    q, r = divmod(n_clips, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def shard_sizes(n_clips: int, world: int) -> List[int]:
    return [shard_bounds(n_clips, world, r)[1] - shard_bounds(n_clips, world, r)[0] for r in range(world)]


def _pack_rows(codes, bits: int):
    """[b, ...] int64 CUDA tensor -> [b, packed_bytes] uint8 (device kernel nc_pack_codes_dev: the BitPacker wire layout, `bits` per
    value, one packed row per clip, values in the tensor's own element order)."""
    import ctypes as C
    import torch
    from . import _lib
    if not codes.is_cuda:
        raise ValueError("bit-packed all-gather payloads are packed by device kernels: the codes must live on the GPU")
    b = int(codes.shape[0])
    per = 1                                    # values per clip from the SHAPE: a rank without clips (ragged shards, world > n_clips)
    for d in codes.shape[1:]:                  # must still size its packed rows like every other rank, or the collective's byte
        per *= int(d)                          # counts differ across ranks
    L = _lib.lib()
    nb = int(L.nc_packed_bytes(per, bits))
    out = torch.empty((b, nb), dtype=torch.uint8, device=codes.device)
    if b:
        c = codes.contiguous().to(torch.int64)
        _lib.check(L.nc_pack_codes_dev(codes.device.index or 0, c.data_ptr(), b, 1, per, bits, out.data_ptr(),
                                       C.c_void_p(torch.cuda.current_stream(codes.device).cuda_stream)))
    return out


def _unpack_rows(packed, tail, bits: int, out=None):
    import ctypes as C
    import torch
    from . import _lib
    b = int(packed.shape[0])
    per = 1
    for d in tail:
        per *= int(d)
    if out is None:
        out = torch.empty((b,) + tuple(tail), dtype=torch.int64, device=packed.device)
    if b:
        _lib.check(_lib.lib().nc_unpack_codes_dev(packed.device.index or 0, packed.contiguous().data_ptr(), b, 1, per, bits, out.data_ptr(),
                                                  C.c_void_p(torch.cuda.current_stream(packed.device).cuda_stream)))
    return out


def all_gather_codes(codes, n_clips: int, group=None, out=None, bits: Optional[int] = None):
    """codes: this rank's [b_local, ...] integer tensor (torch).  Returns the [n_clips, ...] tensor of all ranks in clip order.

    Equal shards use one all_gather_into_tensor (a single RCCL collective); ragged shards are padded to the largest shard,
    gathered, and the padding rows are dropped.  bits = 10 / 12 / ... moves the codes bit-packed (Modules/Encodec/BitPacker.cs
    layout, packed and unpacked on the device: 64 / bits times fewer bytes over xGMI); the result is the same int64 tensor.
    """
    import torch
    import torch.distributed as dist
    if bits:
        tail = tuple(codes.shape[1:])
        packed = all_gather_codes(_pack_rows(codes, int(bits)), n_clips, group=group)
        return _unpack_rows(packed, tail, int(bits), out=out)
    world = dist.get_world_size(group)
    sizes = shard_sizes(n_clips, world)
    if codes.shape[0] != sizes[dist.get_rank(group)]:
        raise ValueError(f"rank holds {codes.shape[0]} clips, expected {sizes[dist.get_rank(group)]}")
    bmax = max(sizes)
    tail = tuple(codes.shape[1:])
    if all(s == bmax for s in sizes):
        if out is None:
            out = torch.empty((n_clips,) + tail, dtype=codes.dtype, device=codes.device)
        dist.all_gather_into_tensor(out, codes.contiguous(), group=group)
        return out
    pad = torch.zeros((bmax,) + tail, dtype=codes.dtype, device=codes.device)
    pad[: codes.shape[0]] = codes
    buf = torch.empty((world * bmax,) + tail, dtype=codes.dtype, device=codes.device)
    dist.all_gather_into_tensor(buf, pad, group=group)
    parts = [buf[r * bmax: r * bmax + sizes[r]] for r in range(world)]
    return torch.cat(parts, 0)


def concat_levels(levels: Sequence, ) -> Tuple["object", List[int]]:
    """SNAC emits one code tensor per level ([B, T'/stride_i]); concatenate them along time so ONE collective moves them."""
    import torch
    widths = [int(l.shape[-1]) for l in levels]
    return torch.cat([l.reshape(l.shape[0], -1) for l in levels], dim=-1), widths


def all_gather_levels(levels: Sequence, n_clips: int, group=None, out=None, bits: Optional[int] = None):
    """SNAC.Encode's List<Tensor> (SNAC.cs:113-150) of this rank -> the [n_clips, sum(widths)] tensor of all ranks (levels of a clip
    side by side, the layout nc_snac_encode emits): one collective for all levels.  split_levels(result, widths) restores the list."""
    flat, _ = concat_levels(levels)
    return all_gather_codes(flat.contiguous(), n_clips, group=group, out=out, bits=bits)


def split_levels(flat, widths: Sequence[int]):
    out, o = [], 0
    for w in widths:
        out.append(flat[:, o:o + w])
        o += w
    return out


class Group:
    """nc_group of the C ABI (include/nc_mi355x.h "multi-GPU groups"): the sharded encode + RCCL all-gather without torch.distributed.

    rank mode  -- Group.rank(world, rank, uid, codec): one process per GPU; `uid` = Group.unique_id() drawn by rank 0 and handed to
                  the other ranks out of band (bench.py broadcasts it over the process group it already has).
    local mode -- Group.local([codec_dev0, codec_dev1, ...]): one process drives all devices (the layout of a single C# host).
    """

    def __init__(self, handle, world, rank, codecs):
        self._g, self.world, self.rank_id, self._codecs = handle, world, rank, codecs

    @staticmethod
    def unique_id() -> bytes:
        import ctypes as C
        from . import _lib
        buf = (C.c_char * 128)()
        _lib.check(_lib.lib().nc_group_unique_id(buf))
        return bytes(buf)

    @classmethod
    def rank(cls, world: int, rank: int, uid: bytes, codec) -> "Group":
        import ctypes as C
        from . import _lib
        g = C.c_void_p()
        _lib.check(_lib.lib().nc_group_create_rank(world, rank, uid, codec._h, C.byref(g)))
        return cls(g, world, rank, [codec])

    @classmethod
    def local(cls, codecs: Sequence, peer_copy: bool = False) -> "Group":
        """peer_copy=True (NC_GROUP_PEER_COPY): the gather is moved by peer copies instead of RCCL; the codecs may then share a device (W
        handles on one GPU split a batch W ways exactly as W GPUs would)."""
        import ctypes as C
        from . import _lib
        arr = (C.c_void_p * len(codecs))(*[c._h for c in codecs])
        g = C.c_void_p()
        _lib.check(_lib.lib().nc_group_create_local_ex(len(codecs), arr, 1 if peer_copy else 0, C.byref(g)))
        return cls(g, len(codecs), -1, list(codecs))

    def set_code_bits(self, bits: int) -> None:
        """0 = int64 payload (default); 1..24 = the all-gather moves bit-packed codes (nc_group_set_code_bits)."""
        from . import _lib
        _lib.check(_lib.lib().nc_group_set_code_bits(self._g, int(bits)))

    def dispose(self):
        if self._g:
            from . import _lib
            _lib.lib().nc_group_destroy(self._g)
            self._g = None

    def __del__(self):
        try:
            self.dispose()
        except Exception:
            pass

    # ---- rank mode: torch CUDA tensors, asynchronous on the codec's stream + the group's side stream -----------------------------
    def dac_encode_allgather(self, pcm, codes_all=None, n_quantizers: int = 0):
        """pcm [B_local,1,T] (torch, cuda) -> (z_local, codes_all [world*B_local, n_q, T'], latents_local); call wait() before
        reading codes_all on the codec's stream."""
        import torch
        from . import _lib
        m = self._codecs[0]
        B, _, T = pcm.shape
        Tz = m.frames(T)
        nq = n_quantizers if 0 < n_quantizers <= m.config.n_codebooks else m.config.n_codebooks
        x = pcm.contiguous().to(torch.float32)
        if codes_all is None:
            codes_all = torch.empty((self.world * B, nq, Tz), dtype=torch.int64, device=x.device)
        z = torch.empty((B, m.latent_dim, Tz), dtype=torch.float32, device=x.device)
        lat = torch.empty((B, nq * m.config.codebook_dim, Tz), dtype=torch.float32, device=x.device)
        m._bind_torch_stream()
        _lib.check(_lib.lib().nc_group_dac_encode_allgather_dev(self._g, x.data_ptr(), B, T, m.config.sample_rate, n_quantizers,
                                                                codes_all.data_ptr(), z.data_ptr(), lat.data_ptr()))
        return z, codes_all, lat

    def snac_encode_allgather(self, pcm, codes_all=None):
        """pcm [B_local,1,T] (torch, cuda) -> codes_all [world*B_local, sum(level widths)] (split_levels restores the level list)."""
        import torch
        from . import _lib
        m = self._codecs[0]
        B, _, T = pcm.shape
        widths = m.query(T)[2]
        x = pcm.contiguous().to(torch.float32)
        if codes_all is None:
            codes_all = torch.empty((self.world * B, sum(widths)), dtype=torch.int64, device=x.device)
        m._bind_torch_stream()
        _lib.check(_lib.lib().nc_group_snac_encode_allgather_dev(self._g, x.data_ptr(), B, T, codes_all.data_ptr()))
        return codes_all, widths

    def encodec_encode_allgather(self, pcm, codes_all=None, scales_all=None):
        """Encodec, rank mode: pcm [B_local, C, T] (torch, cuda) -> (codes_all [world, B_local * n_q * sum T'_f] int64, scales_all
        [world, n_frames, B_local] float32 or None, frame_lens, n_q).  encodec_frames(codes_all[r], scales_all[r], ...) gives rank r's
        List<EncodedFrame> (Models/Encodec.cs:259-285) as views; call wait() before reading on the codec's stream."""
        import torch
        from . import _lib
        m = self._codecs[0]
        B, _, T = pcm.shape
        n_frames, nq, frame_lens, _ = m.query(T)
        per_rank = B * nq * sum(frame_lens)
        x = pcm.contiguous().to(torch.float32)
        if codes_all is None:
            codes_all = torch.empty((self.world, per_rank), dtype=torch.int64, device=x.device)
        if scales_all is None and m.config.normalize:
            scales_all = torch.empty((self.world, n_frames, B), dtype=torch.float32, device=x.device)
        m._bind_torch_stream()
        _lib.check(_lib.lib().nc_group_encodec_encode_allgather_dev(self._g, x.data_ptr(), B, T, codes_all.data_ptr(),
                                                                    scales_all.data_ptr() if scales_all is not None else None))
        return codes_all, scales_all, frame_lens, nq

    def encodec_encode_allgather_local(self, pcm_blocks):
        """Encodec, local mode, device-resident equal blocks: returns (codes_all[d] [ndev, per_rank], scales_all[d] [ndev, n_frames, B] or None,
        frame_lens, n_q) with one copy of the gathered tensors per device."""
        import ctypes as C
        import torch
        from . import _lib
        xs, nb, ptrs, T, bmax = self._local_ptr_arrays(pcm_blocks)
        m0 = self._codecs[0]
        n_frames, nq, frame_lens, _ = m0.query(T)
        per_rank = bmax * nq * sum(frame_lens)
        devs = [torch.device("cuda", m.device_index) for m in self._codecs]
        codes_all = [torch.empty((self.world, per_rank), dtype=torch.int64, device=dv) for dv in devs]
        scales_all = [torch.empty((self.world, n_frames, bmax), dtype=torch.float32, device=dv) for dv in devs] if m0.config.normalize else None
        for m in self._codecs:
            m._bind_torch_stream()
        _lib.check(_lib.lib().nc_group_encodec_encode_allgather_local_dev(
            self._g, ptrs, nb, T, (C.c_void_p * self.world)(*[t.data_ptr() for t in codes_all]),
            (C.c_void_p * self.world)(*[t.data_ptr() for t in scales_all]) if scales_all is not None else None))
        self._keep = (xs,)
        return codes_all, scales_all, frame_lens, nq

    @staticmethod
    def encodec_frames(codes_block, scales_block, B, n_q, frame_lens):
        """One rank's block of the gathered tensors -> its frames [(codes [B, n_q, T'_f], scale [B, 1] or None), ...] as views."""
        out, o = [], 0
        for f, fl in enumerate(frame_lens):
            n = B * n_q * fl
            out.append((codes_block[o:o + n].view(B, n_q, fl), None if scales_block is None else scales_block[f].view(B, 1)))
            o += n
        return out

    def wait(self):
        from . import _lib
        _lib.check(_lib.lib().nc_group_wait(self._g))

    # ---- local mode, device-resident blocks: torch CUDA tensors (one per device), asynchronous ------------------------------------------
    def _local_ptr_arrays(self, pcm_blocks):
        import ctypes as C
        import torch
        if self.rank_id >= 0 or len(pcm_blocks) != self.world:
            raise ValueError("local-mode group: one block of clips per device")
        xs = [None if x is None else x.contiguous().to(torch.float32) for x in pcm_blocks]
        for d, (x, m) in enumerate(zip(xs, self._codecs)):
            if x is not None and (not x.is_cuda or (x.device.index or 0) != m.device_index):
                raise ValueError(f"block {d} must live on device {m.device_index}")
        nb = (C.c_int32 * self.world)(*[0 if x is None else int(x.shape[0]) for x in xs])
        ptrs = (C.c_void_p * self.world)(*[None if x is None or x.shape[0] == 0 else x.data_ptr() for x in xs])
        full = [x for x in xs if x is not None and x.shape[0] > 0]
        if not full:
            raise ValueError("no clips: every block is empty")
        if any(x.dim() != 3 or tuple(x.shape[1:]) != tuple(full[0].shape[1:]) for x in full):
            raise ValueError("every block must be [B_d, C, T] with the same C and T")
        return xs, nb, ptrs, int(full[0].shape[-1]), max(nb)

    def dac_encode_allgather_local(self, pcm_blocks, n_quantizers: int = 0, codes_all=None):
        """pcm_blocks[d]: [B_d,1,T] on device d (None / empty for a device without clips).  Returns (z[d], codes_all[d], latents[d]) lists:
        codes_all[d] = device d's copy of the gathered [ndev*B_max, n_q, T'] tensor (valid on the codec's stream after wait())."""
        import ctypes as C
        import torch
        from . import _lib
        xs, nb, ptrs, T, bmax = self._local_ptr_arrays(pcm_blocks)
        m0 = self._codecs[0]
        Tz = m0.frames(T)
        nq = n_quantizers if 0 < n_quantizers <= m0.config.n_codebooks else m0.config.n_codebooks
        devs = [torch.device("cuda", m.device_index) for m in self._codecs]
        if codes_all is None:
            codes_all = [torch.empty((self.world * bmax, nq, Tz), dtype=torch.int64, device=dv) for dv in devs]
        z = [torch.empty((nb[d], m0.latent_dim, Tz), dtype=torch.float32, device=devs[d]) for d in range(self.world)]
        lat = [torch.empty((nb[d], nq * m0.config.codebook_dim, Tz), dtype=torch.float32, device=devs[d]) for d in range(self.world)]
        arr = lambda ts: (C.c_void_p * self.world)(*[t.data_ptr() if t.numel() else None for t in ts])
        for m in self._codecs:
            m._bind_torch_stream()
        _lib.check(_lib.lib().nc_group_dac_encode_allgather_local_dev(self._g, ptrs, nb, T, m0.config.sample_rate, n_quantizers,
                                                                      (C.c_void_p * self.world)(*[t.data_ptr() for t in codes_all]), arr(z), arr(lat)))
        self._keep = (xs,)                      # (the blocks handed to the asynchronous call stay referenced until the next call)
        return z, codes_all, lat

    def snac_encode_allgather_local(self, pcm_blocks, codes_all=None):
        """SNAC form: codes_all[d] [ndev*B_max, sum(level widths)] on every device; returns (codes_all, widths)."""
        import ctypes as C
        import torch
        from . import _lib
        xs, nb, ptrs, T, bmax = self._local_ptr_arrays(pcm_blocks)
        m0 = self._codecs[0]
        widths = m0.query(T)[2]
        if codes_all is None:
            codes_all = [torch.empty((self.world * bmax, sum(widths)), dtype=torch.int64, device=torch.device("cuda", m.device_index)) for m in self._codecs]
        for m in self._codecs:
            m._bind_torch_stream()
        _lib.check(_lib.lib().nc_group_snac_encode_allgather_local_dev(self._g, ptrs, nb, T, (C.c_void_p * self.world)(*[t.data_ptr() for t in codes_all])))
        self._keep = (xs,)
        return codes_all, widths

    # ---- local mode: numpy in / numpy out, synchronous ----------------------------------------------------------------------------
    def dac_encode_allgather_host(self, pcm, n_quantizers: int = 0, return_z: bool = False):
        import numpy as np
        from . import _lib
        m = self._codecs[0]
        x = np.ascontiguousarray(pcm, dtype=np.float32)
        B, _, T = x.shape
        Tz = m.frames(T)
        nq = n_quantizers if 0 < n_quantizers <= m.config.n_codebooks else m.config.n_codebooks
        codes = np.empty((B, nq, Tz), np.int64)
        z = np.empty((B, m.latent_dim, Tz), np.float32) if return_z else None
        _lib.check(_lib.lib().nc_group_dac_encode_allgather(self._g, x.ctypes.data, B, T, m.config.sample_rate, n_quantizers, codes.ctypes.data,
                                                            z.ctypes.data if z is not None else None))
        return (codes, z) if return_z else codes

    def snac_encode_allgather_host(self, pcm):
        import numpy as np
        from . import _lib
        m = self._codecs[0]
        x = np.ascontiguousarray(pcm, dtype=np.float32)
        B, _, T = x.shape
        widths = m.query(T)[2]
        codes = np.empty((B, sum(widths)), np.int64)
        _lib.check(_lib.lib().nc_group_snac_encode_allgather(self._g, x.ctypes.data, B, T, codes.ctypes.data))
        return split_levels(codes, widths)

"""GPU suite: each HIP kernel against the C oracle, bit for bit (same canonical fma chains on both sides)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from neuralcodecs_amd import ops  # noqa: E402
from oracle import c_oracle  # noqa: E402


def _rand(rng, *shape, scale=1.0):
    return (rng.standard_normal(shape) * scale).astype(np.float32)


def _alpha(rng, c):
    a = (0.5 + 1.5 * rng.random(c)).astype(np.float32)
    a[::5] = 0.0
    return a


CONV_CASES = [
    # (Cin, Cout, K, stride, pad, dil, T, B)       the DAC layer families at reduced width, plus ragged sizes
    (1, 64, 7, 1, 3, 1, 1000, 2),      # stem (Cin=1)
    (64, 64, 7, 1, 3, 1, 777, 2),      # residual unit d=1
    (96, 96, 7, 1, 9, 3, 640, 1),      # d=3, TM=3 tile
    (128, 128, 7, 1, 27, 9, 300, 2),   # d=9 halo 54, TM=4
    (40, 24, 7, 1, 27, 9, 50, 1),      # ragged channels, clip shorter than the halo
    (64, 64, 1, 1, 0, 1, 515, 2),      # 1x1
    (64, 128, 4, 2, 1, 1, 1024, 2),    # down s=2
    (32, 64, 8, 4, 2, 1, 1000, 1),     # down s=4
    (32, 64, 16, 8, 4, 1, 2048, 1),    # down s=8
    (16, 32, 10, 5, 3, 1, 995, 1),     # down s=5 (DAC 16/24 kHz)
    (1024, 8, 1, 1, 0, 1, 87, 2),      # RVQ in_proj
    (8, 1024, 1, 1, 0, 1, 87, 2),      # RVQ out_proj
    (128, 96, 3, 1, 1, 1, 87, 2),      # encoder tail k3
    (96, 1, 7, 1, 3, 1, 900, 2),       # head (Cout=1)
]


@pytest.mark.parametrize("cin,cout,k,s,p,d,T,B", CONV_CASES)
def test_conv1d_bit_exact(cin, cout, k, s, p, d, T, B):
    rng = np.random.default_rng(cin * 1000 + cout + k)
    x = _rand(rng, B, cin, T)
    w = _rand(rng, cout, cin, k, scale=1.0 / np.sqrt(cin * k))
    b = _rand(rng, cout, scale=0.1)
    want = c_oracle.conv1d(x, w, b, s, p, d)
    got = ops.conv1d(x, w, b, s, p, d)
    assert got.shape == want.shape
    assert np.array_equal(got, want), f"max abs diff {np.abs(got - want).max()}"


def test_conv1d_fused_snake_residual_bit_exact():
    """The ResidualUnit fusion: Snake on the input tile, bias, Snake for the next conv; then 1x1 + residual."""
    rng = np.random.default_rng(5)
    B, C, T, d = 2, 96, 700, 3
    x = _rand(rng, B, C, T, scale=1.5)
    a1, a2 = _alpha(rng, C), _alpha(rng, C)
    w7 = _rand(rng, C, C, 7, scale=1.0 / np.sqrt(C * 7)); b7 = _rand(rng, C, scale=0.1)
    w1 = _rand(rng, C, C, 1, scale=1.0 / np.sqrt(C)); b1 = _rand(rng, C, scale=0.1)
    h_ref = c_oracle.snake(c_oracle.conv1d(c_oracle.snake(x, a1), w7, b7, 1, 3 * d, d), a2)
    y_ref = c_oracle.conv1d(h_ref, w1, b1, residual=x)
    h = ops.conv1d(x, w7, b7, 1, 3 * d, d, alpha_in=a1, alpha_out=a2)
    assert np.array_equal(h, h_ref)
    y = ops.conv1d(h, w1, b1, residual=x)
    assert np.array_equal(y, y_ref)


@pytest.mark.parametrize("C,T,d,B", [(64, 700, 1, 2), (96, 523, 3, 1), (128, 300, 9, 2), (64, 40, 9, 1),
                                     (192, 300, 1, 2), (192, 131, 9, 1), (256, 257, 3, 2), (256, 90, 9, 1)])   # wide units: W1 streamed
def test_fused_res_unit_bit_exact(C, T, d, B):
    if C >= 192 and any(os.environ.get(k) == "1" for k in ("NC_NO_WIDE_FUSE", "NC_NO_TILE_ALTS")):
        pytest.skip("the whole-channel fused unit is switched off in this environment (tools/probe/envmatrix.sh row)")
    """Single-launch ResidualUnit (conv7 + Snake + 1x1 on the accumulators + skip) == oracle == two-launch path."""
    rng = np.random.default_rng(C + d)
    x = _rand(rng, B, C, T, scale=1.5)
    a1, a2 = _alpha(rng, C), _alpha(rng, C)
    w7 = _rand(rng, C, C, 7, scale=1.0 / np.sqrt(C * 7)); b7 = _rand(rng, C, scale=0.1)
    w1 = _rand(rng, C, C, 1, scale=1.0 / np.sqrt(C)); b1 = _rand(rng, C, scale=0.1)
    h_ref = c_oracle.snake(c_oracle.conv1d(c_oracle.snake(x, a1), w7, b7, 1, 3 * d, d), a2)
    y_ref = c_oracle.conv1d(h_ref, w1, b1, residual=x)
    y_f = ops.res_unit(x, w7, b7, a1, a2, w1, b1, dil=d, fused=True)
    y_u = ops.res_unit(x, w7, b7, a1, a2, w1, b1, dil=d, fused=False)
    assert np.array_equal(y_u, y_ref)
    assert np.array_equal(y_f, y_ref), f"max abs diff {np.abs(y_f - y_ref).max()}"


def test_conv1d_tanh_head_bit_exact():
    rng = np.random.default_rng(6)
    x = _rand(rng, 2, 96, 1000, scale=2.0)
    a = _alpha(rng, 96)
    w = _rand(rng, 1, 96, 7, scale=0.05); b = _rand(rng, 1, scale=0.1)
    want = c_oracle.tanh(c_oracle.conv1d(c_oracle.snake(x, a), w, b, 1, 3, 1))
    got = ops.conv1d(x, w, b, 1, 3, 1, alpha_in=a, tanh_out=True)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("cin,cout,s,T,B", [(64, 32, 2, 500, 2), (96, 48, 4, 300, 1), (128, 64, 8, 87, 2), (48, 24, 5, 56, 2),
                                            (1536, 768, 8, 20, 1)])
def test_conv_transpose1d_bit_exact(cin, cout, s, T, B):
    rng = np.random.default_rng(cin + s)
    k, p = 2 * s, (s + 1) // 2
    x = _rand(rng, B, cin, T)
    a = _alpha(rng, cin)
    w = _rand(rng, cin, cout, k, scale=1.0 / np.sqrt(cin * 2)); b = _rand(rng, cout, scale=0.1)
    want = c_oracle.conv_transpose1d(c_oracle.snake(x, a), w, b, s, p)
    got = ops.conv1d(x, w, b, s, p, 1, alpha_in=a, transposed=True)
    assert got.shape == want.shape
    assert np.array_equal(got, want), f"max abs diff {np.abs(got - want).max()}"


# Flattened (clip, column) tile axis (launch_conv: short rows are cut into tiles as ONE axis over all clips, a tile touching up to 4
# clips, each with its own halo in LDS).  Shapes whose rows are shorter than / not a multiple of the tile width, with enough clips that
# tiles straddle 2, 3 and 4 clips, the last tile ends inside the last clip, and the clip count is not a multiple of anything.
FLAT_CASES = [
    # (Cin, Cout, K, stride, pad, dil, T, B, transposed)
    (64, 96, 7, 1, 3, 1, 87, 9, False),       # 87-column rows: 256-wide tiles over 4 clips
    (64, 64, 7, 1, 9, 3, 87, 5, False),       # dilation 3: halo 18 per segment
    (32, 64, 7, 1, 27, 9, 150, 7, False),     # dilation 9 at 150 columns (128-wide tiles, 3 segments)
    (48, 64, 3, 1, 1, 1, 50, 11, False),      # 50-column rows (4 segments per 128-wide tile)
    (64, 128, 7, 1, 3, 1, 696, 3, False),     # 696 = 2.7 tiles of 256
    (32, 64, 16, 8, 4, 1, 696, 5, False),     # strided down-conv: 87 output columns, phase-de-interleaved window per segment
    (32, 64, 8, 4, 2, 1, 600, 6, False),      # stride 4, 150 columns
    (64, 48, 16, 8, 4, 1, 87, 6, True),       # sub-pixel up-conv: 88 input columns per clip
    (96, 32, 4, 2, 1, 1, 150, 7, True),       # stride 2
]


@pytest.mark.parametrize("cin,cout,k,s,p,d,T,B,tr", FLAT_CASES)
def test_flattened_column_axis_bit_exact(cin, cout, k, s, p, d, T, B, tr):
    rng = np.random.default_rng(cin * 31 + cout + k + T)
    x = _rand(rng, B, cin, T)
    a = _alpha(rng, cin)
    b = _rand(rng, cout, scale=0.1)
    if tr:
        w = _rand(rng, cin, cout, k, scale=1.0 / np.sqrt(cin * 2))
        want = c_oracle.conv_transpose1d(c_oracle.snake(x, a), w, b, s, p)
        got = ops.conv1d(x, w, b, s, p, 1, alpha_in=a, transposed=True)
    else:
        w = _rand(rng, cout, cin, k, scale=1.0 / np.sqrt(cin * k))
        want = c_oracle.conv1d(c_oracle.snake(x, a), w, b, s, p, d)
        got = ops.conv1d(x, w, b, s, p, d, alpha_in=a)
    assert got.shape == want.shape
    assert np.array_equal(got, want), f"max abs diff {np.abs(got - want).max()}"
    # every clip is computed independently of its neighbours in the tile: permuting the clips permutes the output
    perm = rng.permutation(B)
    got_p = ops.conv1d(x[perm], w, b, s, p, 1 if tr else d, alpha_in=a, transposed=tr)
    assert np.array_equal(got_p, want[perm])


def _random_conv_shapes(n, seed):
    """Seeded random layer shapes around the tiling decisions: short rows x many clips (flattened axis, 2-4 segments per tile), rows
    just above / below the tile widths, every kernel size the models use, strided and transposed forms, ragged channel counts."""
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n):
        kind = rng.choice(["s1", "strided", "up"])
        B = int(rng.integers(1, 12))
        if kind == "s1":
            k = int(rng.choice([1, 3, 7])); d = int(rng.choice([1, 3, 9])) if k == 7 else 1
            s, p, tr = 1, d * (k - 1) // 2, False
            T = int(rng.choice([rng.integers(33, 200), rng.integers(250, 300), rng.integers(500, 800)]))
        elif kind == "strided":
            s = int(rng.choice([2, 4, 5, 8])); k, d, tr = 2 * s, 1, False
            p = (s + 1) // 2
            T = int(rng.integers(40, 160)) * s + int(rng.integers(0, s))
        else:
            s = int(rng.choice([2, 4, 8])); k, d, tr = 2 * s, 1, True
            p = (s + 1) // 2
            T = int(rng.integers(33, 180))
        cin = int(rng.choice([8, 24, 32, 48, 64, 96, 160])); cout = int(rng.choice([16, 32, 40, 64, 96, 128, 192]))
        if tr and (cout * s) % 32:
            cout = 32
        out.append((cin, cout, k, s, p, d, T, B, bool(tr)))
    return out


@pytest.mark.parametrize("cin,cout,k,s,p,d,T,B,tr", _random_conv_shapes(36, seed=20260102))
def test_conv1d_random_shapes_bit_exact(cin, cout, k, s, p, d, T, B, tr):
    rng = np.random.default_rng(cin + 7 * cout + 13 * k + T + B)
    x = _rand(rng, B, cin, T)
    a = _alpha(rng, cin) if rng.random() < 0.5 else None
    b = _rand(rng, cout, scale=0.1)
    xin = c_oracle.snake(x, a) if a is not None else x
    if tr:
        w = _rand(rng, cin, cout, k, scale=1.0 / np.sqrt(cin * 2))
        want = c_oracle.conv_transpose1d(xin, w, b, s, p)
        got = ops.conv1d(x, w, b, s, p, 1, alpha_in=a, transposed=True)
    else:
        w = _rand(rng, cout, cin, k, scale=1.0 / np.sqrt(cin * k))
        res = _rand(rng, *c_oracle.conv1d(xin, w, b, s, p, d).shape) if rng.random() < 0.4 else None
        want = c_oracle.conv1d(xin, w, b, s, p, d, residual=res)
        got = ops.conv1d(x, w, b, s, p, d, alpha_in=a, residual=res)
    assert got.shape == want.shape
    assert np.array_equal(got, want), f"max abs diff {np.abs(got - want).max()}"


@pytest.mark.parametrize("C,res,snake", [(64, True, False), (96, True, True), (96, False, False), (192, True, False)])
def test_pointwise_streaming_variant_bit_exact(C, res, snake):
    """Long narrow rows (>= 2048 column tiles): the tile-per-workgroup pointwise kernel by default, the streaming kernel (weights resident in
    LDS, B ring across column tiles) under NC_PW_STREAM=1 -- tests/test_children_gpu.py runs these cases in that form too."""
    rng = np.random.default_rng(C + res + 2 * snake)
    B, T = 8, 66000 if C < 192 else 33100
    x = _rand(rng, B, C, T)
    w = _rand(rng, C, C, 1, scale=1.0 / np.sqrt(C)); b = _rand(rng, C, scale=0.1)
    r = _rand(rng, B, C, T) if res else None
    a = _alpha(rng, C) if snake else None
    want = c_oracle.conv1d(x, w, b, residual=r)
    if snake:
        want = c_oracle.snake(want, a)
    got = ops.conv1d(x, w, b, residual=r, alpha_out=a)
    assert np.array_equal(got, want), f"max abs diff {np.abs(got - want).max()}"


def test_flattened_column_axis_residual_and_snake_epilogues():
    """Stride-1 k=7 over 87-column rows with the residual / next-Snake epilogues (the full-tile straight-line path and edge tiles)."""
    rng = np.random.default_rng(77)
    B, C, T, d = 10, 96, 87, 3
    x = _rand(rng, B, C, T, scale=1.5)
    a1, a2 = _alpha(rng, C), _alpha(rng, C)
    w = _rand(rng, C, C, 7, scale=1.0 / np.sqrt(C * 7)); b = _rand(rng, C, scale=0.1)
    res = _rand(rng, B, C, T)
    want = c_oracle.snake(c_oracle.conv1d(c_oracle.snake(x, a1), w, b, 1, 3 * d, d, residual=res), a2)
    got = ops.conv1d(x, w, b, 1, 3 * d, d, alpha_in=a1, alpha_out=a2, residual=res)
    assert np.array_equal(got, want), f"max abs diff {np.abs(got - want).max()}"


@pytest.mark.parametrize("N,D,T,B", [(1024, 8, 87, 4), (64, 8, 7, 2), (4096, 8, 100, 1)])
def test_vq_argmin_bit_exact(N, D, T, B):
    rng = np.random.default_rng(N)
    cb = _rand(rng, N, D, scale=0.8)
    z = _rand(rng, B, D, T)
    idx_ref, st_ref, _ = c_oracle.vq_argmin(z, cb)
    idx, st = ops.vq_argmin(z, cb)
    assert np.array_equal(idx, idx_ref)
    assert np.array_equal(st, st_ref)


def test_vq_argmin_tie_break_first_index():
    cb = np.zeros((128, 8), np.float32)
    cb[:, 0] = 1.0
    cb[70, 0] = 0.0
    cb[5, 0] = 0.0      # two exact ties in different lanes/iterations: the lower index must win
    z = np.zeros((1, 8, 3), np.float32)
    idx, _ = ops.vq_argmin(z, cb)
    assert idx.tolist() == [[5, 5, 5]]


# Epilogue / tile-variant matrix: every combination of (residual, Snake-out) over shapes that select the straight-line full-tile
# epilogue, the edge-tile quad emitter, the narrow 96-column variant, the alternate row tiles (Cout >= 128 packs TM = 4/3/2), the
# pointwise kernel's modes, and row-partial tiles (Cout not a multiple of the tile height).
EPI_SHAPES = [
    # (Cin, Cout, K, pad, dil, T, B)
    (96, 96, 7, 3, 1, 512, 2),       # full 256-column tiles, TM=3
    (128, 128, 7, 9, 3, 600, 1),     # full + edge tile, TM=4
    (64, 384, 7, 3, 1, 300, 2),      # Cout 384: alternates packed, edge tile
    (256, 256, 7, 27, 9, 87, 3),     # ~90 frames: narrow variant
    (192, 192, 1, 0, 1, 512, 2),     # pointwise kernel, TM=3
    (256, 128, 1, 0, 1, 258, 2),     # pointwise, edge columns
    (64, 48, 1, 0, 1, 256, 2),       # pointwise, row-partial tile (48 rows of 64)
    (32, 40, 3, 1, 1, 333, 2),       # generic kernel, row-partial + edge
    (64, 64, 1, 0, 1, 255, 1),       # odd length: pointwise kernel not eligible -> generic K=1
]


@pytest.mark.parametrize("with_res", [False, True])
@pytest.mark.parametrize("with_snake", [False, True])
@pytest.mark.parametrize("cin,cout,k,p,d,T,B", EPI_SHAPES)
def test_conv1d_epilogue_matrix_bit_exact(cin, cout, k, p, d, T, B, with_snake, with_res):
    rng = np.random.default_rng(cin + 7 * cout + 13 * T + k)
    x = _rand(rng, B, cin, T)
    w = _rand(rng, cout, cin, k, scale=1.0 / np.sqrt(cin * k))
    b = _rand(rng, cout, scale=0.1)
    res = _rand(rng, B, cout, T) if with_res else None
    ao = _alpha(rng, cout) if with_snake else None
    want = c_oracle.conv1d(x, w, b, 1, p, d, residual=res)
    if with_snake:
        want = c_oracle.snake(want, ao)
    got = ops.conv1d(x, w, b, 1, p, d, alpha_out=ao, residual=res)
    assert np.array_equal(got, want), f"max abs diff {np.abs(got - want).max()}"


@pytest.mark.parametrize("cin,cout,s,T,B", [(256, 128, 8, 87, 2), (128, 256, 2, 90, 1)])
def test_conv_transpose_short_rows_bit_exact(cin, cout, s, T, B):
    """Up-convolutions over ~90-frame rows (the decoder's first block): narrow tiles, all phases, Snake-out."""
    rng = np.random.default_rng(cin + s)
    x = _rand(rng, B, cin, T)
    w = _rand(rng, cin, cout, 2 * s, scale=1.0 / np.sqrt(cin * 2))
    b = _rand(rng, cout, scale=0.1)
    ao = _alpha(rng, cout)
    pad = (s + 1) // 2
    want = c_oracle.snake(c_oracle.conv_transpose1d(x, w, b, s, pad, s % 2), ao)
    got = ops.conv1d(x, w, b, s, pad, 1, alpha_out=ao, transposed=True, out_pad=s % 2)
    assert np.array_equal(got, want), f"max abs diff {np.abs(got - want).max()}"


@pytest.mark.parametrize("cin,cout,s,pad,T,B,snake_out", [(96, 64, 3, 2, 200, 2, True), (64, 32, 5, 0, 150, 3, False), (128, 32, 6, 3, 77, 2, True),
                                                          (256, 128, 5, 0, 150, 4, False), (64, 32, 7, 4, 50, 1, False), (384, 192, 3, 2, 300, 2, True)])
def test_conv_transpose_subpixel_any_stride_bit_exact(cin, cout, s, pad, T, B, snake_out):
    """Sub-pixel form for strides that are not a power of two (SNAC's stride-3 block with output_padding = 1, Encodec's stride-5
    SConvTranspose1d with no padding): rows = (channel, phase) through the multiply-shift map, row tiles that do not start on a channel
    boundary, partial last row tiles, Snake of the consumer in the epilogue."""
    rng = np.random.default_rng(cin + 17 * s)
    op = s % 2 if pad else 0
    x = _rand(rng, B, cin, T)
    w = _rand(rng, cin, cout, 2 * s, scale=1.0 / np.sqrt(cin * 2)); b = _rand(rng, cout, scale=0.1)
    ao = _alpha(rng, cout) if snake_out else None
    want = c_oracle.conv_transpose1d(x, w, b, s, pad, op)
    if snake_out:
        want = c_oracle.snake(want, ao)
    got = ops.conv1d(x, w, b, s, pad, 1, alpha_out=ao, transposed=True, out_pad=op)
    assert got.shape == want.shape
    assert np.array_equal(got, want), f"max abs diff {np.abs(got - want).max()}"


@pytest.mark.parametrize("cin,cout,k,s,p,T,B", [(384, 768, 16, 8, 4, 375, 1), (192, 384, 16, 8, 4, 1500, 1), (96, 192, 8, 4, 2, 500, 2),
                                                (64, 128, 4, 2, 1, 100, 3), (128, 64, 6, 3, 2, 200, 1), (64, 96, 8, 4, 2, 77, 2),
                                                (512, 1024, 16, 8, 4, 87, 1), (256, 512, 16, 8, 0, 150, 1)])
def test_conv1d_short_rows_16x16x4_kernel_bit_exact(cin, cout, k, s, p, T, B):
    """The strided down-convolutions of one- / few-clip batches (conv_small_kernel, nc_conv_small.hip: v_mfma_f32_16x16x4_f32, 16-column
    tiles, weights streamed from a packed image): partial last column tile, partial row tile (Cout = 96), k = 6 / stride 3, zero padding
    on both sides and none at all."""
    rng = np.random.default_rng(cin * 7 + cout + k)
    x = _rand(rng, B, cin, T)
    w = _rand(rng, cout, cin, k, scale=1.0 / np.sqrt(cin * k))
    b = _rand(rng, cout, scale=0.1)
    want = c_oracle.conv1d(x, w, b, s, p, 1)
    got = ops.conv1d(x, w, b, s, p, 1)
    assert got.shape == want.shape
    assert np.array_equal(got, want), f"max abs diff {np.abs(got - want).max()}"


@pytest.mark.parametrize("cin,cout,T,B", [(256, 512, 1200, 8), (512, 1024, 696, 4), (128, 256, 640, 40)])
def test_conv1d_k16_wide_short_row_form_with_snake_out_bit_exact(cin, cout, T, B):
    """k = 16 / stride 8 layers whose template grid would not fill the chip twice take the 32-column form of the 16x16x4 kernel
    (two column tiles per A fragment); the consumer's Snake in the epilogue (DAC's last EncoderBlock, Encoder.cs:44)."""
    rng = np.random.default_rng(cin + cout + T)
    x = _rand(rng, B, cin, T)
    w = _rand(rng, cout, cin, 16, scale=1.0 / np.sqrt(cin * 16)); b = _rand(rng, cout, scale=0.1)
    ao = _alpha(rng, cout)
    want = c_oracle.snake(c_oracle.conv1d(x, w, b, 8, 4, 1), ao)
    got = ops.conv1d(x, w, b, 8, 4, 1, alpha_out=ao)
    assert got.shape == want.shape
    assert np.array_equal(got, want), f"max abs diff {np.abs(got - want).max()}"


@pytest.mark.parametrize("cin,cout,T,B,snake_out", [(512, 128, 150, 4, False), (128, 512, 150, 32, False), (1024, 1536, 87, 4, True), (64, 64, 47, 1, False)])
def test_conv1d_k7_short_row_forms_bit_exact(cin, cout, T, B, snake_out):
    _short_row_stride1(7, cin, cout, T, B, snake_out)


@pytest.mark.parametrize("cin,cout,T,B,snake_out", [(1024, 1024, 87, 4, False), (1024, 1024, 87, 32, True), (64, 128, 33, 2, False)])
def test_conv1d_k3_short_row_forms_bit_exact(cin, cout, T, B, snake_out):
    """k = 3 / stride 1 (DAC's encoder output convolution, Encoder.cs:45): the fourth lane group of a matrix-core step starts in the next
    channel row, and a step can wrap twice."""
    _short_row_stride1(3, cin, cout, T, B, snake_out)


def _short_row_stride1(k, cin, cout, T, B, snake_out):
    """k = 7 / stride 1 plain-input layers (Encodec's 512 <-> 128 convolutions around the quantizer, DAC's decoder input with the first
    DecoderBlock's Snake in the epilogue) on the 16x16x4 kernel: 16 channels per block, 16- and 32-column forms."""
    rng = np.random.default_rng(cin + cout + T)
    x = _rand(rng, B, cin, T)
    w = _rand(rng, cout, cin, k, scale=1.0 / np.sqrt(cin * k)); b = _rand(rng, cout, scale=0.1)
    ao = _alpha(rng, cout) if snake_out else None
    want = c_oracle.conv1d(x, w, b, 1, k // 2, 1)
    if snake_out:
        want = c_oracle.snake(want, ao)
    got = ops.conv1d(x, w, b, 1, k // 2, 1, alpha_out=ao)
    assert got.shape == want.shape
    assert np.array_equal(got, want), f"max abs diff {np.abs(got - want).max()}"


@pytest.mark.parametrize("cin,cout,T,B", [(512, 2048, 1408, 1), (512, 2048, 576, 1), (256, 1024, 100, 3)])
def test_conv1d_pointwise_short_row_form_bit_exact(cin, cout, T, B):
    """Wide pointwise GEMMs over few columns (the chunked LSTM input projections, 512 -> 2048 over 44 steps x 32 rows) on the 16x16x4
    kernel: 64 channels per block, every reduction index its own channel row."""
    rng = np.random.default_rng(cin + cout + T)
    x = _rand(rng, B, cin, T)
    w = _rand(rng, cout, cin, 1, scale=1.0 / np.sqrt(cin)); b = _rand(rng, cout, scale=0.1)
    want = c_oracle.conv1d(x, w, b, 1, 0, 1)
    got = ops.conv1d(x, w, b, 1, 0, 1)
    assert got.shape == want.shape
    assert np.array_equal(got, want), f"max abs diff {np.abs(got - want).max()}"


# ---- Encodec Euclidean RVQ: the all-stages matrix-core launch against the per-stage kernels and the C oracle (round 6: its codebook ring runs
#      on across passes AND stages; N = 512 is one pass per wave, N = 1024 two; frame counts that leave the last workgroup ragged) ----
@pytest.mark.parametrize("N,n_q,B,T", [(1024, 8, 3, 150), (512, 5, 2, 77), (1024, 3, 1, 31), (512, 1, 1, 4), (1024, 2, 5, 32)])
def test_euclid_rvq_matrix_core_form_equals_stagewise_and_oracle(N, n_q, B, T):
    rng = np.random.default_rng(1000 * N + T)
    ze = rng.standard_normal((B, 128, T)).astype(np.float32)
    books = rng.standard_normal((n_q, N, 128)).astype(np.float32)
    c1, _ = ops.euclid_rvq(ze, books, form=1)
    c0, r0 = ops.euclid_rvq(ze, books, form=0)
    assert np.array_equal(c0, c1)
    r = ze.copy()
    for q in range(n_q):                                       # ResidualVectorQuantizer.cs:139-156 over EuclideanCodebook.cs:155-182
        idx = c_oracle.vq_argmin(r, books[q])[0]
        assert np.array_equal(c0[:, q, :], idx), f"stage {q}"
        r = r - books[q][idx].transpose(0, 2, 1)
    assert np.array_equal(r0, r)


def test_euclid_rvq_matrix_core_form_refuses_large_codebooks():
    """The all-stages launch keeps |c|^2 of a stage in 4 KB of LDS: codebooks above 1024 entries take the per-stage kernels (form 0 still serves them)."""
    rng = np.random.default_rng(5)
    ze = rng.standard_normal((1, 128, 9)).astype(np.float32)
    books = rng.standard_normal((2, 2048, 128)).astype(np.float32)
    with pytest.raises(Exception, match="512 or 1024"):
        ops.euclid_rvq(ze, books, form=1)
    c0, _ = ops.euclid_rvq(ze, books, form=0)
    r = ze.copy()
    for q in range(2):
        idx = c_oracle.vq_argmin(r, books[q])[0]
        assert np.array_equal(c0[:, q, :], idx)
        r = r - books[q][idx].transpose(0, 2, 1)

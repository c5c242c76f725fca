"""The pipelined pre-split conv engine (csrc/conv_x3.hip: LDS-DMA ring, counted vmcnt, one barrier per slab) against an
f64 convolution computed outside the kernel, for every tile variant: forward with the fused epilogue options, BN partial
sums, x3 output, dgrad (stride 1 and phase-decomposed stride 2), ragged M / K / C edges and dilations whose taps fall
entirely into the padding."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

# (N, C, H, W, K, ksize, stride, pad, dil)
CASES = [(8, 256, 33, 33, 256, 3, 1, 1, 1),      # layer3 3x3 at full size: M = 8712, 138 tiles of 128 x 128 on 256 CUs
         (2, 304, 33, 33, 256, 3, 1, 1, 1),      # C = 9.5 slabs (zero-padded tail), decoder shape at 33^2
         (2, 256, 33, 33, 256, 3, 1, 6, 6),      # dilation 6 on a 33-map: border tiles skip taps
         (1, 512, 33, 33, 256, 3, 1, 18, 18),    # dilation 18: most taps fall in the padding
         (3, 1024, 17, 17, 256, 1, 1, 0, 1),     # 1x1, M = 867 (ragged last tile)
         (2, 64, 31, 29, 72, 1, 1, 0, 1),        # K = 72: ragged N tile, K % 32 != 0
         (2, 128, 35, 35, 128, 3, 2, 1, 1),      # stride 2
         (2, 48, 19, 23, 40, 3, 1, 1, 1),        # C = 48 (1.5 slabs), K = 40
         (1, 96, 9, 9, 320, 1, 1, 2, 1)]         # 1x1 over a zero-padded border (MobileNet fixed_padding form)
# dass_x3_force_tile codes: tile + 10 * mode + 100 * shape (mode 1 = one tile per workgroup, 2 = stream-K slab ranges, 0 = auto;
# shape 0 = the default MFMA shape (16x16x32), 1 = 32x32x16, 2 = 16x16x32)
TILES = [11, 12, 13, 14, 15, 16, 17, 21, 22, 23, 24, 0, 111, 113, 114, 115, 121, 124, 100]


@pytest.fixture(autouse=True)
def _engine():
    from dass_hip import ops
    from dass_hip._lib import lib

    mode, dt = ops.f32_mma(), ops.compute_dtype()
    ops.set_compute_dtype(torch.float32)
    ops.set_f32_mma("bf16x6")
    yield
    lib.dass_x3_force_tile(0)
    ops.set_f32_mma(mode)
    ops.set_compute_dtype(dt)


def _inputs(case):
    n, c, h, wd, k, ks, stride, pad, dil = case
    g = torch.Generator().manual_seed(c * 7 + k)
    x = torch.randn(n, c, h, wd, generator=g)
    w = torch.randn(k, c, ks, ks, generator=g) * (2.0 / (c * ks * ks)) ** 0.5
    return x, w


def _x3_of(ops, t_nchw):
    n, c, h, w = t_nchw.shape
    rows = t_nchw.permute(0, 2, 3, 1).contiguous().cuda()
    return ops.split3_rows(rows, c, n * h * w, c)


def _decode_x3(buf, rows, c):
    """x3 bytes -> f32 [rows, c] (x0 + x1 + x2), and the raw zero row"""
    cc = (c + 31) // 32
    v = buf[:-16].view(torch.bfloat16).view(rows + 1, cc, 3, 32).float()   # (the last 16 B are the operand's trailer)
    full = (v[:, :, 0] + v[:, :, 1]) + v[:, :, 2]
    return full[:rows].reshape(rows, cc * 32)[:, :c], v[rows]


def _rel(a, ref):
    return (a.double().cpu() - ref).norm().item() / max(ref.norm().item(), 1e-30)


def test_split3_rows_is_exact():
    from dass_hip import ops

    g = torch.Generator().manual_seed(3)
    x = torch.randn(37, 72, generator=g) * torch.logspace(-6, 4, 72)[None]
    x[0, :4] = torch.tensor([0.0, -0.0, 1e-30, 65504.0])
    xr = x.cuda()
    buf = ops.split3_rows(xr, 72, 37, 72)
    back, zero = _decode_x3(buf, 37, 72)
    assert torch.equal(back.cpu(), x), (back.cpu() - x).abs().max()     # three bf16 parts carry all 24 significand bits
    assert float(zero.abs().max()) == 0.0
    cc = 3
    v = buf[:-16].view(torch.bfloat16).view(38, cc, 3, 32)
    assert float(v[:37, 2, :, 8:].float().abs().max()) == 0.0           # channels 72..95 of the last slab are zero
    mask = (torch.rand(1, 72, generator=g) > 0.5).float() * 2.0
    buf2 = ops.split3_rows(xr, 72, 37, 72, nc_scale=mask.cuda(), rows_per_image=37)
    assert torch.equal(_decode_x3(buf2, 37, 72)[0].cpu(), x * mask)


@pytest.mark.parametrize("tile", TILES)
@pytest.mark.parametrize("case", CASES)
def test_x3_forward_vs_f64(case, tile):
    from dass_hip import ops
    from dass_hip._lib import lib

    lib.dass_x3_force_tile(tile)
    n, c, h, wd, k, ks, stride, pad, dil = case
    x, w = _inputs(case)
    oh, ow = ops.conv_out_size(h, ks, stride, pad, dil), ops.conv_out_size(wd, ks, stride, pad, dil)
    x3 = _x3_of(ops, x)
    w3 = ops.prepare_conv_weight(w.permute(0, 2, 3, 1).contiguous().cuda())
    dims = (n, h, wd, c, oh, ow, k, ks, ks, stride, pad, dil)
    ref = F.conv2d(x.double(), w.double(), None, stride, pad, dil).permute(0, 2, 3, 1)
    y = torch.full((n, oh, ow, k), float("nan"), device="cuda")
    ops.conv_x3_launch(x3, w3, y, k, dims)
    assert _rel(y, ref) <= 2e-6, (case, tile, _rel(y, ref))
    assert (y.double().cpu() - ref).abs().max().item() <= 2e-5 * ref.abs().max().item()
    # the classic engine accumulates the same products in the same order: results agree to the last bit or two
    y_old = torch.empty_like(y)
    xr = x.permute(0, 2, 3, 1).contiguous().cuda()
    if c % 4 == 0:
        ops.conv_launch(xr, c, w3, y_old, k, dims)
        assert (y - y_old).abs().max().item() <= 1e-5 * ref.abs().max().item()


@pytest.mark.parametrize("tile", [11, 12, 14, 21, 22, 23, 24])
def test_x3_epilogue_stats_and_x3_output(tile):
    from dass_hip import ops
    from dass_hip._lib import lib

    lib.dass_x3_force_tile(tile)
    case = (4, 96, 41, 39, 136, 3, 1, 2, 2)   # 6396 output rows: enough slabs for real stream-K ranges
    n, c, h, wd, k, ks, stride, pad, dil = case
    x, w = _inputs(case)
    g = torch.Generator().manual_seed(11)
    scale, shift = torch.rand(k, generator=g) + 0.5, torch.randn(k, generator=g)
    oh, ow = ops.conv_out_size(h, ks, stride, pad, dil), ops.conv_out_size(wd, ks, stride, pad, dil)
    res = torch.randn(n, oh, ow, k, generator=g)
    x3, w3 = _x3_of(ops, x), ops.prepare_conv_weight(w.permute(0, 2, 3, 1).contiguous().cuda())
    dims = (n, h, wd, c, oh, ow, k, ks, ks, stride, pad, dil)
    raw = F.conv2d(x.double(), w.double(), None, stride, pad, dil).permute(0, 2, 3, 1)
    ref = torch.relu(raw * scale.double() + shift.double() + res.double())
    m = n * oh * ow
    y = torch.empty((n, oh, ow, k), device="cuda")
    y3 = ops.x3_alloc(m, k, "cuda")
    y3.fill_(0x7f)
    ops.conv_x3_launch(x3, w3, y, k, dims, y3=y3, scale=scale.cuda(), shift=shift.cuda(), residual=res.cuda(), ldr=k, act=ops.ACT_RELU)
    assert _rel(y, ref) <= 2e-6
    back, zero = _decode_x3(y3, m, k)
    assert torch.equal(back.reshape(n, oh, ow, k), y), "the x3 rows are the exact split of the f32 result"
    assert float(zero.abs().max()) == 0.0
    # the x3 result feeds the next conv directly: y3-only output, 1x1 consumer
    w2 = torch.randn(64, k, 1, 1, generator=g) * 0.1
    y2 = torch.empty((n, oh, ow, 64), device="cuda")
    ops.conv_x3_launch(y3, ops.prepare_conv_weight(w2.permute(0, 2, 3, 1).contiguous().cuda()), y2, 64, (n, oh, ow, k, oh, ow, 64, 1, 1, 1, 0, 1))
    ref2 = F.conv2d(y.permute(0, 3, 1, 2).double().cpu(), w2.double()).permute(0, 2, 3, 1)
    assert _rel(y2, ref2) <= 2e-6
    # BatchNorm partial sums of the raw output
    rows = lib.dass_conv2d_igemm_stats_rows(m)
    part = torch.zeros((rows, 2, k), device="cuda")
    yr = torch.empty((n, oh, ow, k), device="cuda")
    nrows = ops.conv_x3_launch(x3, w3, yr, k, dims, stats=part)
    assert 1 <= nrows <= rows
    sums = part[:nrows].double().sum(0).cpu()
    flat = raw.reshape(m, k)
    assert (sums[0] - flat.sum(0)).abs().max().item() <= 1e-4 * flat.abs().sum(0).max().item()
    assert (sums[1] - (flat * flat).sum(0)).abs().max().item() <= 1e-4 * (flat * flat).sum(0).max().item()


@pytest.mark.parametrize("tile", TILES)
@pytest.mark.parametrize("case", [(2, 64, 33, 33, 96, 3, 1, 1, 1), (2, 128, 35, 35, 128, 3, 2, 1, 1), (2, 256, 17, 17, 512, 1, 2, 0, 1),
                                  (1, 64, 33, 33, 64, 3, 1, 4, 4)])
def test_x3_dgrad_vs_f64(case, tile):
    """input gradient = the same kernel over dy (x3) and the transposed weight operand; stride 2 is phase-decomposed"""
    from dass_hip import ops
    from dass_hip._lib import lib

    lib.dass_x3_force_tile(tile)
    n, c, h, wd, k, ks, stride, pad, dil = case
    x, w = _inputs(case)
    oh, ow = ops.conv_out_size(h, ks, stride, pad, dil), ops.conv_out_size(wd, ks, stride, pad, dil)
    g = torch.Generator().manual_seed(5)
    dy = torch.randn(n, k, oh, ow, generator=g)
    xd = x.double().requires_grad_(True)
    F.conv2d(xd, w.double(), None, stride, pad, dil).backward(dy.double())
    ref = xd.grad.permute(0, 2, 3, 1)
    dy3 = _x3_of(ops, dy)
    wt3 = ops.prepare_conv_weight(w.permute(0, 2, 3, 1).contiguous().cuda(), mode=1)
    dx = torch.full((n, h, wd, c), float("nan"), device="cuda")
    pad_t = dil * (ks - 1) - pad
    ops.conv_x3_launch(dy3, wt3, dx, c, (n, oh, ow, k, h, wd, c, ks, ks, 1, pad_t, dil), ustride=stride)
    assert _rel(dx, ref) <= 2e-6, (case, tile, _rel(dx, ref))


def test_x3_stream_k_is_deterministic_and_matches_one_tile_per_workgroup():
    """the fix-up pass adds the partial slabs of a split tile in workgroup order: two runs agree bit for bit, and the
    result equals the one-tile-per-workgroup schedule to the last bit or two (same products, slab sums re-associated)"""
    from dass_hip import ops
    from dass_hip._lib import lib

    case = (8, 256, 33, 33, 256, 3, 1, 1, 1)
    n, c, h, wd, k, ks, stride, pad, dil = case
    x, w = _inputs(case)
    x3, w3 = _x3_of(ops, x), ops.prepare_conv_weight(w.permute(0, 2, 3, 1).contiguous().cuda())
    dims = (n, h, wd, c, h, wd, k, ks, ks, stride, pad, dil)
    outs = []
    for code in (22, 22, 12, 21, 24):
        lib.dass_x3_force_tile(code)
        y = torch.full((n, h, wd, k), float("nan"), device="cuda")
        ops.conv_x3_launch(x3, w3, y, k, dims)
        outs.append(y)
    assert torch.equal(outs[0], outs[1])
    for other in outs[2:]:
        assert (outs[0] - other).abs().max().item() <= 2e-6 * outs[0].abs().max().item()


@pytest.mark.parametrize("case", [(2, 64, 33, 33, 96, 3, 1, 1, 1), (2, 128, 35, 35, 128, 3, 2, 1, 1), (3, 256, 17, 17, 512, 1, 1, 0, 1),
                                  (1, 72, 33, 33, 40, 3, 1, 4, 4), (8, 256, 33, 33, 256, 3, 1, 1, 1), (2, 304, 29, 31, 256, 3, 1, 1, 1),
                                  (2, 96, 9, 9, 320, 1, 1, 2, 1)])
def test_x3_wgrad_vs_f64(case):
    """weight gradient from pre-split operands (csrc/wgrad_x3.hip: transposed LDS reads of DMA-filled [pixel][channel]
    pieces) against an f64 autograd gradient; pixel splits accumulate with f32 atomics, hence the 4e-6 bound"""
    from dass_hip import ops
    from dass_hip._lib import check, lib

    n, c, h, wd, k, ks, stride, pad, dil = case
    x, w = _inputs(case)
    oh, ow = ops.conv_out_size(h, ks, stride, pad, dil), ops.conv_out_size(wd, ks, stride, pad, dil)
    g = torch.Generator().manual_seed(9)
    dy = torch.randn(n, k, oh, ow, generator=g)
    wd64 = w.double().requires_grad_(True)
    F.conv2d(x.double(), wd64, None, stride, pad, dil).backward(dy.double())
    ref = wd64.grad.permute(0, 2, 3, 1)  # [K][R][S][C]
    x3, dy3 = _x3_of(ops, x), _x3_of(ops, dy)
    dw = torch.full((k, ks, ks, c), float("nan"), device="cuda")
    # K, C > 64 -> the 128 x 128 tile (8 waves), otherwise 64 x 64 (4 waves): the cases cover both
    check(lib.dass_conv2d_wgrad_x3(ops._p(x3), ops._p(dy3), ops._p(dw), n, h, wd, c, oh, ow, k, ks, ks, stride, pad, dil, 1, ops._stream()),
          "dass_conv2d_wgrad_x3")
    assert _rel(dw, ref) <= 4e-6, (case, _rel(dw, ref))
    # accumulate form: a second call adds the same gradient again
    check(lib.dass_conv2d_wgrad_x3(ops._p(x3), ops._p(dy3), ops._p(dw), n, h, wd, c, oh, ow, k, ks, ks, stride, pad, dil, 0, ops._stream()),
          "dass_conv2d_wgrad_x3")
    assert _rel(dw, 2 * ref) <= 4e-6


@pytest.mark.parametrize("with_stats", [False, True])
@pytest.mark.parametrize("code", [24, 26, 21, 11])
def test_x3_stream_k_hybrid_rounds_plus_remainder(code, with_stats):
    """more output tiles than resident workgroups: whole rounds run one tile per workgroup, only the remainder is cut into
    slab ranges and fixed up.  (2, 64, 129, 129) -> 128 channels: 1042 tiles of 64 x 64 on 768 / 512 slots (codes 24 / 26),
    131 tiles of 256 x 128 (pure stream-K, code 21).  The x3 output rides along; with BN partial sums requested every
    tile of the stream-K region is fixed up by one block (no whole-round part)."""
    from dass_hip import ops
    from dass_hip._lib import lib

    case = (2, 64, 129, 129, 128, 3, 1, 1, 1)
    n, c, h, wd, k, ks, stride, pad, dil = case
    x, w = _inputs(case)
    x3, w3 = _x3_of(ops, x), ops.prepare_conv_weight(w.permute(0, 2, 3, 1).contiguous().cuda())
    dims = (n, h, wd, c, h, wd, k, ks, ks, stride, pad, dil)
    ref = F.conv2d(x.double().cuda(), w.double().cuda(), None, stride, pad, dil).permute(0, 2, 3, 1)
    m = n * h * wd
    outs = []
    for _ in range(2):
        lib.dass_x3_force_tile(code)
        y = torch.full((n, h, wd, k), float("nan"), device="cuda")
        y3 = ops.x3_alloc(m, k, "cuda")
        stats = torch.full((lib.dass_conv2d_igemm_stats_rows(m), 2, k), float("nan"), device="cuda")
        rows = ops.conv_x3_launch(x3, w3, y, k, dims, y3=y3, stats=stats if with_stats else None)
        outs.append((y, stats[:rows].clone(), y3))
    y, st, y3 = outs[0]
    assert torch.equal(y, outs[1][0]) and torch.equal(st, outs[1][1])
    assert (y.double() - ref).abs().max().item() <= 2e-5 * ref.abs().max().item()
    assert _rel(y, ref.cpu()) <= 2e-6
    flat = ref.reshape(m, k)
    if with_stats:
        assert (st[:, 0].double().sum(0) - flat.sum(0)).abs().max().item() <= 1e-4 * flat.abs().sum(0).max().item()
        assert (st[:, 1].double().sum(0) - (flat * flat).sum(0)).abs().max().item() <= 1e-5 * (flat * flat).sum(0).max().item()
    dec, zero = _decode_x3(y3, m, k)
    assert torch.equal(dec, y.reshape(m, k)) and not zero.any()


@pytest.mark.parametrize("code", [0, 11, 14, 21, 24])
def test_x3_per_image_dropout_sparse_conv(code):
    """dass_conv2d_x3_per_image (+ dass_dropout_compact / dass_split3_rows_packed / dass_w3_pack_per_image): the conv of a
    Dropout2d-masked input with the dropped channels skipped, against the f64 conv of the masked tensor.  Images with all,
    half, three and none of the channels kept; 17 x 19 maps so image boundaries fall inside what would be one tile."""
    from dass_hip import ops
    from dass_hip._lib import lib

    n, c, h, wd, k = 5, 256, 17, 19, 96
    g = torch.Generator().manual_seed(5)
    x = torch.randn(n, c, h, wd, generator=g)
    w = torch.randn(k, c, 3, 3, generator=g) * (2.0 / (c * 9)) ** 0.5
    mask = (torch.rand(n, c, generator=g) < 0.5).float() * 2.0
    mask[0] = 2.0
    mask[2] = 0.0
    mask[2, [7, 100, 255]] = 2.0
    mask[3] = 0.0
    res = torch.randn(n, h, wd, k, generator=g)
    scale, shift = torch.rand(k, generator=g) + 0.5, torch.randn(k, generator=g)
    ref = F.conv2d((x * mask[:, :, None, None]).double(), w.double(), None, 1, 1, 1).permute(0, 2, 3, 1)
    ref = torch.relu(ref * scale.double() + shift.double() + res.double())

    lib.dass_x3_force_tile(code)
    xr = x.permute(0, 2, 3, 1).contiguous().cuda()
    mk = mask.cuda()
    order, lim = ops.dropout_pack(mk)
    kept = [int((mask[i] != 0).sum()) for i in range(n)]
    assert lim.tolist() == [(v + 31) // 32 for v in kept]
    for i in range(n):
        assert order[i, :kept[i]].tolist() == torch.nonzero(mask[i]).flatten().tolist() and (order[i, kept[i]:] == -1).all()
    m = n * h * wd
    x3 = ops.split3_rows_packed(xr, c, m, c, mk, order, lim, h * wd)
    w3 = ops.prepare_conv_weight(w.permute(0, 2, 3, 1).contiguous().cuda())
    w3n = ops.w3_pack_per_image(w3, k * 9, c, order, lim)
    y = torch.full((n, h, wd, k), float("nan"), device="cuda")
    y3 = ops.x3_alloc(m, k, "cuda")
    ops.conv_x3_per_image_launch(x3, w3n, lim, y, k, (n, h, wd, c, h, wd, k, 3, 3, 1, 1, 1), y3=y3, scale=scale.cuda(),
                                 shift=shift.cuda(), residual=res.cuda(), ldr=k, act=ops.ACT_RELU)
    assert (y.double().cpu() - ref).abs().max().item() <= 2e-5 * ref.abs().max().item(), code
    assert _rel(y, ref) <= 2e-6
    dec, zero = _decode_x3(y3, m, k)
    assert torch.equal(dec, y.reshape(m, k)) and not zero.any()
    # the masked dense conv is the same sum in another order
    xd3 = ops.split3_rows(xr, c, m, c, nc_scale=mk, rows_per_image=h * wd)
    yd = torch.empty_like(y)
    ops.conv_x3_launch(xd3, w3, yd, k, (n, h, wd, c, h, wd, k, 3, 3, 1, 1, 1), scale=scale.cuda(), shift=shift.cuda(),
                       residual=res.cuda(), ldr=k, act=ops.ACT_RELU)
    assert (y - yd).abs().max().item() <= 1e-5 * ref.abs().max().item()

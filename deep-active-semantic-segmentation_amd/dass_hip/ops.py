"""Host side of the HIP hot path: torch tensors in, libdass_hip launches on torch's current stream.

PyTorch is plumbing here (device memory, streams, autograd bookkeeping); every arithmetic pass over an
activation is a libdass_hip kernel.  Activations are 4-D torch tensors with NCHW *shape* and NHWC
*memory* (channels_last), in the compute dtype selected by set_compute_dtype().

The autograd Functions mirror what F.conv2d / F.batch_norm / F.relu / F.interpolate / F.max_pool2d /
F.cross_entropy do at the reference's call sites (cited on each class).
"""
import ctypes
import math
import os
import weakref

import torch

from ._lib import check, lib

F32, BF16, F32X3, F32X6, F16X3, BF16X1 = 0, 1, 2, 3, 4, 5
ACT_NONE, ACT_RELU, ACT_RELU6 = 0, 1, 2

_state = {"dtype": torch.float32, "f32_mma": os.environ.get("DASS_F32_MMA", "f16x3"),
          "x3": os.environ.get("DASS_X3", "select"), "mc_sparse": os.environ.get("DASS_MC_SPARSE", "1") != "0"}
assert _state["x3"] in ("off", "infer", "select", "all"), "DASS_X3 must be off, infer, select or all"
assert _state["f32_mma"] in ("f32", "bf16x3", "bf16x6", "f16x3", "bf16x1"), "DASS_F32_MMA must be f32, bf16x3, bf16x6, f16x3 or bf16x1"
_X3_ENGINES = ("bf16x6", "f16x3", "bf16x1")  # engines that can run the pre-split (x3) kernels
_state["x3_f16"] = os.environ.get("DASS_X3_F16", "all")  # where the pre-split kernels run under the f16x3 engine
assert _state["x3_f16"] in ("off", "infer", "select", "all")


def set_compute_dtype(dtype):
    """torch.float32 = parity mode (exact f32 MFMA); torch.bfloat16 = bf16 storage, f32 accumulate."""
    assert dtype in (torch.float32, torch.bfloat16)
    _state["dtype"] = dtype


def compute_dtype():
    return _state["dtype"]


def set_f32_mma(mode):
    """how the dense convs multiply f32 tensors (tensors, BN, loss and every other kernel stay f32 in all three):
      "bf16x6" each operand is split exactly into three bf16 parts (x = x0+x1+x2 to 2^-26) and the six
               products of order <= 2^-18 are accumulated in f32 on v_mfma_f32_32x32x16_bf16 (DASS_F32X6): the f32
               product to below f32 rounding at 3/8 of the matrix-pipe time -- same parity bars as "f32";
      "f32"    v_mfma_f32_32x32x2_f32, the plain f32 fma chain;
      "bf16x3" two parts, three products (DASS_F32X3): 17-bit products (~4.5e-6 per conv), 3/16 of the pipe time;
               does NOT meet the 1e-3 logit bar on MobileNet -- a fast training mode, not a parity mode;
      "f16x3"  (default) the pre-split kernels in their two-part mode (dass_set_x3_parts(2)): every operand tensor is scaled by a power of
               two (from a guaranteed bound of its max |x|) and split into TWO f16 parts -- 23 significant bits -- and three
               products per pair run on the f16 MFMA pipe: the f32 product to 2^-22, half the matrix work of "bf16x6"; same
               parity bars (include/dass_hip.h "dass_set_x3_parts").  Layers the pre-split kernels do not take (<= 32 output
               channels, stems, depthwise) run the classic kernels exactly as under "bf16x6".
      "bf16x1" a PERF engine, not a parity mode (round 4): the same pre-split kernels with ONE bf16 part per element and one
               product per pair (dass_set_x3_parts(1)) -- what autocast-bf16 multiplies -- on f32 tensors: BN, loss, gradients
               and master weights stay f32, every round-3/4 launch structure (grouped weight gradients, fused BN sums, whole-
               tile kernels, rows-only outputs, batched MC tail) carries over.  A third of f16x3's matrix work, half its operand
               bytes.  tests/test_bf16_gpu.py measures its deviation from the parity mode.
    Initial value from the environment variable DASS_F32_MMA."""
    assert mode in ("f32", "bf16x3", "bf16x6", "f16x3", "bf16x1")
    _state["f32_mma"] = mode
    check(lib.dass_set_x3_parts({"f16x3": 2, "bf16x1": 1}.get(mode, 3)), "dass_set_x3_parts")


def x3_parts():
    """parts per element of the x3 operand format in force (3 = bf16 triple, 2 = scaled f16 pair, 1 = one bf16: perf engine)"""
    return {"f16x3": 2, "bf16x1": 1}.get(_state["f32_mma"], 3)


def _x3_mode():
    """effective placement of the pre-split kernels: DASS_X3 under "bf16x6", DASS_X3_F16 (default "all") under "f16x3" -- with
    4 B instead of 6 B per split element and half the products, the all-layers form pays in training as well"""
    return _state["x3_f16"] if _state["f32_mma"] in ("f16x3", "bf16x1") else _state["x3"]


def f32_mma():
    return _state["f32_mma"]


def set_x3_pipeline(mode):
    """bf16x6 engine only -- where the dense convs run the pipelined pre-split kernels (csrc/conv_x3.hip, wgrad_x3.hip:
    activations converted ONCE to three bf16 parts by the producing pass, LDS-DMA ring, no conversion in the MFMA loop)
    instead of the classic kernel that converts inside its loop (csrc/conv_igemm.hip):
      "infer"  forward passes without autograd (pool scoring, validation): measured +8..25 % there;
      "select" (default, DASS_X3) "infer" + in training the forward and input-gradient launches of the 3x3 convs that
               reduce over >= 256 channels (layer-3/4 conv2, ASPP branches, decoder): there the one conversion pass
               (dass_split3_rows; the gradient's split rows come out of the BN-backward pass) costs a few percent of
               the conv it speeds up by 15..40 %; every other layer and all weight gradients stay on the classic kernels;
      "all"    every dense conv in training too (forward, input and weight gradients; the BN passes then also write
               split rows): measured no faster than the classic engine on the R101 train step;
      "off"    never.
    Same six products in the same order either way: results agree to the last bit or two."""
    mode = {True: "all", False: "off"}.get(mode, mode)
    assert mode in ("off", "infer", "select", "all")
    _state["x3_f16" if _state["f32_mma"] in ("f16x3", "bf16x1") else "x3"] = mode


def x3_mode():
    """the placement set_x3_pipeline() would have to be given to restore the current state"""
    return _x3_mode()


def set_mc_sparse(on):
    """MC-dropout tail (pre-split engine): True (default, DASS_MC_SPARSE) skips the input channels the ASPP Dropout2d mask
    zeroes instead of multiplying the zeros (dass_conv2d_x3_per_image); False runs the masked dense conv.  Both are the
    same sum of the surviving products, accumulated in a different order."""
    _state["mc_sparse"] = bool(on)


def mc_sparse():
    return _state.get("mc_sparse", True)


_mc_side = {}


def mc_side_streams(dev):
    """the extra HIP streams the independent stochastic passes of DeepLab.mc_dropout_votes are dealt over
    (DASS_MC_STREAMS = total number of streams, default 2; 1: none)"""
    n = int(os.environ.get("DASS_MC_STREAMS", "2"))
    if n <= 1 or torch.cuda.is_current_stream_capturing():
        return []
    key = (dev.index if dev.index is not None else torch.cuda.current_device())
    sts = _mc_side.setdefault(key, [])
    while len(sts) < n - 1:
        sts.append(torch.cuda.Stream(device=dev))
    return sts[:n - 1]


def extra_streams(dev, n):
    """up to n more HIP streams on `dev` for independent batches (core_set._features); none when DASS_MC_PIPELINE=0"""
    if n <= 0 or os.environ.get("DASS_MC_PIPELINE", "1") != "1" or torch.cuda.is_current_stream_capturing():
        return []
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    key = ("lanes", idx)
    sts = _mc_side.setdefault(key, [])
    if len(sts) < n and os.environ.get("DASS_LANES_REUSE", "1") == "1":
        # side streams this process already has (the MC passes' streams, the weight gradients' side stream) are idle while batches are
        # dealt over lanes, and which hardware queue the runtime gives a NEW stream depends on how many were created before it: a lane
        # that lands on the caller's queue runs nothing concurrently (core-set features 1390 vs 1120 pool img/s)
        for k2 in (idx, ("prefix", idx), ("wgrad", idx)):
            v = _mc_side.get(k2)
            for st in (v if isinstance(v, list) else [v] if v is not None else []):
                if len(sts) < n and all(st is not t for t in sts):
                    sts.append(st)
    while len(sts) < n:
        sts.append(torch.cuda.Stream(device=dev))
    return sts[:n]


def mc_prefix_stream(dev):
    """the HIP stream on which active_selection.mc_dropout runs the deterministic prefix of the NEXT batch under the stochastic
    passes of the current one (DASS_MC_PIPELINE=0: none, batches strictly one after the other)"""
    if os.environ.get("DASS_MC_PIPELINE", "1") != "1" or torch.cuda.is_current_stream_capturing():
        return None
    key = ("prefix", dev.index if dev.index is not None else torch.cuda.current_device())
    st = _mc_side.get(key)
    if st is None:
        st = _mc_side[key] = torch.cuda.Stream(device=dev)
    return st


def set_deterministic(on):
    """True: conv weight gradients are summed by ONE workgroup per tile in a fixed order (no cross-workgroup f32 atomics):
    bit-reproducible training steps, slower on layers with few output tiles.  Default False (DASS_DETERMINISTIC=1 to start on)."""
    check(lib.dass_set_deterministic(1 if on else 0), "dass_set_deterministic")


if os.environ.get("DASS_DETERMINISTIC", "0") == "1":
    lib.dass_set_deterministic(1)
if _state["f32_mma"] in ("f16x3", "bf16x1"):
    lib.dass_set_x3_parts(2 if _state["f32_mma"] == "f16x3" else 1)


def set_rows_only(on):
    """DASS_ROWS_ONLY: may a layer whose only reader is a pre-split conv (conv_bn_act(sole_consumer=True)) skip the f32 copy of its output"""
    global _ROWS_ONLY
    _ROWS_ONLY = bool(on)


def x3_pipeline(training=False):
    """is the pre-split engine on for an inference call site (training=False) / for a call that records autograd"""
    if _state["f32_mma"] not in _X3_ENGINES:
        return False
    mode = _x3_mode()
    return mode == "all" or (mode in ("infer", "select") and not training)


def _x3_train_layer(taps, red_channels):
    """training launches that go to the pre-split engine: all of them ("all") or the long 3x3 reductions ("select")"""
    if _state["f32_mma"] not in _X3_ENGINES:
        return False
    mode = _x3_mode()
    return mode == "all" or (mode == "select" and taps >= _SELECT[0] and red_channels >= _SELECT[1])


_SELECT = (int(os.environ.get("DASS_X3_SELECT_TAPS", "9")), int(os.environ.get("DASS_X3_SELECT_C", "256")))  # measured optimum
_X3_MIN_ROWS = 256  # output rows below which a conv stays on the classic kernel
_FANOUT = os.environ.get("DASS_FANOUT", "1") == "1"  # gradients of multi-consumer tensors summed by one library pass (0: autograd's adds)
_SKIP_DY32 = os.environ.get("DASS_SKIP_DY32", "1") == "1"  # BN backward writes only the split rows of dy when nothing reads its f32 form
_ROWS_ONLY = os.environ.get("DASS_ROWS_ONLY", "1") == "1"  # a layer whose only reader is a pre-split conv writes only its split rows (conv_bn_act(sole_consumer=True))


def _cdt(t):
    """dtype code for the MFMA conv entry points"""
    d = _dt(t)
    if d == F32:
        return {"f32": F32, "bf16x3": F32X3, "bf16x6": F32X6, "f16x3": F32X6, "bf16x1": F32X6}[_state["f32_mma"]]
    return d


def _dt(t):
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.bfloat16:
        return BF16
    raise TypeError("dass_hip: unsupported dtype %s" % t.dtype)


def _epv(dtype):
    return 4 if dtype == torch.float32 else 8


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


_raw_stream = torch._C._cuda_getCurrentRawStream
_cur_device = torch._C._cuda_getDevice


def _stream():
    # (the raw accessor: torch.cuda.current_stream() builds a Stream object per call, ~8 us, and a step asks ~360 times)
    return ctypes.c_void_p(_raw_stream(_cur_device()))


def _require_cuda(t):
    if not t.is_cuda:
        raise RuntimeError("dass_hip ops run on the GPU only (tensor is on %s); there is no CPU fallback" % t.device)


# ----------------------------------------------------------------------------- layout helpers
def new_act(n, c, h, w, dtype, device):
    """NCHW-shaped tensor backed by NHWC memory."""
    return torch.empty((n, h, w, c), dtype=dtype, device=device).permute(0, 3, 1, 2)


def zeros_act(n, c, h, w, dtype, device):
    return torch.zeros((n, h, w, c), dtype=dtype, device=device).permute(0, 3, 1, 2)


def rows(x):
    """-> (tensor, ld): x as NHWC pixel rows with pixel stride ld (a channel slice of a wider
    NHWC buffer is accepted as is); anything else is re-laid out once."""
    n, c, h, w = x.shape
    sn, sc, sh, sw = x.stride()
    ld = sw if w > 1 else (sh // max(w, 1) if h > 1 else (sn // max(h * w, 1) if n > 1 else c))
    ok = (sc == 1 or c == 1) and ld >= c
    ok = ok and (w == 1 or sw == ld) and (h == 1 or sh == w * ld) and (n == 1 or sn == h * w * ld)
    if not ok or ld % 4 != 0 and ld != c:
        x = x.contiguous(memory_format=torch.channels_last)
        if x.stride(1) != 1:  # degenerate shapes: force explicit NHWC storage
            x = x.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
        ld = c
    return x, ld


def _cast_act(x):
    """bring an incoming activation to the compute dtype (torch cast = boundary plumbing only)."""
    dt = compute_dtype()
    return x if x.dtype == dt else x.to(dt)


# ----------------------------------------------------------------------------- raw launches
def conv_launch(x, ldx, w_op, y, ldy, dims, scale=None, shift=None, residual=None, ldr=0, in_scale=None,
                act=ACT_NONE, ustride=1):
    n, h, w, c, oh, ow, k, r, s, stride, pad, dil = dims
    check(lib.dass_conv2d_igemm(_p(x), ldx, _p(w_op), _p(y), ldy, _p(scale), _p(shift), _p(residual), ldr,
                                _p(in_scale), n, h, w, c, oh, ow, k, r, s, stride, pad, dil, ustride, act,
                                _cdt(y), _stream()), "dass_conv2d_igemm")


# ----------------------------------------------------------------------------- pre-split ("x3") operands
def x3_alloc(nrows, c, device):
    """uninitialised x3 buffer for a [rows][c] activation: rows x ceil(c/32) slabs of 192 B + one zero row"""
    return torch.empty((lib.dass_x3_bytes(nrows, c),), dtype=torch.uint8, device=device)


def split3_rows(x, ld, m, c, nc_scale=None, rows_per_image=1):
    """f32 pixel rows -> x3 rows (the operand of dass_conv2d_x3); nc_scale: [N,C] Dropout2d multipliers of the producer"""
    out = x3_alloc(m, c, x.device)
    check(lib.dass_split3_rows(_p(x), ld, _p(out), m, c, _p(nc_scale), rows_per_image, _stream()), "dass_split3_rows")
    return out


def conv_x3_launch(x3, w3, y, ldy, dims, y3=None, scale=None, shift=None, residual=None, ldr=0, act=ACT_NONE, ustride=1,
                   stats=None, y_amax=None):
    """dims as conv_launch; stats: None or a [rows, 2, K] f32 buffer -> returns the number of partial rows written.
    y3 under the two-part format: call x3_prepare_out(y3, ...) first (the output's scale must exist before the launch);
    y_amax: optional zeroed 1-element tensor that receives max |output| (bit pattern) when only f32 rows are written"""
    n, h, w, c, oh, ow, k, r, s, stride, pad, dil = dims
    nrows = ctypes.c_int(0)
    ws = _x3_workspace(x3.device)
    check(lib.dass_conv2d_x3(_p(x3), _p(w3), _p(y), ldy, _p(y3), _p(scale), _p(shift), _p(residual), ldr, n, h, w, c, oh, ow, k,
                             r, s, stride, pad, dil, ustride, act, _p(stats), ctypes.byref(nrows) if stats is not None else None,
                             _p(ws), ws.numel(), _p(y_amax), _stream()), "dass_conv2d_x3")
    return nrows.value


class BnBwdLink:
    """What the input-gradient launch of the NEXT conv needs to add a conv + BN layer's backward sums in its epilogue
    (dass_conv2d_x3_dgrad_bnstats): attached to the layer's output tensor by the forward, claimed by the single dense conv that
    consumes it; that conv's backward fills `sums` and remembers which tensor it returned, the layer's own backward takes the
    sums only if exactly that tensor arrives as its d_out (autograd hands a lone gradient through untouched; a second consumer
    would make it a sum in a new tensor).  The link keeps a STRONG reference to that tensor (`dx`) until the layer's backward
    has looked at it: autograd's InputBuffer adds a second gradient IN PLACE (`old_var.add_(var)`) when it holds the last
    reference to the first one -- same address, different contents -- and a live reference here makes that impossible, so a
    consumer the link cannot see (a `residual=` use, concat, pooling, user code) always shows up as a different pointer."""
    __slots__ = ("y_raw", "mean", "invstd", "gsc", "gsh", "gates", "act", "m", "k", "claimed", "dead", "sums", "dx_ptr", "dx")

    def __init__(self, y_raw, mean, invstd, gsc, gsh, gates, act, m, k):
        self.y_raw, self.mean, self.invstd, self.gsc, self.gsh, self.gates, self.act, self.m, self.k = y_raw, mean, invstd, gsc, gsh, gates, act, m, k
        self.claimed = self.dead = False
        self.sums = None
        self.dx_ptr = 0
        self.dx = None


_DW_FWD_SUMS = os.environ.get("DASS_DW_FWD_SUMS", "1") == "1"  # train-mode BN statistics of a depthwise conv's output in the conv launch
_DW_BN_LINK = os.environ.get("DASS_DW_BN_LINK", "1") == "1"  # ... also in a depthwise conv's input-gradient launch (MobileNetV2 expand layers)
_BN_LINK = os.environ.get("DASS_BN_LINK", "1") == "1"  # BN-backward sums ride in the epilogue of the next layer's input-gradient launch
bn_link_counts = {"asked": 0, "fused": 0, "used": 0}    # launches asked to carry sums / that did / sums a layer's backward took over


def set_bn_link(on):
    global _BN_LINK
    _BN_LINK = bool(on)


def set_dw_bn_link(on, fwd_sums=None):
    """depthwise launches that carry BN sums: `on` = the producing layer's BN-BACKWARD sums in the input-gradient launch
    (dass_dwconv3x3_bwd_data_bnstats), fwd_sums = the batch statistics of the conv's own output in the forward launch (dass_dwconv3x3_fwd_sums)"""
    global _DW_BN_LINK, _DW_FWD_SUMS
    _DW_BN_LINK = bool(on)
    if fwd_sums is not None:
        _DW_FWD_SUMS = bool(fwd_sums)


def conv_x3_dgrad_bnstats(dy3, w_t, dx, dims, link, residual=None, ldr=0):
    """dgrad launch (dims = conv_launch dims of the transposed conv, stride 1) that also produces link's BN-backward sums;
    -> True when fused (link.sums / link.dx_ptr set)"""
    n, h, w, c, oh, ow, k, r, s, stride, pad, dil = dims
    assert stride == 1 and k == link.k and n * oh * ow == link.m
    sums = _bn_sums(k, dx.device)
    fused = ctypes.c_int(0)
    ws = _x3_workspace(dy3.device)
    g = link.gates
    check(lib.dass_conv2d_x3_dgrad_bnstats(_p(dy3), _p(w_t), _p(dx), k, _p(residual), ldr, n, h, w, c, oh, ow, k, r, s, pad, dil,
                                           _p(link.y_raw), _p(link.mean), _p(link.invstd), _p(link.gsc if g is None else None),
                                           _p(link.gsh if g is None else None), _p(g), g.numel() if g is not None else 0, link.act, _p(sums),
                                           ctypes.byref(fused), _p(ws), ws.numel(), _stream()), "dass_conv2d_x3_dgrad_bnstats")
    bn_link_counts["asked"] += 1
    if fused.value:
        bn_link_counts["fused"] += 1
        link.sums, link.dx_ptr, link.dx = sums, dx.data_ptr(), dx
    return bool(fused.value)


_l1_cache = {}


def weight_l1(weight_or_krsc, key=None):
    """[K] f32: sum |w_k| over taps and input channels (the weight-side factor of a fused two-part conv's output bound), cached
    on (storage, version) for parameters; a raw [K][R][S][C] tensor is reduced as is"""
    t = weight_or_krsc
    if key is None and t.dim() == 4 and isinstance(t, torch.nn.Parameter):
        ck = (id(t), t.data_ptr(), t._version)
        hit = _l1_cache.get(id(t))
        if hit is not None and hit[0] == ck and hit[2]() is t:
            return hit[1]
        master = _krsc_master(t)
    else:
        ck, master = None, t.contiguous().float()
    k = master.shape[0]
    l1 = torch.empty((k,), dtype=torch.float32, device=master.device)
    check(lib.dass_weight_l1(_p(master), k, master.numel() // k, _p(l1), _stream()), "dass_weight_l1")
    if ck is not None:
        if len(_l1_cache) > 4096:
            for kk in [kk for kk, v in _l1_cache.items() if v[2]() is None]:
                del _l1_cache[kk]
        _l1_cache[id(t)] = (ck, l1, weakref.ref(t))
    return l1


def x3_amax_ptr(buf):
    """device float: max |x| of a two-part x3 buffer (exact when written by dass_split3_rows or tracked by a conv epilogue)"""
    return ctypes.c_void_p(buf.data_ptr() + buf.numel() - 8)


def x3_prepare_out(y3, out_rows, k, l1, scale, shift, x3_in, in_rows, in_c, res_amax=None, act=ACT_NONE, mask_max=1.0):
    """fix the scale of a two-part y3 before the conv that writes it (include/dass_hip.h dass_x3_prepare_out)"""
    check(lib.dass_x3_prepare_out(_p(y3), out_rows, k, _p(l1), _p(scale), _p(shift), _p(x3_in), in_rows, in_c, res_amax, float(mask_max), act,
                                  _stream()), "dass_x3_prepare_out")


def dropout_pack(mask):
    """Dropout2d mask [N, C] (0 / multiplier) -> (order int32 [N, ceil(C/32)*32], cc_limit int32 [N]) on the device"""
    n, c = mask.shape
    cc = (c + 31) // 32
    order = torch.empty((n, cc * 32), dtype=torch.int32, device=mask.device)
    lim = torch.empty((n,), dtype=torch.int32, device=mask.device)
    check(lib.dass_dropout_compact(_p(mask), n, c, _p(order), _p(lim), _stream()), "dass_dropout_compact")
    return order, lim


def split3_rows_packed(x, ld, m, c, mask, order, lim, rows_per_image, bound=None, bound_mul=1.0):
    """x3 rows of the surviving channels (packed per image, multiplied by the mask); slabs beyond lim[n] stay unwritten.
    bound (two-part format): device float with *bound * bound_mul >= max |x * mask| -- saves the pass that would find it"""
    out = x3_alloc(m, c, x.device)
    if bound is not None and x3_parts() == 2:
        check(lib.dass_split3_rows_packed_bound(_p(x), ld, _p(out), m, c, _p(mask), _p(order), _p(lim), rows_per_image, _p(bound),
                                                float(bound_mul), _stream()), "dass_split3_rows_packed_bound")
    else:
        check(lib.dass_split3_rows_packed(_p(x), ld, _p(out), m, c, _p(mask), _p(order), _p(lim), rows_per_image, _stream()),
              "dass_split3_rows_packed")
    return out


def split3_rows_packed_rep(x, ld, m_out, c, mask, order, lim, rows_per_image, src_images, bound, bound_mul=1.0):
    """split3_rows_packed for m_out = (T x src_images) x rows_per_image output rows over an x of only src_images images: output
    image v packs source image v % src_images with mask / order / lim row v (all T masks of a scoring batch in one launch)"""
    out = x3_alloc(m_out, c, x.device)
    check(lib.dass_split3_rows_packed_rep(_p(x), ld, _p(out), m_out, c, _p(mask), _p(order), _p(lim), rows_per_image, src_images, _p(bound),
                                          float(bound_mul), _stream()), "dass_split3_rows_packed_rep")
    return out


def conv_x3_per_image_rep_launch(x3, w3n, lim, y, ldy, dims, res_images, y3=None, scale=None, shift=None, residual=None, ldr=0, act=ACT_NONE):
    """dass_conv2d_x3_per_image_rep: N = T x res_images "images" (own weight copy + slab limit each); the residual holds only
    res_images images and is shared by the T groups"""
    n, h, w, c, oh, ow, k, r, s, stride, pad, dil = dims
    ws = _x3_workspace(x3.device)
    check(lib.dass_conv2d_x3_per_image_rep(_p(x3), _p(w3n), _p(lim), _p(y), ldy, _p(y3), _p(scale), _p(shift), _p(residual), ldr, res_images,
                                           n, h, w, c, oh, ow, k, r, s, stride, pad, dil, act, _p(ws), ws.numel(), None, _stream()),
          "dass_conv2d_x3_per_image_rep")


def absmax_rows(x, ld, m, c):
    """device float [1] = max |x| over rows (dass_absmax_rows)"""
    b = torch.zeros((1,), dtype=torch.float32, device=x.device)
    check(lib.dass_absmax_rows(_p(x), ld, m, c, None, 1, _p(b), _stream()), "dass_absmax_rows")
    return b


def w3_pack_per_image(w3, rows, c, order, lim):
    """pre-split weight operand -> one copy per image in that image's channel order"""
    n = order.shape[0]
    out = torch.empty((lib.dass_w3_pack_bytes(rows, c, n),), dtype=torch.uint8, device=order.device)
    check(lib.dass_w3_pack_per_image(_p(w3), _p(out), rows, c, n, _p(order), _p(lim), _stream()), "dass_w3_pack_per_image")
    return out


def conv_x3_per_image_launch(x3, w3n, lim, y, ldy, dims, y3=None, scale=None, shift=None, residual=None, ldr=0, act=ACT_NONE):
    """dass_conv2d_x3_per_image: image n uses weight copy n over its first lim[n] channel slabs (Dropout2d-sparse conv)"""
    n, h, w, c, oh, ow, k, r, s, stride, pad, dil = dims
    ws = _x3_workspace(x3.device)
    check(lib.dass_conv2d_x3_per_image(_p(x3), _p(w3n), _p(lim), _p(y), ldy, _p(y3), _p(scale), _p(shift), _p(residual), ldr,
                                       n, h, w, c, oh, ow, k, r, s, stride, pad, dil, act, _p(ws), ws.numel(), None, _stream()),
          "dass_conv2d_x3_per_image")


_x3_ws = {}


def _x3_workspace(device):
    """scratch for the stream-K schedule of dass_conv2d_x3 (one per device and stream: launches on one stream are ordered)"""
    key = (device, _raw_stream(device.index if device.index is not None else _cur_device()))
    ws = _x3_ws.get(key)
    if ws is None:
        ws = _x3_ws[key] = torch.empty((lib.dass_conv2d_x3_workspace_bytes(),), dtype=torch.uint8, device=device)
    return ws


def x3_operand(t, xs, ld, m, c, nc_scale=None, rows_per_image=1):
    """x3 rows of activation `t` (xs = its NHWC row view): taken from the side buffer the producing pass attached to the
    tensor (`t._dass_x3`, valid while the tensor is unmodified) or converted now by dass_split3_rows"""
    if nc_scale is None:
        if hasattr(t, "__dict__"):
            t = t.__dict__.get("_dass_alias_of", t)  # (aliases handed out by fanout() share the rows of their base tensor)
        hit = t.__dict__.get("_dass_x3") if hasattr(t, "__dict__") else None
        if hit is not None and hit[0] == (t.data_ptr(), t._version, m, c, x3_parts()):
            return hit[1]
        if hasattr(t, "__dict__") and t.__dict__.get("_dass_rows_only"):
            raise RuntimeError("dass_hip: split rows of a rows-only activation (conv_bn_act(sole_consumer=True)) are stale or of another "
                               "format, and its f32 values were never written")
        buf = split3_rows(xs, ld, m, c)
        if hasattr(t, "__dict__"):
            attach_x3(t, buf, m, c)  # other consumers of the same tensor (the ASPP branches share their input) reuse the rows
        return buf
    return split3_rows(xs, ld, m, c, nc_scale, rows_per_image)


def attach_x3(t, buf, m, c):
    t.__dict__["_dass_x3"] = ((t.data_ptr(), t._version, m, c, x3_parts()), buf)


def attached_x3(t, m, c):
    """the split rows a producer attached to `t` (None when absent or stale)"""
    if hasattr(t, "__dict__"):
        t = t.__dict__.get("_dass_alias_of", t)
    hit = t.__dict__.get("_dass_x3") if hasattr(t, "__dict__") else None
    return hit[1] if hit is not None and hit[0] == (t.data_ptr(), t._version, m, c, x3_parts()) else None


def x3_bound_ptr(buf):
    """device float: the bound of max |x| kept in the trailer of a two-part x3 buffer (behind its inverse scale)"""
    return ctypes.c_void_p(buf.data_ptr() + buf.numel() - 12)


def bound_of(t, xs, ld, m, c):
    """-> (keepalive, pointer) of a device float >= max |t|: from the trailer of t's attached two-part rows, else by one
    dass_absmax_rows pass (rows xs / ld)"""
    buf = attached_x3(t, m, c) if x3_parts() == 2 else None
    if buf is not None:
        return buf, x3_bound_ptr(buf)
    b = torch.zeros((1,), dtype=torch.float32, device=xs.device)
    check(lib.dass_absmax_rows(_p(xs), ld, m, c, None, 1, _p(b), _stream()), "dass_absmax_rows")
    return b, _p(b)


def channel_stats(x, ld, m, k):
    nrows = lib.dass_stat_rows(m)
    partial = torch.empty((nrows, 2, k), dtype=torch.float32, device=x.device)
    check(lib.dass_channel_stats(_p(x), ld, m, k, _p(partial), _dt(x), _stream()), "dass_channel_stats")
    return partial, nrows


class BNState(object):
    """mean / invstd / scale / shift vectors of one BN application (f32, device)."""

    def __init__(self, k, device):
        buf = torch.empty((4, k), dtype=torch.float32, device=device)
        self.mean, self.invstd, self.scale, self.shift = buf[0], buf[1], buf[2], buf[3]


_pending_counters = []
_defer = {"depth": 0}


class deferred_bn_counters(object):
    """with deferred_bn_counters(): ... -> counters of every BN touched inside are bumped in ONE launch on
    exit of the outermost scope (modules used stand-alone open their own scope)."""

    def __enter__(self):
        _defer["depth"] += 1

    def __exit__(self, *exc):
        _defer["depth"] -= 1
        if _defer["depth"] == 0:
            flush_bn_counters()
        return False


def bn_counter_scope(forward):
    """decorator for nn.Module.forward: run inside a deferred_bn_counters() scope"""
    import functools

    @functools.wraps(forward)
    def wrapped(self, *args, **kwargs):
        with deferred_bn_counters():
            return forward(self, *args, **kwargs)

    return wrapped


def flush_bn_counters():
    """num_batches_tracked += 1 for every BN that normalised with batch statistics since the last flush:
    one multi-tensor launch instead of 113 one-element adds per step (bookkeeping, not arithmetic on
    activations).  Called at the end of DeepLab.forward and from every state_dict()/load hook below."""
    if _pending_counters:
        pend = list(_pending_counters)
        del _pending_counters[:]
        torch._foreach_add_(pend, 1)


_bn_sum_arena = {"buf": None, "off": 0, "on": os.environ.get("DASS_BN_SUMS", "1") == "1"}
_BN_GATES = os.environ.get("DASS_BN_GATES", "1") == "1"  # residual layers keep their activation gates as bits for the backward


def bn_sums_path(bn, k):
    """train-mode BN of this layer through f64 channel accumulators (dass_conv2d_*_sums -> dass_bn_apply_train, backward
    dass_bn_bwd_reduce_sums -> dass_bn_bwd_apply_sums): no finalize launches.  Not for SyncBN across ranks (the sums are
    exchanged there), not in deterministic mode (atomic order), not beyond 2048 channels (LDS table of the apply kernel)."""
    return (_bn_sum_arena["on"] and k <= 2048 and k % 4 == 0 and sync_bn_world(bn) == 1
            and not lib.dass_get_deterministic())


def _bn_sums(k, dev):
    """a zeroed [2][k] f64 slice: cut from an arena cleared by one memset per ~20 train steps (never handed out twice)"""
    a = _bn_sum_arena
    need = (2 * k + (k + 1) // 2 + 63) // 64 * 64  # (+ K floats: the backward reduce keeps every channel's max |dz| there)
    if a["buf"] is None or a["buf"].device != dev or a["off"] + need > a["buf"].numel():
        a["buf"] = torch.zeros((max(1 << 22, need),), dtype=torch.float64, device=dev)
        a["off"] = 0
    out = a["buf"][a["off"]:a["off"] + 2 * k]
    a["off"] += need
    return out


def _bn_running(bn):
    """(momentum or -1, running_mean, running_var) of a train-mode BN call + the num_batches_tracked bookkeeping"""
    mom = -1.0
    rm = rv = None
    if bn.track_running_stats and bn.running_mean is not None:
        mom = 0.1 if bn.momentum is None else float(bn.momentum)
        rm, rv = bn.running_mean, bn.running_var
        _pending_counters.append(bn.num_batches_tracked)
        if _defer["depth"] == 0 or len(_pending_counters) >= 512:
            flush_bn_counters()
    return mom, rm, rv


def bn_train_state(x, ld, m, k, bn, rep=1.0, stats=None):
    """stats: (partial, rows) already produced by the conv epilogue (dass_conv2d_igemm_stats)"""
    partial, nrows = stats if stats is not None else channel_stats(x, ld, m, k)
    st = BNState(k, x.device)
    mom, rm, rv = _bn_running(bn)
    world = sync_bn_world(bn)
    if world > 1:
        # SyncBN: local sums -> one RCCL all-reduce of [2K] floats -> global mean / clamp(var, eps)^-1/2.
        # Ranks are assumed to hold equal element counts (DDP with equal per-GPU batches), so no count exchange.
        import torch.distributed as dist

        sums = torch.empty((2, k), dtype=torch.float32, device=x.device)
        check(lib.dass_bn_bwd_finalize(_p(partial), nrows, k, _p(sums[0]), _p(sums[1]), _stream()), "dass_bn_bwd_finalize")
        if rep != 1.0:
            sums.mul_(rep)
        dist.all_reduce(sums)
        check(lib.dass_bn_finalize_sums(_p(sums), k, float(m) * rep * world, _p(bn.weight), _p(bn.bias), _p(rm), _p(rv), mom,
                                        float(bn.eps), 1, _p(st.mean), _p(st.invstd), _p(st.scale), _p(st.shift),
                                        _stream()), "dass_bn_finalize_sums")
        _running_stats_written(bn, rm, rv)
        return st
    check(lib.dass_bn_finalize(_p(partial), nrows, k, float(m) * rep, float(rep), _p(bn.weight), _p(bn.bias), _p(rm),
                               _p(rv), mom, float(bn.eps), _p(st.mean), _p(st.invstd), _p(st.scale), _p(st.shift),
                               _stream()), "dass_bn_finalize")
    _running_stats_written(bn, rm, rv)
    return st


def _running_stats_written(bn, rm, rv):
    """the finalize kernels update running_mean / running_var through raw pointers: bump their version counters (the
    eval-BN vector cache of bn_eval_state is keyed on them) and drop the cached eval vectors of this layer"""
    if rm is not None:
        torch.autograd.graph.increment_version([rm, rv])
        bn.__dict__.pop("_dass_eval_state", None)


def sync_bn_world(bn):
    """number of ranks a BN layer synchronises its batch statistics over (1 = plain per-GPU BN).
    SynchronizedBatchNorm2d instances (marked `_dass_sync`) synchronise when torch.distributed is initialised."""
    if not getattr(bn, "_dass_sync", False):
        return 1
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized():
        return dist.get_world_size()
    return 1


def _allreduce_bn_grads(sums, world):
    if world > 1:
        import torch.distributed as dist

        dist.all_reduce(sums)


def bn_eval_state(bn, k, device):
    """mean / invstd / scale / shift of a BN layer on its running statistics; cached on the module until one of its four
    tensors changes (version counters), so the T passes of MC-dropout scoring and every pool batch reuse one launch"""
    def ver(t):
        return None if t is None else (t.data_ptr(), t._version)
    key = (ver(bn.weight), ver(bn.bias), ver(bn.running_mean), ver(bn.running_var), float(bn.eps), k, str(device), _wepoch["n"])
    hit = bn.__dict__.get("_dass_eval_state")
    if hit is not None and hit[0] == key:
        return hit[1]
    st = BNState(k, device)
    check(lib.dass_bn_eval_scale_shift(_p(bn.weight), _p(bn.bias), _p(bn.running_mean), _p(bn.running_var),
                                       float(bn.eps), k, _p(st.mean), _p(st.invstd), _p(st.scale), _p(st.shift),
                                       _stream()), "dass_bn_eval_scale_shift")
    bn.__dict__["_dass_eval_state"] = (key, st)
    return st


def written_in_place(t):
    """`t` was (or is about to be) rewritten through its raw pointer: bump its version counter -- every cache keyed on
    (data_ptr, _version) then misses -- and drop the split rows a producer attached to it"""
    if hasattr(t, "__dict__"):
        t.__dict__.pop("_dass_x3", None)
    if not t.is_inference():
        torch.autograd.graph.increment_version(t)


def scale_shift_act(x, ldx, out, ldo, m, k, scale, shift, residual=None, ldr=0, nc_scale=None, rows_per_image=1,
                    act=ACT_NONE, out3=None):
    """out3: optional x3 buffer that receives the same values as three bf16 parts (operand of the next dense conv)"""
    if out is not None and out.data_ptr() == x.data_ptr() and out3 is None:
        written_in_place(out)
    check(lib.dass_scale_shift_act(_p(x), ldx, _p(out), ldo, _p(scale), _p(shift), _p(residual), ldr, _p(nc_scale),
                                   m, k, rows_per_image, act, _dt(x), _p(out3), _stream()), "dass_scale_shift_act")


def x3_alloc_for(m, k, device):
    """x3 buffer a producer kernel fills 4 channels at a time: zero-filled when the last 32-channel slab is ragged"""
    if k % 32 == 0:
        return x3_alloc(m, k, device)
    return torch.zeros((lib.dass_x3_bytes(m, k),), dtype=torch.uint8, device=device)


def bn_use_batch_stats(bn):
    return bn.training or (bn.running_mean is None)


# ----------------------------------------------------------------------------- weight operands
_wcache = {}
_wepoch = {"n": 0}  # bumped by weights_changed(): part of every operand cache key next to the parameter's version counter


def weights_changed():
    """parameters were written behind autograd's back -- a replayed hipGraph that contains the optimizer step (dass_hip/graph.py) --
    so their version counters did not move: every cached weight operand (casts, splits, L1 norms) is stale from now on"""
    _wepoch["n"] += 1
    _l1_cache.clear()


def _krsc_master(weight):
    """f32 KRSC view of an OIHW parameter (zero-copy when the parameter is channels_last)."""
    w = weight.detach()
    if w.dtype != torch.float32:
        w = w.float()
    wp = w.permute(0, 2, 3, 1)
    if not wp.is_contiguous():
        wp = wp.contiguous()
    return wp


def _split_fmt(dtype, x3):
    """pre-split operand format of an f32 conv weight: None (plain), F32X6 (three bf16 parts: classic bf16x6 kernel, pre-split
    kernels in their three-part mode) or F16X3 (two scaled f16 parts: pre-split kernels under the "f16x3" engine)"""
    if dtype != torch.float32 or _state["f32_mma"] not in _X3_ENGINES:
        return None
    if x3 and _state["f32_mma"] == "bf16x1":
        return BF16X1
    return F16X3 if (x3 and _state["f32_mma"] == "f16x3") else F32X6


def weight_operand(weight, mode, dtype, cpad=None, x3=False):
    """mode 0: [K][R][S][Cpad] forward operand; mode 1: [C][R][S][K] flipped dgrad operand; x3: for the pre-split kernels.
    Cached on (storage, version) so eval / MC passes transform once."""
    k, c, r, s = weight.shape
    cdst = c if cpad is None else cpad
    split6 = _split_fmt(dtype, x3)
    key = (weight.data_ptr(), weight._version, mode, dtype, cdst, tuple(weight.shape), split6, _wepoch["n"])
    hit = _wcache.get((id(weight), mode, split6))
    if hit is not None and hit[0] == key and hit[2]() is weight:
        return hit[1]
    master = _krsc_master(weight)
    if split6:
        if master.data_ptr() == weight.data_ptr():  # zero-copy master: eligible for the one-launch refresh of all weights
            return _split6_registered(weight, master, mode, cdst, split6)
        op = _split6_operand(master, k, r, s, c, cdst, mode, split6)
    elif mode == 0 and dtype == torch.float32 and cdst == c:
        op = master
    else:
        if mode == 0:
            op = torch.empty((k, r, s, cdst), dtype=dtype, device=weight.device)
        else:
            op = torch.empty((c, r, s, k), dtype=dtype, device=weight.device)
        check(lib.dass_weight_transform(_p(master), _p(op), k, r, s, c, cdst, mode, F32 if dtype == torch.float32 else BF16,
                                        _stream()), "dass_weight_transform")
    if len(_wcache) > 4096:  # entries of parameters that no longer exist
        for kk in [kk for kk, v in _wcache.items() if v[2]() is None]:
            del _wcache[kk]
    _wcache[(id(weight), mode, split6)] = (key, op, weakref.ref(weight))
    return op


class _SplitEntry(object):
    __slots__ = ("weight", "master", "mode", "cdst", "op", "version", "items", "fmt")


_split_reg = {"entries": {}, "table_key": {}, "table": {}}


def _split6_registered(weight, master, mode, cdst, fmt=F32X6):
    """pre-split operand (format fmt) of a parameter, refreshed together with every other registered conv weight of that format
    in ONE call when its version went stale (an optimizer step bumps them all): dass_weight_split_batch(_f16)."""
    k, c, r, s = weight.shape
    ent = _split_reg["entries"].get((id(weight), mode, fmt))
    if ent is None or ent.weight() is not weight or ent.master.data_ptr() != master.data_ptr() or ent.cdst != cdst:
        ent = _SplitEntry()
        ent.weight, ent.master, ent.mode, ent.cdst, ent.version, ent.fmt = weakref.ref(weight), master, mode, cdst, -1, fmt
        rows, red = (k, cdst) if mode == 0 else (c, k)
        ent.op = torch.empty((lib.dass_weight_operand_bytes(rows, r, s, red, fmt),), dtype=torch.uint8, device=weight.device)
        ent.items = ((rows + 31) // 32) * r * s * ((red + 31) // 32)  # tiles of 32 rows x one slab
        _split_reg["entries"][(id(weight), mode, fmt)] = ent
    if ent.version != (weight._version, _wepoch["n"]):
        dead = [kk for kk, e in _split_reg["entries"].items() if e.weight() is None]
        for kk in dead:  # parameters that no longer exist: release their operands
            del _split_reg["entries"][kk]
        # (tried in round 4 and dropped: refreshing the input-gradient operands (mode 1) on the side stream beside the forward pass --
        #  0.3 of the 0.6 ms this refresh costs at the head of a step -- left the train step where it was and cost the MC-dropout leg 20 %)
        stale = []
        for e in _split_reg["entries"].values():
            wt = e.weight()
            if e.fmt == fmt and e.version != (wt._version, _wepoch["n"]) and wt.device == weight.device and wt.data_ptr() == e.master.data_ptr():
                stale.append((e, wt))
        key = tuple((e.master.data_ptr(), e.mode, e.op.data_ptr()) for e, _ in stale)
        if _split_reg["table_key"].get(fmt) != key:
            rows_, start, acc = [], [], 0
            for e, wt in stale:
                kk, cc, rr, ss = wt.shape
                rows_.append([e.master.data_ptr(), e.op.data_ptr(), kk, rr, ss, cc, e.cdst, e.mode])
                start.append(acc)
                acc += e.items
            _split_reg["table"][fmt] = (torch.tensor(rows_, dtype=torch.int64).to(weight.device), torch.tensor(start, dtype=torch.int64).to(weight.device), acc)
            _split_reg["table_key"][fmt] = key
        desc, start_t, total = _split_reg["table"][fmt]
        fn = {F16X3: lib.dass_weight_split_batch_f16, BF16X1: lib.dass_weight_split_batch_bf16}.get(fmt, lib.dass_weight_split_batch)
        check(fn(_p(desc), _p(start_t), len(stale), total, _stream()), "dass_weight_split_batch")
        for e, wt in stale:
            e.version = (wt._version, _wepoch["n"])
    return ent.op


def _split6_operand(master, k, r, s, c, cdst, mode, fmt=F32X6):
    """pre-split conv operand of one weight tensor: DASS_F32X6 (three bf16 parts, dass_weight_transform) or DASS_F16X3 (two
    scaled f16 parts + trailer: a one-entry dass_weight_split_batch_f16)"""
    rows, red = (k, cdst) if mode == 0 else (c, k)
    op = torch.empty((lib.dass_weight_operand_bytes(rows, r, s, red, fmt),), dtype=torch.uint8, device=master.device)
    if fmt in (F16X3, BF16X1):
        desc = torch.tensor([[master.data_ptr(), op.data_ptr(), k, r, s, c, cdst, mode]], dtype=torch.int64).to(master.device)
        start = torch.zeros((1,), dtype=torch.int64, device=master.device)
        total = ((rows + 31) // 32) * r * s * ((red + 31) // 32)
        if fmt == F16X3:
            check(lib.dass_weight_split_batch_f16(_p(desc), _p(start), 1, total, _stream()), "dass_weight_split_batch_f16")
        else:
            check(lib.dass_weight_split_batch_bf16(_p(desc), _p(start), 1, total, _stream()), "dass_weight_split_batch_bf16")
        op._dass_keep = (master, desc, start)  # (the launches read these asynchronously)
    else:
        check(lib.dass_weight_transform(_p(master), _p(op), k, r, s, c, cdst, mode, F32X6, _stream()), "dass_weight_transform")
    return op


def prepare_conv_weight(w_krsc, mode=0, x3=False):
    """tools / bench: the operand conv_launch() (x3=True: conv_x3_launch()) wants for a raw [K][R][S][C] f32 (or bf16) weight
    tensor in the current engine (identity except for the split engines, which multiply pre-split weights)"""
    fmt = _split_fmt(w_krsc.dtype, x3)
    if fmt:
        k, r, s, c = w_krsc.shape
        return _split6_operand(w_krsc.contiguous(), k, r, s, c, c, mode, fmt)
    return w_krsc


def _pad_to(v, m):
    return (v + m - 1) // m * m


def conv_out_size(h, k, stride, pad, dil):
    return (h + 2 * pad - dil * (k - 1) - 1) // stride + 1


# ----------------------------------------------------------------------------- fused conv + BN + act
class ConvSpec(object):
    """static description of one conv (+BN +act) site."""

    def __init__(self, conv, bn=None, act=ACT_NONE, extra_pad=0):
        self.conv, self.bn, self.act = conv, bn, act
        self.stride = conv.stride[0]
        self.dil = conv.dilation[0]
        self.pad = conv.padding[0] + extra_pad
        self.depthwise = conv.groups > 1
        if self.depthwise:
            assert conv.groups == conv.in_channels == conv.out_channels and conv.kernel_size == (3, 3)
        else:
            assert conv.groups == 1


def _dw_weight(weight):
    w = weight.detach()
    return w.reshape(w.shape[0], 9).contiguous().float()


def _conv_forward_raw(spec, x, ldx, n, h, w, c, weight, y, ldy, oh, ow, **epi):
    k = weight.shape[0]
    if spec.depthwise:
        assert not epi
        check(lib.dass_dwconv3x3_fwd(_p(x), ldx, _p(_dw_weight(weight)), _p(y), ldy, n, h, w, c, oh, ow, spec.stride,
                                     spec.pad, spec.dil, _dt(y), _stream()), "dass_dwconv3x3_fwd")
    elif getattr(spec, "rowtap", False):
        assert not any(v is not None and v != 0 for v in epi.values()), "row-tap stem has no fused epilogue"
        r, s = weight.shape[2], weight.shape[3]
        check(lib.dass_conv2d_rowtap(_p(x), _p(_krsc_master(weight)), _p(y), ldy, n, h, w, c, oh, ow, k, r, s, spec.stride,
                                     spec.pad, _dt(y), _stream()), "dass_conv2d_rowtap")
    else:
        r, s = weight.shape[2], weight.shape[3]
        w_op = weight_operand(weight, 0, y.dtype, cpad=c)
        conv_launch(x, ldx, w_op, y, ldy, (n, h, w, c, oh, ow, k, r, s, spec.stride, spec.pad, spec.dil), **epi)


class _ConvBnAct(torch.autograd.Function):
    """conv (groups=1 or depthwise 3x3) -> [BatchNorm2d] -> [ReLU|ReLU6] with optional bias, residual
    add before the activation (resnet.py:42-43) and Dropout2d channel mask after it (aspp.py:89).
    x may be the NCHW f32 network input (stem): it is re-laid out to NHWC (C padded) on the fly."""

    @staticmethod
    def forward(ctx, x, weight, gamma, beta, bias, residual, nc_scale, spec, image_input):
        _require_cuda(x)
        dt = compute_dtype()
        dev = x.device
        n, c_in, h, w = x.shape
        k = weight.shape[0]
        r = weight.shape[2]
        rowtap = (image_input and dt == torch.float32 and weight.shape[3] * c_in <= 32 and spec.dil == 1
                  and not spec.depthwise)
        spec.rowtap = rowtap
        if image_input:
            c = c_in if rowtap else _pad_to(c_in, _epv(dt))   # row-tap stem reads the dense 3-channel image
            xr = torch.empty((n, h, w, c), dtype=dt, device=dev)
            xin = x.contiguous().float()
            check(lib.dass_nchw_to_nhwc(_p(xin), _p(xr), n, c_in, h, w, c, _dt(xr), _stream()), "dass_nchw_to_nhwc")
            ldx = c
            xs = xr
        else:
            xs, ldx = rows(_cast_act(x))
            c = c_in
        x_rows_only = bool(hasattr(x, "__dict__") and x.__dict__.get("_dass_rows_only"))
        oh = conv_out_size(h, r, spec.stride, spec.pad, spec.dil)
        ow = conv_out_size(w, weight.shape[3], spec.stride, spec.pad, spec.dil)
        m = n * oh * ow
        bn = spec.bn
        # grad mode is always off INSIDE forward and needs_input_grad ignores no_grad(): the caller's
        # grad mode is sampled in conv_bn_act() and carried on the spec
        need_grad = spec.grad_enabled and any(ctx.needs_input_grad)
        res_t = ldr = None
        if residual is not None:
            res_t, ldr = rows(_cast_act(residual))
            rl = residual.__dict__.get("_dass_bnlink") if hasattr(residual, "__dict__") else None
            if rl is not None:
                rl.dead = True  # a consumer the link's launch does not see: the layer's d_out will be a sum
        kpad = _pad_to(k, _epv(dt))
        if kpad != k:  # e.g. the 19-class classifier: keep a zero pad column so rows stay 16-B aligned
            # (only the pad columns are zeroed: the conv writes the k real ones -- a full-tensor fill per classifier call was 2.5 % of
            # the MC-dropout leg's kernel time)
            buf = new_act(n, kpad, oh, ow, dt, dev)
            buf[:, k:].zero_()
            out = buf[:, :k]
        else:
            out = new_act(n, k, oh, ow, dt, dev)
        ldo = kpad
        into = getattr(spec, "out_into", None)
        if into is not None:  # a channel slice of the caller's wider buffer (every launch below takes (pointer, ld))
            wide, off = into
            if kpad != k or wide.dtype != dt or tuple(wide.shape) != (n, wide.shape[1], oh, ow) or off % 4 or off + k > wide.shape[1] or rows(wide)[0] is not wide:
                raise RuntimeError("dass_hip: out_into needs an NHWC buffer [N, C_total, OH, OW] of the compute dtype and a 16-byte aligned channel offset")
            out = wide[:, off:off + k]
            ldo = wide.shape[1]
        state = None
        y_raw = None
        x3_saved = None  # split rows of the input, kept for the weight gradient (dass_conv2d_wgrad_x3)
        gates = None     # activation gate bits of a residual layer (train-mode BN through the sums path)
        batch_stats = bn is not None and bn_use_batch_stats(bn)
        fuse = (not spec.depthwise) and (not rowtap) and (bn is None or (not batch_stats and not need_grad))
        # pipelined pre-split engine: dense convs of the bf16x6 engine with enough output channels for its tiles
        taps = r * weight.shape[3]
        x3_on = x3_pipeline(training=need_grad)  # the whole network runs pre-split: producers hand split rows on
        x3_fwd = x3_on or (need_grad and _x3_train_layer(taps, c))
        # (a handful of rows -- the ASPP image-pool 1x1 over [N, C, 1, 1] -- stays on the classic kernel: its tiles would be
        # padding, and under "f16x3" its operands then stay exact: the two-sample-per-channel BN behind it amplifies every
        # rounding of that conv by ~1e3, tests/test_grad_parity_gpu.py)
        use_x3 = (x3_fwd and dt == torch.float32 and not spec.depthwise and not rowtap and not image_input and k > 32
                  and m >= _X3_MIN_ROWS)
        dims = (n, h, w, c, oh, ow, k, r, weight.shape[3], spec.stride, spec.pad, spec.dil)
        if x_rows_only and not (use_x3 and attached_x3(x, n * h * w, c) is not None):
            raise RuntimeError("dass_hip: this activation was produced with sole_consumer=True (split rows only, its f32 values were never "
                               "written) but its consumer does not take split rows -- drop sole_consumer at the producer or set DASS_ROWS_ONLY=0")
        rows_only = bool(getattr(spec, "rows_only", False))
        if fuse and use_x3:
            scale = shift = None
            if bn is not None:
                state = bn_eval_state(bn, k, dev)
                scale, shift = state.scale, state.shift
            elif bias is not None:
                shift = bias.detach().float()
            in_scale = getattr(spec, "in_scale", None)
            if in_scale is not None:
                assert not need_grad and tuple(in_scale.shape) == (n, c) and in_scale.dtype == torch.float32
            x3 = x3_operand(x, xs, ldx, n * h * w, c, in_scale, h * w)
            w_op = weight_operand(weight, 0, dt, cpad=c, x3=True)
            want_y3 = getattr(spec, "emit_x3", False) and nc_scale is None and kpad == k
            y3 = x3_alloc_for(m, k, dev) if want_y3 else None
            if y3 is not None and x3_parts() == 2:
                # two-part format: the output's scale is fixed BEFORE the launch from a bound of the result (weights' L1 norms x
                # the input's true maximum + shift + residual); the epilogue then tracks the true maximum for the next layer
                res_keep, res_amax = None, None
                if res_t is not None:
                    rbuf = attached_x3(residual, m, k)
                    if rbuf is not None:
                        res_keep, res_amax = rbuf, x3_amax_ptr(rbuf)
                    else:
                        res_keep, res_amax = bound_of(residual, res_t, ldr, m, k)
                x3_prepare_out(y3, m, k, weight_l1(weight), scale, shift, x3, n * h * w, c, res_amax, spec.act)
            skip_f32 = rows_only and y3 is not None and res_t is None
            conv_x3_launch(x3, w_op, None if skip_f32 else out, ldo, dims, y3=y3, scale=scale, shift=shift, residual=res_t, ldr=ldr or 0, act=spec.act)
            if y3 is not None:
                attach_x3(out, y3, m, k)
                if skip_f32:
                    out.__dict__["_dass_rows_only"] = True
            if nc_scale is not None:
                scale_shift_act(out, ldo, out, ldo, m, k, None, None, nc_scale=nc_scale, rows_per_image=oh * ow)
        elif fuse:
            scale = shift = None
            if bn is not None:
                state = bn_eval_state(bn, k, dev)
                scale, shift = state.scale, state.shift
            elif bias is not None:
                shift = bias.detach().float()
            in_scale = getattr(spec, "in_scale", None)
            if in_scale is not None:
                assert not need_grad and tuple(in_scale.shape) == (n, c) and in_scale.dtype == torch.float32
            _conv_forward_raw(spec, xs, ldx, n, h, w, c, weight, out, ldo, oh, ow, scale=scale, shift=shift,
                              residual=res_t, ldr=ldr or 0, act=spec.act, in_scale=in_scale)
            if nc_scale is not None:
                scale_shift_act(out, ldo, out, ldo, m, k, None, None, nc_scale=nc_scale, rows_per_image=oh * ow)
        else:
            if getattr(spec, "in_scale", None) is not None:
                raise RuntimeError("conv_bn_act(in_scale=...) folds a Dropout2d mask into an INFERENCE conv loader: it needs "
                                   "eval-mode BN (or no BN), a dense non-stem conv and no gradient; this call would drop it")
            assert k % 4 == 0, "BN epilogue needs K % 4 == 0"
            y_raw = new_act(n, k, oh, ow, dt, dev)
            fused_stats = None
            sums = _bn_sums(k, dev) if batch_stats and bn_sums_path(bn, k) else None
            if sums is not None:
                # statistics as f64 accumulators the conv adds to; scale / shift derived inside the apply launch below
                if spec.depthwise or rowtap:
                    rc = 3
                    if spec.depthwise and dt == torch.float32 and _DW_FWD_SUMS:
                        # the depthwise strip kernel owns fixed channels per thread: the BN statistics ride along (no pass over y_raw)
                        rc = lib.dass_dwconv3x3_fwd_sums(_p(xs), ldx, _p(_dw_weight(weight)), _p(y_raw), k, n, h, w, c, oh, ow, spec.stride,
                                                         spec.pad, spec.dil, _p(sums), _stream())
                        if rc not in (0, 3):  # (3 = DASS_ERR_UNSUPPORTED: outside the specialisation -- the two passes below)
                            check(rc, "dass_dwconv3x3_fwd_sums")
                    if rc != 0:
                        _conv_forward_raw(spec, xs, ldx, n, h, w, c, weight, y_raw, k, oh, ow)
                        check(lib.dass_channel_sums(_p(y_raw), k, m, k, _p(sums), _dt(y_raw), _stream()), "dass_channel_sums")
                elif use_x3:
                    x3 = x3_operand(x, xs, ldx, n * h * w, c)
                    x3_saved = x3 if x3_on else None
                    ws = _x3_workspace(dev)
                    check(lib.dass_conv2d_x3_sums(_p(x3), _p(weight_operand(weight, 0, dt, cpad=c, x3=True)), _p(y_raw), k, n, h, w, c, oh, ow, k, r,
                                                  weight.shape[3], spec.stride, spec.pad, spec.dil, _p(sums), _p(ws), ws.numel(),
                                                  _stream()), "dass_conv2d_x3_sums")
                else:
                    check(lib.dass_conv2d_igemm_sums(_p(xs), ldx, _p(weight_operand(weight, 0, dt, cpad=c)), _p(y_raw), k, n, h, w, c, oh,
                                                     ow, k, r, weight.shape[3], spec.stride, spec.pad, spec.dil, _cdt(y_raw), _p(sums),
                                                     _stream()), "dass_conv2d_igemm_sums")
            elif use_x3:
                x3 = x3_operand(x, xs, ldx, n * h * w, c)
                x3_saved = x3 if x3_on else None  # "select": the weight gradient stays on the classic kernel (f32 rows)
                w_op = weight_operand(weight, 0, dt, cpad=c, x3=True)
                partial = None
                if batch_stats:
                    partial = torch.empty((lib.dass_conv2d_igemm_stats_rows(m), 2, k), dtype=torch.float32, device=dev)
                nrows = conv_x3_launch(x3, w_op, y_raw, k, dims, stats=partial)
                if batch_stats:
                    fused_stats = (partial, nrows)
            elif batch_stats and not spec.depthwise and not rowtap:
                # train-mode BN: the conv epilogue also emits the per-tile channel sums (no second read of y_raw)
                rmax = lib.dass_conv2d_igemm_stats_rows(m)
                partial = torch.empty((rmax, 2, k), dtype=torch.float32, device=dev)
                nrows = ctypes.c_int(0)
                w_op = weight_operand(weight, 0, dt, cpad=c)
                check(lib.dass_conv2d_igemm_stats(_p(xs), ldx, _p(w_op), _p(y_raw), k, n, h, w, c, oh, ow, k, r,
                                                  weight.shape[3], spec.stride, spec.pad, spec.dil, _cdt(y_raw), _p(partial),
                                                  ctypes.byref(nrows), _stream()), "dass_conv2d_igemm_stats")
                fused_stats = (partial, nrows.value)
            else:
                _conv_forward_raw(spec, xs, ldx, n, h, w, c, weight, y_raw, k, oh, ow)
            if bn is not None and sums is None:
                state = (bn_train_state(y_raw, k, m, k, bn, stats=fused_stats) if batch_stats
                         else bn_eval_state(bn, k, dev))
                scale, shift = state.scale, state.shift
            elif bn is None:
                scale, shift = None, (bias.detach().float() if bias is not None else None)
            out3 = None
            if ((x3_on or (need_grad and getattr(spec, "x3_consumer", False))) and dt == torch.float32
                    and getattr(spec, "emit_x3", True) and k >= 32 and kpad == k and (x3_parts() != 2 or sums is not None)):
                out3 = x3_alloc_for(m, k, dev)  # the consumer is (almost always) the next dense conv: hand it split rows
            if sums is not None:
                state = BNState(k, dev)
                mom, rm, rv = _bn_running(bn)
                if need_grad and res_t is not None and nc_scale is None and spec.act != ACT_NONE and _BN_GATES:
                    # residual layers cannot re-derive the activation gate from the conv output alone: keep it as one byte
                    # per 4-channel group, which the backward reads instead of the 16 bytes of `out`
                    gates = torch.empty((m, k // 4), dtype=torch.uint8, device=dev)
                res_keep, res_bound = None, None
                if out3 is not None and x3_parts() == 2 and res_t is not None:  # the output's bound includes the residual's
                    res_keep, res_bound = bound_of(residual, res_t, ldr, m, k)
                skip_f32 = (rows_only and out3 is not None and res_t is None and gates is None
                            and (not need_grad or (y_raw is not None and spec.act != ACT_NONE)))  # (backward: gate from y_raw, never from `out`)
                check(lib.dass_bn_apply_train(_p(y_raw), k, None if skip_f32 else _p(out), ldo, _p(sums), float(m), _p(bn.weight), _p(bn.bias), _p(rm), _p(rv),
                                              mom, float(bn.eps), _p(state.mean), _p(state.invstd), _p(state.scale), _p(state.shift),
                                              _p(res_t), ldr or 0, _p(nc_scale), m, k, oh * ow, spec.act, _dt(out), _p(out3), _p(gates),
                                              gates.numel() if gates is not None else 0, res_bound, _stream()), "dass_bn_apply_train")
                _running_stats_written(bn, rm, rv)
            else:
                scale_shift_act(y_raw, k, out, ldo, m, k, scale, shift, residual=res_t, ldr=ldr or 0,
                                nc_scale=nc_scale, rows_per_image=oh * ow, act=spec.act, out3=out3)
            if out3 is not None:
                attach_x3(out, out3, m, k)
                if sums is not None and skip_f32:
                    out.__dict__["_dass_rows_only"] = True
            if (_BN_LINK and need_grad and sums is not None and nc_scale is None and dt == torch.float32 and ldo == k and k % 4 == 0
                    and state is not None and sync_bn_world(bn) == 1):
                # (the gate is re-derived from y_raw unless the layer has a residual, whose gate bits `gates` holds; a residual
                # layer without stored gates reads `out` in its backward: no link)
                if res_t is None or gates is not None or spec.act == ACT_NONE:
                    ctx.out_link = out.__dict__["_dass_bnlink"] = BnBwdLink(y_raw, state.mean, state.invstd, state.scale, state.shift, gates,
                                                                            spec.act, m, k)
        in_link = None
        if need_grad and _BN_LINK and hasattr(x, "__dict__"):
            in_link = x.__dict__.get("_dass_bnlink")
            if in_link is not None:
                if in_link.claimed:  # a second consumer: the gradient will be a sum of two launches' outputs
                    in_link.dead = True
                    in_link = None
                else:
                    in_link.claimed = True
                    if (image_input or spec.stride != 1 or in_link.k != c or in_link.m != n * h * w or c != c_in
                            or getattr(spec, "rowtap", False)):
                        in_link = None
                    elif spec.depthwise and (not _DW_BN_LINK or dt != torch.float32 or spec.dil not in (1, 2) or in_link.gates is not None):
                        in_link = None  # (the depthwise input-gradient kernel carries the sums at stride 1, dilation 1 / 2, gate from y_raw)
        if need_grad:
            ctx.in_link = in_link
            ctx.spec = spec
            ctx.x3_on = x3_on
            ctx.x_rows_only = x_rows_only
            ctx.x3_dgrad = (x3_on or _x3_train_layer(taps, k)) and n * h * w >= _X3_MIN_ROWS  # the input gradient reduces over k x taps
            ctx.image_input = image_input
            ctx.dims = (n, h, w, c, oh, ow, k, ldx, ldo, c_in)
            ctx.train_stats = batch_stats
            ctx.bn_sums = batch_stats and bn_sums_path(bn, k)
            ctx.sync_world = sync_bn_world(bn) if batch_stats else 1
            ctx.has_bn = bn is not None
            ctx.has_bias = bias is not None
            ctx.has_res = residual is not None
            ctx.x_dtype = x.dtype
            ctx.save_for_backward(xs, weight, gamma, y_raw, out, nc_scale,
                                  state.mean if state is not None else None,
                                  state.invstd if state is not None else None,
                                  state.scale if state is not None else None,
                                  state.shift if state is not None else None, x3_saved, gates)
        if getattr(spec, "fork", False):
            # the input is handed back as a second output: the gradients of both uses of x (this conv and the
            # identity branch of a residual block) then arrive in ONE backward call, where the input-gradient conv adds
            # the identity gradient in its epilogue instead of autograd running a separate add pass over the tensor
            return out, x
        return out

    @staticmethod
    def backward(ctx, dout, d_fork=None):
        xs, weight, gamma, y_raw, out, nc_scale, mean, invstd, bn_scale, bn_shift, x3_in, gates = ctx.saved_tensors
        spec = ctx.spec
        if dout is None:  # only the forked alias of the input was used downstream
            return d_fork, None, None, None, None, None, None, None, None
        n, h, w, c, oh, ow, k, ldx, ldo, c_in = ctx.dims
        dt = out.dtype
        dev = out.device
        m = n * oh * ow
        dout_r, lddo = rows(_cast_act(dout))
        dgamma = dbeta = dbias = dres = None
        simple = (not ctx.has_bn) and spec.act == ACT_NONE and nc_scale is None and not ctx.has_res
        kp = _pad_to(k, _epv(dt))
        if simple:
            # conv (+bias): dy = dout.  (K may be unaligned: work on a zero-padded copy of width kp)
            if lddo % 4 != 0 or kp != k:
                dy = zeros_act(n, kp, oh, ow, dt, dev)
                dy[:, :k].copy_(dout_r)
                lddy = kp
            else:
                dy, lddy = dout_r, lddo
            if ctx.has_bias:
                partial = torch.empty((lib.dass_stat_rows(m), 2, kp), dtype=torch.float32, device=dev)
                dbias_p = torch.empty((kp,), dtype=torch.float32, device=dev)
                check(lib.dass_colsum(_p(dy), lddy, m, kp, _p(partial), _p(dbias_p), _dt(dy), _stream()), "dass_colsum")
                dbias = dbias_p[:k]
        else:
            assert k % 4 == 0
            dy = new_act(n, k, oh, ow, dt, dev)
            lddy = k
            if ctx.has_res:
                dres = new_act(n, k, oh, ow, dt, dev)
            src = y_raw if y_raw is not None else out
            ones = None
            if not ctx.has_bn:
                ones = torch.ones((k,), dtype=torch.float32, device=dev)
                mean_v, invstd_v, gamma_v = torch.zeros_like(ones), ones, None
            else:
                mean_v, invstd_v, gamma_v = mean, invstd, gamma
            db = dg = None
            need_red = ctx.has_bn or ctx.has_bias
            # no residual, f32, conv output kept: the ReLU gate is re-derived from y_raw with the forward's own fma
            # (bit-identical), so neither pass reads `out`
            gate = (ctx.has_bn and not ctx.has_res and y_raw is not None and dt == torch.float32 and bn_scale is not None
                    and spec.act != ACT_NONE)
            bsums = None
            if need_red and ctx.has_bn and ctx.train_stats and getattr(ctx, "bn_sums", False) and ctx.sync_world == 1:
                # (sum dz, sum dz * xhat) as f64 accumulators: the apply launch reads them, no finalize launch
                no_out = gate or gates is not None
                link = getattr(ctx, "out_link", None)
                link_ok = (link is not None and link.sums is not None and not link.dead and link.dx_ptr == dout_r.data_ptr() and lddo == k
                           and nc_scale is None and (no_out or spec.act == ACT_NONE))
                if link is not None:
                    link.dx = None  # (held only so that autograd could not accumulate a second gradient into it in place)
                if link_ok:
                    bsums = link.sums  # already added by the consumer's input-gradient launch (dass_conv2d_x3_dgrad_bnstats)
                    bn_link_counts["used"] += 1
                else:
                    bsums = _bn_sums(k, dev)
                    check(lib.dass_bn_bwd_reduce_sums(_p(dout_r), lddo, _p(None if no_out else out), ldo, _p(y_raw), k, _p(mean_v), _p(invstd_v),
                                                      _p(bn_scale if gate else None), _p(bn_shift if gate else None), _p(nc_scale), m, k,
                                                      oh * ow, spec.act, _p(bsums), _p(gates), gates.numel() if gates is not None else 0,
                                                      _dt(out), _stream()), "dass_bn_bwd_reduce_sums")
                pg = torch.empty((2, k), dtype=torch.float32, device=dev)
                dbeta, dgamma = pg[0], pg[1]
            elif need_red:
                nrows = lib.dass_stat_rows(m)
                partial = torch.empty((nrows, 2, k), dtype=torch.float32, device=dev)
                if gate:
                    check(lib.dass_bn_bwd_reduce_gate(_p(dout_r), lddo, _p(y_raw), k, _p(mean_v), _p(invstd_v), _p(bn_scale), _p(bn_shift),
                                                      _p(nc_scale), m, k, oh * ow, spec.act, _p(partial), _dt(out), _stream()),
                          "dass_bn_bwd_reduce_gate")
                else:
                    check(lib.dass_bn_bwd_reduce(_p(dout_r), lddo, _p(out), ldo, _p(src), k, _p(mean_v), _p(invstd_v),
                                                 _p(nc_scale), m, k, oh * ow, spec.act, _p(partial), _dt(out), _stream()),
                          "dass_bn_bwd_reduce")
                sums = torch.empty((2, k), dtype=torch.float32, device=dev)
                db, dg = sums[0], sums[1]
                check(lib.dass_bn_bwd_finalize(_p(partial), nrows, k, _p(db), _p(dg), _stream()), "dass_bn_bwd_finalize")
                if ctx.has_bn:
                    dbeta, dgamma = db, dg
                else:
                    dbias = db
                if ctx.has_bn and ctx.train_stats and ctx.sync_world > 1:
                    # SyncBN: the dx formula needs the sums over the GLOBAL batch; the parameter gradients returned to
                    # autograd stay this rank's LOCAL sums (like torch.nn.SyncBatchNorm), so that the gradient
                    # averager scales BN affine gradients exactly like conv weight gradients
                    dbeta, dgamma = db.clone(), dg.clone()
                    _allreduce_bn_grads(sums, ctx.sync_world)
            dy3 = None
            if (dt == torch.float32 and not spec.depthwise and not ctx.image_input and k >= 32
                    and ((ctx.x3_dgrad and ctx.needs_input_grad[0] and c > 32)
                         or (ctx.x3_on and ctx.needs_input_grad[1] and x3_in is not None))
                    and (x3_parts() != 2 or bsums is not None)):  # two-part rows need max |dz|: the sums path supplies it
                dy3 = x3_alloc_for(m, k, dev)  # dy also as split rows: operand of the input- and weight-gradient launches
            if bsums is not None:
                if dy3 is not None and not spec.depthwise and not getattr(spec, "rowtap", False):
                    # the f32 form of dy is dead weight when both gradient launches read the split rows: write dy3 only
                    dg3 = bool(ctx.x3_dgrad and c > 32 and ctx.needs_input_grad[0] and not ctx.image_input)
                    wg3 = bool(ctx.x3_on and x3_in is not None and x3_in.numel() == lib.dass_x3_bytes(n * h * w, c))
                    need32 = ((ctx.needs_input_grad[0] and not ctx.image_input and not dg3)
                              or (ctx.needs_input_grad[1] and not wg3))
                    if not need32 and _SKIP_DY32:
                        dy = torch.empty((0,), dtype=dt, device=dev)
                check(lib.dass_bn_bwd_apply_sums(_p(dout_r), lddo, _p(None if (gate or gates is not None) else out), ldo, _p(y_raw), k,
                                                 _p(mean_v), _p(invstd_v), _p(gamma_v.detach()), _p(bsums), _p(dbeta), _p(dgamma),
                                                 _p(bn_scale if gate else None), _p(bn_shift if gate else None), _p(nc_scale),
                                                 _p(dy) if dy.numel() else None, lddy,
                                                 _p(dres), k, m, k, oh * ow, float(m), spec.act, _p(gates), gates.numel() if gates is not None else 0,
                                                 _dt(out), _p(dy3), _stream()),
                      "dass_bn_bwd_apply_sums")
            elif gate:
                check(lib.dass_bn_bwd_apply_gate(_p(dout_r), lddo, _p(y_raw), k, _p(mean_v), _p(invstd_v), _p(gamma_v.detach()), _p(db), _p(dg),
                                                 _p(bn_scale), _p(bn_shift), _p(nc_scale), _p(dy), lddy, m, k, oh * ow,
                                                 float(m) * ctx.sync_world, 1 if ctx.train_stats else 0, spec.act, _dt(out), _p(dy3),
                                                 _stream()), "dass_bn_bwd_apply_gate")
            else:
                check(lib.dass_bn_bwd_apply(_p(dout_r), lddo, _p(out), ldo, _p(src), k, _p(mean_v), _p(invstd_v),
                                            _p(gamma_v.detach() if gamma_v is not None else None), _p(db), _p(dg),
                                            _p(nc_scale), _p(dy), lddy, _p(dres), k, m, k, oh * ow, float(m) * ctx.sync_world,
                                            1 if ctx.train_stats else 0, spec.act, _dt(out), _p(dy3), _stream()), "dass_bn_bwd_apply")
            if dy3 is not None:
                attach_x3(dy, dy3, m, k)
        dx = dw = None
        kk = kp if simple else k
        if spec.depthwise:
            wdw = _dw_weight(weight)
            if ctx.needs_input_grad[0]:
                dx = new_act(n, c, h, w, dt, dev)
                link = getattr(ctx, "in_link", None)
                fused = False
                if link is not None and not link.dead and d_fork is None and ctx.x_dtype == dx.dtype and lddy % 4 == 0:
                    # dx is the d_out of the layer that produced this conv's input (MobileNetV2: the expand 1x1 + BN + ReLU6): its
                    # BN-backward sums ride in this launch (no separate dass_bn_bwd_reduce_sums pass over dx and that layer's conv output)
                    sums = _bn_sums(c, dev)
                    rc = lib.dass_dwconv3x3_bwd_data_bnstats(_p(dy), lddy, _p(wdw), _p(dx), c, n, h, w, c, oh, ow, spec.stride, spec.pad, spec.dil,
                                                             _p(link.y_raw), _p(link.mean), _p(link.invstd), _p(link.gsc), _p(link.gsh), link.act,
                                                             _p(sums), _stream())
                    bn_link_counts["asked"] += 1
                    if rc == 0:
                        fused = True
                        bn_link_counts["fused"] += 1
                        link.sums, link.dx_ptr, link.dx = sums, dx.data_ptr(), dx
                    elif rc != 3:  # (DASS_ERR_UNSUPPORTED: outside the kernel's specialisation -- the two passes below)
                        check(rc, "dass_dwconv3x3_bwd_data_bnstats")
                if not fused:
                    check(lib.dass_dwconv3x3_bwd_data(_p(dy), lddy, _p(wdw), _p(dx), c, n, h, w, c, oh, ow, spec.stride,
                                                      spec.pad, spec.dil, _dt(dx), _stream()), "dass_dwconv3x3_bwd_data")
            if ctx.needs_input_grad[1]:
                dwf = torch.empty((c, 9), dtype=torch.float32, device=dev)
                check(lib.dass_dwconv3x3_bwd_weight(_p(xs), ldx, _p(dy), lddy, _p(dwf), n, h, w, c, oh, ow,
                                                    spec.stride, spec.pad, spec.dil, _dt(dy), _stream()),
                      "dass_dwconv3x3_bwd_weight")
                dw = dwf.view(c, 1, 3, 3)
        else:
            r, s = weight.shape[2], weight.shape[3]
            wsrc = weight
            if kk != k:  # zero-pad the out-channel axis of the master weight (classifier: 19 -> 20)
                wsrc = torch.zeros((kk, weight.shape[1], r, s), dtype=torch.float32, device=dev)
                wsrc[:k].copy_(weight.detach())
            # The weight gradient is independent of the input gradient: launch it on a side HIP stream so the
            # two kernels share the chip (small layers fill only part of the 256 CUs on their own); the main
            # stream re-joins before this backward returns, so no tensor outlives its users.
            join = None
            want_dx = ctx.needs_input_grad[0] and not ctx.image_input
            generic_dw = ctx.needs_input_grad[1] and not getattr(spec, "rowtap", False)
            if generic_dw:
                dwk = _zeroed_dw(kk * r * s * c, dev).view(kk, r, s, c)
                side = _side_stream(dev) if want_dx else None
                if side is not None:
                    _EV_FORK.record()
                    side.wait_event(_EV_FORK)
                    wstream = ctypes.c_void_p(side.cuda_stream)
                else:
                    wstream = _stream()
                dy3_w = None
                if x3_in is not None and kk == k and ctx.x3_on and x3_in.numel() == lib.dass_x3_bytes(n * h * w, c):
                    dy3_w = attached_x3(dy, m, k)
                    if dy3_w is None and x3_parts() <= 2 and lddy % 4 == 0:  # (no BN pass emitted them: convert once, both gradients use them)
                        dy3_w = x3_operand(dy, dy, lddy, m, k)
                if dy3_w is not None and _wg["on"] and side is None and c == c_in:
                    # a weight gradient has no consumer before the optimizer step: queue it; ALL queued layers are computed by
                    # one grouped launch per tile class when this backward pass ends (_wgrad_flush), which then sets .grad
                    _wgrad_enqueue(weight, x3_in, dy3_w, dwk, (n, h, w, c, oh, ow, kk, r, s, spec.stride, spec.pad, spec.dil), k, c_in)
                    generic_dw = False
                elif dy3_w is not None:
                    # both operands already exist as split rows (forward producer / BN-backward pass): copy + MFMA only
                    check(lib.dass_conv2d_wgrad_x3(_p(x3_in), _p(dy3_w), _p(dwk), n, h, w, c, oh, ow, kk, r, s, spec.stride, spec.pad,
                                                   spec.dil, 0, wstream), "dass_conv2d_wgrad_x3")
                else:
                    if getattr(ctx, "x_rows_only", False):
                        # the input was produced with sole_consumer=True: its f32 buffer was never written, and this weight gradient
                        # would read it (gradient rows not 16-byte aligned, or the conv engine changed between forward and backward)
                        raise RuntimeError("dass_hip: weight gradient of a conv whose input holds split rows only (sole_consumer=True) cannot "
                                           "take the f32 path -- keep the conv engine fixed between forward and backward, or set DASS_ROWS_ONLY=0")
                    check(lib.dass_conv2d_wgrad_acc(_p(xs), ldx, _p(dy), lddy, _p(dwk), n, h, w, c, oh, ow, kk, r, s,
                                                    spec.stride, spec.pad, spec.dil, _cdt(dy), wstream), "dass_conv2d_wgrad_acc")
                if side is not None:
                    _EV_JOIN.record(side)
                    join = _EV_JOIN
            if want_dx:
                dg_x3 = bool(ctx.x3_dgrad and dt == torch.float32 and c > 32 and kk == k and lddy % 4 == 0)
                w_t = weight_operand(wsrc, 1, dt, x3=dg_x3) if wsrc is weight else _dgrad_operand_uncached(wsrc, dt, x3=dg_x3)
                dx = new_act(n, c, h, w, dt, dev)
                pad_t = spec.dil * (r - 1) - spec.pad
                # forked input: its other gradient rides in as the epilogue's residual (stride-1 launches)
                add_t = add_ld = None
                if d_fork is not None and spec.stride == 1 and d_fork.dtype == dx.dtype:
                    add_t, add_ld = rows(d_fork)
                    if add_ld % 4 != 0:
                        add_t = None
                    else:
                        d_fork = None
                # dgrad = stride-1 conv over dy with flipped/transposed taps; ustride re-inserts the stride
                if dg_x3:
                    dy3 = x3_operand(dy, dy, lddy, m, kk)
                    link = getattr(ctx, "in_link", None)
                    if (link is not None and not link.dead and spec.stride == 1 and d_fork is None and ctx.x_dtype == dx.dtype
                            and x3_parts() <= 2):
                        # dx is the d_out of the layer that produced this conv's input: its BN-backward sums ride in this
                        # launch's epilogue (no separate pass over dx and that layer's conv output)
                        conv_x3_dgrad_bnstats(dy3, w_t, dx, (n, oh, ow, kk, h, w, c, r, s, 1, pad_t, spec.dil), link, residual=add_t,
                                              ldr=add_ld or 0)
                    else:
                        conv_x3_launch(dy3, w_t, dx, c, (n, oh, ow, kk, h, w, c, r, s, 1, pad_t, spec.dil), ustride=spec.stride,
                                       residual=add_t, ldr=add_ld or 0)
                else:
                    check(lib.dass_conv2d_igemm(_p(dy), lddy, _p(w_t), _p(dx), c, None, None, _p(add_t), add_ld or 0, None, n, oh, ow,
                                                kk, h, w, c, r, s, 1, pad_t, spec.dil, spec.stride, ACT_NONE, _cdt(dx),
                                                _stream()), "dass_conv2d_igemm(dgrad)")
            if ctx.needs_input_grad[1] and getattr(spec, "rowtap", False):
                dwk = torch.empty((k, r, s, c_in), dtype=torch.float32, device=dev)
                check(lib.dass_conv2d_rowtap_wgrad(_p(xs), _p(dy), lddy, _p(dwk), n, h, w, c_in, oh, ow, k, r, s, spec.stride,
                                                   spec.pad, _dt(dy), _stream()), "dass_conv2d_rowtap_wgrad")
                dw = dwk.permute(0, 3, 1, 2)
            elif generic_dw:
                if join is not None:
                    torch.cuda.current_stream().wait_event(join)
                dw = dwk[:k, :, :, :c_in].permute(0, 3, 1, 2)
                if kk != k or c != c_in:
                    dw = dw.contiguous(memory_format=torch.channels_last)
        if dx is not None and ctx.x_dtype != dx.dtype:
            dx = dx.to(ctx.x_dtype)
        if d_fork is not None and ctx.needs_input_grad[0]:  # not folded into the launch above
            dx = d_fork if dx is None else dx + d_fork.to(dx.dtype)
        return dx, dw, dgamma, dbeta, dbias, dres, None, None, None


# ---- deferred, grouped weight gradients (csrc/wgrad_x3.hip "GROUPED form").  A conv weight gradient is consumed only by the
# optimizer step, so the pre-split weight-gradient launches of a backward pass are queued and run as ONE grouped launch per tile
# class when the pass ends (autograd's final callback): hundreds of output tiles fill the chip without cutting the 8712-pixel
# reductions into atomically-added pieces, and the fill / drain of ~100 small launches is paid once.  The gradient is then
# written to `weight.grad` (accumulated if one exists) and the parameter's post-accumulate-grad hooks are fired, exactly what
# autograd's AccumulateGrad node would have done; the Function itself returns None for the weight.  DASS_WGRAD_DEFER=0 /
# set_deferred_wgrad(False): per-layer launches inside backward.
_wg = {"on": os.environ.get("DASS_WGRAD_DEFER", "1") == "1", "queue": [], "armed": False, "pending": set(),
       "chunk": int(os.environ.get("DASS_WGRAD_CHUNK", "16")), "side": os.environ.get("DASS_WGRAD_SIDE", "1") == "1",
       "side_capture": os.environ.get("DASS_WGRAD_SIDE_CAPTURE", "1") == "1"}
if not hasattr(torch._C, "_current_graph_task_id"):
    _wg["on"] = False  # without the graph-task id a failed backward pass could not be told from the next one: per-layer launches


def set_deferred_wgrad(on):
    _wg["on"] = bool(on)


def deferred_wgrad():
    return _wg["on"]


def wgrad_pending(p):
    """is the weight gradient of parameter `p` still queued for the grouped launch of the running backward pass?  (autograd may
    run the parameter's AccumulateGrad node -- and its post-accumulate hooks -- with an undefined gradient when a Function
    returns None: hook owners that count gradients in, like dist.GradientAverager, ask here and wait for _wgrad_flush's call)"""
    return id(p) in _wg["pending"]


def set_wgrad_chunk(n, side=None):
    """> 0 (default 16, DASS_WGRAD_CHUNK): the queue of deferred weight gradients is also flushed whenever it holds n layers -- one
    grouped launch per chunk, on a side stream beside the rest of backward (_wgrad_flush) -- and, with more than one process,
    the gradient all-reduce of the early chunks overlaps the rest of the pass instead of starting when it ends.
    0: one launch when the pass ends.  side: also switch the side stream on / off."""
    _wg["chunk"] = max(0, int(n))
    if side is not None:
        _wg["side"] = bool(side)


def _wgrad_enqueue(weight, x3, dy3, dwk, dims, k, c_in):
    task = torch._C._current_graph_task_id() if hasattr(torch._C, "_current_graph_task_id") else None
    if task is not None and task != _wg.get("task"):
        # first deferred gradient of a NEW backward pass.  Whatever is still queued (or in flight on the side stream) belongs to
        # a pass that never reached its final callback (an exception inside backward): drop it BEFORE anything is flushed -- a
        # flush of the stale state would add the failed pass's gradients into .grad after the caller's zero_grad and fire
        # hooks for it -- and arm the callback again for this pass
        _wg["task"] = task
        _wg["queue"], _wg["inflight"], _wg["pending"], _wg["armed"] = [], [], set(), False
    # (tried in round 4 and dropped: one more flush when only a few layers are left, so that less of the last chunk trails the pass --
    #  the 0.5 ms tail moved under the end of backward and slowed that by as much: the chip is busy either way, tools/step_timeline.py)
    if _wg["chunk"] and len(_wg["queue"]) >= _wg["chunk"]:
        # the layers already queued have returned from their backward, and autograd has run their (empty-handed) AccumulateGrad
        # nodes -- those have the highest priority in the engine's ready queue -- so their hooks may fire for real now
        _wgrad_flush(final=False)
    _wg["queue"].append((weight, x3, dy3, dwk, dims, k, c_in))
    _wg["pending"].add(id(weight))
    if not _wg["armed"]:
        _wg["armed"] = True
        torch.autograd.Variable._execution_engine.queue_callback(_wgrad_flush)


def _wgrad_finish(q):
    """a chunk's gradients are complete on the caller's stream: hand them to .grad and fire the parameters' hooks"""
    for item in q:
        _wg["pending"].discard(id(item[0]))
    with torch.no_grad():
        done = {}
        for weight, x3, dy3, dwk, dims, k, c_in in q:
            dw = dwk[:k, :, :, :c_in].permute(0, 3, 1, 2)
            if weight.grad is None:
                weight.grad = dw
            else:
                weight.grad.add_(dw)
            done[id(weight)] = weight
        for weight in done.values():  # (a module applied twice in one forward queued twice: its hooks fire once, after both)
            hooks = getattr(weight, "_post_accumulate_grad_hooks", None)
            if hooks:
                for hook in list(hooks.values()):
                    hook(weight)


def _wgrad_flush(final=True):
    """Runs at the end of the backward pass that queued work (autograd's final callback, on the caller's stream), and for every
    full chunk before that (set_wgrad_chunk).  A chunk flushed DURING backward is launched on a side HIP stream: the grouped
    launch (long workgroups, bound by load latency at 31 % MFMA busy) then shares the chip with the chain of short
    input-gradient and BN launches that backward keeps issuing -- measured 32.7 -> 30.8 ms per R101 step with chunks of 12-20
    layers; every launch has a critical path of one 512-slab workgroup (~1.3 ms), so chunks of 8 or fewer make the side stream
    the bottleneck (41.8 ms).  DASS_WGRAD_SIDE=0: chunks run on the caller's stream.  Its gradients are handed over (.grad, hooks) at the NEXT flush,
    after the caller's stream has waited for the side launch; the final flush hands over everything."""
    import numpy as np

    q = _wg["queue"]
    _wg["queue"] = []
    if final:
        _wg["armed"] = False
    inflight = _wg.setdefault("inflight", [])
    dev = q[0][3].device if q else (inflight[0][1][0][3].device if inflight else None)
    if dev is None:
        return
    main = torch.cuda.current_stream(dev)
    ready, inflight[:] = list(inflight), []
    for ev, qp in ready:          # launched at an earlier flush: long done in practice, the wait is what makes it formal
        main.wait_event(ev)
        _wgrad_finish(qp)
    if not q:
        return
    items = np.zeros((len(q), 16), dtype=np.int64)
    for i, (weight, x3, dy3, dwk, dims, k, c_in) in enumerate(q):
        items[i, :3] = (x3.data_ptr(), dy3.data_ptr(), dwk.data_ptr())
        items[i, 3:15] = dims
    nbytes = lib.dass_conv2d_wgrad_x3_group_scratch_bytes(len(q)) + 128
    scratch = torch.empty((nbytes,), dtype=torch.uint8, device=dev)
    side = None
    # (under stream capture the side stream joins the capture through the fork event and re-joins at the next flush: the graph keeps
    #  the two-stream shape of the eager step; DASS_WGRAD_SIDE_CAPTURE=0: one stream inside a graph)
    if not final and _wg["side"] and (_wg["side_capture"] or not torch.cuda.is_current_stream_capturing()):
        key = ("wgrad", dev.index if dev.index is not None else torch.cuda.current_device())
        side = _mc_side.get(key)
        if side is None:
            # (DASS_WGRAD_SIDE_PRIO: HIP stream priority of the side stream, default 0; a positive value = below the caller's stream,
            #  where the runtime offers such a level)
            prio = int(os.environ.get("DASS_WGRAD_SIDE_PRIO", "0"))
            try:
                side = torch.cuda.Stream(device=dev, priority=prio)
            except Exception:  # noqa: BLE001  (priority outside the device's range)
                side = torch.cuda.Stream(device=dev)
            _mc_side[key] = side
        ev = torch.cuda.Event()
        ev.record(main)           # operands (and the zeroed gradient arena) are complete on the caller's stream
        side.wait_event(ev)
        for weight, x3, dy3, dwk, dims, k, c_in in q:
            for t in (x3, dy3, dwk):
                t.record_stream(side)   # allocated on the caller's stream, read / written by the side launch
        scratch.record_stream(side)
    stream_ptr = ctypes.c_void_p(side.cuda_stream) if side is not None else _stream()
    check(lib.dass_conv2d_wgrad_x3_group(items.ctypes.data_as(ctypes.c_void_p), len(q), _p(scratch), scratch.numel(), stream_ptr),
          "dass_conv2d_wgrad_x3_group")
    if side is not None:
        done = torch.cuda.Event()
        done.record(side)
        inflight.append((done, q))
    else:
        _wgrad_finish(q)


def graph_capture_begin():
    """call right before capturing a step into a hipGraph (dass_hip/graph.py): the zeroed arenas (conv weight gradients, BN f64
    sums) are dropped, so the capture allocates -- and ZEROES, as graph nodes -- fresh ones: every replay starts from zeros"""
    _dw_arena["buf"], _dw_arena["off"] = None, 0
    _bn_sum_arena["buf"], _bn_sum_arena["off"] = None, 0
    flush_bn_counters()


def graph_capture_end():
    """after the capture: eager steps must not cut from arenas that belong to the graph's memory pool"""
    _dw_arena["buf"], _dw_arena["off"] = None, 0
    _bn_sum_arena["buf"], _bn_sum_arena["off"] = None, 0


_dw_arena = {"buf": None, "off": 0, "size": 1 << 21, "on": os.environ.get("DASS_DW_ARENA", "1") == "1"}


def _zeroed_dw(numel, dev):
    """a zeroed f32 slice for one conv weight gradient (the wgrad kernels accumulate with atomics).  Slices are cut from
    an arena cleared by ONE memset instead of one per layer; an arena is never handed out twice -- parameters' .grad
    tensors are views of it and keep it alive -- so gradient accumulation across steps stays correct."""
    a = _dw_arena
    if not a["on"]:
        return torch.zeros((numel,), dtype=torch.float32, device=dev)
    need = (numel + 63) // 64 * 64
    if a["buf"] is None or a["buf"].device != dev or a["off"] + need > a["buf"].numel():
        a["size"] = min(max(2 * a["size"], need), max(1 << 27, need))  # grows to >= one backward pass (R101: 59 M floats)
        a["buf"] = torch.zeros((a["size"],), dtype=torch.float32, device=dev)
        a["off"] = 0
    out = a["buf"][a["off"]:a["off"] + numel]
    a["off"] += need
    return out


_side = {}
_overlap = {"on": os.environ.get("DASS_OVERLAP_WGRAD", "0") == "1"}  # measured: no gain on MI355X (67.6 vs 66.9 ms/step), kept as an experiment switch
_EV_FORK = None
_EV_JOIN = None


def set_overlap_wgrad(on):
    """run conv weight gradients on a side HIP stream next to the input gradient (default off)"""
    _overlap["on"] = bool(on)


def _side_stream(dev):
    global _EV_FORK, _EV_JOIN
    if not _overlap["on"] or torch.cuda.is_current_stream_capturing():
        return None
    key = dev.index if dev.index is not None else torch.cuda.current_device()
    st = _side.get(key)
    if st is None:
        st = _side[key] = torch.cuda.Stream(device=dev)
        _EV_FORK, _EV_JOIN = torch.cuda.Event(), torch.cuda.Event()
    return st


def _dgrad_operand_uncached(wsrc, dtype, x3=False):
    k, c, r, s = wsrc.shape
    master = _krsc_master(wsrc)
    if _split_fmt(dtype, x3):
        return _split6_operand(master, k, r, s, c, c, 1, _split_fmt(dtype, x3))
    op = torch.empty((c, r, s, k), dtype=dtype, device=wsrc.device)
    check(lib.dass_weight_transform(_p(master), _p(op), k, r, s, c, c, 1, F32 if dtype == torch.float32 else BF16,
                                    _stream()), "dass_weight_transform")
    return op


def conv_bn_act(x, conv, bn=None, act=ACT_NONE, residual=None, nc_scale=None, extra_pad=0, image_input=False,
                in_scale=None, emit_x3=True, fork=False, consumer=None, sole_consumer=False, out_into=None):
    """out_into = (wide, offset): write the output into channels [offset, offset + K) of the NHWC buffer `wide` instead of a buffer of
    its own (the ASPP branches write straight into the 1280-wide tensor the merge conv reads: aspp.py:83's torch.cat without the copy).
    in_scale: [N,C] f32 multipliers applied to the INPUT while it is staged (inference only): a
    Dropout2d mask of the producer folded into this conv's loader (MC-dropout tail, SURVEY 8a note iii).
    fork=True -> (out, x'): x' is x again, to be used for the OTHER consumer of x (the identity branch of a residual
    block): both gradients of x then meet in this op's backward and are summed inside the input-gradient launch."""
    spec = ConvSpec(conv, bn, act, extra_pad)
    spec.grad_enabled = torch.is_grad_enabled()
    spec.fork = bool(fork) and spec.grad_enabled and x.requires_grad
    if fork and not spec.fork:
        return conv_bn_act(x, conv, bn, act, residual, nc_scale, extra_pad, image_input, in_scale, emit_x3, consumer=consumer,
                           sole_consumer=sole_consumer), x
    # consumer: the nn.Conv2d that reads this output.  If the train step runs THAT conv on the pre-split kernels
    # (DASS_X3=select: long 3x3 reductions), the BN-apply pass of this layer writes the split rows along with the f32
    # ones (6 more bytes per element) instead of the consumer running a conversion pass (4 read + 6 written)
    spec.x3_consumer = bool(consumer is not None and spec.grad_enabled and consumer.groups == 1
                            and consumer.out_channels > 32
                            and _x3_train_layer(consumer.kernel_size[0] * consumer.kernel_size[1], conv.out_channels))
    spec.emit_x3 = emit_x3  # False where the consumer is not a dense conv (concat / pool / upsample / classifier)
    spec.out_into = out_into
    assert out_into is None or (not fork and not sole_consumer)
    # sole_consumer: `consumer` is the ONLY reader of this output (conv1 -> conv2 -> conv3 inside a bottleneck).  If that conv is sure
    # to take the split rows this layer emits (two-part engine, dense, > 32 output channels, enough output rows for the pre-split
    # kernels), the f32 copy of the output is never read by anyone -- this layer's own backward re-derives its gate from the conv
    # output -- and is NOT WRITTEN: 4 of the 12 bytes per element the BN-apply / fused epilogue pass moves.  The tensor handed back
    # then carries only its attached rows and is flagged; a reader that wants its f32 values fails loudly (x3_operand, _ConvBnAct).
    spec.rows_only = False
    if (sole_consumer and _ROWS_ONLY and consumer is not None and emit_x3 and residual is None and nc_scale is None and act != ACT_NONE
            and compute_dtype() == torch.float32 and x3_parts() <= 2 and x3_pipeline(training=spec.grad_enabled)
            and consumer.groups == 1 and consumer.out_channels > 32 and consumer.out_channels % 4 == 0 and conv.out_channels % 32 == 0
            and not image_input):
        oh = conv_out_size(x.shape[2], conv.kernel_size[0], spec.stride, spec.pad, spec.dil)
        ow = conv_out_size(x.shape[3], conv.kernel_size[1], spec.stride, spec.pad, spec.dil)
        ch = conv_out_size(oh, consumer.kernel_size[0], consumer.stride[0], consumer.padding[0], consumer.dilation[0])
        cw = conv_out_size(ow, consumer.kernel_size[1], consumer.stride[1], consumer.padding[1], consumer.dilation[1])
        spec.rows_only = x.shape[0] * ch * cw >= _X3_MIN_ROWS and x.shape[0] * oh * ow >= _X3_MIN_ROWS and conv.out_channels > 32
    if in_scale is not None:
        assert bn is None or not bn_use_batch_stats(bn), "in_scale needs eval-mode BN"
        spec.in_scale = in_scale.contiguous()
    gamma = bn.weight if bn is not None else None
    beta = bn.bias if bn is not None else None
    res = _ConvBnAct.apply(x, conv.weight, gamma, beta, conv.bias, residual, nc_scale, spec, image_input)
    if spec.fork:
        # the alias of x handed back for the identity branch is a new tensor object over the same memory: let it keep the split
        # rows (and with them the bound of max |x|) the producer attached to x
        src = x.__dict__.get("_dass_alias_of", x) if hasattr(x, "__dict__") else x
        hit = src.__dict__.get("_dass_x3") if hasattr(src, "__dict__") else None
        alias = res[1]
        if hit is not None and hit[0][:2] == (x.data_ptr(), x._version) and alias.data_ptr() == x.data_ptr():
            alias.__dict__["_dass_x3"] = ((alias.data_ptr(), alias._version) + tuple(hit[0][2:]), hit[1])
    return res


# ----------------------------------------------------------------------------- max pool (resnet.py:68)
class _MaxPool3x3s2(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, grad_enabled):
        xs, ld = rows(_cast_act(x))
        n, c, h, w = xs.shape
        if ld != c:
            xs = xs.contiguous(memory_format=torch.channels_last)
        oh, ow = (h + 2 - 3) // 2 + 1, (w + 2 - 3) // 2 + 1
        y = new_act(n, c, oh, ow, xs.dtype, xs.device)
        need = grad_enabled and ctx.needs_input_grad[0]
        idx = torch.empty((n, oh, ow, c), dtype=torch.uint8, device=xs.device) if need else None
        check(lib.dass_maxpool3x3s2_fwd(_p(xs), _p(y), _p(idx), n, h, w, c, oh, ow, _dt(y), _stream()),
              "dass_maxpool3x3s2_fwd")
        if need:
            ctx.save_for_backward(idx)
            ctx.dims = (n, c, h, w, oh, ow)
        return y

    @staticmethod
    def backward(ctx, dy):
        (idx,) = ctx.saved_tensors
        n, c, h, w, oh, ow = ctx.dims
        dyr, ld = rows(_cast_act(dy))
        if ld != c:
            dyr = dyr.contiguous(memory_format=torch.channels_last)
        dx = new_act(n, c, h, w, dyr.dtype, dyr.device)
        check(lib.dass_maxpool3x3s2_bwd(_p(dyr), _p(idx), _p(dx), n, h, w, c, oh, ow, _dt(dx), _stream()),
              "dass_maxpool3x3s2_bwd")
        return dx, None


def maxpool3x3s2(x):
    return _MaxPool3x3s2.apply(x, torch.is_grad_enabled())


# ----------------------------------------------------------------------------- residual add without BN (mobilenet.py:74)
class _Add(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        ar, lda = rows(_cast_act(a))
        br, ldb = rows(_cast_act(b))
        n, c, h, w = ar.shape
        out = new_act(n, c, h, w, ar.dtype, ar.device)
        m = n * h * w
        check(lib.dass_copy_channels(_p(ar), lda, _p(out), c, m, c, _dt(out), _stream()), "dass_copy_channels")
        check(lib.dass_add_channels(_p(br), ldb, _p(out), c, m, c, _dt(out), _stream()), "dass_add_channels")
        return out

    @staticmethod
    def backward(ctx, g):
        return g, g


def add(a, b):
    return _Add.apply(a, b)


# ----------------------------------------------------------------------------- one tensor, several consumers
class _Fanout(torch.autograd.Function):
    """x -> n aliases of x, one per consumer (aspp.py:76-80: five branches read the backbone output; resnet.py:36-44: conv1 and
    the downsample conv read the block input; decoder.py:41: the low-level features feed the decoder AND layer 2).  Forward is
    free; backward adds the n gradients in ONE dass_sum_channels pass instead of autograd's n - 1 at::add launches."""

    @staticmethod
    def forward(ctx, x, n):
        ctx.n = n
        ctx.shape, ctx.dt = x.shape, x.dtype
        return tuple(x.view_as(x) for _ in range(n))

    @staticmethod
    def backward(ctx, *grads):
        gs = [g for g in grads if g is not None]
        if not gs:
            return None, None
        if len(gs) == 1:
            return gs[0], None
        rws = [rows(_cast_act(g)) for g in gs]
        n, c, h, w = rws[0][0].shape
        if c % 4 or any(ld % 4 for _, ld in rws) or len(gs) > 8:
            out = gs[0]
            for g in gs[1:]:
                out = out + g
            return out, None
        dst = new_act(n, c, h, w, rws[0][0].dtype, rws[0][0].device)
        ptrs = (ctypes.c_void_p * len(rws))(*[t.data_ptr() for t, _ in rws])
        lds = (ctypes.c_int64 * len(rws))(*[ld for _, ld in rws])
        check(lib.dass_sum_channels(ptrs, lds, len(rws), _p(dst), c, n * h * w, c, _dt(dst), _stream()), "dass_sum_channels")
        return dst, None


def fanout(x, n):
    """n aliases of x for n consumers (gradients summed by one library pass); a no-op list when autograd is not recording.
    Split rows attached to x, or by the first consumer that converts it, are shared by all aliases."""
    if n <= 1 or not _FANOUT or not (torch.is_grad_enabled() and x.requires_grad):
        return [x] * n
    outs = _Fanout.apply(x, n)
    base = x.__dict__.get("_dass_alias_of", x)  # (an alias of an alias points at the first tensor: one lookup finds its rows)
    for o in outs:
        o.__dict__["_dass_alias_of"] = base
    return list(outs)


# ----------------------------------------------------------------------------- concat (aspp.py:83)
class _Concat(torch.autograd.Function):
    @staticmethod
    def forward(ctx, *xs):
        xs = [rows(_cast_act(x)) for x in xs]
        n, _, h, w = xs[0][0].shape
        cs = [x.shape[1] for x, _ in xs]
        ct = sum(cs)
        out = new_act(n, ct, h, w, xs[0][0].dtype, xs[0][0].device)
        m = n * h * w
        off = 0
        for (x, ld), c in zip(xs, cs):
            dst = out[:, off:off + c]
            check(lib.dass_copy_channels(_p(x), ld, _p(dst), ct, m, c, _dt(out), _stream()), "dass_copy_channels")
            off += c
        ctx.cs = cs
        return out

    @staticmethod
    def backward(ctx, g):
        outs = []
        off = 0
        for c in ctx.cs:
            outs.append(g[:, off:off + c])  # channel-slice views; consumers read them with ld = C_total
            off += c
        return tuple(outs)


def concat(*xs):
    return _Concat.apply(*xs)


class _ConcatShared(torch.autograd.Function):
    """torch.cat over tensors that ALREADY are the channel slices of `wide`, in order (their producers wrote them there:
    conv_bn_act(out_into=...), broadcast_bn(out_into=...)): forward is free, backward hands every producer its slice of the gradient"""

    @staticmethod
    def forward(ctx, wide, *xs):
        off = 0
        for x in xs:
            c = x.shape[1]
            if x.data_ptr() != wide[:, off:off + c].data_ptr() or x.stride() != wide[:, off:off + c].stride():
                raise RuntimeError("dass_hip.concat_shared: input %d is not channel slice [%d, %d) of the shared buffer" % (len(ctx.cs) if hasattr(ctx, "cs") else 0, off, off + c))
            off += c
        if off != wide.shape[1]:
            raise RuntimeError("dass_hip.concat_shared: the slices cover %d of %d channels" % (off, wide.shape[1]))
        ctx.cs = [x.shape[1] for x in xs]
        return wide.view_as(wide)

    @staticmethod
    def backward(ctx, g):
        outs, off = [None], 0
        for c in ctx.cs:
            outs.append(g[:, off:off + c])
            off += c
        return tuple(outs)


def concat_shared(wide, *xs):
    return _ConcatShared.apply(wide, *xs)


# ----------------------------------------------------------------------------- ASPP image-pool branch (aspp.py:62-65,79-81)
class _GlobalAvgPool(torch.autograd.Function):
    """AdaptiveAvgPool2d((1,1)) -> [N,C,1,1]"""

    @staticmethod
    def forward(ctx, x):
        xs, ld = rows(_cast_act(x))
        n, c, h, w = xs.shape
        y = new_act(n, c, 1, 1, xs.dtype, xs.device)
        check(lib.dass_global_avgpool_fwd(_p(xs), ld, _p(y), n, h * w, c, _dt(y), _stream()), "dass_global_avgpool_fwd")
        ctx.dims = (n, c, h, w)
        return y

    @staticmethod
    def backward(ctx, g):
        n, c, h, w = ctx.dims
        gr, _ = rows(_cast_act(g))
        gr = gr.reshape(n, c).contiguous()
        dx = new_act(n, c, h, w, gr.dtype, gr.device)
        check(lib.dass_broadcast_rows(_p(gr), _p(dx), c, n, h * w, c, 1.0 / (h * w), _dt(dx), _stream()),
              "dass_broadcast_rows")
        return dx


def global_avgpool(x):
    return _GlobalAvgPool.apply(x)


class _BroadcastBN(torch.autograd.Function):
    """bilinear 1x1 -> HxW (a broadcast) followed by BatchNorm2d over the broadcast map.
    Batch statistics over N*H*W copies equal statistics over N rows with count N*H*W (only the
    unbiased running_var correction sees H*W), so BN runs on the [N,C] vector and is broadcast.
    The rows are post-ReLU and N is the batch size: statistics and the backward run two-pass in f64
    (dass_bn_rows_fwd / _bwd); only SyncBN, which must exchange sums, goes through the sum / sum-of-squares kernels."""

    @staticmethod
    def forward(ctx, x, gamma, beta, bn, h, w, out_into=None):
        xs, _ = rows(_cast_act(x))
        n, c = xs.shape[0], xs.shape[1]
        xv = xs.reshape(n, c).contiguous()
        train = bn_use_batch_stats(bn)
        world = sync_bn_world(bn) if train else 1
        rows_f64 = xv.dtype == torch.float32 and world == 1
        if train and rows_f64:
            st = BNState(c, xv.device)
            mom, rm, rv = -1.0, None, None
            if bn.track_running_stats and bn.running_mean is not None:
                mom = 0.1 if bn.momentum is None else float(bn.momentum)
                rm, rv = bn.running_mean, bn.running_var
                _pending_counters.append(bn.num_batches_tracked)
                if _defer["depth"] == 0 or len(_pending_counters) >= 512:
                    flush_bn_counters()
            check(lib.dass_bn_rows_fwd(_p(xv), n, c, float(h * w), _p(bn.weight), _p(bn.bias), _p(rm), _p(rv), mom, float(bn.eps),
                                       _p(st.mean), _p(st.invstd), _p(st.scale), _p(st.shift), _stream()), "dass_bn_rows_fwd")
            _running_stats_written(bn, rm, rv)
        elif train:
            st = bn_train_state(xv, c, n, c, bn, rep=float(h * w))
        else:
            st = bn_eval_state(bn, c, xv.device)
        yv = torch.empty_like(xv)
        scale_shift_act(xv, c, yv, c, n, c, st.scale, st.shift)
        if out_into is not None:
            wide, off = out_into
            out = wide[:, off:off + c]
            ldo = wide.shape[1]
        else:
            out = new_act(n, c, h, w, xv.dtype, xv.device)
            ldo = c
        check(lib.dass_broadcast_rows(_p(yv), _p(out), ldo, n, h * w, c, 1.0, _dt(out), _stream()), "dass_broadcast_rows")
        ctx.save_for_backward(xv, yv, gamma, st.mean, st.invstd)
        ctx.dims = (n, c, h, w)
        ctx.train_stats = train
        ctx.sync_world = world
        ctx.rows_f64 = rows_f64
        return out

    @staticmethod
    def backward(ctx, g):
        xv, yv, gamma, mean, invstd = ctx.saved_tensors
        n, c, h, w = ctx.dims
        gr, ldg = rows(_cast_act(g))
        gs = torch.empty((n, c), dtype=gr.dtype, device=gr.device)
        check(lib.dass_reduce_rows(_p(gr), ldg, _p(gs), n, h * w, c, _dt(gs), _stream()), "dass_reduce_rows")
        if ctx.rows_f64:
            dx = torch.empty((n, c), dtype=torch.float32, device=gs.device)
            sums = torch.empty((2, c), dtype=torch.float32, device=gs.device)
            check(lib.dass_bn_rows_bwd(_p(gs), _p(xv), _p(mean), _p(invstd), _p(gamma.detach()), n, c, 1 if ctx.train_stats else 0,
                                       _p(dx), _p(sums[1]), _p(sums[0]), _stream()), "dass_bn_rows_bwd")
            return dx.view(n, 1, 1, c).permute(0, 3, 1, 2), sums[1], sums[0], None, None, None, None
        nrows = lib.dass_stat_rows(n)
        partial = torch.empty((nrows, 2, c), dtype=torch.float32, device=gr.device)
        check(lib.dass_bn_bwd_reduce(_p(gs), c, _p(yv), c, _p(xv), c, _p(mean), _p(invstd), None, n, c, 1, ACT_NONE,
                                     _p(partial), _dt(gs), _stream()), "dass_bn_bwd_reduce")
        sums = torch.empty((2, c), dtype=torch.float32, device=gr.device)
        check(lib.dass_bn_bwd_finalize(_p(partial), nrows, c, _p(sums[0]), _p(sums[1]), _stream()), "dass_bn_bwd_finalize")
        dgamma, dbeta = sums[1], sums[0]
        if ctx.train_stats and ctx.sync_world > 1:
            dgamma, dbeta = sums[1].clone(), sums[0].clone()  # parameter gradients stay local sums (see _ConvBnAct.backward)
            _allreduce_bn_grads(sums, ctx.sync_world)
        dx = torch.empty((n, c), dtype=gs.dtype, device=gs.device)
        check(lib.dass_bn_bwd_apply(_p(gs), c, _p(yv), c, _p(xv), c, _p(mean), _p(invstd), _p(gamma.detach()),
                                    _p(sums[0]), _p(sums[1]), None, _p(dx), c, None, 0, n, c, 1, float(n) * ctx.sync_world,
                                    1 if ctx.train_stats else 0, ACT_NONE, _dt(dx), None, _stream()), "dass_bn_bwd_apply")
        return dx.view(n, 1, 1, c).permute(0, 3, 1, 2), dgamma, dbeta, None, None, None, None


def broadcast_bn(x, bn, h, w, out_into=None):
    return _BroadcastBN.apply(x, bn.weight, bn.bias, bn, h, w, out_into)


# ----------------------------------------------------------------------------- bilinear (align_corners=True)
class _UpsampleCat(torch.autograd.Function):
    """F.interpolate(x, low.size()[2:], bilinear, align_corners=True); torch.cat((x, low), 1)
    (decoder.py:45-46): the resampled rows are written straight into the 304-channel buffer."""

    @staticmethod
    def forward(ctx, x, low):
        xs, ldx = rows(_cast_act(x))
        ls, ldl = rows(_cast_act(low))
        n, c, ih, iw = xs.shape
        _, cl, oh, ow = ls.shape
        ct = c + cl
        out = new_act(n, ct, oh, ow, xs.dtype, xs.device)
        check(lib.dass_bilinear_fwd(_p(xs), ldx, _p(out), ct, n, ih, iw, c, oh, ow, 0, _dt(out), _stream()),
              "dass_bilinear_fwd")
        check(lib.dass_copy_channels(_p(ls), ldl, _p(out[:, c:]), ct, n * oh * ow, cl, _dt(out), _stream()),
              "dass_copy_channels")
        ctx.dims = (n, c, ih, iw, cl, oh, ow)
        return out

    @staticmethod
    def backward(ctx, g):
        n, c, ih, iw, cl, oh, ow = ctx.dims
        gr, ldg = rows(_cast_act(g))
        dx = new_act(n, c, ih, iw, gr.dtype, gr.device)
        check(lib.dass_bilinear_bwd(_p(gr), ldg, _p(dx), c, n, ih, iw, c, oh, ow, 0, _dt(dx), _stream()),
              "dass_bilinear_bwd")
        return dx, gr[:, c:]


def upsample_cat(x, low):
    return _UpsampleCat.apply(x, low)


class _UpsampleToNCHW(torch.autograd.Function):
    """final F.interpolate(low_res_logits, size=input HW, bilinear, align_corners=True) (deeplab.py:59):
    NHWC low-res logits in, NCHW f32 logits out (what the reference returns)."""

    @staticmethod
    def forward(ctx, x, oh, ow):
        xs, ldx = rows(_cast_act(x))
        n, c, ih, iw = xs.shape
        y = torch.empty((n, c, oh, ow), dtype=torch.float32, device=xs.device)
        check(lib.dass_bilinear_fwd(_p(xs), ldx, _p(y), 0, n, ih, iw, c, oh, ow, 1, _dt(xs), _stream()),
              "dass_bilinear_fwd")
        ctx.dims = (n, c, ih, iw, oh, ow)
        ctx.dt = xs.dtype
        return y

    @staticmethod
    def backward(ctx, g):
        n, c, ih, iw, oh, ow = ctx.dims
        g = g.contiguous().float()
        cp = _pad_to(c, _epv(ctx.dt))
        dx = zeros_act(n, cp, ih, iw, ctx.dt, g.device) if cp != c else new_act(n, c, ih, iw, ctx.dt, g.device)
        check(lib.dass_bilinear_bwd(_p(g), 0, _p(dx), cp, n, ih, iw, c, oh, ow, 1, _dt(dx), _stream()),
              "dass_bilinear_bwd")
        return dx[:, :c], None, None


def upsample_to_nchw(x, oh, ow):
    return _UpsampleToNCHW.apply(x, oh, ow)


class _Upsample(torch.autograd.Function):
    """generic NHWC -> NHWC bilinear align_corners resample"""

    @staticmethod
    def forward(ctx, x, oh, ow):
        xs, ldx = rows(_cast_act(x))
        n, c, ih, iw = xs.shape
        y = new_act(n, c, oh, ow, xs.dtype, xs.device)
        check(lib.dass_bilinear_fwd(_p(xs), ldx, _p(y), c, n, ih, iw, c, oh, ow, 0, _dt(y), _stream()), "dass_bilinear_fwd")
        ctx.dims = (n, c, ih, iw, oh, ow)
        return y

    @staticmethod
    def backward(ctx, g):
        n, c, ih, iw, oh, ow = ctx.dims
        gr, ldg = rows(_cast_act(g))
        dx = new_act(n, c, ih, iw, gr.dtype, gr.device)
        check(lib.dass_bilinear_bwd(_p(gr), ldg, _p(dx), c, n, ih, iw, c, oh, ow, 0, _dt(dx), _stream()), "dass_bilinear_bwd")
        return dx, None, None


def upsample(x, oh, ow):
    return _Upsample.apply(x, oh, ow)


# ----------------------------------------------------------------------------- Dropout2d as a channel mask
def dropout2d_mask(n, c, p, device, generator=None):
    """[N,C] f32 multipliers {0, 1/(1-p)} -- the exact values nn.Dropout2d applies per (n,c) plane."""
    keep = torch.rand((n, c), device=device, generator=generator) >= p
    return keep.to(torch.float32) * (1.0 / (1.0 - p))


class _ChannelScale(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, mask):
        xs, ld = rows(_cast_act(x))
        n, c, h, w = xs.shape
        out = new_act(n, c, h, w, xs.dtype, xs.device)
        scale_shift_act(xs, ld, out, c, n * h * w, c, None, None, nc_scale=mask, rows_per_image=h * w)
        ctx.save_for_backward(mask)
        return out

    @staticmethod
    def backward(ctx, g):
        (mask,) = ctx.saved_tensors
        gr, ld = rows(_cast_act(g))
        n, c, h, w = gr.shape
        dx = new_act(n, c, h, w, gr.dtype, gr.device)
        scale_shift_act(gr, ld, dx, c, n * h * w, c, None, None, nc_scale=mask, rows_per_image=h * w)
        return dx, None


def channel_scale(x, mask):
    return _ChannelScale.apply(x, mask)


# ----------------------------------------------------------------------------- cross entropy (utils/loss.py:39-51)
class _CrossEntropy(torch.autograd.Function):
    """nn.CrossEntropyLoss(weight, ignore_index, reduction='mean')(logit, target.long())"""

    @staticmethod
    def forward(ctx, logit, target, weight, ignore_index, mean):
        _require_cuda(logit)
        lg = logit.contiguous().float()
        n, c, h, w = lg.shape
        tgt = target.contiguous()
        is_float = 1 if tgt.dtype == torch.float32 else 0
        if not is_float and tgt.dtype != torch.int64:
            tgt = tgt.long()
        wt = weight.contiguous().float() if weight is not None else None
        hw = h * w
        blocks = lib.dass_ce_blocks(n * hw)
        partial = torch.empty((blocks, 2), dtype=torch.float32, device=lg.device)
        acc = torch.empty((2,), dtype=torch.float32, device=lg.device)
        check(lib.dass_ce_fwd(_p(lg), _p(tgt), is_float, _p(wt), n, c, hw, int(ignore_index), _p(partial), _stream()),
              "dass_ce_fwd")
        check(lib.dass_ce_finalize(_p(partial), blocks, _p(acc), _stream()), "dass_ce_finalize")
        ctx.meta = (n, c, hw, is_float, int(ignore_index))
        if mean:
            ctx.save_for_backward(lg, tgt, wt, acc)
            return acc[0] / acc[1]
        norm = torch.ones((2,), dtype=torch.float32, device=lg.device)  # backward divides by norm[1] = 1
        ctx.save_for_backward(lg, tgt, wt, norm)
        count = acc[1].clone()
        ctx.mark_non_differentiable(count)
        return acc[0].clone(), count

    @staticmethod
    def backward(ctx, g, *unused):
        lg, tgt, wt, acc = ctx.saved_tensors
        n, c, hw, is_float, ignore = ctx.meta
        gs = g.detach().float().reshape(1).contiguous()
        d = torch.empty_like(lg)
        check(lib.dass_ce_bwd(_p(lg), _p(tgt), is_float, _p(wt), n, c, hw, ignore, _p(acc), _p(gs), _p(d), _stream()),
              "dass_ce_bwd")
        return d, None, None, None, None


def cross_entropy(logit, target, weight=None, ignore_index=255):
    return _CrossEntropy.apply(logit, target, weight, ignore_index, True)


def cross_entropy_sum(logit, target, weight=None, ignore_index=255):
    """sum over valid pixels of w[t]*nll (reduction='none' summed), used by the sample-weighted loss"""
    return _CrossEntropy.apply(logit, target, weight, ignore_index, False)[0]


def cross_entropy_parts(logit, target, weight=None, ignore_index=255):
    """-> (sum over valid pixels of w[t]*nll  [autograd], sum of w[t] over valid pixels [no grad]): the numerator and
    denominator of reduction='mean', kept apart so that ranks can exchange them (utils/loss.py under DDP)"""
    return _CrossEntropy.apply(logit, target, weight, ignore_index, False)


# ----------------------------------------------------------------------------- scoring launches (no autograd)
def upsample_argmax(low, oh, ow, votes, t):
    """votes[:, t] = argmax_c bilinear(low)[.., c]  (votes: uint8 [N,T,OH,OW])"""
    xs, ldx = rows(low)
    n, c, ih, iw = xs.shape
    tt = votes.shape[1]
    dst = votes[:, t]
    check(lib.dass_upsample_argmax(_p(xs), ldx, _p(dst), tt * oh * ow, n, ih, iw, c, oh, ow, _dt(xs), _stream()),
          "dass_upsample_argmax")


def argmax_nchw(logits, votes, t):
    lg = logits.contiguous().float()
    n, c, h, w = lg.shape
    tt = votes.shape[1]
    check(lib.dass_argmax_nchw(_p(lg), _p(votes[:, t]), tt * h * w, n, c, h * w, _stream()), "dass_argmax_nchw")


def vote_entropy(votes, label, num_classes, want_map=True):
    """-> (entropy_map [N,H,W] f32 or None, image_mean [N] f32) ; mc_dropout.py:43-49,189"""
    n, t, h, w = votes.shape
    dev = votes.device
    lab = label.contiguous().float() if label is not None else None
    emap = torch.empty((n, h, w), dtype=torch.float32, device=dev) if want_map else None
    partial = torch.empty((n, lib.dass_score_blocks()), dtype=torch.float32, device=dev)
    sums = torch.empty((n,), dtype=torch.float32, device=dev)
    check(lib.dass_vote_entropy(_p(votes), _p(lab), n, t, h * w, num_classes, _p(emap), _p(partial), _p(sums), _stream()),
          "dass_vote_entropy")
    return emap, sums / float(h * w)


def softmax_scores(logits, label, num_classes, mode, want_map=False):
    lg = logits.contiguous().float()
    n, c, h, w = lg.shape
    dev = lg.device
    lab = label.contiguous().float() if label is not None else None
    smap = torch.empty((n, h, w), dtype=torch.float32, device=dev) if want_map else None
    partial = torch.empty((n, lib.dass_score_blocks()), dtype=torch.float32, device=dev)
    sums = torch.empty((n,), dtype=torch.float32, device=dev)
    check(lib.dass_softmax_scores(_p(lg), _p(lab), n, c, h * w, num_classes, mode, _p(smap), _p(partial), _p(sums),
                                  _stream()), "dass_softmax_scores")
    return smap, sums / float(h * w)


def weak_labels(logits, label, num_classes):
    lg = logits.contiguous().float()
    n, c, h, w = lg.shape
    out = torch.empty((n, h, w), dtype=torch.uint8, device=lg.device)
    check(lib.dass_weak_labels(_p(lg), _p(label.contiguous().float()), n, c, h * w, num_classes, _p(out), _stream()),
          "dass_weak_labels")
    return out


def avgpool_features(feat, k, s):
    """F.avg_pool2d(feat, k, s).flatten(1) with the reference's channel-major order (core_set.py:61-63)"""
    xs, ld = rows(feat)
    n, c, h, w = xs.shape
    ph, pw = (h - k) // s + 1, (w - k) // s + 1
    out = torch.empty((n, c * ph * pw), dtype=torch.float32, device=xs.device)
    check(lib.dass_avgpool_features(_p(xs), ld, _p(out), n, h, w, c, k, s, ph, pw, _dt(xs), _stream()),
          "dass_avgpool_features")
    return out


def kcenter_greedy(features, selected, count):
    """k-center greedy (core_set.py:17-38) fully enqueued on the device: returns int64 [count] picks."""
    feats = features.contiguous().float()
    n, d = feats.shape
    dev = feats.device
    min_dist = torch.empty((n,), dtype=torch.float64, device=dev)
    nb = lib.dass_argmax_blocks(n)
    pval = torch.empty((nb,), dtype=torch.float64, device=dev)
    pidx = torch.empty((nb,), dtype=torch.int64, device=dev)
    picks = torch.empty((max(count, 1),), dtype=torch.int64, device=dev)
    centers = torch.as_tensor(list(selected), dtype=torch.int64, device=dev)
    for i in range(centers.numel()):
        check(lib.dass_kcenter_update(_p(feats), n, d, _p(centers[i:i + 1]), _p(min_dist), 1 if i == 0 else 0,
                                      _stream()), "dass_kcenter_update")
    if centers.numel() == 0:
        min_dist.fill_(float("inf"))
    for j in range(count):
        check(lib.dass_argmax_f64(_p(min_dist), n, _p(pval), _p(pidx), _p(picks[j:j + 1]), None, _stream()),
              "dass_argmax_f64")
        check(lib.dass_kcenter_update(_p(feats), n, d, _p(picks[j:j + 1]), _p(min_dist), 0, _stream()),
              "dass_kcenter_update")
    return picks[:count], min_dist


def kcenter_update(feats, centers, min_dist):
    """min_dist[i] = min(min_dist[i], ||f_i - f_c||) for every c in centers, in place (f64; core_set.py:32-38)"""
    idx = torch.as_tensor(list(centers), dtype=torch.int64, device=feats.device)
    n, d = feats.shape
    for i in range(idx.numel()):
        check(lib.dass_kcenter_update(_p(feats), n, d, _p(idx[i:i + 1]), _p(min_dist), 0, _stream()), "dass_kcenter_update")
    return min_dist


def box_sum(maps, r):
    n, h, w = maps.shape
    out = torch.empty((n, h - r + 1, w - r + 1), dtype=torch.float32, device=maps.device)
    tmp = torch.empty((n, h, w - r + 1), dtype=torch.float32, device=maps.device)
    check(lib.dass_box_sum(_p(maps.contiguous()), _p(out), _p(tmp), n, h, w, r, _stream()), "dass_box_sum")
    return out


def zero_rect(maps, i, r0, r1, c0, c1):
    n, h, w = maps.shape
    r0, c0, r1, c1 = max(r0, 0), max(c0, 0), min(r1, h), min(c1, w)
    written_in_place(maps)
    check(lib.dass_zero_rect(_p(maps), i, h, w, r0, r1, c0, c1, _stream()), "dass_zero_rect")


def minmax(maps):
    """-> device tensor [min, max] over every element (an empty tensor gives [+inf, -inf], the identities of min / max)"""
    nel = maps.numel()
    if nel == 0:
        return torch.tensor([float("inf"), float("-inf")], dtype=torch.float32, device=maps.device)
    partial = torch.empty((lib.dass_minmax_blocks(nel), 2), dtype=torch.float32, device=maps.device)
    mm = torch.empty((2,), dtype=torch.float32, device=maps.device)
    check(lib.dass_minmax(_p(maps), nel, _p(partial), _p(mm), _stream()), "dass_minmax")
    return mm


def minmax_normalize_(maps, mm=None):
    """x.add_(-min).mul_(1/(max-min)) over ALL maps (mc_dropout.py:152-155); mm: [min, max] computed elsewhere (the
    GLOBAL extrema when the maps are one rank's shard of the pool)"""
    if mm is None:
        mm = minmax(maps)
    if maps.numel():
        check(lib.dass_affine_inplace(_p(maps), maps.numel(), _p(mm), _stream()), "dass_affine_inplace")
    return mm


def square_nms(score_maps, region_size, max_selection_count):
    """mc_dropout.py:82-108 on the device; returns (selected_regions per image, selection_count)"""
    n, h, w = score_maps.shape
    dev = score_maps.device
    max_picks = int(math.ceil(max_selection_count))
    imax = torch.empty((n,), dtype=torch.float32, device=dev)
    iarg = torch.empty((n,), dtype=torch.int32, device=dev)
    picks = torch.zeros((max(max_picks, 1), 3), dtype=torch.int32, device=dev)
    count = torch.zeros((1,), dtype=torch.int32, device=dev)
    if max_picks > 0:
        check(lib.dass_square_nms(_p(score_maps), n, h, w, int(region_size), max_picks, _p(imax), _p(iarg), _p(picks),
                                  _p(count), _stream()), "dass_square_nms")
    cnt = int(count.item())
    regions = [[] for _ in range(n)]
    for i, r, c in picks[:cnt].tolist():
        regions[i].append((r, c, region_size, region_size))
    return regions, cnt


def confusion_accumulate(cm, target, pred_or_logits, num_class):
    """cm[gt][pred] += 1 on the device (utils/metrics.py:37-42); pred_or_logits: [N,H,W] class map or [N,C,H,W] logits"""
    tgt = target.contiguous().float()
    if pred_or_logits.dim() == 4:
        lg = pred_or_logits.contiguous().float()
        n, c, h, w = lg.shape
        check(lib.dass_confusion_accumulate(_p(lg), None, _p(tgt), n, c, h * w, num_class, _p(cm), _stream()),
              "dass_confusion_accumulate")
    else:
        pr = pred_or_logits.contiguous().to(torch.uint8)
        n, h, w = pr.shape
        check(lib.dass_confusion_accumulate(None, _p(pr), _p(tgt), n, 0, h * w, num_class, _p(cm), _stream()),
              "dass_confusion_accumulate")


def max_representative(all_features, candidate_features, count):
    """greedy facility location (max_subset.py:17-39) on the device: each pick maximises
    -sum_i min(mind_i, D[i][j]) over unselected candidates (first max wins, like the reference's strict '>')."""
    a = all_features.contiguous().float()
    b = candidate_features.contiguous().float()
    n, d = a.shape
    m = b.shape[0]
    dev = a.device
    dist = torch.empty((n, m), dtype=torch.float64, device=dev)
    check(lib.dass_pairwise_dist_f64(_p(a), n, _p(b), m, d, _p(dist), _stream()), "dass_pairwise_dist_f64")
    mind = torch.full((n,), float("inf"), dtype=torch.float64, device=dev)
    selected = torch.zeros((m,), dtype=torch.uint8, device=dev)
    scores = torch.empty((m,), dtype=torch.float64, device=dev)
    nb = lib.dass_argmax_blocks(m)
    pval = torch.empty((nb,), dtype=torch.float64, device=dev)
    pidx = torch.empty((nb,), dtype=torch.int64, device=dev)
    picks = torch.empty((max(count, 1),), dtype=torch.int64, device=dev)
    for j in range(count):
        check(lib.dass_facility_scores(_p(dist), n, m, _p(mind), _p(selected), _p(scores), _stream()), "dass_facility_scores")
        check(lib.dass_argmax_f64(_p(scores), m, _p(pval), _p(pidx), _p(picks[j:j + 1]), None, _stream()), "dass_argmax_f64")
        check(lib.dass_facility_update(_p(dist), n, m, _p(picks[j:j + 1]), _p(mind), _p(selected), _stream()),
              "dass_facility_update")
    return picks[:count]


def add_noise_(x, std, generator=None):
    """x += N(0, std) (deeplab.py:39-56, mc_noise.py:24-25): torch draws the gaussians, the add is dass_add_channels"""
    xs, ld = rows(x)
    n, c, h, w = xs.shape
    noise = torch.randn((n, h, w, c), device=xs.device, dtype=torch.float32, generator=generator).mul_(std).to(xs.dtype)
    if c % 4 == 0 and ld % 4 == 0:
        check(lib.dass_add_channels(_p(noise), c, _p(xs), ld, n * h * w, c, _dt(xs), _stream()), "dass_add_channels")
        written_in_place(xs)  # (split rows attached to the un-noised tensor are stale now)
        if xs is not x:
            written_in_place(x)
        return xs
    return xs + noise.permute(0, 3, 1, 2)

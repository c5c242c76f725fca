"""ctypes binding of libdass_hip.so (the C-ABI declared in include/dass_hip.h).

Prototypes are parsed from the header itself so the Python side can never drift from the ABI:
pointers -> c_void_p, int64_t -> c_int64, int -> c_int, float -> c_float, double -> c_double.
There is NO fallback: if the shared library is missing the import fails loudly with the build hint
(`python -c "import __graft_entry__ as g; g.build()"` or `make -C deep-active-semantic-segmentation_amd/csrc`).
"""
import ctypes
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(os.path.dirname(_HERE))
HEADER = os.path.join(_ROOT, "include", "dass_hip.h")
LIB_PATH = os.environ.get("DASS_HIP_LIB", os.path.join(_HERE, "libdass_hip.so"))  # DASS_HIP_LIB: an instrumented debug build

_CTYPES = {
    "int": ctypes.c_int,
    "int64_t": ctypes.c_int64,
    "float": ctypes.c_float,
    "double": ctypes.c_double,
}


ARG_NAMES = {}  # {entry point: [parameter names]} (filled by parse_header; KernelTimer finds shapes and the stream by name)


def parse_header(path=HEADER):
    """-> {name: (restype, [argtypes])} for every function prototype in the header."""
    src = open(path).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    protos = {}
    ARG_NAMES.clear()
    for m in re.finditer(r"\b(int64_t|int|const char \*)\s*(dass_\w+)\s*\(([^;{]*?)\)\s*;", src, flags=re.S):
        ret, name, args = m.group(1), m.group(2), m.group(3).strip()
        argtypes = []
        names = ARG_NAMES.setdefault(name, [])
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                names.append(re.sub(r"[\s*]", " ", a).split()[-1])
                if "*" in a:
                    argtypes.append(ctypes.c_void_p)
                else:
                    base = a.replace("const", "").split()[0]
                    argtypes.append(_CTYPES[base])
        restype = {"int": ctypes.c_int, "int64_t": ctypes.c_int64}.get(ret, ctypes.c_char_p)
        protos[name] = (restype, argtypes)
    return protos


PROTOTYPES = parse_header()


def load(path=LIB_PATH):
    if not os.path.exists(path):
        raise ImportError(
            "dass_hip: %s not found -- the HIP extension is mandatory (no CPU fallback). Build it with "
            "`make -C %s` (hipcc --offload-arch=gfx950)." % (path, os.path.join(os.path.dirname(_HERE), "csrc"))
        )
    lib = ctypes.CDLL(path)
    for name, (restype, argtypes) in PROTOTYPES.items():
        fn = getattr(lib, name)  # AttributeError here = header/library mismatch: fail loudly
        fn.restype = restype
        fn.argtypes = argtypes
    return lib


lib = load()

ERRORS = {1: "DASS_ERR_ARG (bad shape / alignment / null pointer)", 2: "DASS_ERR_LAUNCH", 3: "DASS_ERR_UNSUPPORTED"}


def check(rc, name):
    if rc != 0:
        raise RuntimeError("libdass_hip: %s failed: %s" % (name, ERRORS.get(rc, rc)))


# ----------------------------------------------------------------------------- in-step kernel timing (bench.py `roofline`)
CONV_ENTRY_POINTS = ("dass_conv2d_igemm", "dass_conv2d_igemm_stats", "dass_conv2d_igemm_sums", "dass_conv2d_wgrad", "dass_conv2d_wgrad_acc",
                     "dass_conv2d_rowtap", "dass_conv2d_rowtap_wgrad", "dass_conv2d_x3", "dass_conv2d_x3_sums", "dass_conv2d_x3_dgrad_bnstats",
                     "dass_conv2d_x3_per_image", "dass_conv2d_wgrad_x3", "dass_conv2d_wgrad_x3_group")
BN_ENTRY_POINTS = ("dass_bn_apply_train", "dass_bn_bwd_apply_sums", "dass_bn_bwd_reduce_sums")


def _val(a):
    return a.value if hasattr(a, "value") else a


def _conv_gflop(name, args):
    """2 x output rows x K x R x S x C of the conv this call computes (input rows for a strided input gradient), from its arguments"""
    idx = {n: i for i, n in enumerate(ARG_NAMES[name])}
    if name == "dass_conv2d_wgrad_x3_group":
        import numpy as np

        n = int(_val(args[idx["n"]]))
        ptr = ctypes.cast(args[idx["items"]], ctypes.POINTER(ctypes.c_int64 * (16 * n)))
        it = np.frombuffer(ptr.contents, dtype=np.int64).reshape(n, 16)
        nn, h, w, c, oh, ow, k, r, s = (it[:, 3 + j].astype(np.float64) for j in range(9))
        return float((2.0 * nn * oh * ow * k * r * s * c).sum() / 1e9)
    g = lambda key: float(_val(args[idx[key]]))  # noqa: E731
    us = g("ustride") if "ustride" in idx else 1.0
    out_rows = g("N") * (g("H") * g("W") if us > 1 else g("OH") * g("OW"))
    return 2.0 * out_rows * g("K") * g("R") * g("S") * g("C" if "C" in idx else "Cin") / 1e9


def _bn_gbytes(name, args):
    """ALGORITHMIC bytes of one train-mode BN pass in f32: every [M][K] tensor the call reads or writes once, 4 B per element
    (two f16 parts of a split-row output: 4 B as well), gate bits 1 B per 4 channels; a null pointer = that tensor is not touched"""
    idx = {n: i for i, n in enumerate(ARG_NAMES[name])}
    has = lambda key: key in idx and _val(args[idx[key]]) not in (None, 0)  # noqa: E731
    e = float(_val(args[idx["M"]])) * float(_val(args[idx["K"]]))
    tensors = {"dass_bn_apply_train": ("x", "out", "residual", "out3"),
               "dass_bn_bwd_apply_sums": ("dout", "out", "x", "dx", "dres", "dx3"),
               "dass_bn_bwd_reduce_sums": ("dout", "out", "x")}[name]
    b = sum(4.0 * e for t in tensors if has(t))
    if has("gates"):
        b += e / 4.0
    return b / 1e9


class KernelTimer(object):
    """Kernel-exact timing of real steps through the library's own profile (include/dass_hip.h dass_prof_*): while the context is
    open every kernel launch of libdass_hip carries a start / stop HIP event pair bound to that dispatch, ON THE STREAM IT IS
    LAUNCHED ON (the chunked weight gradients run on a side stream; torch.cuda.Event would only see torch's current stream).

        with KernelTimer() as kt:
            train_step()
        kernels, calls = kt.results()
        # kernels: [(kernel name, ms, workgroups, stream)] for EVERY launch, in launch order -- a rocprofv3 kernel trace, live
        # calls:   [(entry point, tag, work, ms, [kernel names])] per call of a conv / BN entry point: ms = sum of the kernels
        #          that call enqueued (a phase-decomposed input gradient is up to four, a stream-K conv has its fix-up pass);
        #          work = GFLOP (conv: 2 x output rows x K x R x S x C from the call's own arguments) or GB (BN passes);
        #          tag = the pre-split conv kernel's tile class (dass_x3_last_pick)

    Launches on two streams overlap in time; durations are summed, not merged (as in a rocprofv3 kernel table)."""

    def __init__(self, conv=CONV_ENTRY_POINTS, bn=BN_ENTRY_POINTS):
        self.work = {n: _conv_gflop for n in conv if n in PROTOTYPES}
        self.work.update({n: _bn_gbytes for n in bn if n in PROTOTYPES})
        self.calls = []
        self.shapes = []  # per call: (N, OH, OW, C, K, R) of a conv entry point, else None (diagnostics: tools/step_shapes.py)
        self.saved = {}

    def _wrap(self, name, fn):
        pick = lib.dass_x3_last_pick if ("x3" in name and "wgrad" not in name and "conv2d" in name) else None
        work = self.work[name]

        idx = {n: i for i, n in enumerate(ARG_NAMES[name])}
        dims = [idx[k] for k in ("N", "OH", "OW", "C", "K", "R") if k in idx]

        def timed(*args):
            i0 = lib.dass_prof_count()
            rc = fn(*args)
            self.calls.append((name, pick() if pick is not None else 0, work(name, args), i0, lib.dass_prof_count()))
            self.shapes.append(tuple(int(_val(args[i])) for i in dims) if len(dims) == 6 else None)   # (N, OH, OW, C, K, R) of a conv call
            return rc

        return timed

    def __enter__(self):
        check(lib.dass_prof_begin(), "dass_prof_begin")
        for n in self.work:
            self.saved[n] = getattr(lib, n)
            setattr(lib, n, self._wrap(n, self.saved[n]))
        return self

    def __exit__(self, *exc):
        lib.dass_prof_end()
        for n, fn in self.saved.items():
            setattr(lib, n, fn)
        self.saved = {}

    def restart(self):
        """forget what was recorded so far (e.g. an untimed first step) and keep recording"""
        self.calls = []
        self.shapes = []
        check(lib.dass_prof_begin(), "dass_prof_begin")

    def results(self):
        n = lib.dass_prof_count()
        buf = ctypes.create_string_buffer(512)
        ms, grid, st = ctypes.c_float(0.0), ctypes.c_int64(0), ctypes.c_void_p()
        kernels = []
        for i in range(n):
            check(lib.dass_prof_get(i, buf, 512, ctypes.byref(ms), ctypes.byref(grid), ctypes.byref(st)), "dass_prof_get")
            kernels.append((buf.value.decode("ascii", "replace"), float(ms.value), int(grid.value), st.value or 0))
        calls = [(name, tag, work, sum(k[1] for k in kernels[i0:i1]), [k[0] for k in kernels[i0:i1]]) for name, tag, work, i0, i1 in self.calls]
        return kernels, calls

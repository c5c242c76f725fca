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


class KernelTimer(object):
    """HIP events around every call of the named C-ABI entry points, recorded ON THE STREAM EACH CALL LAUNCHES ON (its last
    argument) -- torch.cuda.Event would only see torch's current stream, and the chunked weight gradients run on a side stream.

        with KernelTimer(CONV_ENTRY_POINTS) as kt:
            train_step()
        rows = kt.rows()   # [(entry point, tag, gflop, ms)] per call, after one device synchronize

    A call's duration is stop-event minus start-event on its stream: the kernels that call enqueued (a phase-decomposed input
    gradient is up to four, a stream-K conv has its fix-up pass), plus whatever the stream waited for in between -- nothing, for
    launches that are issued back to back.  Launches on two streams overlap in time; their durations are summed, not merged
    (the same convention as a rocprofv3 kernel table).  gflop = 2 x output rows x K x R x S x C of the conv the call computes
    (input rows for a strided input gradient), from the call's own arguments.  tag = the pre-split kernel's tile pick of that
    call (dass_x3_last_pick) where the library reports one."""

    def __init__(self, names=CONV_ENTRY_POINTS):
        self.names = [n for n in names if n in PROTOTYPES]
        self.hip = ctypes.CDLL("libamdhip64.so")
        self.hip.hipEventCreate.argtypes = [ctypes.POINTER(ctypes.c_void_p)]
        self.hip.hipEventRecord.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        self.hip.hipEventSynchronize.argtypes = [ctypes.c_void_p]
        self.hip.hipEventElapsedTime.argtypes = [ctypes.POINTER(ctypes.c_float), ctypes.c_void_p, ctypes.c_void_p]
        self.hip.hipEventDestroy.argtypes = [ctypes.c_void_p]
        self.calls = []
        self.saved = {}

    def _event(self):
        ev = ctypes.c_void_p()
        if self.hip.hipEventCreate(ctypes.byref(ev)) != 0:
            raise RuntimeError("hipEventCreate failed")
        return ev

    @staticmethod
    def _val(a):
        return a.value if hasattr(a, "value") else a

    def _gflop(self, name, args):
        idx = {n: i for i, n in enumerate(ARG_NAMES[name])}
        if name == "dass_conv2d_wgrad_x3_group":
            import numpy as np

            n = int(self._val(args[idx["n"]]))
            ptr = ctypes.cast(args[idx["items"]], ctypes.POINTER(ctypes.c_int64 * (16 * n)))
            it = np.frombuffer(ptr.contents, dtype=np.int64).reshape(n, 16)
            nn, h, w, c, oh, ow, k, r, s = (it[:, 3 + j].astype(np.float64) for j in range(9))
            return float((2.0 * nn * oh * ow * k * r * s * c).sum() / 1e9)
        g = lambda key: float(self._val(args[idx[key]]))  # noqa: E731
        us = g("ustride") if "ustride" in idx else 1.0
        out_rows = g("N") * (g("H") * g("W") if us > 1 else g("OH") * g("OW"))
        return 2.0 * out_rows * g("K") * g("R") * g("S") * g("C" if "C" in idx else "Cin") / 1e9

    def _wrap(self, name, fn):
        pick = getattr(lib, "dass_x3_last_pick", None) if "x3" in name and "wgrad" not in name else None

        def timed(*args):
            stream = self._val(args[-1])
            e0, e1 = self._event(), self._event()
            self.hip.hipEventRecord(e0, stream)
            rc = fn(*args)
            self.hip.hipEventRecord(e1, stream)
            self.calls.append((name, pick() if pick is not None else 0, self._gflop(name, args), e0, e1))
            return rc

        return timed

    def __enter__(self):
        for n in self.names:
            self.saved[n] = getattr(lib, n)
            setattr(lib, n, self._wrap(n, self.saved[n]))
        return self

    def __exit__(self, *exc):
        for n, fn in self.saved.items():
            setattr(lib, n, fn)
        self.saved = {}

    def rows(self):
        out = []
        ms = ctypes.c_float(0.0)
        for name, tag, gflop, e0, e1 in self.calls:
            self.hip.hipEventSynchronize(e1)
            if self.hip.hipEventElapsedTime(ctypes.byref(ms), e0, e1) != 0:
                raise RuntimeError("hipEventElapsedTime failed")
            out.append((name, tag, gflop, float(ms.value)))
            self.hip.hipEventDestroy(e0)
            self.hip.hipEventDestroy(e1)
        self.calls = []
        return out

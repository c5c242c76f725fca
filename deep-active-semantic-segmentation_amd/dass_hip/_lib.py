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


def parse_header(path=HEADER):
    """-> {name: (restype, [argtypes])} for every function prototype in the header."""
    src = open(path).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    protos = {}
    for m in re.finditer(r"\b(int64_t|int|const char \*)\s*(dass_\w+)\s*\(([^;{]*?)\)\s*;", src, flags=re.S):
        ret, name, args = m.group(1), m.group(2), m.group(3).strip()
        argtypes = []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
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

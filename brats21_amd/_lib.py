"""ctypes binding of libbrats_hip.so (the C ABI declared in include/brats_hip.h).

There is NO fallback: if the shared library is missing the import of any compute entry point fails
loudly (``BratsHipError``).  Build it with ``python -c "import __graft_entry__ as g; g.build()"`` or
``make -C brats21_amd/csrc -j8``.
"""
import ctypes
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("BRATS_HIP_LIB") or os.path.join(_HERE, "libbrats_hip.so")  # override: A/B builds only
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "brats_hip.h")

F32, BF16, F16, X3_BF16, X3_F16 = 0, 1, 2, 3, 4
ACT_NONE, ACT_RELU, ACT_LEAKY, ACT_ELU, ACT_SWISH, ACT_MISH = 0, 1, 2, 3, 4, 5
PACK_FWD, PACK_DGRAD = 0, 1


class BratsHipError(RuntimeError):
    pass


_C = {"p": ctypes.c_void_p, "i": ctypes.c_int, "f": ctypes.c_float, "d": ctypes.c_double, "z": ctypes.c_size_t}

def _parse_header():
    """Derive every prototype's ctypes signature from include/brats_hip.h itself, so the binding can
    never drift from the declared ABI: p = pointer / stream, i = int, f = float, d = double, z = size_t."""
    txt = open(HEADER_PATH).read()
    txt = re.sub(r"/\*.*?\*/", " ", txt, flags=re.S)
    sigs = {}
    for m in re.finditer(r"(const char\s*\*|size_t|int)\s+(brats_[a-z0-9_]+)\s*\(([^)]*)\)\s*;", txt):
        ret, name, args = m.group(1), m.group(2), m.group(3).strip()
        spec = ""
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                if "*" in a or "brats_stream_t" in a:
                    spec += "p"
                elif a.startswith("size_t"):
                    spec += "z"
                elif a.startswith("double"):
                    spec += "d"
                elif a.startswith("float"):
                    spec += "f"
                elif a.startswith("int"):
                    spec += "i"
                else:
                    raise BratsHipError(f"cannot bind argument '{a}' of {name}")
        sigs[name] = ("s" if "char" in ret else ("z" if ret == "size_t" else "i"), spec)
    return sigs


_lib = None


def declared_symbols():
    """Every function name declared in include/brats_hip.h (used by the CPU symbol-export test)."""
    return sorted(_parse_header().keys())


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise BratsHipError(
            f"{LIB_PATH} is missing: the HIP extension has not been built (run __graft_entry__.build()). "
            "brats21_amd has no CPU / PyTorch fallback on purpose.")
    l = ctypes.CDLL(LIB_PATH)
    override = bool(os.environ.get("BRATS_HIP_LIB"))
    for name, (res, spec) in _parse_header().items():
        try:
            fn = getattr(l, name)
        except AttributeError:
            if not override:
                raise  # the in-tree library must export every declared symbol (tests/test_abi_cpu.py)
            # an OLDER build loaded for a same-box A/B (scripts/ab.sh): entry points it predates fail when called
            def missing(*a, _n=name, **k):
                raise BratsHipError(f"{LIB_PATH} (BRATS_HIP_LIB override) does not export {_n}")
            setattr(l, name, missing)
            continue
        fn.restype = ctypes.c_char_p if res == "s" else _C[res]
        fn.argtypes = [_C[c] for c in spec]
    # the binding above is derived from THIS tree's header: a library of another ABI version (an older A/B build loaded
    # through BRATS_HIP_LIB whose entry points changed signature) would be called with the wrong argument lists
    want = _header_abi_version()
    got = l.brats_abi_version()
    if got != want:
        raise BratsHipError(f"{LIB_PATH} reports ABI version {got}, include/brats_hip.h declares {want}: rebuild the library "
                            "(an A/B library must come from a tree with the same entry-point signatures)")
    _lib = l
    return l


def _header_abi_version():
    """``#define BRATS_ABI_VERSION N`` of include/brats_hip.h (the one place the version is written; abi.hip returns it)."""
    m = re.search(r"^#define\s+BRATS_ABI_VERSION\s+(\d+)\s*$", open(HEADER_PATH).read(), flags=re.M)
    if not m:
        raise BratsHipError(f"{HEADER_PATH} does not define BRATS_ABI_VERSION")
    return int(m.group(1))


def check(rc, what):
    if rc != 0:
        msg = lib().brats_last_error()
        raise BratsHipError(f"{what} failed ({rc}): {msg.decode() if msg else '?'}")

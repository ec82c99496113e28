"""ctypes binding of libsimpledet_ops_hip.so (the C ABI in include/simpledet_ops.h).

The argtypes are generated from the header itself, so the header is the single source of truth for
the boundary.  There is NO CPU fallback: if the shared library is missing or a symbol is absent the
import fails loudly.
"""
import ctypes
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
HEADER = os.path.join(os.path.dirname(_HERE), "include", "simpledet_ops.h")
# SIMPLEDET_AMD_LIB: load another build of the same ABI (tools/ use the -DSD_PROFILING build)
LIB_PATH = os.environ.get("SIMPLEDET_AMD_LIB") or os.path.join(_HERE, "libsimpledet_ops_hip.so")

_SCALARS = {
    "int": ctypes.c_int,
    "long": ctypes.c_long,
    "float": ctypes.c_float,
    "double": ctypes.c_double,
    "size_t": ctypes.c_size_t,
    "int32_t": ctypes.c_int32,
    "uint32_t": ctypes.c_uint32,
    "uint8_t": ctypes.c_uint8,
    "int64_t": ctypes.c_int64,
    "uint64_t": ctypes.c_uint64,
    "unsigned": ctypes.c_uint,
}


def parse_header(path=HEADER):
    """Return {symbol: (restype, [(argname, ctype)])} for every function the header declares."""
    src = open(path).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    src = re.sub(r"//[^\n]*", "", src)
    src = re.sub(r"^\s*#.*$", "", src, flags=re.M)
    protos = {}
    for m in re.finditer(r"([A-Za-z_][\w\s\*]*?)\b(sd_\w+)\s*\(([^;{}]*?)\)\s*;", src, flags=re.S):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()
        if "*" in ret:
            restype = ctypes.c_char_p if "char" in ret else ctypes.c_void_p
        else:
            restype = _SCALARS.get(ret.replace("const", "").strip(), ctypes.c_int)
        argl = []
        if args and args != "void":
            for a in args.split(","):
                a = " ".join(a.split())
                mm = re.match(r"(.*?)(\w+)$", a)
                typ, an = mm.group(1).strip(), mm.group(2)
                if "*" in typ:
                    ct = ctypes.c_char_p if re.match(r"(const\s+)?char\s*\*$", typ) else ctypes.c_void_p
                else:
                    ct = _SCALARS[typ.replace("const", "").strip()]
                argl.append((an, ct))
        protos[name] = (restype, argl)
    return protos


def header_abi_version(path=HEADER):
    m = re.search(r"^\s*#define\s+SD_ABI_VERSION\s+(\d+)", open(path).read(), flags=re.M)
    if not m:
        raise ImportError("SD_ABI_VERSION missing from %s" % path)
    return int(m.group(1))


class SimpleDetOpsError(RuntimeError):
    """code: the SD_ERR_* value the entry point returned (include/simpledet_ops.h)."""
    code = 0


SD_ERR_UNSUPPORTED = -2


class _Lib:
    def __init__(self):
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "simpledet_amd: %s not found. Build it with `python -c 'import __graft_entry__ as g; "
                "g.build()'` or `make -C simpledet_amd/csrc` (needs hipcc, gfx950). There is no CPU "
                "fallback." % LIB_PATH)
        # torch ships its own libamdhip64.so.7; import it first so this library binds to the SAME
        # HIP runtime (one runtime per process: streams and device pointers are shared with torch)
        try:
            import torch  # noqa: F401
        except Exception:  # pragma: no cover - torch is optional for pure-C users
            pass
        self.cdll = ctypes.CDLL(LIB_PATH)
        self.protos = parse_header()
        for name, (restype, args) in self.protos.items():
            try:
                fn = getattr(self.cdll, name)
            except AttributeError as e:
                raise ImportError("simpledet_amd: %s does not export %s (header/library mismatch)"
                                  % (LIB_PATH, name)) from e
            fn.restype = restype
            fn.argtypes = [t for _, t in args]
        # buffer layouts and size contracts are part of the ABI: a library older or newer than the
        # header this wrapper marshals for is refused
        want = header_abi_version()
        got = int(self.cdll.sd_abi_version())
        if got != want:
            raise ImportError("simpledet_amd: %s has ABI version %d, include/simpledet_ops.h %d: rebuild "
                              "(`make -C simpledet_amd/csrc`)" % (LIB_PATH, got, want))

    def call(self, name, *args):
        """Call an int-returning entry point; raise SimpleDetOpsError on a non-zero code."""
        rc = getattr(self.cdll, name)(*args)
        if rc != 0:
            msg = self.cdll.sd_last_error()
            err = SimpleDetOpsError("%s failed (%d): %s" % (name, rc, (msg or b"").decode()))
            err.code = int(rc)
            raise err
        return rc

    def set_tuning(self, key, value):
        self.call("sd_set_tuning", key.encode(), int(value))

    def get_tuning(self, key):
        """value of a kernel-variant knob, -1 when it was never set (= the library's default)."""
        v = ctypes.c_int(-1)
        self.call("sd_get_tuning", key.encode(), ctypes.byref(v))
        return int(v.value)


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = _Lib()
    return _lib

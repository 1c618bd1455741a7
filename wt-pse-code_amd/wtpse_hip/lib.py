"""ctypes binding of libwtpse_hip.so (the C ABI declared in include/wtpse_hip.h).

The prototypes are read from the header itself, so the header is the single source of truth for the
boundary.  There is no fallback: if the library is missing or a call fails, this raises.
"""
import ctypes
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libwtpse_hip.so")
HEADER_PATH = os.path.normpath(os.path.join(_HERE, "..", "..", "include", "wtpse_hip.h"))

_CTYPES = {
    "int": ctypes.c_int,
    "long long": ctypes.c_longlong,
    "unsigned long long": ctypes.c_ulonglong,
    "float": ctypes.c_float,
    "double": ctypes.c_double,
}


class WtpseError(RuntimeError):
    pass


def parse_header(path=HEADER_PATH):
    """-> {name: [ctypes arg types]} for every `int wtpse_*(...)` declaration."""
    text = open(path).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    protos = {}
    for m in re.finditer(r"\bint\s+(wtpse_\w+)\s*\(([^)]*)\)\s*;", text):
        name, args = m.group(1), m.group(2)
        types = []
        for a in args.split(","):
            a = " ".join(a.split())
            if not a or a == "void":
                continue
            if "*" in a:
                types.append(ctypes.c_void_p)
                continue
            base = a.rsplit(" ", 1)[0].replace("const ", "").strip()
            types.append(_CTYPES[base])
        protos[name] = types
    return protos


class _Lib:
    def __init__(self):
        if not os.path.isfile(LIB_PATH):
            raise WtpseError(
                "libwtpse_hip.so is missing (%s). Build it with `python -c 'import __graft_entry__ as g; g.build()'`; "
                "the WT-PSE MI355X path has no CPU fallback." % LIB_PATH)
        # When a GPU is present, let torch create its HIP context first: loading a HIP code object into a process
        # whose runtime has not been initialised yet left the first launch with hipErrorNoDevice (100) on the GPU box.
        try:
            import torch
            if torch.cuda.is_available():
                torch.cuda.init()
        except ImportError:
            pass
        self._dll = ctypes.CDLL(LIB_PATH)
        # a library built from other sources than the ones in the tree (stale .so after an edit, changed signature)
        # would be a silent ABI mismatch: only missing symbols are detected by the binding below
        from . import build
        fn = getattr(self._dll, "wtpse_source_hash", None)
        have = None
        if fn is not None:
            fn.restype = ctypes.c_char_p
            have = fn().decode()
        want = build.source_hash()
        if have != want:
            raise WtpseError("libwtpse_hip.so was built from different sources (library %s, tree %s): rebuild it with "
                             "`python -c 'import __graft_entry__ as g; g.build()'`" % (have, want))
        self.protos = parse_header()
        for name, argtypes in self.protos.items():
            fn = getattr(self._dll, name)        # AttributeError if the header and the library disagree
            fn.argtypes = argtypes
            fn.restype = ctypes.c_int
            setattr(self, "_raw_" + name, fn)

    def raw(self, name):
        return getattr(self, "_raw_" + name)

    def call(self, name, *args):
        rc = getattr(self, "_raw_" + name)(*args)
        if rc != 0:
            raise WtpseError("%s failed with status %d%s" % (name, rc, " (invalid argument)" if rc == -1 else " (hipError_t)"))

    def query(self, name, *args):
        """For the int-valued sizing helpers (wtpse_*_blocks / _ksplit / _nsplit / _split)."""
        return getattr(self, "_raw_" + name)(*args)


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = _Lib()
    return _lib

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


class PlanArg(ctypes.Union):
    """One argument slot of a recorded call (csrc/plan.hip)."""
    _fields_ = [("p", ctypes.c_void_p), ("i", ctypes.c_longlong), ("u", ctypes.c_ulonglong), ("d", ctypes.c_double)]


_SLOT = {ctypes.c_void_p: "p", ctypes.c_int: "i", ctypes.c_longlong: "i", ctypes.c_ulonglong: "u", ctypes.c_float: "d",
         ctypes.c_double: "d"}


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
        # launch plans: entry points that can be recorded, by name
        self._dll.wtpse_plan_create.restype = ctypes.c_void_p
        self._dll.wtpse_plan_fn_name.restype = ctypes.c_char_p
        self._plan_fn = {self._dll.wtpse_plan_fn_name(i).decode(): i for i in range(self._dll.wtpse_plan_fn_count())}
        self._rec = None                         # the plan being recorded (a c_void_p value) or None

    # ---- launch plans (csrc/plan.hip) ----------------------------------------------------------------------------
    def plan_begin(self):
        """Start recording every call() into a new plan (the calls are still issued: under stream capture they only enter
        the capture).  -> opaque plan handle."""
        assert self._rec is None, "a plan is already being recorded"
        self._rec = self._dll.wtpse_plan_create()
        return self._rec

    def plan_end(self):
        plan, self._rec = self._rec, None
        return plan

    def plan_wait(self, waiter, waited):
        """Record: stream `waiter` waits for stream `waited` (raw hipStream_t values).  No-op when not recording."""
        if self._rec is not None and waiter != waited:
            rc = self._raw_wtpse_plan_add_wait(self._rec, waiter, waited)
            if rc:
                raise WtpseError("wtpse_plan_add_wait failed with status %d" % rc)

    def plan_replay(self, plan):
        rc = self._raw_wtpse_plan_replay(plan)
        if rc == -2:
            raise WtpseError("wtpse_plan_replay refused: wtpse_x3_terms / wtpse_x3r_enable / wtpse_x3_xcd changed since the plan was "
                             "recorded (the recorded buffer sizes and packed weights belong to the old setting): record a new plan")
        if rc:
            raise WtpseError("wtpse_plan_replay failed with status %d" % rc)

    def plan_destroy(self, plan):
        self._raw_wtpse_plan_destroy(plan)

    def _record(self, name, args):
        types = self.protos[name]
        n = len(types) - 1                       # the trailing stream is stored separately
        arr = (PlanArg * max(n, 1))()
        for k in range(n):
            v = args[k]
            setattr(arr[k], _SLOT[types[k]], 0 if v is None else v)
        rc = self._raw_wtpse_plan_add_call(self._rec, self._plan_fn[name], arr, n, args[n])
        if rc:
            raise WtpseError("cannot record %s (status %d)" % (name, rc))

    def raw(self, name):
        return getattr(self, "_raw_" + name)

    def call(self, name, *args):
        if self._rec is not None and name in self._plan_fn:
            self._record(name, args)
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

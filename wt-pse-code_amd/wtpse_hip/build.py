"""Builds libwtpse_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
import glob
import hashlib
import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
OUT = os.path.join(_HERE, "libwtpse_hip.so")


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


HEADER = os.path.normpath(os.path.join(_HERE, "..", "..", "include", "wtpse_hip.h"))


def source_hash():
    """Fingerprint of everything the library is compiled from plus the header ctypes reads its prototypes from.  It is
    compiled into the library (`wtpse_source_hash()`), so a stale .so — e.g. one built before a signature changed — is
    recognised by content, not by file times (which a snapshot copy to the GPU box does not preserve)."""
    h = hashlib.sha256()
    for f in sources() + sorted(glob.glob(os.path.join(CSRC, "*.h"))) + [HEADER]:
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:32]


def built_hash():
    """The fingerprint stored next to the library by build() (None: no library)."""
    try:
        with open(OUT + ".hash") as f:
            return f.read().strip()
    except OSError:
        return None


def needs_build():
    return not os.path.isfile(OUT) or built_hash() != source_hash()


def build(force=False, verbose=False):
    if not force and not needs_build():
        return OUT
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    digest = source_hash()
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wno-unused-result",
           "-DWTPSE_SRC_HASH=\"%s\"" % digest, "-I", CSRC] + sources() + ["-o", OUT + ".tmp"]
    if verbose:
        print(" ".join(cmd))
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + res.stdout + res.stderr)
    os.replace(OUT + ".tmp", OUT)
    with open(OUT + ".hash", "w") as f:
        f.write(digest + "\n")
    return OUT


if __name__ == "__main__":
    print(build(force=True, verbose=True))

"""Build libnerfhip.so (hand-written HIP for gfx950) in-tree with hipcc.

    python -m nerf_pl_amd.build            # incremental
    python -m nerf_pl_amd.build --force

No torch headers, no hipify, no cmake: each csrc/*.hip is compiled to an object with
`hipcc --offload-arch=gfx950` (in parallel) and linked into nerf_pl_amd/libnerfhip.so, which
exports the plain-C ABI declared in include/nerfhip.h.  hipcc cross-compiles without a GPU.
"""
import concurrent.futures
import hashlib
import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
OBJ = os.path.join(PKG, "build")
LIB = os.path.join(PKG, "libnerfhip.so")
ARCH = "gfx950"
FLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-Wall", "-Wno-unused-function"]


def _hipcc():
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (need ROCm >= 7.0 for gfx950)")


def _sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _deps_digest():
    h = hashlib.sha256()
    for d in (CSRC, os.path.join(ROOT, "include")):
        for f in sorted(os.listdir(d)):
            if f.endswith((".h", ".hip")):
                h.update(f.encode())
                with open(os.path.join(d, f), "rb") as fh:
                    h.update(fh.read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def _compile(src):
    obj = os.path.join(OBJ, os.path.basename(src)[:-4] + ".o")
    cmd = [_hipcc()] + FLAGS + ["-c", src, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed: %s\n%s\n%s" % (" ".join(cmd), r.stdout, r.stderr))
    return obj


def build(force=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    stamp = os.path.join(OBJ, "digest.txt")
    digest = _deps_digest()
    if not force and os.path.exists(LIB) and os.path.exists(stamp) and open(stamp).read() == digest:
        return LIB
    srcs = _sources()
    if verbose:
        print("[nerf_pl_amd.build] hipcc %s: %d sources" % (ARCH, len(srcs)), flush=True)
    with concurrent.futures.ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        objs = list(ex.map(_compile, srcs))
    cmd = [_hipcc(), "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + objs
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed: %s\n%s" % (" ".join(cmd), r.stderr))
    with open(stamp, "w") as f:
        f.write(digest)
    if verbose:
        print("[nerf_pl_amd.build] wrote", LIB, flush=True)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)

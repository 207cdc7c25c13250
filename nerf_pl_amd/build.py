"""Build libnerfhip.so (hand-written HIP for gfx950) in-tree with hipcc.

    python -m nerf_pl_amd.build            # incremental
    python -m nerf_pl_amd.build --force

No torch headers, no hipify, no cmake: each csrc/*.hip is compiled to an object with
`hipcc --offload-arch=gfx950` (in parallel) and linked into nerf_pl_amd/libnerfhip.so, which
exports the plain-C ABI declared in include/nerfhip.h.  hipcc cross-compiles without a GPU.

Objects are rebuilt individually: an object's stamp is the hash of its source, of the headers it
(transitively) includes and of its flags.  `mlp_fwd_variant.hip` is compiled once per kernel
instantiation (-DNH_PREC/-DNH_MODE/-DNH_VARIANT): the 12 fully unrolled forward kernels take ~12
minutes in one translation unit and ~3 when built side by side.
"""
import concurrent.futures
import hashlib
import os
import re
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
TAG = os.environ.get("NERFHIP_BUILD_TAG", "")          # A/B kernel builds: objects in build_<tag>/, library in variants/
OBJ = os.path.join(PKG, "build" + ("_" + TAG if TAG else ""))
LIB = os.path.join(PKG, "variants", "libnerfhip_%s.so" % TAG) if TAG else os.path.join(PKG, "libnerfhip.so")
ARCH = "gfx950"
FLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-Wall", "-Wno-unused-function"]
EXTRA = [f for f in os.environ.get("NERFHIP_EXTRA_FLAGS", "").split() if f]     # A/B kernel builds (-DNERFHIP_...=)

# (source, object suffix, extra -D flags): sources compiled more than once
# The fp8-storage activation-saving forward (variant 3) sits exactly at the 256-register limit of its 2-waves-per-SIMD
# launch bounds: hipcc's default scheduling strategy spills 25-37 VGPRs there, `max-memory-clause` 2 (the other variants do not
# spill under either).  NERFHIP_V3_SCHED= (empty) builds it with the default strategy (A/B).
V3_SCHED = os.environ.get("NERFHIP_V3_SCHED", "max-memory-clause")
MULTI = {"mlp_fwd_variant.hip": [("_p%dm%dv%d" % (p, m, v), ["-DNH_PREC=%d" % p, "-DNH_MODE=%d" % m, "-DNH_VARIANT=%d" % v]
                                  + (["-mllvm", "-amdgpu-sched-strategy=" + V3_SCHED] if (v == 3 and V3_SCHED) else []))
                                 for p in (0, 1) for m in (0, 1) for v in (0, 1, 2, 3) if not (v == 3 and p == 0)],
         # the single-launch render kernels: inference / training forward / test_time inference (sv 3), per precision (e4m3 storage:
         # bf16 only, same scheduling note)
         "mlp_render_variant.hip": [("_p%ds%d" % (p, sv), ["-DNH_PREC=%d" % p, "-DNH_SV=%d" % sv]
                                     + (["-mllvm", "-amdgpu-sched-strategy=" + V3_SCHED] if (sv == 2 and V3_SCHED) else []))
                                    for p in (0, 1) for sv in (0, 1, 2, 3) if not (sv == 2 and p == 0)]}


# per-source extra flags.  The backward chain at its 256-register budget: hipcc's default strategy spills 30-56 VGPRs in the
# fp8-storage variant, `max-memory-clause` none (and the code shrinks from 79.8 to 61.0 KB, under the 64 KiB instruction cache).
CHAIN_SCHED = os.environ.get("NERFHIP_CHAIN_SCHED", "max-memory-clause")        # empty = hipcc's default strategy (A/B)
PER_FILE = {"mlp_bwd_chain.hip": ["-mllvm", "-amdgpu-sched-strategy=" + CHAIN_SCHED] if CHAIN_SCHED else []}


def _hipcc():
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (need ROCm >= 7.0 for gfx950)")


def _jobs():
    """[(source path, object path, flags)]"""
    out = []
    for f in sorted(os.listdir(CSRC)):
        if not f.endswith(".hip"):
            continue
        src = os.path.join(CSRC, f)
        for suffix, defs in MULTI.get(f, [("", [])]):
            out.append((src, os.path.join(OBJ, f[:-4] + suffix + ".o"), FLAGS + EXTRA + defs + PER_FILE.get(f, [])))
    return out


_INC = re.compile(r'^\s*#\s*include\s+"([^"]+)"', re.M)


def _closure(path, seen):
    """The file and every quoted include below it (relative to the including file)."""
    path = os.path.normpath(path)
    if path in seen or not os.path.exists(path):
        return
    seen.add(path)
    with open(path) as fh:
        text = fh.read()
    for inc in _INC.findall(text):
        _closure(os.path.join(os.path.dirname(path), inc), seen)


def _digest(src, flags):
    files = set()
    _closure(src, files)
    h = hashlib.sha256()
    for f in sorted(files):
        h.update(os.path.relpath(f, ROOT).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    h.update(" ".join(flags).encode())
    return h.hexdigest()


def source_digest():
    """One digest of everything the MLP kernels are built from (csrc/mlp_*, the headers they include, the flags): measurements
    taken on one build (profiles/pmc_traffic.json: HBM traffic of the MLP kernels) are stamped with it, and bench.py refuses
    counters stamped with another."""
    h = hashlib.sha256()
    files = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC))
             if f.startswith("mlp_") or f in ("common.h", "f8_store.h", "adam_math.h")]
    for f in files:
        h.update(os.path.relpath(f, ROOT).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    h.update(" ".join(FLAGS + [V3_SCHED] + [x for v in PER_FILE.values() for x in v]).encode())
    return h.hexdigest()


def _compile(job):
    src, obj, flags = job
    cmd = [_hipcc()] + flags + ["-c", src, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed: %s\n%s\n%s" % (" ".join(cmd), r.stdout, r.stderr))
    with open(obj + ".stamp", "w") as f:
        f.write(_digest(src, flags))
    return obj


def build(force=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    jobs = _jobs()
    todo = []
    for job in jobs:
        src, obj, flags = job
        stamp = obj + ".stamp"
        if force or not (os.path.exists(obj) and os.path.exists(stamp) and open(stamp).read() == _digest(src, flags)):
            todo.append(job)
    objs = [j[1] for j in jobs]
    link_stamp = os.path.join(OBJ, "link.stamp")
    link_key = hashlib.sha256("\n".join(open(o + ".stamp").read() if os.path.exists(o + ".stamp") else "?" for o in objs).encode()).hexdigest()
    if not todo and os.path.exists(LIB) and os.path.exists(link_stamp) and open(link_stamp).read() == link_key:
        return LIB
    if verbose:
        print("[nerf_pl_amd.build] hipcc %s: %d of %d objects to compile" % (ARCH, len(todo), len(jobs)), flush=True)
    # the long forward-kernel objects first, so they overlap everything else
    todo.sort(key=lambda j: 0 if "mlp_fwd_variant" in j[0] else 1)
    with concurrent.futures.ThreadPoolExecutor(max_workers=max(1, min(os.cpu_count() or 4, 16, len(todo) or 1))) as ex:
        list(ex.map(_compile, todo))
    for o in os.listdir(OBJ):                          # objects of sources that no longer exist must not be linked
        if o.endswith(".o") and os.path.join(OBJ, o) not in objs:
            os.remove(os.path.join(OBJ, o))
    cmd = [_hipcc(), "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + objs
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed: %s\n%s" % (" ".join(cmd), r.stderr))
    link_key = hashlib.sha256("\n".join(open(o + ".stamp").read() for o in objs).encode()).hexdigest()
    with open(link_stamp, "w") as f:
        f.write(link_key)
    if verbose:
        print("[nerf_pl_amd.build] wrote", LIB, flush=True)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)

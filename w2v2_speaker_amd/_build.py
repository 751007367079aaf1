"""Build libw2v2hip.so (gfx950) in-tree with hipcc.  Used by ``__graft_entry__.build()`` and by the
loader when the library is missing and a compiler is present.  No fallback of any kind: if the build
fails the product path raises."""
from __future__ import annotations

import os
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libw2v2hip.so")
SOURCES = ["api.hip", "gemm.hip", "gemm_ring.hip", "gemm_phased.hip", "gemm_f32.hip", "gemm_f32_dma.hip", "wgrad.hip", "wgrad_phased.hip", "norm.hip", "elementwise.hip", "conv0.hip", "posconv.hip", "posconv_direct.hip", "posconv_wgrad.hip",
           "softmax.hip", "attention.hip", "pool.hip", "asp.hip", "tdnn.hip", "skinny.hip", "heads.hip", "optim.hip", "comm.hip"]
HEADERS = ["common.h", "gemm_common.h", "wgrad_common.h"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result"]


def _hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: cannot build libw2v2hip.so")
    return exe


def source_hash() -> str:
    """sha256 over the kernel sources + the C header (what the profiled counters under profiles/ were measured on)."""
    import hashlib
    h = hashlib.sha256()
    for f in sorted(SOURCES) + HEADERS:
        h.update(open(os.path.join(CSRC, f), "rb").read())
    h.update(open(os.path.join(HERE, "..", "include", "w2v2_hip.h"), "rb").read())
    return h.hexdigest()[:16]


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.join(HERE, "..", "include", "w2v2_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build_locked(force: bool = False, verbose: bool = False) -> str:
    """build() under an exclusive file lock: concurrent ranks of a data-parallel job build once."""
    import fcntl
    os.makedirs(os.path.join(HERE, "build"), exist_ok=True)
    with open(os.path.join(HERE, "build", ".lock"), "w") as lk:
        fcntl.flock(lk, fcntl.LOCK_EX)
        try:
            return build(force=force, verbose=verbose)
        finally:
            fcntl.flock(lk, fcntl.LOCK_UN)


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not needs_build():
        return LIB
    import time
    t_start = time.time()
    hipcc = _hipcc()
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)

    def compile_one(src: str) -> str:
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        srcp = os.path.join(CSRC, src)
        hdrs = [os.path.join(CSRC, "common.h"), os.path.join(HERE, "..", "include", "w2v2_hip.h")]
        if src.startswith("gemm"):
            hdrs.append(os.path.join(CSRC, "gemm_common.h"))
        if src.startswith("wgrad"):
            hdrs.append(os.path.join(CSRC, "wgrad_common.h"))
        if (not force and os.path.exists(obj)
                and os.path.getmtime(obj) > max(os.path.getmtime(p) for p in [srcp] + hdrs)):
            return obj
        cmd = [hipcc] + FLAGS + ["-c", srcp, "-o", obj]
        t_obj = time.time()
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{r.stderr[-4000:]}")
        os.utime(obj, (t_obj, t_obj))
        if verbose:
            print("compiled", src)
        return obj

    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    # the library's timestamp must not run ahead of a source that was edited WHILE the objects were compiling (a later
    # needs_build() would then skip the rebuild): it gets the time at which this build STARTED reading the sources
    tmp = LIB + f".tmp{os.getpid()}"
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", tmp] + objs + ["-ldl"],
                       capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stderr[-4000:]}")
    os.utime(tmp, (t_start, t_start))
    os.replace(tmp, LIB)
    return LIB


if __name__ == "__main__":
    print(build(force=False, verbose=True))

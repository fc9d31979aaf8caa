"""Builds libdca_hip.so (hand-written HIP kernels, gfx950 only) in-tree with hipcc: one object per .hip source
(compiled in parallel, re-compiled only when the source or a header changed), then one link."""
import concurrent.futures
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "build")
LIB = os.path.join(HERE, "libdca_hip.so")
HEADER = os.path.join(HERE, "..", "include", "dca_hip.h")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value"] + os.environ.get("DCA_EXTRA_CFLAGS", "").split()


def sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _headers():
    return [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + [HEADER]


def _newer(path, than):
    return (not os.path.exists(than)) or os.path.getmtime(path) > os.path.getmtime(than)


FLAGS_STAMP = os.path.join(OBJ, ".flags")   # the compile flags the objects in OBJ were built with


def _flags_changed() -> bool:
    """objects built with other flags (an ablation / stamp build through DCA_EXTRA_CFLAGS) must not be reused"""
    try:
        return open(FLAGS_STAMP).read() != " ".join(FLAGS)
    except OSError:
        return True


def _stale() -> bool:
    deps = [os.path.join(CSRC, f) for f in sources()] + _headers()
    return _flags_changed() or any(_newer(d, LIB) for d in deps if os.path.exists(d))


def build(force: bool = False, verbose: bool = False) -> str:
    """hipcc --offload-arch=gfx950 -c per source, then -shared; cross-compiles without a GPU."""
    if not force and not _stale():
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    os.makedirs(OBJ, exist_ok=True)
    force = force or _flags_changed()
    jobs = []
    for src in sources():
        s, o = os.path.join(CSRC, src), os.path.join(OBJ, src[:-4] + ".o")
        if force or _newer(s, o) or any(_newer(h, o) for h in _headers() if os.path.exists(h)):
            jobs.append([hipcc] + FLAGS + ["-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True, cwd=CSRC)

    with concurrent.futures.ThreadPoolExecutor(max_workers=min(6, os.cpu_count() or 1)) as ex:
        list(ex.map(run, jobs))
    objs = [os.path.join(OBJ, s[:-4] + ".o") for s in sources()]
    run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs)
    with open(FLAGS_STAMP, "w") as f:
        f.write(" ".join(FLAGS))
    return LIB


if __name__ == "__main__":
    import sys
    print(build(force="--force" in sys.argv, verbose=True))

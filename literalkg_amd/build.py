"""Build liblkg_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

One object per source under ``lib/obj`` (re-made only when the source, a header or the flags changed; the objects are
compiled side by side), then one link: a kernel edit costs one translation unit, not ten."""
import hashlib
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "lib", "liblkg_hip.so")
OBJ = os.path.join(HERE, "lib", "obj")
SOURCES = ["lkg_graph_host.cpp", "lkg_spmm.hip", "lkg_attention.hip", "lkg_score.hip", "lkg_rowwise.hip",
           "lkg_gemm.hip", "lkg_batch.hip", "lkg_csr_device.hip", "lkg_gemm_tall.hip", "lkg_gemm_wgrad.hip", "lkg_layer.hip"]
FLAGS = ["-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17"]


def _headers():
    return [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith(".h")] + \
           [os.path.join(HERE, "..", "include", "literalkg_hip.h")]


TAG_FILE = os.path.join(HERE, "lib", "flags.tag")


def _stale(tag):
    if not os.path.exists(LIB):
        return True
    if os.path.exists(TAG_FILE) and open(TAG_FILE).read().strip() != tag:     # linked from objects of other flags
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "literalkg_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def _flag_tag(extra):
    return hashlib.sha256(" ".join(FLAGS + extra).encode()).hexdigest()[:12]


def build(force=False, verbose=True):
    extra = os.environ.get("LKG_EXTRA_HIPCC_FLAGS", "").split()      # kernel A/B experiments only
    tag = _flag_tag(extra)
    if not force and not _stale(tag):
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: cannot build literalkg_amd/lib/liblkg_hip.so")
    os.makedirs(OBJ, exist_ok=True)
    newest_header = max(os.path.getmtime(h) for h in _headers())
    todo, objs = [], []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(OBJ, f"{os.path.splitext(s)[0]}.{tag}.o")
        objs.append(obj)
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), newest_header):
            todo.append((src, obj))

    def compile_one(job):
        src, obj = job
        cmd = [hipcc] + FLAGS + extra + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.run(cmd, check=True)

    workers = max(1, min(len(todo), int(os.environ.get("LKG_BUILD_JOBS", "6"))))
    if todo:
        with ThreadPoolExecutor(workers) as pool:
            list(pool.map(compile_one, todo))
    cmd = [hipcc, "--offload-arch=gfx950", "-fPIC", "-shared", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True)
    with open(TAG_FILE, "w") as f:
        f.write(tag + "\n")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)

"""Build liblkg_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "lib", "liblkg_hip.so")
SOURCES = ["lkg_graph_host.cpp", "lkg_spmm.hip", "lkg_attention.hip", "lkg_score.hip", "lkg_rowwise.hip",
           "lkg_gemm.hip", "lkg_batch.hip", "lkg_csr_device.hip", "lkg_gemm_tall.hip", "lkg_gemm_wgrad.hip"]


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "literalkg_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    if not force and not _stale():
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: cannot build literalkg_amd/lib/liblkg_hip.so")
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    extra = os.environ.get("LKG_EXTRA_HIPCC_FLAGS", "").split()      # kernel A/B experiments only
    cmd = [hipcc, "-O3", "--offload-arch=gfx950", "-fPIC", "-shared", "-std=c++17", "-o", LIB] + extra + \
          [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)

"""Build libe2fgvi_hip.so (gfx950) in-tree with hipcc.  ``python -m e2fgvi_amd.build``."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(CSRC, "libe2fgvi_hip.so")
SOURCES = ["error.hip", "conv.hip", "conv_bf16.hip", "conv_wino.hip", "mdcn.hip", "attention.hip", "misc.hip", "video.hip", "metrics.hip"]
# -packed-fp32-ops: no kernel of this library may contain v_pk_{mul,add,fma}_f32.  Measured on MI355X (tools/probe/
# overlap_probe.hip, profiles/r02_overlap_probe_*.txt, DESIGN.md "Stream overlap"): the results of packed-fp32 VALU
# instructions of a wave are corrupted in lanes 48-63 when that wave shares a SIMD with the bf16 implicit-GEMM tile
# that keeps four v_mfma_f32_32x32x16_bf16 back to back -- independent of memory waits and cache policy, never with the
# scalar-fp32 form of the same arithmetic.  The side-stream kernels (SPyNet next to the encoder) were the victims.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=fast",
         "-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    return "hipcc"


def _stale(out, deps):
    return (not os.path.exists(out)) or any(os.path.getmtime(d) > os.path.getmtime(out) for d in deps)


def build(force=False, verbose=False):
    """Compile every HIP source for gfx950 and link the shared library.  Returns its path."""
    hipcc = _hipcc()
    objdir = os.path.join(CSRC, "build")
    os.makedirs(objdir, exist_ok=True)
    headers = [os.path.join(CSRC, "common.h"), os.path.join(HERE, "..", "include", "e2fgvi_hip.h")]
    jobs = []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(objdir, s.replace(".hip", ".o"))
        if force or _stale(obj, [src] + headers):
            jobs.append([hipcc] + FLAGS + ["-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)

    with ThreadPoolExecutor(max_workers=min(4, max(1, len(jobs)))) as ex:
        list(ex.map(run, jobs))
    objs = [os.path.join(objdir, s.replace(".hip", ".o")) for s in SOURCES]
    if force or jobs or _stale(LIB, objs):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))

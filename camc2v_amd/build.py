"""Build libccv_hip.so (gfx950) in-tree with hipcc.  No GPU needed (cross-compiles).

    python -m camc2v_amd.build [--force] [--verbose]
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libccv_hip.so")
SOURCES = ["ccv_gemm.hip", "ccv_fused.hip", "ccv_attn.hip", "ccv_norm.hip", "ccv_misc.hip", "ccv_pose.hip"]
HEADERS = [os.path.join(CSRC, "ccv_common.h"), os.path.join(os.path.dirname(HERE), "include", "ccv.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function"]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    raise RuntimeError("hipcc not found")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


VARIANTS = {            # library -> (object directory, extra compile flags)
    "libccv_hip.so": ("build", []),                                   # bf16 MFMA operands (the default)
    "libccv_hip_f16.so": ("build_f16", ["-DCCV_OPERANDS_F16"]),       # fp16 MFMA operands (CCV_OPERANDS=f16; csrc/ccv_common.h: ccv_opnd_t)
}


def build(force=False, verbose=False, variants=None):
    """Compile every variant of the library (all sources of all variants in one pool).  Returns the default library's path."""
    hipcc = _hipcc()
    jobs = []
    for libname, (objsub, extra) in VARIANTS.items():
        if variants is not None and libname not in variants:
            continue
        objdir = os.path.join(HERE, objsub)
        os.makedirs(objdir, exist_ok=True)
        for src in SOURCES:
            s = os.path.join(CSRC, src)
            o = os.path.join(objdir, src.replace(".hip", ".o"))
            if force or _stale(o, [s] + HEADERS):
                jobs.append((s, o, extra))

    def compile_one(job):
        s, o, extra = job
        cmd = [hipcc] + FLAGS + extra + ["-c", s, "-o", o]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {s}:\n{r.stdout}\n{r.stderr}")
        if verbose and r.stderr.strip():
            print(r.stderr)
        return o

    if jobs:
        jobs.sort(key=lambda j: -os.path.getsize(j[0]))          # the long compiles (ccv_gemm.hip: ~4 min) first
        with ThreadPoolExecutor(max_workers=min(max(2, (os.cpu_count() or 4) - 2), len(jobs))) as ex:
            list(ex.map(compile_one, jobs))
    for libname, (objsub, _) in VARIANTS.items():
        if variants is not None and libname not in variants:
            continue
        objs = [os.path.join(HERE, objsub, s.replace(".hip", ".o")) for s in SOURCES]
        lib = os.path.join(HERE, libname)
        if force or _stale(lib, objs):
            cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs
            if verbose:
                print(" ".join(cmd), flush=True)
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="--verbose" in sys.argv or "-v" in sys.argv))

"""Builds libmmvae_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libmmvae_hip.so")
SOURCES = ["gemm.hip", "convres.hip", "wgrad_ring.hip", "gemm_small.hip", "elementwise.hip", "text.hip", "thin.hip", "dec_last.hip", "conv1.hip", "multimnist.hip", "mnist.hip", "mnist_f32.hip", "gemm_f32.hip", "celeba.hip", "coco.hip", "coco_text.hip", "coco_text_bf16.hip", "mlp_tail.hip", "capi.cpp", "comm.cpp", "util.cpp"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wno-unused-result", "-Wno-unused-value", "-ffp-contract=fast"]


def _stale(out, deps):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers.append(os.path.join(os.path.dirname(HERE), "include", "mmvae_hip.h"))
    jobs = []
    objs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(objdir, src + ".o")
        objs.append(o)
        if force or _stale(o, [s] + headers):
            jobs.append([hipcc, *FLAGS, "-x", "hip", "-c", s, "-o", o])

    def run(cmd):
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            errs = [l for l in r.stderr.splitlines() if "error" in l]
            raise RuntimeError("hipcc failed: %s\n%s\n%s" % (" ".join(cmd), "\n".join(errs[:20]), r.stderr[-2000:]))
        if verbose and r.stderr:
            sys.stderr.write(r.stderr)

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(run, jobs))
    # (an object compiled by hand, outside this function, must reach the library too: relink whenever any object is newer than it)
    if jobs or not os.path.exists(LIB) or any(os.path.getmtime(o) > os.path.getmtime(LIB) for o in objs):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs, "-ldl"])
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))

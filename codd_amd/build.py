"""Build libcodd_hip.so (gfx950) in-tree with hipcc.  No torch involved: the library is a plain
C-ABI shared object (include/codd_hip.h) loaded through ctypes by codd_amd._abi."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(CSRC, "libcodd_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# files built without SLP vectorisation: the Gauss-Newton builder feeds its VALU from SGPRs (scalar
# loads); v_pk_* packing forces ~50 v_mov per neighbour to assemble register pairs
NO_SLP = {"motion.hip"} if os.environ.get("CODD_NO_SLP", "1") == "1" else set()


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def headers():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")) + [
        os.path.join(ROOT, "include", "codd_hip.h")]


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = sources() + headers()
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    if not force and not needs_build():
        if verbose:  # (the driver's "does it build" check: say which of the two happened)
            print(f"codd_amd.build: REUSED {os.path.relpath(LIB, ROOT)} -- up to date against {len(sources())} .hip sources and "
                  f"{len(headers())} headers (nothing compiled; build(force=True) or `python -m codd_amd.build --force` recompiles)", flush=True)
        return LIB
    from concurrent.futures import ThreadPoolExecutor
    objs, jobs = [], []
    newest_hdr = max(os.path.getmtime(h) for h in headers())
    for src in sources():
        obj = src[:-4] + ".o"
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), newest_hdr):
            cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", src, "-o", obj,
                   "-I", os.path.join(ROOT, "include"), "-I", CSRC, "-Wno-unused-result", "-Wno-pass-failed",
                   # keep MFMA accumulators in VGPRs: without it hipcc (ROCm 7.2) copies all
                   # accumulators VGPR<->AGPR around EVERY k-step of the conv loop (10 VALU per MFMA)
                   "-mllvm", "-amdgpu-mfma-vgpr-form"]
            cmd += os.environ.get("CODD_EXTRA_FLAGS", "").split()  # dev builds, e.g. -DCONV_PROFILE
            if os.path.basename(src) in NO_SLP:
                cmd.append("-fno-slp-vectorize")
            jobs.append(cmd)
        objs.append(obj)

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)

    with ThreadPoolExecutor(max_workers=max(1, min(len(jobs), os.cpu_count() or 1, 16))) as ex:
        list(ex.map(run, jobs))  # translation units are independent: compile them in parallel
    tmp = LIB + ".tmp.%d" % os.getpid()  # link under a private name, then rename: no reader ever sees a partial file
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", tmp] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    os.replace(tmp, LIB)
    if verbose:
        print(f"codd_amd.build: COMPILED {len(jobs)} of {len(objs)} translation units with {HIPCC} --offload-arch=gfx950 and linked "
              f"{os.path.relpath(LIB, ROOT)} ({os.path.getsize(LIB)} bytes)", flush=True)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)

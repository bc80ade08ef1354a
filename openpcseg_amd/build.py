"""Compile openpcseg_amd/csrc/*.hip into openpcseg_amd/lib/libpcseg_hip.so for gfx950.

hipcc cross-compiles without a GPU. The library has no torch dependency: it is the C ABI of
include/pcseg_hip.h and links only the HIP runtime.
"""
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libpcseg_hip.so")

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-munsafe-fp-atomics", "-fPIC",
         "-Wno-unused-value"]
# per-file additions. The wave kernels: no SLP vectorisation -- it packs the 48 accumulate adds of the ticket-ordered commit
# into v_pk_add_f32 at the price of two register moves each, inside the one serial chain of a workgroup
EXTRA_FLAGS = {"conv_wave5.hip": ["-fno-slp-vectorize", "-Wno-array-bounds"],
               "conv_wave5h.hip": ["-fno-slp-vectorize", "-Wno-array-bounds"],
               "conv_wave6h.hip": ["-fno-slp-vectorize", "-Wno-array-bounds"]}


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def _stale():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = sources() + glob.glob(os.path.join(CSRC, "*.h")) + [
        os.path.join(HERE, "..", "include", "pcseg_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    """Build the shared library if it is missing or older than its sources."""
    if not force and not _stale():
        return LIB_PATH
    os.makedirs(LIB_DIR, exist_ok=True)
    for stale in glob.glob(os.path.join(LIB_DIR, "*.hip.o")):   # objects of sources that left csrc/ must not be linked
        if os.path.basename(stale)[:-2] not in {os.path.basename(s) for s in sources()}:
            os.remove(stale)
    objs = []
    procs = []
    for s in sources():
        o = os.path.join(LIB_DIR, os.path.basename(s) + ".o")
        objs.append(o)
        cmd = [HIPCC] + FLAGS + EXTRA_FLAGS.get(os.path.basename(s), []) + ["-c", s, "-o", o]
        if verbose:
            print(" ".join(cmd))
        procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for s, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError("hipcc failed on %s:\n%s" % (s, out.decode()))
        if verbose and out:
            print(out.decode())
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", LIB_PATH]
    subprocess.check_call(cmd)
    return LIB_PATH


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))

#!/usr/bin/env python3
"""Build the reference's OWN CPU backend into oracle/_ref/ (test infrastructure only).

TEST INFRASTRUCTURE -- never imported by the product package `openpcseg_amd`.

What this does (all offline, nothing is copied into the repo's history):
  1. unzips /root/reference/package/torchsparse.zip (vendored torchsparse 1.4.0) and
     /root/reference/package/sparsehash.zip into a throw-away temp dir;
  2. runs sparsehash's `./configure && make src/sparsehash/internal/sparseconfig.h`
     ONLY to generate its one config header (header-only library; needed by the
     reference's query_cpu.cpp, TS:torchsparse/backend/others/query_cpu.cpp:6);
  3. compiles the reference's 7 CPU translation units
     (TS:torchsparse/backend/{convolution,devoxelize,hash,hashmap,others,voxelize}/*_cpu.cpp
      + pybind_cpu.cpp, listed in TS:setup.py:17-19) with plain g++ -- NOT the
     reference's setup.py -- into ONE python extension
        oracle/_ref/ref_backend.cpython-310-x86_64-linux-gnu.so
     exporting the reference's 10 `*_cpu` functions (TS:.../backend/pybind_cpu.cpp:12-23).

oracle/_ref/ is git-ignored (binary only, no sources) but travels to the GPU box.
If /root/reference is absent (GPU box) this script is a no-op.
"""
import glob
import os
import shutil
import subprocess
import sys
import sysconfig
import tempfile
import zipfile

HERE = os.path.dirname(os.path.abspath(__file__))
OUT_DIR = os.path.join(HERE, "_ref")
REF_PKG = "/root/reference/package"
MOD = "ref_backend"


def out_path():
    return os.path.join(OUT_DIR, MOD + sysconfig.get_config_var("EXT_SUFFIX"))


def build(force=False, verbose=False):
    if not os.path.isdir(REF_PKG):
        return None  # GPU box: use the prebuilt binary if it travelled
    so = out_path()
    if os.path.exists(so) and not force:
        return so
    os.makedirs(OUT_DIR, exist_ok=True)
    import torch.utils.cpp_extension as ce

    tmp = tempfile.mkdtemp(prefix="pcs_ref_build_")
    try:
        for z in ("torchsparse.zip", "sparsehash.zip"):
            with zipfile.ZipFile(os.path.join(REF_PKG, z)) as zf:
                members = [m for m in zf.namelist() if "__MACOSX" not in m and "/.git/" not in m]
                zf.extractall(tmp, members)
        # zip extraction drops the exec bit
        sh_dir = os.path.join(tmp, "sparsehash-master")
        for f in ("configure", "install-sh", "missing", "depcomp", "config.guess", "config.sub"):
            p = os.path.join(sh_dir, f)
            if os.path.exists(p):
                os.chmod(p, 0o755)
        log = open(os.path.join(tmp, "sparsehash.log"), "w")
        subprocess.check_call(["sh", "./configure", "--prefix=" + os.path.join(tmp, "sphash")],
                              cwd=sh_dir, stdout=log, stderr=log)
        subprocess.check_call(["make", "-j8"], cwd=sh_dir, stdout=log, stderr=log)
        subprocess.check_call(["make", "install"], cwd=sh_dir, stdout=log, stderr=log)
        sp_inc = os.path.join(tmp, "sphash", "include")

        be = os.path.join(tmp, "torchsparse", "torchsparse", "backend")
        srcs = sorted(set(glob.glob(os.path.join(be, "**", "*_cpu.cpp"), recursive=True)))
        incs = ce.include_paths() + [sysconfig.get_paths()["include"], sp_inc, be]
        objs = []
        procs = []
        for s in srcs:
            o = os.path.join(tmp, os.path.basename(s) + ".o")
            objs.append(o)
            cmd = ["g++", "-O3", "-fopenmp", "-fPIC", "-std=c++17", "-w",
                   "-DTORCH_EXTENSION_NAME=" + MOD, "-DTORCH_API_INCLUDE_EXTENSION_H",
                   "-D_GLIBCXX_USE_CXX11_ABI=1"]
            for i in incs:
                cmd += ["-I", i]
            cmd += ["-c", s, "-o", o]
            procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        for s, p in procs:
            outp, _ = p.communicate()
            if p.returncode != 0:
                raise RuntimeError("compile failed: %s\n%s" % (s, outp.decode()))
        libdir = ce.library_paths()[0]
        link = ["g++", "-shared", "-fopenmp"] + objs + [
            "-L" + libdir, "-Wl,-rpath," + libdir,
            "-lc10", "-ltorch", "-ltorch_cpu", "-ltorch_python", "-o", so]
        subprocess.check_call(link)
        if verbose:
            print("built", so)
        return so
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def load():
    """Import the prebuilt reference backend (or None if it does not exist)."""
    so = out_path()
    if not os.path.exists(so):
        return None
    import importlib.util
    import torch  # noqa: F401  (libtorch must be loaded first)
    spec = importlib.util.spec_from_file_location(MOD, so)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


if __name__ == "__main__":
    p = build(force="--force" in sys.argv, verbose=True)
    print(p if p else "reference not present; nothing built")

#!/usr/bin/env python3
"""Stage the reference files the `-m gpu` model tests import into tests/_refsrc/ (git-ignored, test-only).

/root/reference does not exist on the GPU box, but the claim "the reference's segmentors load unmodified on this
package" has to be checked THERE, on the HIP kernels. So `__graft_entry__.build()` (which runs in the build container,
where the reference tree is present) copies exactly the reference modules that importing the four segmentors pulls in
-- found by importing them and reading sys.modules, not by a hand-written list -- into tests/_refsrc/, which travels
to the GPU box with the snapshot like oracle/_ref does and never enters git history. Nothing in openpcseg_amd/
reads this directory.
"""
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = "/root/reference"
DST = os.path.join(ROOT, "tests", "_refsrc")
MODELS = ["pcseg.model.segmentor.voxel.minkunet.minkunet", "pcseg.model.segmentor.fusion.spvcnn.spvcnn",
          "pcseg.model.segmentor.voxel.cylinder3d.cylinder_ts", "pcseg.model.segmentor.fusion.rpvnet.rpvnet"]
# data-side reference code the cylinder front-end goldens are generated from (pure NumPy, imported by make_golden.py)
EXTRA = ["pcseg/data/dataset/semantickitti/semantickitti_cylinder.py"]


def reference_root():
    """Where the reference's Python sources can be imported from: the real tree, else the staged copy."""
    if os.path.isdir(os.path.join(REF, "pcseg")) and os.environ.get("PCS_REFSRC") != "staged":
        return REF
    if os.path.isdir(os.path.join(DST, "pcseg")):
        return DST
    return None


TS_ZIP = os.path.join(REF, "package", "torchsparse.zip")
TS_FUNCTIONAL = ["hash", "query", "count", "voxelize", "devoxelize", "conv"]   # TS:torchsparse/nn/functional/<name>.py


def functional_source(name):
    """Source text of the reference's torchsparse/nn/functional/<name>.py: from the reference's zip where it exists, else from
    the staged copy (tests/_refsrc/ts_functional/, git-ignored, travels to the GPU box). None when neither is there."""
    if os.path.isfile(TS_ZIP) and os.environ.get("PCS_REFSRC") != "staged":
        import zipfile
        with zipfile.ZipFile(TS_ZIP) as z:
            return z.read("torchsparse/torchsparse/nn/functional/%s.py" % name).decode()
    p = os.path.join(DST, "ts_functional", name + ".py")
    return open(p).read() if os.path.isfile(p) else None


def stage_functional():
    if not os.path.isfile(TS_ZIP):
        return
    import zipfile
    d = os.path.join(DST, "ts_functional")
    os.makedirs(d, exist_ok=True)
    with zipfile.ZipFile(TS_ZIP) as z:
        for name in TS_FUNCTIONAL:
            with open(os.path.join(d, name + ".py"), "wb") as f:
                f.write(z.read("torchsparse/torchsparse/nn/functional/%s.py" % name))


def stage(verbose=False):
    if not os.path.isdir(os.path.join(REF, "pcseg")):
        return None
    # import in a subprocess: the import stubs / aliases must not leak into the caller's sys.modules
    import subprocess
    code = (
        "import sys, os\n"
        "sys.path[:0] = [%r, %r, %r]\n"
        "import openpcseg_amd; openpcseg_amd.install_reference_aliases()\n"
        "import make_golden\n"
        "for d in %r: make_golden.import_reference_model(d)\n"
        "fs = sorted({m.__file__ for m in list(sys.modules.values()) if getattr(m, '__file__', None) "
        "and m.__file__.startswith(%r + '/')})\n"
        "print('\\n'.join(fs))\n"
    ) % (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden"), MODELS, REF)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, check=True).stdout
    files = [l for l in out.splitlines() if l.startswith(REF + "/")]
    files += [os.path.join(REF, e) for e in EXTRA]
    if os.path.isdir(DST):
        shutil.rmtree(DST)
    for f in sorted(set(files)):
        rel = os.path.relpath(f, REF)
        dst = os.path.join(DST, rel)
        os.makedirs(os.path.dirname(dst), exist_ok=True)
        shutil.copyfile(f, dst)
        if verbose:
            print("staged", rel)
    stage_functional()
    return DST


if __name__ == "__main__":
    print(stage(verbose=True))

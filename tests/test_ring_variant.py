"""The experimental column-parallel "ring" convolution kernels (round 4: parity-green, 0.47-0.70x the wave kernels,
profiles/round4_ring.md, profiles/DESIGN_rounds1-5.md section 5d) are NOT part of libpcseg_hip.so: sources under tools/experimental/csrc/, built only
into the variant library `tools/build_variant_lib.sh ring` (-DPCS_WITH_RING=1). This test keeps them from rotting: it builds the
variant on the GPU box (hipcc is there) and runs their oracle-parity cases (tests/test_dense_parity.py::test_ring_conv_*) in a
subprocess bound to that library."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_product_library_has_no_ring_kernels():
    """CPU: neither the C ABI nor the product sources carry the ring kernels."""
    hdr = open(os.path.join(ROOT, "include", "pcseg_hip.h")).read()
    assert "pcs_conv_ring_enable(" not in hdr and "pcs_conv_ring_applies(" not in hdr
    assert not [f for f in os.listdir(os.path.join(ROOT, "openpcseg_amd", "csrc")) if f.startswith("conv_ring6")]
    lib = os.path.join(ROOT, "openpcseg_amd", "lib", "libpcseg_hip.so")
    if os.path.exists(lib):
        syms = subprocess.run(["nm", "-D", lib], capture_output=True, text=True).stdout
        assert "pcs_conv_ring" not in syms and "conv_ring6" not in syms


@pytest.mark.gpu
def test_ring_kernels_in_the_variant_library(hip):
    r = subprocess.run(["bash", os.path.join(ROOT, "tools", "build_variant_lib.sh"), "ring"], capture_output=True, text=True, timeout=900)
    lib = os.path.join(ROOT, "openpcseg_amd", "lib", "dbg", "ring.so")
    if r.returncode != 0 or not os.path.exists(lib):
        pytest.skip("variant library did not build here: " + r.stderr[-300:])
    try:
        env = dict(os.environ, PCS_LIB_PATH=lib, PCS_RING_VARIANT="1")
        p = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_dense_parity.py"), "-q", "-x", "-m", "gpu",
                            "-k", "test_ring_conv", "-p", "no:cacheprovider"], env=env, capture_output=True, text=True, timeout=1200, cwd=ROOT)
        tail = (p.stdout + p.stderr)[-1500:]
        assert p.returncode == 0 and " passed" in p.stdout and "skipped" not in p.stdout.splitlines()[-1], tail
    finally:
        os.remove(lib)

"""Static resource check of the built gfx950 code objects (CPU only, reads libdrgnn.so's fat binary): no kernel may keep a
private copy of its argument block in scratch memory.  The fused step kernels take a 2.4 KB argument block with run-time
indexed arrays; when the compiler materialises it per lane the kernel runs ~5x slower (round 3: the generic-width sGAT step
kernels did, 151.7 vs 27.6 us per step, profiles/r03_kernarg_scratch.txt) -- visible in the code object's metadata as a
private segment of about the size of the block.  Genuine register spills of the generic kernels are 20 - 204 bytes."""
import os
import re
import shutil
import subprocess
import tempfile

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "..", "deeprank-gnn_amd", "csrc", "libdrgnn.so")
LLVM = "/opt/rocm/lib/llvm/bin"
LIMIT = 512       # bytes of private segment per lane


def _kernels():
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        shutil.copy(LIB, os.path.join(tmp, "lib.so"))
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", "lib.so"], cwd=tmp, check=True,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        for f in sorted(os.listdir(tmp)):
            if "amdgcn" not in f:
                continue
            notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", f], cwd=tmp, check=True,
                                   stdout=subprocess.PIPE, stderr=subprocess.DEVNULL).stdout.decode()
            name = None
            for line in notes.splitlines():
                m = re.match(r"\s+\.name:\s+(\S+)", line)
                if m:
                    name = m.group(1)
                m = re.match(r"\s+\.private_segment_fixed_size:\s+(\d+)", line)
                if m and name:
                    out[name] = int(m.group(1))
    return out


@pytest.mark.skipif(not (os.path.exists(LIB) and os.path.exists(os.path.join(LLVM, "llvm-readelf"))),
                    reason="needs the built library and the ROCm LLVM tools")
def test_no_kernel_copies_its_argument_block_to_scratch():
    kern = _kernels()
    steps = {k: v for k, v in kern.items() if "k_step" in k}
    assert len(steps) >= 44, sorted(kern)            # 3 kinds x 5 widths x 2 modes + the one-workgroup layouts
    fat = {k: v for k, v in kern.items() if v > LIMIT}
    assert not fat, "kernels with a large private segment (argument block copied to scratch?): %s" % fat

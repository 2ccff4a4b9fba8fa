"""The C ABI from a caller that is not python: tests/native/abi_harness.cpp (plain C++ / HIP, no torch) is compiled against
lib/libflmr_hip.so and run -- index open from HOST arrays (FLMR_MEM_HOST), batched search at two policies with ragged q_lens,
taps, deferred checks, op-level calls, every argument error, three index shapes.  (profiles/sanitize_host.sh runs the same
program against the AddressSanitizer + UBSan build of the library.)"""
import os
import shutil
import subprocess

import pytest

from conftest import ROOT


@pytest.mark.gpu
def test_cpp_caller_of_the_c_abi(tmp_path):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import ravqa_amd
    from ravqa_amd import _native
    lib = ravqa_amd.build_native()
    hipcc = _native.hipcc_path()
    if shutil.which(hipcc) is None and not os.path.exists(hipcc):
        pytest.fail("hipcc not found: the harness cannot be built")
    exe = str(tmp_path / "abi_harness")
    libdir = os.path.dirname(lib)
    link = str(tmp_path / "libflmr_hip.so")   # -lflmr_hip resolves here; rpath points at the real directory
    os.symlink(lib, link)
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O1", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "native", "abi_harness.cpp"), "-o", exe,
                           "-L" + str(tmp_path), "-lflmr_hip", "-Wl,-rpath," + str(tmp_path) + ":" + libdir])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "abi harness: all cases passed" in out.stdout

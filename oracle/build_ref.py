#!/usr/bin/env python3
"""Compile the reference's four CPU C++ extensions IN PLACE into oracle/_ref/*.so.

TEST INFRASTRUCTURE ONLY (see flmr_oracle.c).  Sources are read where they lie under
$FLMR_REFERENCE_ROOT (default /root/reference) and are never copied into this repository;
only the binaries land in oracle/_ref/ (git-ignored; they travel to the GPU box with the
snapshot).  g++ is invoked directly -- the reference's own build path (torch JIT `load`)
is not used.  The only dependencies are the torch/pybind11 headers and libs shipped in this
image (the same image runs on the GPU box).

  TPC/search/filter_pids.cpp            -> filter_pids_cpp.so
  TPC/search/decompress_residuals.cpp   -> decompress_residuals_cpp.so
  TPC/search/segmented_lookup.cpp       -> segmented_lookup_cpp.so
  TPC/modeling/segmented_maxsim.cpp     -> segmented_maxsim_cpp.so
"""
import os
import subprocess
import sys
import sysconfig

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("FLMR_REFERENCE_ROOT", "/root/reference")
TPC = os.path.join(REF, "third_party", "ColBERT", "colbert")
OUT = os.path.join(HERE, "_ref")

SOURCES = {
    "filter_pids_cpp": "search/filter_pids.cpp",
    "decompress_residuals_cpp": "search/decompress_residuals.cpp",
    "segmented_lookup_cpp": "search/segmented_lookup.cpp",
    "segmented_maxsim_cpp": "modeling/segmented_maxsim.cpp",
}


def main():
    if not os.path.isdir(TPC):
        print(f"[build_ref] {TPC} not present: skipping (prebuilt oracle/_ref/*.so are used if they exist)")
        return 0
    import torch
    from torch.utils import cpp_extension as ce

    os.makedirs(OUT, exist_ok=True)
    inc = [f"-I{p}" for p in ce.include_paths()] + [f"-I{sysconfig.get_paths()['include']}"]
    libdirs = ce.library_paths()
    abi = int(torch._C._GLIBCXX_USE_CXX11_ABI)
    for name, rel in SOURCES.items():
        src = os.path.join(TPC, rel)
        dst = os.path.join(OUT, name + ".so")
        if os.path.exists(dst) and os.path.getmtime(dst) >= os.path.getmtime(src):
            continue
        cmd = (["g++", "-O3", "-std=c++17", "-fPIC", "-shared", "-w", f"-DTORCH_EXTENSION_NAME={name}",
                "-DTORCH_API_INCLUDE_EXTENSION_H", f"-D_GLIBCXX_USE_CXX11_ABI={abi}"] + inc + [src, "-o", dst]
               + [f"-L{d}" for d in libdirs] + [f"-Wl,-rpath,{d}" for d in libdirs]
               + ["-lc10", "-ltorch_cpu", "-ltorch", "-ltorch_python", "-lpthread"])
        print("[build_ref]", name)
        subprocess.check_call(cmd)
    return 0


if __name__ == "__main__":
    sys.exit(main())

"""Build a development variant of the HIP library beside the product one:
    python profiles/build_variant.py s1sprof -DS1S_PROFILE     ->  lib/libflmr_hip_s1sprof.so
(profiles/s1s_profile.sh swaps it in for one short bench run on the GPU box)"""
import os, subprocess, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import ravqa_amd._native as n
P = os.path.dirname(n.LIB_PATH)
srcs = [os.path.join(n.CSRC, s) for s in n.SOURCES]
out = os.path.join(P, f"libflmr_hip_{sys.argv[1]}.so")
subprocess.check_call([n.hipcc_path(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-w"] + sys.argv[2:] +
                      ["-I" + os.path.join(R, "include"), "-I" + n.CSRC] + srcs + ["-o", out])
print(out)

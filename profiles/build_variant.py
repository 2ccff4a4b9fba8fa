"""Build a development variant of the HIP library beside the product one:
    python profiles/build_variant.py s1sprof -DS1S_PROFILE               ->  lib/libflmr_hip_s1sprof.so
    python profiles/build_variant.py exp -DFLMR_EXPERIMENTAL_VARIANTS    ->  the measured-loser kernel forms (FLMR_S2_IMPL=regs|ldsb)
Select it for a run with FLMR_HIP_LIB=<path> (ravqa_amd/_native.py), or swap it in with profiles/ab_lib.sh."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import ravqa_amd._native as n
out = os.path.join(os.path.dirname(n.LIB_PATH), f"libflmr_hip_{sys.argv[1]}.so")
print(n.build_native(extra_flags=tuple(sys.argv[2:]), lib_path=out))

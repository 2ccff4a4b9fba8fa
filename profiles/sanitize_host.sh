#!/bin/bash
# Host side of the C-ABI library under AddressSanitizer + UBSan.  Build here (no GPU needed), run on the GPU box from the repo root:
#   python profiles/build_variant.py asan  -fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-gpu-sanitize -g
#   python profiles/build_variant.py ubsan -fsanitize=undefined -fno-sanitize-recover=undefined -fno-gpu-sanitize -g
#   L=retrieval-augmented-visual-question-answering_amd/lib
#   hipcc --offload-arch=gfx950 -O1 -g -fsanitize=address,undefined -fno-gpu-sanitize -shared-libsan -Iinclude tests/native/abi_harness.cpp \
#         -o tests/native/abi_harness_asan -L$L -lflmr_hip_asan -Wl,-rpath,'$ORIGIN/../../retrieval-augmented-visual-question-answering_amd/lib'
#   bash profiles/sanitize_host.sh                      (GPU box)
# The device code is compiled as always (-fno-gpu-sanitize); what is checked is every host function behind include/flmr_hip.h --
# argument validation, workspace carving, index open / close, option tables, the launchers' host arithmetic.
#   1. ASan + UBSan: tests/native/abi_harness.cpp, a plain C++ caller of the C ABI (torch's HIP start-up does not survive a
#      preloaded ASan runtime; a plain HIP program does).
#   2. UBSan alone, preloaded into python: the whole GPU parity suite drives the instrumented library.
set -u
RTD=$(dirname $(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1))
L=$(pwd)/retrieval-augmented-visual-question-answering_amd/lib
[ -f "$L/libflmr_hip_asan.so" ] && [ -f "$L/libflmr_hip_ubsan.so" ] && [ -x tests/native/abi_harness_asan ] || { echo "build the variants first (see the header of this script)"; exit 2; }
echo "== 1. AddressSanitizer + UBSan: C++ harness over the C ABI =="
LD_LIBRARY_PATH=$RTD:${LD_LIBRARY_PATH:-} ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0:halt_on_error=1 \
  UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 timeout 600 ./tests/native/abi_harness_asan 2>&1 | tail -40
echo "exit code ${PIPESTATUS[0]}"
echo "== 2. UBSan: GPU parity suite on the instrumented library =="
LD_PRELOAD=$RTD/libclang_rt.ubsan_standalone-x86_64.so UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
  FLMR_HIP_LIB=$L/libflmr_hip_ubsan.so timeout 1200 python -m pytest tests/test_hip_parity.py tests/test_baseline_shapes.py -m gpu -q -x 2>&1 \
  | grep -E "passed|failed|runtime error|Error|SUMMARY" | head -20

#!/bin/bash
# Stage 2's dominant kernel (filter_stage2_xcd_kernel<hi-only>, 620 us per 256-query launch at BASELINE's shape) taken apart: the
# stand-alone harness (profiles/microbench/s2_xcd_probe.hip: 1024 queries x 1024 survivors x 128 uniformly random codes, K = 131072)
# built with the kernel's development ablation switches -- no row gathers (-DX2_NO_DMA), no matrix products (-DX2_NO_MFMA), no
# fold (-DX2_NO_FOLD) and their pairs -- so that the full launch can be read as a sum (or a maximum) of its parts.
#   usage (GPU box, repo root):  bash profiles/s2_ceiling.sh > profiles/r06/s2_ceiling.txt
set -u
R=$(pwd)
B=/tmp/s2c; mkdir -p $B
build() { hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-inline-asm $2 -Iinclude -Iretrieval-augmented-visual-question-answering_amd/csrc -o $B/$1 profiles/microbench/s2_xcd_probe.hip 2>&1 | grep -E "error" ; }
build full "" & build nodma "-DX2_NO_DMA" & build nomfma "-DX2_NO_MFMA" & build nofold "-DX2_NO_FOLD" &
build dma_only "-DX2_NO_MFMA -DX2_NO_FOLD" & build mfma_only "-DX2_NO_DMA -DX2_NO_FOLD" & build fold_only "-DX2_NO_DMA -DX2_NO_MFMA" & build skeleton "-DX2_NO_DMA -DX2_NO_MFMA -DX2_NO_FOLD" &
wait
echo "filter_stage2_xcd_kernel<hi only>, 1024 queries x 1024 survivors x 128 tokens, K = 131072: ms per launch of 1024 queries (best of 3), sliced kernel + combine"
for v in full nodma nomfma nofold dma_only mfma_only fold_only skeleton; do
  t=$(timeout 120 $B/$v 1024 131072 1 | grep "sliced stage 2" | awk '{print $(NF-1)}' | sort -n | head -1)
  printf "  %-10s %s\n" $v "$t"
done

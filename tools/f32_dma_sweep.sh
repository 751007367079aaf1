#!/bin/bash
# Per-shape table of the ECAPA step's f32 products under each forced kernel / tile of the exact-f32 GEMM
# (w2v2_tune_gemm_f32_tile): register-staged kernel (W2V2_F32_NO_DMA=1), the library's choice, and the LDS-DMA kernel
# at every tile height.   gpurun -- 'bash tools/f32_dma_sweep.sh'
mkdir -p gpurun_out
out=gpurun_out/f32_dma_sweep.txt
: > $out
echo "##### W2V2_F32_NO_DMA=1 (register-staged kernel, round-5 rule)" >> $out
W2V2_F32_NO_DMA=1 python3 tools/ecapa_gemm_shapes.py >> $out 2>&1
echo "##### default" >> $out
python3 tools/ecapa_gemm_shapes.py >> $out 2>&1
for t in ${CODES:-11 12 13 14 15 115}; do
  echo "##### F32_TILE=$t" >> $out
  F32_TILE=$t python3 tools/ecapa_gemm_shapes.py >> $out 2>&1
done

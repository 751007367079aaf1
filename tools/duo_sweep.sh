#!/bin/bash
# stand-alone sweeps of the two-workgroups-per-CU GEMM kernel (family 5): start offset of the second residents, and the
# kernel without its epilogue (W2V2_DUO_DBG=1) -- one process per setting (the switches are read once)
export FAMILIES=${FAMILIES:-2,4,5} TRIALS=3
for f in ${SHAPES:-ffn1 dh ffn2 dx1 conv3 conv4}; do
  for st in ${STAGGERS:-default 0}; do
    if [ "$st" = default ]; then unset W2V2_DUO_STAGGER_US; else export W2V2_DUO_STAGGER_US=$st; fi
    echo "## stagger=$st"; python tools/gemm_shapes.py "$f" 2>/dev/null | grep -v "^#"
  done
  unset W2V2_DUO_STAGGER_US
  echo "## no epilogue (duo only), stagger default"
  W2V2_DUO_DBG=1 python tools/gemm_shapes.py "$f" 2>/dev/null | grep -v "^#"
done

#!/bin/bash
# Adam launch over the w2v2-base (99.4 M) and wav2vec2-large (323.5 M) arenas: vectors in flight x grid cap
for N in 99400000 323500000; do for U in 1 2 4; do for NB in 4096 8192 16384 32768; do
  ADAM_N=$N W2V2_ADAM_U=$U W2V2_ADAM_BLOCKS=$NB python tools/adam_bench.py 2>&1 | grep adam | sed "s/^/n=$N /"
done; done; done

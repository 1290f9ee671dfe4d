#!/bin/bash
# development: atomic vs owner-computes train path on narrow-row shapes
for path in atomic tiled; do
  export AMDKGE_TRAIN_PATH=$path
  bash scripts/gpu_variants.sh default "--config C1" "--model TransE --k 100" "--model TransE --k 52" "--model TransE --k 128" 2>&1 | sed "s/^/$path /"
done
unset AMDKGE_TRAIN_PATH
bash scripts/gpu_variants.sh default "--model TransE" "--model TransE --k 352" "--model TransE --loss pairwise"

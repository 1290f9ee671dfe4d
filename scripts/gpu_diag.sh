#!/bin/bash
export TMPDIR=/tmp
for v in "" $@; do
  echo "== lib ${v:-default}"
  if [ -n "$v" ]; then export AMDKGE_LIB=$PWD/build_variants/$v/libamdkge.so; fi
  timeout 200 python scripts/screen_diag.py 2>&1 | grep -v amdgpu.ids
done

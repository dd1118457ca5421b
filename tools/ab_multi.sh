#!/bin/bash
# Same-box timing of several builds of the C-ABI library: tools/ab_multi.sh <reps> <script> <lib1.so> <lib2.so> ...
n="$1"; script="$2"; shift 2
for i in $(seq $n); do
  for v in "$@"; do
    echo "[$(basename $(dirname $v))] $(MODALTUNE_HIP_LIB=$PWD/$v python $script 2>/dev/null | tr '\n' ' ')"
  done
done

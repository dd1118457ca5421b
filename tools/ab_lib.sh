#!/bin/bash
# Same-box A/B of two builds of the C-ABI library: tools/ab_lib.sh <libA.so> <libB.so> <reps> <script> [args]
a="$1"; b="$2"; n="$3"; shift 3
for i in $(seq $n); do
  for v in "$a" "$b"; do
    echo "[$(basename $(dirname $v))] $(MODALTUNE_HIP_LIB=$PWD/$v python "$@" 2>/dev/null | tr '\n' ' ')"
  done
done

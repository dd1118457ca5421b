#!/bin/bash
# A/B the bench under two argument sets on ONE box: tools/ab_args.sh "--no-dropout" "" [reps]
a="$1"; b="$2"; n=${3:-3}
for i in $(seq $n); do
  for v in "$a" "$b"; do
    ms=$(python bench.py --no-cpu-baseline $v 2>/dev/null | python -c "import json,sys; print(round(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'],2))")
    echo "[$v] $ms ms"
  done
done

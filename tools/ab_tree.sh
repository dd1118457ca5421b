#!/bin/bash
# Same-box A/B of two TREES (python + library): tools/ab_tree.sh <treeA> <treeB> <reps> <bench args...>; prints ms_per_step (and slides/s)
a="$1"; b="$2"; n="$3"; shift 3
for i in $(seq $n); do
  for t in "$a" "$b"; do
    r=$(cd $t && python bench.py "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],2), round(d['value'],2))")
    echo "[$t] $r"
  done
done

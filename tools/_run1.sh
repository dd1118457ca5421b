cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
for i in 1 2; do for v in "" hl lh; do
MT_SPLIT_PRIO=$v python bench.py --no-cpu-baseline --no-legs --steps 12 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('prio=[$v]', round(d['ms_per_step'],2))"
done; done

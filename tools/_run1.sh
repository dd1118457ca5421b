cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_model_gpu.py -x -q -k "golden or graph or tiny or stochastic or optimizer" > gpurun_out/t_split.txt 2>&1; echo rc=$? >> gpurun_out/t_split.txt
tail -n 15 gpurun_out/t_split.txt
for i in 1 2; do
for v in 0 1; do
MT_SPLIT_PASSES=$v python bench.py --no-cpu-baseline --no-legs --steps 12 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('split=$v replay', round(d['ms_per_step'],2), d['launch'])"
MT_SPLIT_PASSES=$v python bench.py --no-cpu-baseline --no-legs --steps 12 --eager 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('split=$v eager', round(d['ms_per_step'],2), d['launch'])"
done; done

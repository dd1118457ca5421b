cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_model_gpu.py tests/test_dp_gpu.py -x -q -k "two_pass_groups or bridge or speculation or recycles or reference_loop or distributed_data" > gpurun_out/t_split.txt 2>&1; echo rc=$? >> gpurun_out/t_split.txt
tail -n 8 gpurun_out/t_split.txt
for i in 1 2; do for v in 0 1; do
MT_SPLIT_PASSES=$v python bench.py --api module --optim fused --steps 10 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('module split=$v', round(d['ms_per_step'],2), 'host', round(d['host_enqueue_ms_per_step'],2))"
done; done

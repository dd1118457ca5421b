cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
python -m pytest tests/test_model_gpu.py -x -q -k "fused_adamw or speculation or bridge or fresh_constructor or module or recycles" > gpurun_out/t_r5a.txt 2>&1; echo rc=$? >> gpurun_out/t_r5a.txt
tail -n 4 gpurun_out/t_r5a.txt
for i in 1 2; do
python bench.py --api module --optim fused --steps 10 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fused', round(d['ms_per_step'],2), 'host', round(d['host_enqueue_ms_per_step'],2), d['task_tokens_read_back'])"
python bench.py --no-cpu-baseline --no-legs --steps 10 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('trainstep', round(d['ms_per_step'],2))"
done

cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_model_gpu.py tests/test_titan_gpu.py -x -q > gpurun_out/t_model.txt 2>&1; echo rc=$? >> gpurun_out/t_model.txt
tail -n 6 gpurun_out/t_model.txt
python bench.py --no-cpu-baseline --no-legs --kernel-times > gpurun_out/b_stage.json 2> gpurun_out/b_stage_kt.txt
python -c "
import json;d=json.loads(open('gpurun_out/b_stage.json').read().strip().splitlines()[-1]);print(d['ms_per_step'])"
grep -n "token_side\|sgemm\|stage" gpurun_out/b_stage_kt.txt | head
python bench.py --config titan --patches 4096 --ragged --steps 16 --warmup 8 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('titan',d['value'],d['ms_per_step'],d.get('step_mfma_frac'))"
python bench.py --api module --optim fused --steps 10 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('module fused', round(d['ms_per_step'],2), 'host', round(d['host_enqueue_ms_per_step'],2))"

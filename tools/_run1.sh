cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_model_gpu.py -x -q -k "gradient_exceptions or tiny_bags or reference_loop or golden or fused_adamw" > gpurun_out/t_model.txt 2>&1; echo rc=$? >> gpurun_out/t_model.txt
tail -n 5 gpurun_out/t_model.txt
python bench.py --no-cpu-baseline --no-legs --kernel-times > gpurun_out/b_tok.json 2> gpurun_out/b_tok_kt.txt
python -c "
import json;d=json.loads(open('gpurun_out/b_tok.json').read().strip().splitlines()[-1]);print(d['ms_per_step'], d['token_side_in_graph'], d['skipped_steps'])"
grep -n "token_side\|elementwise\|copy_rows" gpurun_out/b_tok_kt.txt | head

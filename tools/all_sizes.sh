for a in "--patches 512" "--patches 4096" "--patches 25000" "--pathways real" "--no-dropout"; do
  python bench.py --no-cpu-baseline $a 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$a', round(d['value'],2), round(d['ms_per_step'],2), round(d['step_mfma_frac'],3))"
done
python tools/eval_bench.py 2>/dev/null | tail -2
python tools/pipeline_bench.py 2>/dev/null | tail -3

cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_titan_gpu.py tests/test_model_gpu.py -x -q -k "titan or captured or eval or embedding" > gpurun_out/t_model.txt 2>&1; echo rc=$? >> gpurun_out/t_model.txt
tail -n 5 gpurun_out/t_model.txt

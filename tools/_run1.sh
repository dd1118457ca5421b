cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_dp_gpu.py -x -q -k "pass_groups or matches_gradient" > gpurun_out/t_split.txt 2>&1; echo rc=$? >> gpurun_out/t_split.txt
tail -n 12 gpurun_out/t_split.txt

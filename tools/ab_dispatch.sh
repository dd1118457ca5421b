for i in 1 2; do for v in build_variants/origdisp/libmodaltune_hip.so build_variants/olddisp/libmodaltune_hip.so modaltune_amd/_C/libmodaltune_hip.so; do
  a=$(MODALTUNE_HIP_LIB=$PWD/$v python bench.py --no-legs --no-cpu-baseline 2>/dev/null | python -c "import sys,json; print(round(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'],3))")
  b=$(MODALTUNE_HIP_LIB=$PWD/$v python bench.py --patches 4096 --no-legs --no-cpu-baseline 2>/dev/null | python -c "import sys,json; print(round(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'],3))")
  c=$(MODALTUNE_HIP_LIB=$PWD/$v python bench.py --config titan --patches 4096 --ragged --steps 16 --warmup 8 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; print(round(json.loads(sys.stdin.read().strip().splitlines()[-1])['value'],2))")
  echo "[$(basename $(dirname $v))] L10000 $a ms  L4096 $b ms  titan $c slides/s"
done; done

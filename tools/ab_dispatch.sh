#!/bin/bash
# Same-box A/B of three builds of the library inside the step (headline, L = 4 096, TITAN), two rounds: the gemm_nt dispatch before round 6's
# sweep, after its first pair of rule changes, and the tree's (profiles/r06_gemm_candidates.txt, "IN THE STEP").  The two older builds were made
# from history: `git show <commit>:modaltune_amd/csrc/gemm.hip` (and gemm_ps.hip) compiled with the build's flags into build_variants/<name>/ and
# linked with the other objects of modaltune_amd/_C/ (as tools/build_variant.sh does); build_variants/ is not kept in the repository.
for i in 1 2; do for v in build_variants/origdisp/libmodaltune_hip.so build_variants/olddisp/libmodaltune_hip.so modaltune_amd/_C/libmodaltune_hip.so; do
  a=$(MODALTUNE_HIP_LIB=$PWD/$v python bench.py --no-legs --no-cpu-baseline 2>/dev/null | python -c "import sys,json; print(round(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'],3))")
  b=$(MODALTUNE_HIP_LIB=$PWD/$v python bench.py --patches 4096 --no-legs --no-cpu-baseline 2>/dev/null | python -c "import sys,json; print(round(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'],3))")
  c=$(MODALTUNE_HIP_LIB=$PWD/$v python bench.py --config titan --patches 4096 --ragged --steps 16 --warmup 8 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; print(round(json.loads(sys.stdin.read().strip().splitlines()[-1])['value'],2))")
  echo "[$(basename $(dirname $v))] L10000 $a ms  L4096 $b ms  titan $c slides/s"
done; done

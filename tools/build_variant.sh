#!/bin/bash
# Build a variant of the C-ABI library for same-box A/B timing: tools/build_variant.sh <name> <file.hip> [extra hipcc flags...]
# Compiles modaltune_amd/csrc/<file.hip> with the extra flags and links it with the other
# objects of the regular build (python __graft_entry__.py first) into build_variants/<name>/ (binaries: ignored by git, shipped by gpurun).
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
name="$1"; src="$2"; shift 2
out="$ROOT/build_variants/$name"; mkdir -p "$out"
base="$(basename "$src" .hip)"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -mllvm -amdgpu-mfma-vgpr-form -I"$ROOT/include" -I"$ROOT/modaltune_amd/csrc" "$@" -c "$ROOT/modaltune_amd/csrc/$src" -o "$out/$base.o"
objs=$(ls "$ROOT"/modaltune_amd/_C/*.o | grep -v "/$base.o" | grep -v "/build_id.o")
# (the variant names itself: its build id never matches the tree's)
echo "extern \"C\" const char* mt_build_id(void) { return \"variant:$name\"; }" > "$out/build_id.cpp"
/opt/rocm/bin/hipcc -O1 -fPIC -c "$out/build_id.cpp" -o "$out/build_id.o"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$out/libmodaltune_hip.so" "$out/$base.o" "$out/build_id.o" $objs
echo "$out/libmodaltune_hip.so"

#!/bin/bash
# Build libgad_hip.so for gfx950 in-tree (the .so travels to the GPU box with the snapshot).
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
OUT="$HERE/gad/libgad_hip.so"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$HERE/../include -I$HERE/csrc -Wno-unused-result"
mkdir -p "$HERE/build"
pids=()
for f in gemm_f32 wino4_fused attention norm elementwise optim transformer half; do
  src="$HERE/csrc/$f.hip"; obj="$HERE/build/$f.o"
  if [ ! -f "$obj" ] || [ "$src" -nt "$obj" ] || [ "$HERE/csrc/gad_common.h" -nt "$obj" ] || [ "$HERE/csrc/gad_reduce.h" -nt "$obj" ] || [ "$HERE/../include/gad.h" -nt "$obj" ] || [ "$HERE/csrc/gemm_dev.h" -nt "$obj" ]; then
    extra=""
    # attention.hip: MFMA results feed the vector ALUs directly (softmax, dS), so keep them in VGPRs: with the default AGPR
    # form hipcc moved every score / gradient tile through v_accvgpr_read / _write (50-150 instructions per K/V tile)
    [ "$f" = attention ] && extra="-mllvm -amdgpu-mfma-vgpr-form=1"
    # wino4_fused.hip: its accumulator folds run beside MFMAs, where packed f32 VALU (what SLP makes of them) is slower than scalar
    [ "$f" = wino4_fused ] && extra="-fno-slp-vectorize"
    $HIPCC $FLAGS $extra -c "$src" -o "$obj" &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait $p; done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o "$OUT" "$HERE"/build/{gemm_f32,wino4_fused,attention,norm,elementwise,optim,transformer,half}.o
echo "built $OUT"

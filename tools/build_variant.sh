#!/bin/bash
# build_variant.sh <name> <extra flags...> : builds tools/_build/libncg_<name>.so from the current sources with extra flags
set -e
NAME=$1; shift
SRC=/root/repo/noble-curves_amd/csrc
OUT=/root/repo/tools/_build
B=${TMPDIR:-/tmp}/abbuild/$NAME
mkdir -p $B $OUT
cd $SRC
pids=""
for f in api comm mulvar mulvar_inl mulvar_endo ubench msm msm_g1 msm_precomp msm_endo ecdsa ed25519 mulbase decode ntt h2c; do
  fl=""
  case $f in msm) fl="-DNCG_MUL_INLINE=1 -DNCG_FE29_COLS_PAIRED=1";; msm_g1) fl="-DNCG_MUL_INLINE=1 -DNCG_FE29_COLS_PAIRED=1 -mllvm -amdgpu-sched-strategy=max-ilp";; mulbase|ed25519) fl="-DNCG_MUL_INLINE=1";; esac   # as in the Makefile
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Wno-unused-function $fl "$@" -c $f.hip -o $B/$f.o ) &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libncg_$NAME.so $B/*.o -ldl
ls -la $OUT/libncg_$NAME.so

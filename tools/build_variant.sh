#!/bin/bash
# usage: tools/build_variant.sh NAME file.hip [-DDEFS ...]  -> mapperatorinator_amd/lib/libmapperhip_NAME.so
# (only `file.hip` is recompiled with the defines; every other object comes from build/obj -- run `make` first)
set -e
cd "$(dirname "$0")/.."
name=$1; src=$2; shift 2
mkdir -p build/var_$name
base=$(basename $src .hip)
extra=; if [ $base = attention ]; then extra=-fno-slp-vectorize; fi
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -Wno-pass-failed -mllvm -amdgpu-kernarg-preload-count=14 $extra "$@" -c mapperatorinator_amd/csrc/$src -o build/var_$name/$base.o
objs=$(ls build/obj/*.o | grep -v "/$base.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs build/var_$name/$base.o -o mapperatorinator_amd/lib/libmapperhip_$name.so
echo built mapperatorinator_amd/lib/libmapperhip_$name.so

#!/bin/bash
# A/B builds of ONE kernel file: tools/build_variant.sh <name> <file.hip> <extra flags...>  ->  tools/_bin/<name>/libr2l_hip.so
# (the file compiled with the extra flags, linked with the objects of the regular build in r2l_amd/lib/obj)
set -e
cd "$(dirname "$0")/.."
name=$1; src=$2; shift 2
python -m r2l_amd.build > /dev/null
mkdir -p tools/_bin/$name
base=$(basename $src .hip)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function "$@" -I include -c r2l_amd/csrc/$src -o tools/_bin/$name/$base.o
objs=$(ls r2l_amd/lib/obj/*.o | grep -v "/$base.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/_bin/$name/libr2l_hip.so $objs tools/_bin/$name/$base.o -lz
echo "built tools/_bin/$name/libr2l_hip.so"

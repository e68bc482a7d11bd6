set -e
cd /root/repo
name=$1; shift
mkdir -p tools/_bin/$name
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function -I include"
objs=""
for f in r2l_amd/lib/obj/*.o; do objs="$objs $f"; done
for spec in "$@"; do
  src=${spec%%:*}; flags=${spec#*:}
  base=$(basename $src .hip)
  /opt/rocm/bin/hipcc $FL $flags -c r2l_amd/csrc/$src -o tools/_bin/$name/$base.o
  objs=$(echo $objs | tr ' ' '\n' | grep -v "/$base.o" | tr '\n' ' ')
  objs="$objs tools/_bin/$name/$base.o"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/_bin/$name/libr2l_hip.so $objs -lz
echo built $name

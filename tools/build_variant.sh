#!/bin/bash
# build container: tools/ab/<name>.so = the library with pk_forest_q.hip (or the file named by
# SRC=) compiled with extra flags, the other objects as they are.
# usage: tools/build_variant.sh <name> [hipcc flags, e.g. '-DPK_QR_AT(pos)=((pos)+2)']
set -e
root="$(cd "$(dirname "$0")/.." && pwd)"
name="$1"; shift
src="${SRC:-pk_forest_q}"
mkdir -p "$root/tools/ab"
cd "$root/peakachu_amd/csrc"
make -s -j8 >/dev/null
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math \
    -Wno-unused-function -Wno-unused-value -Wno-unused-result -Wno-pass-failed "$@" \
    -c "$src.hip" -o "$root/tools/ab/$name.$src.o" 2>&1 | grep -v "argument unused" || true
objs=""
for o in pk_api pk_extract pk_forest pk_forest_img pk_image pk_forest_q pk_qimage pk_compact pk_comm pk_hostio; do
  if [ "$o" = "$src" ]; then objs="$objs $root/tools/ab/$name.$src.o"; else objs="$objs $o.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$root/tools/ab/$name.so" $objs -L/opt/rocm/lib -lrccl -ldl -lpthread
rm -f "$root/tools/ab/$name.$src.o"
echo "built tools/ab/$name.so"

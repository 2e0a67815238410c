#!/bin/bash
# tools/mkvariant.sh <name> "<extra hipcc flags>" [file ...]  --  an A/B build of the library: the named device files (default
# dx_qv) compiled with the extra flags into build_var/<name>/, linked with the main build's other objects into
# tools/variants/libdexgpu_<name>.so (DEXGPU_LIB selects it; tools/ab.sh, tools/microbench/hist_time.py)
set -e
name=$1; extra=$2; shift 2
files=${@:-dx_qv}
make lib > /dev/null
mkdir -p build_var/$name tools/variants
objs=""
for o in dx_ctx dx_pack2 dx_qv dx_qv_decode dx_synth dx_index dx_qv_walk; do
  if [[ " $files " == *" $o "* ]]; then
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Iinclude -Idextractor_amd/csrc -Wall -Wno-unused-function $extra \
        -Rpass-analysis=kernel-resource-usage -c dextractor_amd/csrc/$o.hip -o build_var/$name/$o.o 2> build_var/$name/$o.res || { grep -v "remark" build_var/$name/$o.res | head -20; exit 1; }
    objs="$objs build_var/$name/$o.o"
  else
    objs="$objs build/$o.o"
  fi
done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o tools/variants/libdexgpu_$name.so $objs build/dx_host.o build/dx_files.o build/dx_compat.o \
    -Wl,-rpath,/opt/rocm/lib -Wl,-soname,libdexgpu.so -lpthread
for f in $files; do
  sed -n 's/.*remark: *//p' build_var/$name/$f.res | sed 's/ \[-Rpass-analysis=kernel-resource-usage\]//' | \
    awk '/^Function Name:/ { if (n) print n, v, s, l, o; n = $3 } /^VGPRs:/ { v = "vgprs=" $2 } /^ScratchSize/ { s = "scratch=" $NF } /^LDS Size/ { l = "lds=" $NF } /^Occupancy/ { o = "waves_per_simd=" $NF } END { if (n) print n, v, s, l, o }' \
    | grep -i "${GREP:-hist}" | sed "s/^/$name: /"
done

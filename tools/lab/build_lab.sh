# lab build: libmeant_hip with extra -D flags into tools/lab/lib_<TAG>.so (objects in /tmp): bash tools/lab/build_lab.sh TAG -DATTN_LAB_STAMP
set -e
TAG=$1; shift
R=$(cd $(dirname $0)/../.. && pwd); C=$R/meant_amd/csrc; O=/tmp/lab_$TAG; mkdir -p $O
for f in $C/*.hip; do
  b=$(basename $f .hip); x=""
  case $b in attn_bf16|attn_bwd1) x="-fno-slp-vectorize";; esac
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-unused-variable $x "$@" -c $f -o $O/$b.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $O/*.o -o $R/tools/lab/lib_$TAG.so
ls -la $R/tools/lab/lib_$TAG.so

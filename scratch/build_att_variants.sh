#!/bin/bash
# Build libbsr variants with extra -D flags into scratch/libatt_<name>.so (timed by scratch/att_diag.py on the GPU box):
#   bash scratch/build_att_variants.sh name1 "-DX=1" name2 "-DX=2 -DY=1" ...
cd "$(dirname "$0")/../blindshadowremoval_amd" || exit 1
rm -f ../scratch/libatt_*.so
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -Wno-unused-value $flags -o ../scratch/libatt_$name.so csrc/bsr_api.hip &
done
wait
ls -la ../scratch/libatt_*.so

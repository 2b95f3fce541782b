#!/bin/bash
# Build kernel variants side by side (in this container), to compare them on ONE GPU box in ONE gpurun call:
#   tools/ab_build.sh name1 "-DFLAG=1" name2 "-DOTHER=0" ...   ->  orb_slam3_detailed_comments_kor_amd/liborbfe_<name>.so
# run with ORBFE_LIB=$PWD/orb_slam3_detailed_comments_kor_amd/liborbfe_<name>.so python bench.py ...
set -e
cd "$(dirname "$0")/../orb_slam3_detailed_comments_kor_amd/csrc"
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -Wall -Wno-unused-result \
      $flags -shared -o ../liborbfe_$name.so orbfe_extractor.hip orbfe_matcher.hip orbfe_multicam.hip -ldl -lrt -pthread &
done
wait
ls -la ../liborbfe_*.so

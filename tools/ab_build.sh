#!/bin/bash
# A/B of compile-time variants on ONE box: a second copy of the package under ab/<name>/ built with extra compiler flags
# (e.g. -DSELLA_NO_ARG_BATCH); tools that honour SELLA_AB_ROOT (eigh_only.py) import the package from there.
#   tools/ab_build.sh noarg -DSELLA_NO_ARG_BATCH ;  SELLA_AB_ROOT=ab/noarg python tools/eigh_only.py 3072 6
set -e
cd "$(dirname "$0")/.."
NAME=$1; shift
D=ab/$NAME/sella_amd
rm -rf ab/$NAME; mkdir -p ab/$NAME
cp -r sella_amd ab/$NAME/; rm -rf $D/_obj $D/libsella_hip.so $D/__pycache__ $D/*/__pycache__; mkdir -p $D/_obj
mkdir -p ab/$NAME/include; cp include/*.h ab/$NAME/include/
for f in $D/csrc/*.hip; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-result -Xarch_host -march=x86-64-v3 -Xarch_host -ffp-contract=off "$@" -c $f -o $D/_obj/$(basename ${f%.hip}).o &
done
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $D/_obj/*.o -o $D/libsella_hip.so
rm -rf $D/_obj
echo "built $D/libsella_hip.so ($*)"

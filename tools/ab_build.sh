#!/bin/bash
# Builds the library from the working tree into build_ab/<name>.so with extra compiler flags: bash tools/ab_build.sh <name> [flags...]
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
tmp=$(mktemp -d)
mkdir -p $tmp/visual-slam_amd $tmp/include
cp -r $root/visual-slam_amd/csrc $tmp/visual-slam_amd/
cp $root/include/*.h $root/include/*.inc $tmp/include/
(cd $tmp/visual-slam_amd/csrc && rm -f *.o *.so && make -j8 CXXFLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function $*" 2>&1 | grep -E "error|warning: unused" || true)
mkdir -p $root/build_ab
cp $tmp/visual-slam_amd/csrc/liborbx.so $root/build_ab/$name.so
rm -rf $tmp
ls -la $root/build_ab/$name.so

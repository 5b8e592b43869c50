#!/bin/bash
# Build a variant of the library for same-box A/B timing (tools/ab_libs.sh): tools/build_variant.sh <name> [-DKNOB=value ...]
# -> abl_libs/lib_<name>.so.  The tree's own build/ and libshotfpfh.so are not touched.
set -e
cd "$(dirname "$0")/.."
name=$1; shift
work=/tmp/sf_variant_$name
mkdir -p $work/shot_fpfh_amd/csrc $work/include abl_libs
cp shot_fpfh_amd/csrc/*.hip shot_fpfh_amd/csrc/*.h shot_fpfh_amd/csrc/Makefile $work/shot_fpfh_amd/csrc/
cp include/shotfpfh.h $work/include/
# (DESC_EXTRA in the environment: flags for shot.hip alone)
make -s -C $work/shot_fpfh_amd/csrc -j8 DESC_EXTRA="${DESC_EXTRA:-}" CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -Wall -Wno-unused-result -I/opt/rocm/include $*" > $work/log 2>&1 || { tail -20 $work/log; exit 1; }
cp $work/shot_fpfh_amd/libshotfpfh.so abl_libs/lib_$name.so
echo "abl_libs/lib_$name.so  ($*)"

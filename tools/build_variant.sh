#!/bin/bash
# Build an experimental variant of libafcm_hip.so: tools/build_variant.sh NAME SOURCE.hip "EXTRA FLAGS"
# -> afcm_amd/csrc/variants/NAME.so (select at run time with AFCM_HIP_LIB=<path>).  Only SOURCE is recompiled.
set -e
cd "$(dirname "$0")/../afcm_amd/csrc"
name=$1; src=$2; shift 2
mkdir -p variants
per_file=""; [ "$src" = upfirdn2d.hip ] && per_file="-fno-slp-vectorize"      # the Makefile's per-file flags
make -s -j8 >/dev/null 2>&1 || make
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wno-unused-function -mllvm -amdgpu-mfma-vgpr-form=1 $per_file "$@" -c $src -o variants/$name.o 2> variants/$name.log || { tail -20 variants/$name.log; exit 1; }
objs=$(ls *.o | grep -v "^${src%.hip}.o$")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o variants/$name.so $objs variants/$name.o
echo built variants/$name.so

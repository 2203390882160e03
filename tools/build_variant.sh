#!/bin/bash
# build a library variant from a source tree for same-box A/B runs:  tools/build_variant.sh <csrc dir> <out .so> [extra flags]
# (product flags of build.py; the probe kernels are left out like in the product)
SRC=$1; OUT=$2; shift 2
FILES=$(ls $SRC/*.hip | grep -v ssal_probe.hip)
hipcc --offload-arch=${ARCH:-gfx950:xnack-} -O3 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden -Wall -shared "$@" -o $OUT $FILES

#!/bin/bash
# One-file variant of the library: recompile ONE source with extra flags and link it with the main build's other objects.
#   bash scripts/exp/variant_lib.sh <name> <source.hip> "<flags>" [other-version-of-that-source]   -> basedet_amd/lib/libbd_<name>.so
#   (run python -m basedet_amd.build first; the optional 4th argument, e.g. a file written by `git show REV:path`, is compiled in place of the source)
cd "$(dirname "$0")/../.."
N=$1; SRC=$2; FLAGS=$3; ALT=${4:-basedet_amd/csrc/$SRC}; L=basedet_amd/lib
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-result -I include -I basedet_amd/csrc $FLAGS -x hip -c $ALT -o /tmp/variant_$N.o || exit 1
OBJS=$(ls $L/*.o | grep -v "/libbd_" | grep -v "/${SRC%.hip}.o")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $L/libbd_$N.so $OBJS /tmp/variant_$N.o && echo $L/libbd_$N.so

#!/bin/bash
# usage: tools/build_variant.sh <name> "<extra hipcc flags>"  ->  variants/<name>/{librt_hip.so,librt_host.so}
# (a full build of the kernel library with other -D settings, beside the default one; bench with RT_HIP_LIB_DIR=variants/<name>)
set -e
name=$1; flags=$2
make -j8 hip OBJDIR=build/obj_$name LIBDIR=variants/$name EXTRA_HIPFLAGS="$flags" > /dev/null
make host OBJDIR=build/obj_$name LIBDIR=variants/$name > /dev/null
ls -la variants/$name

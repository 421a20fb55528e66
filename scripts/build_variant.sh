#!/bin/bash
# usage: scripts/build_variant.sh <name> <file.hip> [extra hipcc flags]  ->  neurad_studio_amd/lib/variants/lib_<name>.so
# The library with ONE source recompiled under extra flags (experiment hooks are -D macros); run anything against it with
# NEURAD_HIP_LIB=neurad_studio_amd/lib/variants/lib_<name>.so.  Needs a prior `python __graft_entry__.py`.
set -e
name=$1; src=$2; shift; shift
R=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $R/neurad_studio_amd/lib/variants
base=$(basename $src .hip)
objs=$(ls $R/neurad_studio_amd/lib/obj/*.o | grep -v "/$base\.")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -ffp-contract=off -Wno-unused-result \
  -I$R/include "$@" -c $R/neurad_studio_amd/csrc/$base.hip -o /tmp/variant_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs /tmp/variant_$name.o -o $R/neurad_studio_amd/lib/variants/lib_$name.so
echo built $R/neurad_studio_amd/lib/variants/lib_$name.so

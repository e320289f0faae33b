#!/bin/bash
# tools/build_variant.sh NAME [-DFLAG=V ...]: a second build of the same ABI under variants/libenarf_NAME.so
# (git-ignored; used through bench.py --allow-variant --variant PATH / ENARF_VARIANT=NAME of the tools for A/B runs and the phase-timer diagnostic build)
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/variants; mkdir -p $out/obj_$name
csrc=$root/enarf-gan_amd/csrc
for f in enarf_render enarf_render_bwd enarf_sampler enarf_raysample enarf_gan_ops; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -Wall -Wno-unused-function -I$root/include -I$csrc "$@" -c $csrc/$f.hip -o $out/obj_$name/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out/libenarf_$name.so $out/obj_$name/*.o
echo $out/libenarf_$name.so

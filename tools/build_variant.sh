#!/bin/bash
# build an engine variant for in-session A/B runs: tools/build_variant.sh <name> [ENV=VALUE ...] [-- hipcc flags]
# the generator options (KASM_*) go through the environment, compiler defines after "--"
NAME=$1; shift
ENVS=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do ENVS+=("$1"); shift; done; [ "$1" == "--" ] && shift
cd $(dirname $0)/..
env "${ENVS[@]}" python tools/gen_walk_asm.py > /dev/null || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared "$@" -o kangaroo_amd/lib/libkangaroo_hip_$NAME.so kangaroo_amd/csrc/kng_engine.hip 2>&1 | grep -E "error|rror:" 
python tools/gen_walk_asm.py > /dev/null   # restore the default header
ls -la kangaroo_amd/lib/libkangaroo_hip_$NAME.so

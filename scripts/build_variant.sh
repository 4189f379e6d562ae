# Builds match.hip with extra flags into sfm_amd/lib/ab/<name>.so (the other objects are the tree's):
#   scripts/build_variant.sh <name> [flags...]
set -e
name=$1; shift
cd "$(dirname "$0")/../sfm_amd/csrc"
mkdir -p ../lib/ab
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -Wno-unused-function "$@" -c match.hip -o /tmp/match_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/ab/$name.so context.o /tmp/match_$name.o ransac.o pipeline.o chain.o hamming.o comm.o -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib
echo built ../lib/ab/$name.so

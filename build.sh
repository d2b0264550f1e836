#!/bin/bash
# Builds the product library (HIP, gfx950) and the CPU oracle (plain C, test infrastructure).
# Called by __graft_entry__.build(); safe to run by hand.  hipcc cross-compiles without a GPU.
#   air_rs_amd/lib/libadsb_hip.so               the product: ONE i8 scan kernel (floor(sqrt) per sample) + CS16's
#   air_rs_amd/lib/variants/libadsb_hip_ab.so   the same sources with -DADSB_AB_KERNELS=1: also the A/B scan kernels (code, nsq,
#                                               reg) round 3-4 measured against the product's; loaded only by
#                                               tests/test_gpu_ab_kernels.py (one parity smoke each) and tools/gpu/ab.sh
set -euo pipefail
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
mkdir -p air_rs_amd/lib/variants
SRC=air_rs_amd/csrc
FILES="$SRC/adsb_kernels.hip $SRC/adsb_track.hip $SRC/adsb_api.cpp $SRC/adsb_group.cpp
    $SRC/host/adsb_packet.cpp $SRC/host/adsb_aircraft.cpp $SRC/host/adsb_threads.cpp $SRC/host/adsb_host_api.cpp"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -pthread -Wall -Wno-unused-function"
$HIPCC $FLAGS $FILES -o air_rs_amd/lib/libadsb_hip.so &
$HIPCC $FLAGS -DADSB_AB_KERNELS=1 $FILES -o air_rs_amd/lib/variants/libadsb_hip_ab.so &
gcc -O3 -std=c99 -fPIC -shared -Wall -Wextra oracle/adsb_oracle.c -o oracle/libadsb_oracle.so -lm
wait
test -s air_rs_amd/lib/libadsb_hip.so && test -s air_rs_amd/lib/variants/libadsb_hip_ab.so
echo "built air_rs_amd/lib/libadsb_hip.so air_rs_amd/lib/variants/libadsb_hip_ab.so oracle/libadsb_oracle.so"

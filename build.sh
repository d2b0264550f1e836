#!/bin/bash
# Builds the product library (HIP, gfx950) and the CPU oracle (plain C, test infrastructure).
# Called by __graft_entry__.build(); safe to run by hand.  hipcc cross-compiles without a GPU.
set -euo pipefail
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
mkdir -p air_rs_amd/lib
SRC=air_rs_amd/csrc
$HIPCC --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -pthread \
    -Wall -Wno-unused-function \
    $SRC/adsb_kernels.hip $SRC/adsb_track.hip $SRC/adsb_api.cpp $SRC/adsb_group.cpp \
    $SRC/host/adsb_packet.cpp $SRC/host/adsb_aircraft.cpp $SRC/host/adsb_threads.cpp $SRC/host/adsb_host_api.cpp \
    -o air_rs_amd/lib/libadsb_hip.so
gcc -O3 -std=c99 -fPIC -shared -Wall -Wextra oracle/adsb_oracle.c -o oracle/libadsb_oracle.so -lm
echo "built air_rs_amd/lib/libadsb_hip.so oracle/libadsb_oracle.so"

#!/bin/bash
# tools/build_variant.sh NAME [extra hipcc flags...] -> air_rs_amd/lib/variants/libadsb_hip_NAME.so
set -euo pipefail
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p air_rs_amd/lib/variants
SRC=air_rs_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -pthread -Wno-unused-function "$@" \
    $SRC/adsb_kernels.hip $SRC/adsb_track.hip $SRC/adsb_api.cpp $SRC/adsb_group.cpp $SRC/host/adsb_packet.cpp $SRC/host/adsb_aircraft.cpp $SRC/host/adsb_threads.cpp $SRC/host/adsb_host_api.cpp \
    -o air_rs_amd/lib/variants/libadsb_hip_$name.so
echo built $name

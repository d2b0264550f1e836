#!/bin/bash
# usage: mkvar.sh NAME FLAGS...
set -e
cd /root/repo
name=$1; shift
SRC=air_rs_amd/csrc
FILES="$SRC/adsb_kernels.hip $SRC/adsb_track.hip $SRC/adsb_api.cpp $SRC/adsb_group.cpp $SRC/host/adsb_packet.cpp $SRC/host/adsb_aircraft.cpp $SRC/host/adsb_threads.cpp $SRC/host/adsb_host_api.cpp"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -pthread -Wall -Wno-unused-function "$@" $FILES -o air_rs_amd/lib/variants/libadsb_hip_$name.so
echo built $name

#!/bin/bash
# Development tool: the kernel sources under the host-simulation shim (tests/hostsim) built with AddressSanitizer, and a part of the
# CPU test tier / the structure fuzz run on it.  The "device" arrays of the host simulation are heap blocks: an element read or
# written outside its array is reported with the source line of the kernel (round 5: the (Z, z) pair of slack 0 read by the GEN row
# functions in a batch without slacks -- one element per instance allocated, a GPU fault only where the array ended on a page).
#     bash tools/asan_hostsim.sh build
#     bash tools/asan_hostsim.sh pytest tests/test_host_logic.py -x -q          (any pytest arguments)
#     bash tools/asan_hostsim.sh fuzz 8000 8030                                  (tools/fuzz_parity.py hostsim <lo> <hi>)
root=$(cd "$(dirname "$0")/.." && pwd)
lib=$root/tools/ab/libgqp_hostsim_asan.so
# (libstdc++ beside it: ASan resolves __cxa_throw when it starts -- inside python, which has no C++ runtime yet, the first exception
# of the library would end in "CHECK failed: real___cxa_throw != 0")
asan="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libstdc++.so.6)"
mkdir -p $root/tools/ab
build() {
    (cd $root/tests/hostsim && g++ -O1 -g -fsanitize=address -fno-omit-frame-pointer -std=c++17 -fPIC -shared -x c++ -Wno-unknown-pragmas \
        -Iinclude -I../../include -I../../acados_amd/csrc ../../acados_amd/csrc/gpu_batch.hip ../../acados_amd/csrc/gpu_shapes_large.hip \
        ../../acados_amd/csrc/ocp_qp_host.cpp ../../acados_amd/csrc/ocp_qp_xcond.cpp -o $lib)
}
case "$1" in
    build) build ;;
    pytest) shift; [ -f $lib ] || build
        LD_PRELOAD="$asan" ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 ACADOS_AMD_HOSTSIM_LIB=$lib python -m pytest "$@" ;;
    fuzz) [ -f $lib ] || build
        LD_PRELOAD="$asan" ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 FUZZ_HOSTSIM_LIB=$lib python $root/tools/fuzz_parity.py hostsim $2 $3 ;;
    *) echo "usage: $0 build | pytest <args> | fuzz <lo> <hi>" ;;
esac

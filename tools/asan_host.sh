#!/bin/bash
# CPU container: the library's HOST side (ordering, symbolic + numeric factor, assembly, sharding, all-reduce hooks) under AddressSanitizer +
# UBSan.  The host translation units are rebuilt with g++ -fsanitize=address,undefined and linked with the (unsanitized) device unit
# into tests/_build/asan/libadmm_hip_asan.so; then the CPU tests that go through the library's host-only mode (device_id = -1) run against it.
#   tools/asan_host.sh [pytest args]       default: tests/test_factor_host.py tests/test_generic_host.py tests/test_abi.py tests/test_sharding.py
cd "$(dirname "$0")/.."
P=admm-elastic-sca_amd; B=tests/_build/asan; mkdir -p $B
python3 -c "import __graft_entry__ as g; g.build()" > /dev/null || exit 1       # the device unit's object (_build/admm_hip.o)
SAN="-fsanitize=address,undefined -fno-omit-frame-pointer -g -O1 -std=c++17 -fPIC -fopenmp -Wno-unknown-pragmas"
for f in dense factor; do g++ $SAN -mavx2 -mfma -c $P/csrc/$f.cpp -o $B/$f.o || exit 1; done
for f in host_setup partition comm; do g++ $SAN -x c++ -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -c $P/csrc/$f.cpp -o $B/$f.o || exit 1; done
g++ -shared -fsanitize=address,undefined -o $B/libadmm_hip_asan.so $B/{dense,factor,host_setup,partition,comm}.o $P/_build/admm_hip.o -L/opt/rocm/lib -lamdhip64 -lgomp -Wl,-rpath,/opt/rocm/lib || exit 1
args=("$@"); [ ${#args[@]} -eq 0 ] && args=(tests/test_factor_host.py tests/test_generic_host.py tests/test_abi.py tests/test_sharding.py)
ASAN_OPTIONS=detect_leaks=0:abort_on_error=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)" ADMM_HIP_LIB=$PWD/$B/libadmm_hip_asan.so \
  python3 -m pytest "${args[@]}" -x -q -m "not gpu" -p no:cacheprovider || exit 1
# the header-only host classes (Comm.hpp: shared-memory all-reduce, rendezvous file; System.hpp) inside sanitized test programs
[ $# -eq 0 ] && ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 ADMM_TEST_CXXFLAGS="-fsanitize=address,undefined -fno-omit-frame-pointer -g" \
  python3 -m pytest tests/test_cpp_host.py tests/test_scene_ingest.py -x -q -m "not gpu" -p no:cacheprovider
rc=$?
rm -rf $B tests/_build/*_san      # sanitized binaries are large and would travel with every gpurun snapshot
exit $rc

#!/bin/bash
# AddressSanitizer + UndefinedBehaviorSanitizer over the CPU-side native code (the oracle and the host library: scene / OBJ
# loaders, PNG / HDR writers).  CPU build only: GPU sanitizers are not available on the pool.
#   bash tests/run_sanitizers.sh      -> "sanitizer run finished" and exit 0 when clean
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
export PT_SAN_DIR=${PT_SAN_DIR:-/tmp/pt_san}
mkdir -p "$PT_SAN_DIR"
SAN="-O1 -g -ffp-contract=off -fno-fast-math -fPIC -shared -fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer"
g++ -std=c++11 $SAN -o "$PT_SAN_DIR/libptoracle.so" "$ROOT/oracle/pt_oracle.cpp"
H="$ROOT/project3-cuda-path-tracer_amd/host"
g++ -std=c++11 $SAN -I"$ROOT/include" -o "$PT_SAN_DIR/libpt_host.so" "$H/scene.cpp" "$H/utilities.cpp" "$H/image.cpp" "$H/c_api.cpp"
LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0 UBSAN_OPTIONS=print_stacktrace=1 \
    python3 "$ROOT/tests/sanitizer_driver.py"

#!/usr/bin/env bash
# AddressSanitizer + UBSan over the CPU builds (the C oracle and the g++ emulator of the kernel phases).
# GPU sanitizers are not available on the pool; the phase code is the same source the device compiles.
set -euo pipefail
cd "$(dirname "$0")/.."
SAN="-fsanitize=address,undefined -fno-omit-frame-pointer -fno-sanitize-recover=undefined"
make -s -C tests/emu clean
make -s -C tests/emu CXXFLAGS="-O1 -g -fPIC -std=c++17 -ffp-contract=off -fno-fast-math -Wno-unknown-pragmas $SAN"
make -s -C oracle clean
make -s -C oracle CFLAGS="-O1 -g -fPIC -std=c11 -ffp-contract=off -fno-fast-math $SAN"
LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)" ASAN_OPTIONS=detect_leaks=0 \
    python -m pytest tests/test_kernel_emulated.py tests/test_oracle_golden.py tests/test_fuzz_emulated.py \
    tests/test_capi.py tests/test_actor.py tests/test_geo_emulated.py tests/test_oracle_geo.py tests/test_compat_class.py \
    tests/test_home_block_emulated.py tests/test_ctor_kwargs.py \
    -x -q -p no:cacheprovider
# back to the normal builds
make -s -C tests/emu clean && make -s -C tests/emu
make -s -C oracle clean && make -s -C oracle

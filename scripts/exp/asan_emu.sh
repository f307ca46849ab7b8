#!/bin/bash
# EXPERIMENT RECORD / TEST TOOL (round 5): the kernel emulator under AddressSanitizer, every "device" allocation filled with
# garbage (what freed device memory holds in a long-lived process; a fresh process sees zero pages) and the launch geometry of an
# MI355X (KD_EMU_CUS=256: persistent kernels with hundreds of idle workgroups).  Usage:
#   scripts/exp/asan_emu.sh <fill byte 0..255> <python script and its arguments>      (the script must load $EMULIB as its library)
# e.g. the 300 structured --realign files of profiles/r05_unexplained_fault.txt ran clean and identical under fill 255 / 1 / 127.
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
OUT=${EMULIB:-/tmp/libkindel_emu_asan.so}
if [ ! -f "$OUT" ] || [ "$ROOT/tests/emu/emu_lib.cpp" -nt "$OUT" ]; then
  g++ -O1 -g -fsanitize=address -fno-omit-frame-pointer -std=c++20 -shared -fPIC -pthread -Wno-unknown-pragmas \
      "$ROOT/tests/emu/emu_lib.cpp" "$ROOT/kindel_amd/csrc/kd_decode.cpp" -lz -o "$OUT"
fi
FILL=$1; shift
export EMULIB=$OUT KD_EMU_CUS=${KD_EMU_CUS:-256}
LD_PRELOAD=$(g++ -print-file-name=libasan.so) \
ASAN_OPTIONS=detect_leaks=0:detect_stack_use_after_return=0:halt_on_error=1:max_malloc_fill_size=4000000000:malloc_fill_byte=$FILL \
  python "$@"

#!/bin/bash
# Compile tools/repro_wavectx.hip to gfx950 ISA and count the scratch accesses of the two callees
# (16-dword context: register-passed; 17-dword context: passed by reference through scratch).
set -e
here="$(cd "$(dirname "$0")" && pwd)"
tmp="$(mktemp -d)"
cd "$tmp"
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -save-temps -c "$here/repro_wavectx.hip" -o repro.o > /dev/null 2>&1
s=$(ls *gfx950*.s | head -1)
for n in 0 1; do
  awk -v pat="calleeILi${n}E" '$0 ~ "^_Z.*" pat ".*:" {on=1} on && /^\.Lfunc_end/ {on=0} on' "$s" > callee$n.s
  echo "callee<$n> ($((16 + n)) dwords): $(grep -c scratch_load callee$n.s) scratch_load, $(grep -c scratch_store callee$n.s) scratch_store, $(grep -cE '^\s+v_|^\s+s_' callee$n.s) instructions"
done
rm -rf "$tmp"

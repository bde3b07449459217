#!/bin/bash
# K1 under library variants (tools/build_variant.py): `bash tools/k1_ab.sh main a1 a2 -- cfg2:10000 test1`
V=(); while [ "$1" != "--" ]; do V+=("$1"); shift; done; shift
for v in "${V[@]}"; do
  echo "=== variant $v"
  if [ "$v" = main ]; then unset CAFEHIP_LIB; else export CAFEHIP_LIB=tools/_variants/$v/libcafehip.so; fi
  timeout 600 python tools/k2c_ab.py "$@" -- default 2>&1 | grep -v WARNING | grep -v amdgpu.ids
done

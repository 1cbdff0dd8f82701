#!/bin/bash
# A/B of weight-gradient library builds: tools/ws_probe.sh <lib-suffix>...  (stlpose_amd/libstlpose_hip<suffix>.so)
set -e
for v in "$@"; do
  echo "== lib$v"
  for ns in 64 256 512; do
    STLPOSE_HIP_LIB=$PWD/stlpose_amd/libstlpose_hip$v.so timeout -k 10 120 python tools/wgrad_probe.py 32,96,72,32,32,3,$ns,128 2>&1 | grep -v amdgpu.ids
  done
done

#!/bin/bash
# A/B of two builds of the library on ONE box: tools/ab_lib.sh <other .so> <command ...> runs the command alternately with the
# in-tree library and with the other build (VSDE_HIP_LIB), twice each.
other=$1; shift
for i in 1 2; do
  for lib in "" "$other"; do
    VSDE_HIP_LIB=$lib "$@" 2>&1 | sed "s|^|[${lib:+B}${lib:-A}] |" | sed "s|\[B[^]]*\]|[B]|"
  done
done

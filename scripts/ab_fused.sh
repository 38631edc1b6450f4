#!/bin/bash
# usage: ab_fused.sh variant...   (alternates base and the variants three times)
for r in 1 2 3; do
  echo -n "base   : "; python scripts/ab_fused.py
  for v in "$@"; do echo -n "$v : "; MPB_LIB_PATH=$PWD/build_variants/$v.so python scripts/ab_fused.py; done
done

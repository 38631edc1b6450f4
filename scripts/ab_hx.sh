#!/bin/bash
# usage: ab_hx.sh variant...   (alternates the product library and the variants, twice: H = 128 persistent kernel and the MPPI entry)
for r in 1 2; do
  echo -n "base   : "; python scripts/bench_hx.py 2>/dev/null | grep "H=128 d=14" | head -1
  for v in "$@"; do echo -n "$v : "; MPB_LIB_PATH=$PWD/build_variants/$v.so python scripts/bench_hx.py 2>/dev/null | grep "H=128 d=14" | head -1; done
done

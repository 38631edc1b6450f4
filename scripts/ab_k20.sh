#!/bin/bash
# usage: ab_k20.sh variant...   (alternates the product library and build_variants/<variant>.so three times; scripts/ab_k20.py)
for r in 1 2 3; do
  echo -n "base   : "; python scripts/ab_k20.py
  for v in "$@"; do echo -n "$v : "; MPB_LIB_PATH=$PWD/build_variants/$v.so python scripts/ab_k20.py; done
done

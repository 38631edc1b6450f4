for r in 1 2 3; do
  echo -n "base    : "; python scripts/ab_gpmp2.py 2>&1 | grep GPMP2
  for v in skip64 skip256; do echo -n "$v : "; MPB_LIB_PATH=$PWD/build_variants/$v.so python scripts/ab_gpmp2.py 2>&1 | grep GPMP2; done
done

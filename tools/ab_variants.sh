#!/bin/bash
# Build variant libraries for A/B timing: the named translation units are recompiled with extra -D flags, the others are taken
# from the default build.  Usage: tools/ab_variants.sh [-u "trace shade"] name1 "-DFOO=1" name2 "..."  -> build/ab/libhiprz_<name>.so
# (units: api trace shade batch sort build; default "trace shade").  Load one with HIPRZ_LIB=build/ab/libhiprz_<name>.so.
set -eo pipefail
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=$R/build/ab; mkdir -p $OUT
UNITS="trace shade"
if [ "$1" = "-u" ]; then UNITS=$2; shift 2; fi
cd $R/rayzath_amd/csrc
make -s -j8 libhiprz.so
FLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -I../../include -I. --offload-arch=gfx950 -fhip-fp32-correctly-rounded-divide-sqrt -fno-gpu-rdc"
src() { case $1 in api) echo hiprz_api;; sort) echo hiprz_sort;; build) echo hiprz_build;; *) echo hiprz_launch_$1;; esac; }
while [ $# -ge 2 ]; do
  name=$1; defs=$2; shift 2
  (
    objs="hiprz_host.o"
    for u in api trace shade batch sort build; do
      if [[ " $UNITS " == *" $u "* ]]; then
        /opt/rocm/bin/hipcc $FLAGS $defs -c $(src $u).hip -o $OUT/${u}_$name.o
        objs="$objs $OUT/${u}_$name.o"
      else
        objs="$objs $(src $u).o"
      fi
    done
    /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $OUT/libhiprz_$name.so $objs
  ) &
done
wait
ls $OUT/*.so

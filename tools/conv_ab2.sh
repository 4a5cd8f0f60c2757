#!/bin/bash
for args in "--cin 256 --cout 256 --t 16 --h 240 --w 416" "--cin 512 --cout 512 --t 16 --h 120 --w 208" "--cin 1024 --cout 1024 --t 8 --h 60 --w 104"; do
  for lib in "" prio sb priosb; do
    echo -n "lib ${lib:-default}: "; L=""; [ -n "$lib" ] && L=fairygen_amd/csrc/build/ab/libfgc_$lib.so
    FAIRYGEN_HIP_LIB=$L timeout -k 10 120 python tools/microbench.py conv $args --iters 10 2>/dev/null
  done
done

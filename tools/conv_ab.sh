#!/bin/bash
# conv 256x256 kernel forms on the main VAE shapes (one process per form: the override is read once)
for args in "--cin 256 --cout 256 --t 16 --h 240 --w 416" "--cin 512 --cout 512 --t 16 --h 120 --w 208" "--cin 1024 --cout 1024 --t 8 --h 60 --w 104" "--cin 1024 --cout 512 --t 16 --h 120 --w 208" "--cin 512 --cout 512 --t 16 --h 120 --w 208 --kt 1"; do
  for form in ${FORMS:-2568 25681 2560}; do
    echo -n "form $form: "; FAIRYGEN_CONV_TILE=$form timeout -k 10 120 python tools/microbench.py conv $args --iters 10 2>&1 | tail -1
  done
done

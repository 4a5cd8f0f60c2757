"""CPU oracle for the FairyGen animation hot path (Wan2.2-TI2V-5B denoise loop + VAE38 decode).

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product: only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it, and there only as
the checker / the CPU baseline — never as the thing measured or shipped.  ``fairygen_amd`` never
imports this package and fails loudly when its HIP library is missing.

The oracle is a from-scratch, functional (state-dict in, tensor out) restatement in plain PyTorch-on-CPU
of the reference's algorithms; every function cites the reference file:line it follows
(paths relative to ``/root/reference/animation/``).  It is pinned by ``tests/golden/*.safetensors``,
which ``oracle/gen_golden.py`` produced in the build container by importing and running the
reference's own Python modules on seeded inputs (the reference ships no tests or golden vectors for
this path, SURVEY.md §4 / §8c, so those generated vectors are the only pin).
"""

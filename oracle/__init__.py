"""CPU oracle for the voicepuppet hot path (TEST INFRASTRUCTURE ONLY).

This package is a plain-numpy restatement of the reference algorithm
(`/root/reference`, TF1.x graph code) for the PixReferNet G+D step and the
log-mel -> BFMNet audio front-end.  It exists so that the HIP path in
`voicepuppet_amd/` can be checked against an independent CPU computation.

PARITY UNPINNED: the reference ships no tests, golden vectors or fixtures for
this path and TensorFlow 1.x cannot be imported in the build container, so the
oracle is pinned only by (a) identities / finite differences / a torch-CPU
float64 second opinion in `tests/`, and (b) the bit-exact raster goldens from
the compiled reference C++ (`oracle/_ref`).

Only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of
`bench.py` may import this package.  The product path never does.
"""

"""CPU oracle for the manifold_dimension hot path -- TEST INFRASTRUCTURE ONLY.

This package is a CPU restatement (PyTorch fp32 on the host, numpy float64 for
the integer-ID rule, plain C for the two native ops) of what GBATZOLIS/ID-diff
computes on the path

    main.py --mode manifold_dimension -> dim_reduction.get_manifold_dimension
    -> score_fn (fcn / ncsnpp / BeatGANsUNet) -> centred score matrix -> SVD
    -> spectrum -> integer intrinsic dimension.

It exists so that the HIP product in ``id-diff_amd/`` can be checked against
it.  Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg
of ``bench.py`` may import it; nothing under ``id-diff_amd/`` does, and the
product fails loudly when its HIP library is missing instead of falling back
to this code.

Pinning: the reference ships no tests or golden files, so the
oracle is pinned against outputs of the reference itself, produced in the
build container by ``tests/golden/make_golden.py`` (which imports
/root/reference read-only) and committed as ``tests/golden/*.npz``;
``tests/test_oracle_golden.py`` holds the comparison.  Every function cites the
reference file:line it follows.
"""

"""CPU oracle for the HiKapok/DAN detector hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``dan_amd/`` may import this package: only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg use it,
and there only as the checker, never as the thing measured or shipped.

What it is: a restatement, in PyTorch-CPU fp32 (floating-point graph ops), numpy fp32
(box / anchor arithmetic) and C++ (``extra_lib.cpp``: the two CPU custom ops), of the
algorithms in the reference's hot path.  Each function cites the reference file:line it
follows (paths relative to the reference checkout).

Pinning status ("parity unpinned" where stated):
  * The reference holds NO asserted golden vectors (its tests only print).  The KATs that
    exist as *inputs* in ``cpp/ExtraLib/test_op.py`` are pinned here with outputs obtained
    by hand-tracing the reference C++ (tests/golden/kats.json, derivations in DESIGN.md).
  * TensorFlow 1.8 (the reference's only third-party arithmetic dependency, named in its
    README.md:54; no lock file) is absent offline, and the reference C++/CUDA includes TF
    headers, so neither can be built or imported here: conv/pool/resize/softmax/top_k/NMS
    numerics are "parity unpinned" against TF itself and pinned instead by hand-computed
    micro-KATs of the documented TF semantics (tests/test_oracle_tf_semantics.py).
"""

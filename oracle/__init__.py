"""CPU oracle for the HoRoPose image->pose hot path.  TEST INFRASTRUCTURE ONLY.

A plain PyTorch fp32 (CPU) restatement of the reference algorithm, written from the reference's
behaviour (each function cites the reference file:line it follows).  It is the checker that the
HIP path is compared against; it is never the thing measured or shipped:

  * only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg import it;
  * the product package never imports ``oracle`` and has no CPU fallback - it raises when the HIP
    extension is missing.

Parity pin: the reference repository has no tests or golden vectors of its own ("parity
unpinned" by the reference).  The pin used here is the reference itself, imported on CPU in the
build container by ``tests/golden/gen_golden.py`` (harness: ``tests/golden/ref_harness.py``);
the fixtures it wrote are committed under ``tests/golden/*.npz`` and ``tests/test_oracle_golden.py``
checks this oracle against every one of them.

Arithmetic note: convolution / batch-norm / linear / softmax arithmetic in the reference is
PyTorch ATen (``torch==1.13.1+cu117`` pinned in the reference's requirements.txt:138); the oracle
runs the same ATen ops of the torch in this image on CPU.  Forward kinematics is in-tree in the
reference (``lib/utils/urdfpytorch``) and is restated in ``oracle/fk.py``.
"""

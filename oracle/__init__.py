"""CPU oracle for the CPD / FilterReg EM hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``probreg_amd/`` may import this
package; only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` use it, and only as the checker / reported baseline.

Contents
--------
``cpd_numpy``      numpy fp64 restatement of probreg's CPD E/M steps
                   (reference ``probreg/cpd.py:71-88,160-192,219-244,284-303``),
                   chunked so that it also runs at sizes where the reference
                   would need an 80 GB M x N temporary.
``filterreg_numpy`` numpy restatement of the FilterReg rigid pt2pt iteration
                   (reference ``probreg/filterreg.py:78-196``) on top of the C
                   permutohedral restatement in ``permutohedral_oracle.c`` and a
                   restatement of ``probreg/cc/kabsch.cc``.
``ref_import``     loads the *unmodified* reference modules from
                   ``/root/reference`` through stub modules (container only; the
                   GPU box has no /root/reference).  Used by
                   ``tests/golden/make_golden.py`` to pin the restatements.

Parity status: PINNED - every restatement here is checked against the
reference's own code executed in the build container (see
``tests/golden/make_golden.py`` and ``tests/test_oracle_golden.py``); the
reference ships no golden vectors of its own.
"""

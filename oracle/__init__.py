"""oracle/ — TEST INFRASTRUCTURE ONLY.  Not part of the product path.

A CPU restatement (plain PyTorch fp32 ops on the host, plus a small C file for
the whitening loss) of the WT-PSE training hot path of tonyckc/WT-PSE-code,
written to be read side by side with the reference (every function cites the
reference file:line it follows).

Who may import this package
---------------------------
Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` — and there only as the *checker* or the *timed CPU baseline*,
never as the thing shipped.  Nothing under ``wt-pse-code_amd/`` imports it; the
product path raises if the HIP library is missing instead of falling back here.

How the oracle is pinned
------------------------
``oracle/make_golden.py`` imports the reference itself (``/root/reference``, in
the build container only — it is not present on the GPU box) with three shims
that do not touch arithmetic, runs it on CPU on seeded inputs
with name-keyed deterministic weights (``oracle/filler.py``) and injected
sampling noise, and writes the small fixtures under ``tests/golden/``.
``tests/test_oracle_golden.py`` checks every function of ``oracle/wtpse_cpu.py``
against those fixtures, so parity of the HIP path against this oracle is parity
against the reference.  ASD/HD95 (un-vendored ``medpy``) are *parity unpinned*.
"""

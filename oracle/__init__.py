"""CPU oracle for the AFCM `--model stylegan3` generator hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``afcm_amd/`` imports this package; only
``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` do,
and there only as the checker / the timed CPU baseline -- never as the product path.

Contents
--------
``aten_ops``      plain-aten (torch CPU, fp32/fp64) restatement of the reference's ``impl='ref'``
                  ops: ``bias_act``, ``upfirdn2d``, ``filtered_lrelu``, ``modulated_conv2d``.
``direct_np``     a second, definition-level numpy restatement (explicit zero-insert / pad /
                  FIR / decimate index arithmetic) used to cross-check ``aten_ops`` on small cases.
``generator``     functional restatement of the conditional StyleGAN3 generator
                  (mapping + U-shaped synthesis network) built only on ``aten_ops``.

Parity pinning: every function is checked against golden vectors captured from the real
reference running in the build container (``tools/gen_golden.py`` -> ``tests/golden/*.npz``,
see ``tests/test_oracle_golden.py``).  The reference itself ships no tests or golden
vectors, so these captured outputs are the pin.
"""

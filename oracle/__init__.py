"""CPU oracle for the CamContextI2V DDIM denoising hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it, and only as the checker / the timed CPU baseline.
The product path (``camc2v_amd`` + the ``lvdm.*`` / ``model.*`` import paths)
never routes through this package and fails loudly without the HIP library.

Contents (all plain fp32 torch / numpy on CPU, each function cites the
reference file:line it restates):

  unet_oracle.py      lvdm 3D-UNet forward incl. camera conditioning
  ddim_oracle.py      DDIM schedule, CFG + rescale, x_prev update, sampling loop
  geometry_oracle.py  relative poses, fundamental matrices, epipolar masks
  gen_golden.py       (build container only) imports /root/reference to pin
                      the three files above; writes tests/golden/*.npz

Parity status: PINNED.  ``gen_golden.py`` ran the reference's own Python
(imported from /root/reference in the build container) on seeded inputs and
the outputs are committed under ``tests/golden``; ``tests/test_oracle_golden.py``
checks this restatement against them.
"""

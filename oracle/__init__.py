"""CPU oracle for the PoET encoder-decoder hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``poet_amd/`` or ``deformable_attention/``
may import this package.  The only legitimate importers are ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` -- and
there only as the checker / the timed CPU baseline, never as the thing shipped.

Contents
--------
poet_ref.py     op-for-op fp32 PyTorch-CPU restatement of the reference's hot path
                (models/deformable_transformer.py, models/position_encoding.py,
                models/pose_estimation_transformer.py, models/matcher.py) plus a
                restatement of the un-vendored ``deformable_attention.MSDeformAttn``
                (fundamentalvision/Deformable-DETR models/ops, version unpinned by
                the reference: README.md:48, docker/README.md:3).
msda_explicit.py closed-form (no grid_sample, no autograd) forward/backward of the
                MSDA core in float64 numpy -- the formulas the HIP kernels implement.
formula.py      closed-form weight fill + seeded synthetic inputs, so fixtures need
                to carry outputs only.
gen_golden.py   runs the REAL reference (imported from /root/reference, in the build
                container only) and writes tests/golden/*.npz.

Pinning status
--------------
* Everything that lives in /root/reference (transformer stacks, encodings, PoET
  glue, heads, 6D->R, matcher, losses) is pinned: ``tests/golden`` holds outputs of
  the imported reference itself on formula weights/inputs, and
  ``tests/test_oracle_golden.py`` checks this restatement against them.
* The MSDA core (bilinear sampling + weighted sum) is NOT in /root/reference and the
  reference holds no test or vector for it: **parity unpinned** against upstream's
  CUDA kernel.  It is anchored instead on (a) the reference's own call sites
  (deformable_transformer.py:24,58-59,177,201,248,283), (b) upstream's documented
  pure-PyTorch formulation (grid_sample, bilinear, zeros, align_corners=False),
  (c) an independent third-party restatement of the same op (HF transformers
  5.15.0 ``multi_scale_deformable_attention``), whose outputs are committed as
  ``tests/golden/msda_core_hf.npz``, and (d) the float64 closed form in
  msda_explicit.py, which agrees with autograd of (b) to 1e-12.
"""

"""CPU oracle for the CausalDiffAE diffusion hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``causaldiffae_amd/`` or
``improved_diffusion/`` may import this package; only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg do, and
there only as the checker / reported CPU baseline, never as the product.

It is a from-scratch functional restatement (plain torch-CPU fp32 ops and
numpy float64 tables) of the reference algorithm in
``/root/reference/improved_diffusion/{unet,nn,gaussian_diffusion,respace}.py``;
each function cites the reference file:line it follows.

Parity pin: the reference has no tests / golden vectors of its own
(SURVEY.md §4), so the oracle is pinned against outputs of the reference
itself, generated in the build container by ``tools/gen_golden.py`` (the only
file that ever imports ``/root/reference``) and committed as ``tests/golden/*.npz``.
``tests/test_oracle_golden.py`` checks the oracle against every fixture.
"""

"""CPU oracle for the StyleGAN2 few-shot hot path.

TEST INFRASTRUCTURE ONLY.  This package is a plain-PyTorch (CPU, any float
dtype, differentiable to any order) restatement of the reference's fallback
algorithm for upfirdn2d / bias_act / conv2d_resample / modulated_conv2d /
minibatch-stddev and of the generator / discriminator / loss built on them.
Only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of
`bench.py` may import it, and only as the checker.  The product package
(`ga-gan_amd/`) never imports anything from here.

Parity status: PINNED.  Every function is checked against outputs of the
reference itself (imported on CPU in the build container by
`tools/gen_golden.py`) that are committed as fixtures under `tests/golden/`.
The reference ships no tests or golden vectors of its own for this path.
"""

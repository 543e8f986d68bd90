"""Minimal stand-in for the un-vendored `diffusers` dependency of the reference.

TEST INFRASTRUCTURE ONLY (see oracle/README.md).  The reference imports six
classes from `diffusers` (unpinned in /root/reference/requirements.txt:11):

  temporal_denoiser.py:16   Timesteps, TimestepEmbedding
  utils/block.py:12-14      FeedForward, Attention, FP32LayerNorm
  (Attention(qk_norm="rms_norm") instantiates RMSNorm)

This package restates the published semantics of exactly those classes so the
reference's own unmodified modules can be imported in the build container by
`oracle/make_golden.py` to generate the fixtures under tests/golden/.  Nothing
in the product path imports it.
"""
__version__ = "0.0.0-shim"

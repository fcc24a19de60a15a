"""Drop-in mirror of the reference's `src` import surface for the MultiModalVAE hot path
(reference src/models/__init__.py:4-17, src/utils/__init__.py:4-6), backed by the MI355X
kernels in ../mmvae.  Put the directory that contains this package on sys.path where the
reference repo root used to be."""

"""ORACLE — test infrastructure, NOT product code.

CPU restatement of the reference's InfiniteDiffusion hot path, pinned to golden
vectors captured from the reference's own modules (tests/golden/make_golden.py, run in the build
container where /root/reference exists).  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this package; the product (terrain_diffusion_amd) never does.
"""

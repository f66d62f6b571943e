"""ORACLE package -- TEST INFRASTRUCTURE, NOT PRODUCT.

CPU restatements of the reference's hot path (fbank -> ECAPA/ResNet/CAM++ forward -> PLDA LLR),
each citing the reference file:line it follows.  Only `tests/`, `__graft_entry__.smoke()` and
`bench.py`'s `cpu_baseline` leg may import from here, and only as the checker / the timed CPU
baseline -- never as part of the shipped path.  `wespeaker_amd/` must not import this package.
"""

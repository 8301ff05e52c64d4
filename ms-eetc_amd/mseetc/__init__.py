"""
mseetc -- host-side mirror of the reference package's solve path, backed by the
MI355X HIP solver in ../csrc (loaded through the C ABI in include/mseetc_hip.h).
"""

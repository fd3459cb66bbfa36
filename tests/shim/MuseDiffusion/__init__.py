"""Test shim: the package layout of INTEGRATION.md section 1 - the reference's four model modules replaced by re-exports of
musediffusion_amd (plus the north_star alias modules).  tests/test_shim_*.py import the hot path through THESE names, the way the
reference's run/sample.py, run/train.py and utils/initialization.py:110-112 do."""

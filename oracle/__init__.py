"""ORACLE - TEST INFRASTRUCTURE ONLY.

CPU restatement of the reference's SMIL fitting inner loop, used as the parity checker by
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg.  Nothing in
``smilify_amd/`` may import this package: the product path is the HIP library only.
"""

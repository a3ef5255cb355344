"""oracle/ -- TEST INFRASTRUCTURE ONLY (CPU checker for the HIP hot path).

* ``oracle.oracle``      ctypes binding of oracle/liboracle.so, this repository's plain-C
                         restatement of the reference algorithm (oracle/c2ray_oracle.c).
* ``oracle.ref_fortran`` ctypes binding of oracle/_ref/libc2ray_ref.so, the reference's own
                         Fortran sources compiled by oracle/Makefile (present only where that
                         build ran; it travels to the GPU box as a built artefact).

Only tests/, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of bench.py may import
this package.  Nothing under pyc2ray_amd/ imports it; the product path fails loudly when the
HIP library is missing instead of falling back to anything here.
"""

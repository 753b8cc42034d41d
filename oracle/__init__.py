"""oracle/ -- TEST INFRASTRUCTURE, not product code.

Two independent checkers for the MI355X hot path:

* ``oracle.ref``     raw ctypes access to ``oracle/_ref/libaudioflux_ref.so``, the
                     reference's own C implementation compiled from /root/reference
                     by ``oracle/Makefile`` (built-in FFT, naive double-accumulating
                     matmul, OpenMP).  This is the parity authority.
* ``oracle.restate`` a numpy restatement of the reference algorithms, each function
                     citing the reference file:line it follows; pinned against
                     ``oracle.ref`` and the golden fixtures by tests/test_oracle.py.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package.  Nothing under audioflux_amd/ does.
"""

"""CPU oracle for the TransCodon forward/decode path.

TEST INFRASTRUCTURE ONLY.  Nothing in the shipped package imports this
directory; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg may use it, and only as the checker / the timed CPU baseline.

Pinning status: the reference ships no tests or golden vectors, so the oracle
is pinned against outputs of the reference's own modules imported in the build
container (tools/gen_golden.py -> tests/golden/*.npz).  The rotary embedding
arithmetic lives in the un-vendored third-party package
rotary-embedding-torch==0.8.6 (reference environment.yml:205), which is not
installable offline; its published algorithm is restated here, so parity is
UNPINNED at the RoPE boundary.
"""

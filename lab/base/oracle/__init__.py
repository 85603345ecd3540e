"""CPU oracle for the MusicGen / EnCodec generation hot path.

THIS PACKAGE IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.

It is a from-scratch restatement (plain PyTorch-CPU fp32 functional ops + numpy / python loops) of
the algorithms the reference (facebookresearch/audiocraft @ /root/reference) runs on this path;
every function cites the reference file:line it follows.  Only

  * tests/                       (as the checker),
  * __graft_entry__.smoke()      (as the checker),
  * bench.py's `cpu_baseline` leg (timed next to the GPU path, never as the thing shipped)

may import it.  `audiocraft_amd/` never imports it and raises if its HIP library is missing.

Parity pinning: the reference's own tests hold no golden vectors for this path, so
the oracle is pinned against outputs of the reference itself, run in the build container through
`oracle/refstubs.py`: `tests/golden/make_golden.py` imports the unmodified reference modules,
runs them on seeded tiny models and commits inputs + weights + outputs as `tests/golden/*.npz`;
`tests/test_oracle_golden.py` checks every oracle function against those files (CPU suite), and
`oracle/validate_against_reference.py` re-checks at larger sizes whenever /root/reference exists.
"""

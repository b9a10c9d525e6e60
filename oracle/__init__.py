"""CPU oracle: TEST INFRASTRUCTURE ONLY.

A restatement of the reference's per-frame tracking path in plain torch-CPU fp32 /
numpy, pinned against golden vectors that were produced by importing the reference in the build
container (tests/golden/make_golden.py).  Only tests/, __graft_entry__.smoke() and bench.py's
`cpu_baseline` leg may import this package; the product (mo_yolo_amd) never does and fails loudly
if its HIP library is missing.
"""

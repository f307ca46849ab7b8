"""oracle/ -- TEST INFRASTRUCTURE ONLY.

CPU restatement of the reference hot path (kindel_oracle.c), the tooling that pins it
against the unmodified reference (/root/reference, when present) and the golden-vector
generator.  Nothing in the product package (kindel_amd/) may import, link or execute
anything in here; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do.
"""

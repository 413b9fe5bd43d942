"""oracle — CPU restatement of the reference hot path. TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
metagym_amd/ never does; the product path fails loudly when its HIP library is missing instead
of falling back to anything in here.
"""

"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's hot path (see halo_oracle.py).
Importable only from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg."""

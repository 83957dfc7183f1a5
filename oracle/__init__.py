"""TEST INFRASTRUCTURE ONLY: CPU restatement of the reference's hot path (the checker).

Importable only from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
"""

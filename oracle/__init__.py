"""ORACLE — CPU restatement of exprgrad's LLVM CPU path (test infrastructure only).

Nothing under exprgrad_amd/ may import this package; only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg do, and only as the checker.
"""

"""Import shim used ONLY by tests/golden/make_golden.py (see its docstring)."""

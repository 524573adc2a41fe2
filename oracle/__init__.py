"""ORACLE — CPU restatement of the reference algorithm; test infrastructure only (see scan_ref.py)."""

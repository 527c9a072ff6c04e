"""CPU oracle (test infrastructure only; see same_oracle.py)."""

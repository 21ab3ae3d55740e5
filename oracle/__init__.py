"""Test infrastructure only: CPU restatements of the reference algorithm (see rtgs_ref.py)."""

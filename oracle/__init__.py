"""Test infrastructure: CPU restatement of the reference algorithm (see microdit_ref.py). Not product code."""

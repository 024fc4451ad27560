"""CPU oracle for the DMEL hot path -- test infrastructure only (see dmel_oracle.c)."""

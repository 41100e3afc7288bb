"""CPU oracle (test infrastructure only). See o2v_oracle.c for scope and pinning status."""

"""CPU oracle -- TEST INFRASTRUCTURE ONLY (see oracle/parc_oracle.c)."""

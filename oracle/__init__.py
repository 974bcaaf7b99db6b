"""CPU oracle (test infrastructure only): see oracle/oracle.py and oracle/rt2d_oracle.c."""

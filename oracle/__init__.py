"""CPU oracle — TEST INFRASTRUCTURE, not product (see rsx_oracle.c header)."""

"""CPU oracle package -- test infrastructure only (see dvdgan_cpu.py header)."""

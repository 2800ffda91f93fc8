"""Runs the sampler sweep of tests/fuzz_all.py on its own (progress line per case)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fuzz_all
sys.exit(1 if fuzz_all.run_mcmc(int(sys.argv[1]) if len(sys.argv) > 1 else 30, int(sys.argv[2]) if len(sys.argv) > 2 else 2) else 0)

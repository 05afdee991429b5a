"""Stand-in package for `gurobi_optimods` (gurobi-optimods==1.1.0, absent here; needs a licence).

TEST INFRASTRUCTURE ONLY -- used by oracle/refrun/gen_golden.py to run the *reference* code in this
container so that golden vectors can be frozen.  Never imported by the product path.
"""

#!/usr/bin/env python3
"""Container-only check: the USE_MI355X binding of INTEGRATION.md compiled into the reference's own
main.cpp / solver_harness.hpp / solver.hpp and linked against libits_hip.so.  The recipe lives with the other reference
builds: oracle/build_integrated.py (all arguments are passed on)."""
import os
import runpy
import sys

sys.argv[0] = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "build_integrated.py")
runpy.run_path(sys.argv[0], run_name="__main__")

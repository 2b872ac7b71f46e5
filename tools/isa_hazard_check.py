#!/usr/bin/env python3
"""DPP hazard check of a gfx950 .s file (see a1-qp-mpc-controller_amd/isa_check.py; build() runs the same check on every build).
usage: isa_hazard_check.py file.s [kernel_substring]      exit status 1 if a hazard is found"""
import importlib.util
import os
import sys

spec = importlib.util.spec_from_file_location("isa_check", os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "a1-qp-mpc-controller_amd", "isa_check.py"))
mod = importlib.util.module_from_spec(spec)
spec.loader.exec_module(mod)
bad = mod.dpp_hazards(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "")
print("\n".join(bad[:50]))
print(f"{'HAZARDS: %d' % len(bad) if bad else 'no DPP / transcendental read hazards'} ({sys.argv[1]})")
sys.exit(1 if bad else 0)

"""DESIGN.md section 6 is generated from the tracked profiles of the round and must stay equal to them; the profiles must not
contradict each other (tools/check_design_numbers.py)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_design_numbers_are_the_tracked_profiles():
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'check_design_numbers.py')], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-1000:]

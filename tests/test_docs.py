"""INTEGRATION.md's "Numerical differences from upstream" table is generated from the committed parity table of the GPU suite
(profiles/r06_parity_table.json); this keeps the two equal."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_integration_parity_table_equals_the_committed_parity_table():
    run = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "parity_section.py"), "--check"], capture_output=True, text=True)
    assert run.returncode == 0, run.stderr
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    assert "## Numerical differences from upstream" in doc and "cooktorrance.py:216" in doc and "brdf_math.hpp:231-250" in doc

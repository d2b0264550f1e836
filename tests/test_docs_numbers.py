"""DESIGN.md / README.md quote measured figures only through the block tools/design_numbers.py generates from profiles/:
the committed block must equal what the script prints from the committed files (CPU tier)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _block(path):
    s = open(path).read()
    a, b = "<!-- numbers:begin -->\n", "<!-- numbers:end -->"
    assert a in s and b in s, path
    return s[s.index(a) + len(a):s.index(b)]


def test_design_and_readme_numbers_are_the_generated_ones():
    want = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "design_numbers.py")], capture_output=True, text=True, check=True).stdout
    assert len(want) > 1500 and "profiles/r04_bench_driver_flags.json" in want
    for name in ("DESIGN.md", "README.md"):
        assert _block(os.path.join(ROOT, name)) == want, f"{name}: run tools/design_numbers.py --write"

#!/usr/bin/env python3
"""Writes tests/golden/ec_<case>.md5 for the cases of tests/ec_cases.py with the REFERENCE decoder built with error concealment
(oracle/_ref/ref_md5_ec, `make -C oracle ref`; dev container only: it is built from /root/reference)."""
import os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, ".."))
import tempfile
from ec_cases import CASES, materialize, tool_args
TOOL = os.path.join(HERE, "..", "..", "oracle", "_ref", "ref_md5_ec")
for name, (fixture, lose, cut) in CASES.items():
    out = os.path.join(HERE, f"ec_{name}.md5")
    with tempfile.TemporaryDirectory() as tmp:
        subprocess.run([TOOL, "--damage"] + tool_args(lose, cut) + [materialize(fixture, HERE, tmp), out], check=True,
                       stderr=subprocess.DEVNULL)
    print(name, len(open(out).read().splitlines()), "lines")

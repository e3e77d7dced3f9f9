#!/usr/bin/env python
"""Prints the markdown table of DESIGN.md section 4 "Switches" from the library's own table (eg_switch_table,
csrc/switches.cpp): python tools/switch_table.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from exprgrad_amd import _lib

rows = _lib.switch_table()
print("| Switch | Class | Purpose |")
print("|---|---|---|")
for name, cls, purpose in rows:
    print(f"| `{name}` | {cls} | {purpose} |")

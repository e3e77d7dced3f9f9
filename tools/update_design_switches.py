#!/usr/bin/env python
"""Rewrites the switch table of DESIGN.md section 4 from the library's own table (tools/switch_table.py)."""
import os
import subprocess
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
table = subprocess.run([sys.executable, os.path.join(root, "tools", "switch_table.py")], capture_output=True, text=True, check=True).stdout
path = os.path.join(root, "DESIGN.md")
s = open(path).read()
a = s.index("| Switch | Class | Purpose |")
b = s.index("## 5. Oracle")
open(path, "w").write(s[:a] + table + "\n" + s[b:])

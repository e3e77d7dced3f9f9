#!/usr/bin/env python
"""Print the launch sequence of the XOR train target (BASELINE configs[2]) (GPU box): tools/show_xor_plan.py [batch]"""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import exprgrad_amd as eg
from exprgrad_amd import examples
from exprgrad_amd import model as egm

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
ctx = eg.newGpuContext(0)
m = egm.compile(*examples.xor_from_scratch(), gpu=ctx)
x = np.random.randint(0, 2, (batch, 2)).astype(np.float32)
y = (x[:, :1] != x[:, 1:]).astype(np.float32)
m.apply("train", {"x": x, "y": y})
print(m.launch_plan("train"))

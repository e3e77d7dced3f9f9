#!/usr/bin/env python
"""Print the launch sequence of the dense-net train target (BASELINE configs[4]) at one batch size
(GPU box): tools/show_plan.py [batch]"""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import exprgrad_amd as eg
from exprgrad_amd import examples
from exprgrad_amd import model as egm

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
ctx = eg.newGpuContext(0)
m = egm.compile(*examples.dense_softmax_net(), gpu=ctx)
x = np.random.rand(batch, 784).astype(np.float32)
y = np.eye(10, dtype=np.float32)[np.random.randint(0, 10, size=batch)]
m.apply("train", {"x": x, "y": y})
print(m.launch_plan("train"))

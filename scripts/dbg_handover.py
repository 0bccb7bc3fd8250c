import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from xivo_amd import synth
from xivo_amd.lib import Context
N, F, B = 64, 8, 4
P, H, inn, dR = synth.s_level(N, F, B, seed=1)
with Context(N, 2 * F, B) as ctx:
    ctx.upload_P(P)
    ctx.set_measurements(H, inn, dR)
    print("ok")

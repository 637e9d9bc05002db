import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, 'tests')
import torch
print("torch", torch.__version__, torch.cuda.is_available(), torch.cuda.get_device_name(0))
import numpy as np
from rgbd_odometry_amd import DvoContext, SynthScene
x = torch.ones(4, device='cuda') * 2
print(x.sum().item())
maps=[l for l in open('/proc/self/maps') if 'amdhip' in l or 'libdvo' in l or 'hsa-runtime' in l]
print(sorted(set(m.split()[-1] for m in maps)))
import __graft_entry__ as g
g.smoke()
# torch stream + events with the library
ctx = DvoContext(4)
s = torch.cuda.current_stream().cuda_stream
ctx.set_stream(s)
print("set_stream ok", s)
ctx.close()

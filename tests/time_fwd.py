import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nerf_fl_amd
from nerf_fl_amd import NeRF, rendering as rnd
from oracle import nerfw_oracle as orc
dev = torch.device("cuda", 0)
R, F = 4096, 128
nerf_fl_amd.set_precision(sys.argv[1] if len(sys.argv) > 1 else "f16x3")
m = NeRF("fine"); m.load_state_dict(orc.make_field_params(orc.FieldSpec("fine"), 12, "sharp")); m = m.to(dev)
f = rnd._field(m, 10, 4, dev)
rays = orc.make_rays(R, 100).to(dev)
z = torch.sort(2 + 4 * torch.rand(R, F, device=dev), dim=1)[0]
fn = lambda: rnd._run_pass(f, rays, F, z=z, noise=None, noise_std=0.0, white_back=True)
for _ in range(3): fn()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); e0.record()
for _ in range(20): fn()
e1.record(); torch.cuda.synchronize()
print("fwd ms %.4f" % (e0.elapsed_time(e1) / 20))

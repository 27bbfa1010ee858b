"""Run the fine-pass forward kernel a few times (for rocprofv3 --pmc runs)."""
import os, sys, torch
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
noise = torch.randn(R, F, device=dev)
for _ in range(6):
    rnd._run_pass(f, rays, F, z=z, noise=noise, noise_std=1.0, white_back=True)
torch.cuda.synchronize()

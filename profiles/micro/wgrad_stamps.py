"""In-kernel timeline of the weight-gradient kernel on a 512 -> 512 layer of the PointMaze net (diagnostic build:
hipcc ... -DDAD_WG_STAMPS -o profiles/micro/_lib_stamps.so; DAD_LIB=that python3 profiles/micro/wgrad_stamps.py).
Block (0,0,0), thread 0, s_memrealtime (100 MHz): entry, first chunk staged, every chunk's end, after the K-group
reduction, after the stores."""
import os, sys, ctypes as C
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from dynamics_aware_diffusion_amd import GaussianDiffusion, TemporalUnet, _engine
from dynamics_aware_diffusion_amd.utils import synth
dev = torch.device("cuda:0")
od, ad, dim, mults, T = synth.ARCHS["pointmaze"]; td = od + ad
unet = TemporalUnet(td, dim=dim, dim_mults=mults)
unet.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_unet_state(td, dim, mults, seed=0).items()})
diff = GaussianDiffusion(unet, 32, od, ad, n_timesteps=T).to(dev)
x = torch.randn(256, 32, td, device=dev).clamp(-1, 1)
for _ in range(3):
    for p in diff.parameters(): p.grad = None
    diff.loss(x).backward()
torch.cuda.synchronize()
lib = _engine.load_library()
buf = (C.c_ulonglong * 32)()
lib.dad_debug_wgrad_stamps(buf)
s = list(buf)
print("staged first chunk  +%.2f us" % ((s[1] - s[0]) * 0.01))
prev = s[1]
for k in range(2, 22):
    if s[k] <= prev: break
    print("chunk %2d end        +%.2f us   (t = %.2f)" % (k - 2, (s[k] - prev) * 0.01, (s[k] - s[0]) * 0.01)); prev = s[k]
print("K-groups reduced    +%.2f us   (t = %.2f)" % ((s[28] - prev) * 0.01, (s[28] - s[0]) * 0.01))
print("stored              +%.2f us   (t = %.2f)" % ((s[29] - s[28]) * 0.01, (s[29] - s[0]) * 0.01))

import os, sys, ctypes as C
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from dynamics_aware_diffusion_amd import GaussianDiffusion, GuidedPolicy, TemporalUnet, _engine
from dynamics_aware_diffusion_amd.utils import synth
dev = torch.device("cuda:0")
od, ad, dim, mults, T = synth.ARCHS["pointmaze"]; td = od + ad
unet = TemporalUnet(td, dim=dim, dim_mults=mults)
unet.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_unet_state(td, dim, mults, seed=0).items()})
diff = GaussianDiffusion(unet, 32, od, ad, n_timesteps=T).to(dev)
x = torch.randn(256, 32, td, device=dev)
with torch.no_grad():
    for _ in range(5): diff.model(x, 5)
lib = _engine.load_library()
buf = (C.c_ulonglong * 32)()
lib.dad_debug_chain_stamps(buf)
s = list(buf)
names = ["entry", "staged", "sync"] + [f"conv{i}.{w}" for i in range(5) for w in ("kloop", "epilogue")]
for k in range(1, 13):
    print(f"{names[k]:16s} +{(s[k]-s[k-1])*0.01:6.2f} us   (t = {(s[k]-s[0])*0.01:6.2f})")

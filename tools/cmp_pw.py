"""Measurement aid: denoiser output at a given shape / precision written to a file (compare runs with VMM_C3_PERSISTENT=0 / 1 / 2)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import videometamaterials_amd as vm
T, H, prec, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
B = int(sys.argv[5]) if len(sys.argv) > 5 else 1
kw = dict(dim=64, dim_mults=(1, 2, 4, 8), channels=3, cond_attention="self-stacked", cond_attention_tokens=16, use_temporal_attention_cond=True, per_frame_cond=False)
torch.manual_seed(0)
m = vm.Unet3D(**kw).cuda().eval()
m.precision = prec
g = torch.Generator().manual_seed(1)
x = torch.randn(B, 3, T, H, H, generator=g).cuda()
t = torch.randint(0, 256, (B,), generator=g).cuda()
cond = (torch.rand(B, 51, generator=g) * 2 - 1).cuda()
with torch.no_grad():
    y = m.forward_with_guidance_scale(x, t, cond=cond, guidance_scale=5.0)
    taps = {}
torch.save(y.cpu(), out)
print(out, float(y.abs().mean()))

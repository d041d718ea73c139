import argparse, sys, torch
sys.path.insert(0, '.')
from oracle import synth, unet
from brats21_amd import get_model
torch.manual_seed(0)
width = int(sys.argv[1]) if len(sys.argv) > 1 else 48
size = (16, 16, 16)
mode = sys.argv[2] if len(sys.argv) > 2 else "closed"
sd = synth.fill_state_dict(unet.equiunet_state_shapes(width))
if mode == "rand":
    sd = {k: (v + 0.02 * torch.randn_like(v)) for k, v in sd.items()}
    x = synth.random_image(2, 4, size)
else:
    x = synth.closed_form_image(2, 4, size)
t = synth.nested_spheres(2, size)
m = get_model(argparse.Namespace(model="equiunet", width=width, norm="group", act="relu", num_classes=3, dropout=0))
m.load_state_dict(sd); m.precision = "fp32"; m = m.cuda().train()
sd_ref = {k: v.clone().double().requires_grad_(True) for k, v in sd.items()}
out_ref = unet.equiunet_forward(sd_ref, x.double())
loss_ref = unet.deep_supervision_loss(out_ref, t.double()); loss_ref.backward()
sd32 = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
out32 = unet.equiunet_forward(sd32, x)
loss32 = unet.deep_supervision_loss(out32, t); loss32.backward()
out, deeps = m(x.cuda())
loss = unet.deep_supervision_loss((out, deeps), t.cuda()); loss.backward()
print("logit err hip vs f64:", float((out.detach().cpu().double() - out_ref[0].detach()).abs().max()),
      " cpu32 vs f64:", float((out32[0].detach().double() - out_ref[0].detach()).abs().max()))
for k, p in m.named_parameters():
    ref = sd_ref[k].grad
    e_hip = float((p.grad.cpu().double() - ref).norm() / (ref.norm() + 1e-30))
    e_cpu = float((sd32[k].grad.double() - ref).norm() / (ref.norm() + 1e-30))
    print(f"{k:40s} hip {e_hip:.2e}  cpu32 {e_cpu:.2e}  |ref| {float(ref.norm()):.3e}")

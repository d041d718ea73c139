import argparse, sys, torch
sys.path.insert(0, '.')
from oracle import synth, unet
from brats21_amd import get_model
size = (int(sys.argv[1]),) * 3 if len(sys.argv) > 1 else (16, 16, 16)
g = torch.Generator().manual_seed(7)
sd = {k: v + 0.02 * torch.randn(v.shape, generator=g) for k, v in synth.fill_state_dict(unet.equiunet_state_shapes(48)).items()}
x = synth.random_image(2, 4, size); t = synth.nested_spheres(2, size)
m = get_model(argparse.Namespace(model="equiunet", width=48, norm="group", act="relu", num_classes=3, dropout=0))
m.load_state_dict(sd); m = m.cuda().train()
sd_ref = {k: v.clone().double().requires_grad_(True) for k, v in sd.items()}
loss_ref = unet.deep_supervision_loss(unet.equiunet_forward(sd_ref, x.double()), t.double()); loss_ref.backward()
# torch's own bf16 autocast on CPU as a yardstick for "what bf16 costs"
sd_b = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
with torch.autocast("cpu", dtype=torch.bfloat16):
    out_b = unet.equiunet_forward(sd_b, x)
lb = unet.deep_supervision_loss(out_b, t); lb.backward()
with torch.autocast("cuda", dtype=torch.bfloat16):
    out = m(x.cuda()); loss = unet.deep_supervision_loss(out, t.cuda())
loss.backward()
rows = []
for k, p in m.named_parameters():
    ref = sd_ref[k].grad
    rows.append((float((p.grad.cpu().double() - ref).norm() / ref.norm()), float((sd_b[k].grad.double() - ref).norm() / ref.norm()), k))
rows.sort()
print("loss hip", loss.item(), "cpu-bf16", lb.item(), "f64", loss_ref.item())
for r in rows[-12:]: print(f"{r[2]:40s} hip_bf16 {r[0]:.3f}   torch_cpu_bf16_autocast {r[1]:.3f}")
import statistics
print("median hip", statistics.median(r[0] for r in rows), "median torch-cpu-bf16", statistics.median(r[1] for r in rows))

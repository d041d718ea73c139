"""Width-64 sanity (BASELINE configs[4] uses width 64): f32 logits of both models against the oracle on random weights."""
import sys, argparse, contextlib, io, torch
sys.path.insert(0, '.')
from brats21_amd import get_model
from oracle import synth, unet
dev = torch.device("cuda:0")
for model, fwd in (("equiunet", unet.equiunet_forward), ("equiunet_assp_evo", unet.assp_evo_forward)):
    torch.manual_seed(0)
    with contextlib.redirect_stdout(io.StringIO()):
        m = get_model(argparse.Namespace(model=model, width=64, norm="group", act="relu", num_classes=3, dropout=0)).to(dev).train()
    m.precision = "fp32"
    sd = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    x = synth.random_image(1, 4, (32, 32, 32), seed=2)
    out = m(x.to(dev))
    with torch.no_grad():
        ref = fwd(sd, x)
    print(model, "w64 f32 logit max err", float((out[0].detach().cpu() - ref[0]).abs().max()))
    (out[0].mean() + sum(d.mean() for d in out[1])).backward()
    print("  grads finite", all(bool(torch.isfinite(p.grad).all()) for p in m.parameters() if p.grad is not None))

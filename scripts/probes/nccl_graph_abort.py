"""Reproduction attempt for the abort seen when a hipGraph with captured RCCL work is replayed late in a long-lived process
(DESIGN.md section 5, round-6 caveat): first some unrelated graph captures and big allocations (a sliding-window inference with
hipGraph patch steps), then the body of tests/_nccl_world1_worker.py::graphed in the SAME process.
  python scripts/probes/nccl_graph_abort.py [n_inference_rounds]"""
import argparse, contextlib, io, os, sys, warnings
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from brats21_amd import get_model, synth, tta
from brats21_amd.evaluate import Evaluator
dev = torch.device("cuda:0")
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 2
with contextlib.redirect_stdout(io.StringIO()), warnings.catch_warnings():
    warnings.simplefilter("ignore")
    m = get_model(argparse.Namespace(model="equiunet", width=48, norm="group", act="relu", num_classes=3, dropout=0)).to(dev).eval()
m.skip_deep_heads_in_eval = True
vol = synth.random_image(1, 4, (240, 240, 155), seed=99, device=dev)
for r in range(rounds):
    ev = Evaluator(m, tta_transforms=tta.flip8()[: 2 + r], sliding_window_size=(128,) * 3, sw_batch_size=2 + r, overlap=0.5, k_divisible=8, amp=True, use_graph=True)
    with torch.no_grad():
        ev(vol); ev(vol)
    del ev
torch.cuda.synchronize()
print("inference graphs done", flush=True)
import _nccl_world1_worker as w
w.graphed()
torch.cuda.synchronize()
print("OK graphed after inference graphs", flush=True)

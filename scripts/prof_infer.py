"""The inference leg of bench.py alone (for rocprofv3 --kernel-trace --stats): python scripts/prof_infer.py"""
import argparse, contextlib, io, sys
import torch
sys.path.insert(0, '.')
import bench
from brats21_amd import get_model

dev = torch.device("cuda:0")
args = argparse.Namespace(width=48, model="equiunet", sw_batch=4, precision="bf16", fp8=None, infer_headline_only="--all" not in sys.argv)
ns = argparse.Namespace(model="equiunet", width=48, norm="group", act="relu", num_classes=3, dropout=0)
torch.manual_seed(0)
with contextlib.redirect_stdout(io.StringIO()):
    model = get_model(ns).to(dev)
print(bench.inference_bench(model, dev, args))

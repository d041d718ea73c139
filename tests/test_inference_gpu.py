"""-m gpu: on-GPU sliding-window stitching and TTA against the golden vectors produced by the reference's
utils/inferers.py / tta package, and end-to-end against the CPU oracle with the real network."""
import argparse
import os

import numpy as np
import pytest
import torch

from oracle import inference as oinf
from oracle import synth, unet

pytestmark = pytest.mark.gpu


def _analytic_predictor(dev):
    w = synth.closed_form("swpred", (3, 4), 0.5).to(dev)

    def predictor(p):
        zz = torch.arange(p.shape[2], dtype=torch.float32, device=p.device).view(1, 1, -1, 1, 1) * 0.01
        out = torch.einsum("oc,ncdhw->nodhw", w, p) + zz
        return out, [out * 2]
    return predictor


def test_sliding_window_matches_reference_golden(golden_dir):
    from brats21_amd.inferers import sliding_window_inference
    g = np.load(os.path.join(golden_dir, "inference.npz"))
    dev = torch.device("cuda:0")
    x = synth.closed_form_image(1, 4, (20, 27, 17), "swx").to(dev)
    pred = _analytic_predictor(dev)
    for mode in ("constant", "gaussian"):
        for ov in (0.25, 0.5):
            y = sliding_window_inference(x, (16, 16, 16), 1, pred, overlap=ov, mode=mode)
            np.testing.assert_allclose(y.cpu().numpy(), g[f"sw_{mode}_{int(ov * 100)}"], atol=2e-6)
    y = sliding_window_inference(x[..., :12].contiguous(), (16, 16, 16), 2, pred, overlap=0.5)  # roi > image
    np.testing.assert_allclose(y.cpu().numpy(), g["sw_pad"], atol=2e-6)
    for pm in ("reflect", "replicate", "circular"):  # PytorchPadMode (utils/inferers.py:34,109), roi > image in two dims
        y = sliding_window_inference(x[:, :, :, :11, :12].contiguous(), (16, 16, 16), 2, pred, overlap=0.5, padding_mode=pm)
        np.testing.assert_allclose(y.cpu().numpy(), g[f"sw_pad_{pm}"], atol=2e-6)
    with pytest.raises(ValueError):
        sliding_window_inference(x, (16, 16, 16), 1, pred, padding_mode="mirror")
    y = sliding_window_inference(x, (16, 16, 16), 4, pred, overlap=0.5, device=torch.device("cpu"))
    assert y.device.type == "cpu"
    np.testing.assert_allclose(y.numpy(), g["sw_constant_50"], atol=2e-6)
    with pytest.raises(AssertionError):
        sliding_window_inference(x, (16, 16, 16), 1, pred, overlap=1.0)


def test_tta_kernels_match_reference_golden(golden_dir):
    from brats21_amd import tta
    g = np.load(os.path.join(golden_dir, "inference.npz"))
    dev = torch.device("cuda:0")
    comp = tta.Compose([tta.OnAxes(axes=["zxy", "xyz"]), tta.HorizontalFlip(), tta.Rotate90(angles=[0, 90, 180, 270])])
    v = synth.closed_form("ttav", (1, 2, 4, 6, 6)).to(dev)
    acc = torch.zeros_like(v)
    for i, tr in enumerate(comp):
        a = tr.augment_image(v)
        np.testing.assert_array_equal(a.cpu().numpy().ravel(), g["tta_aug"][i])
        np.testing.assert_array_equal(tr.deaugment_mask(a).cpu().numpy(), g["tta_roundtrip"][i])
        tr.accumulate_probability(a, acc)
    torch.testing.assert_close(acc / 16, torch.sigmoid(v), atol=1e-6, rtol=1e-6)


def test_network_sliding_window_tta_graph_vs_oracle():
    """EquiUnet w8 through hipGraph-captured patch steps, on-GPU stitching and 16-pass TTA vs the CPU oracle."""
    from brats21_amd import get_model, tta
    from brats21_amd.inferers import GraphedPredictor, sliding_window_inference, tta_predict
    dev = torch.device("cuda:0")
    sd = synth.fill_state_dict(unet.equiunet_state_shapes(8))
    m = get_model(argparse.Namespace(model="equiunet", width=8, norm="group", act="relu", num_classes=3, dropout=0))
    m.load_state_dict(sd)
    m.precision = "fp32"
    m = m.to(dev).eval()
    m.skip_deep_heads_in_eval = True
    x = synth.closed_form_image(1, 4, (24, 24, 24), "swnet")
    ref_pred = lambda p: unet.equiunet_forward(sd, p)  # noqa: E731
    with torch.no_grad():
        y_ref = oinf.sliding_window_inference(x, (16, 16, 16), 1, ref_pred, overlap=0.5)
        graphed = GraphedPredictor(m)
        y = sliding_window_inference(x.to(dev), (16, 16, 16), 1, graphed, overlap=0.5)
        assert len(graphed.graphs) == 1  # one capture, 8 replays
        assert float((y.cpu() - y_ref).abs().max()) < 1e-3
        # TTA on a cubic patch: 16 passes, probabilities averaged on the GPU
        comp = tta.Compose([tta.OnAxes(axes=["zxy", "xyz"]), tta.HorizontalFlip(), tta.Rotate90(angles=[0, 90, 180, 270])])
        xp = x[..., :16, :16, :16].contiguous()
        p = tta_predict(xp.to(dev), graphed, comp)
        p_ref = oinf.tta_predict(xp, ref_pred)
        assert float((p.cpu() - p_ref).abs().max()) < 5e-4


def test_packed_weight_cache_tracks_weight_updates():
    """Inference caches packed weights under no_grad; an optimizer step / in-place update must invalidate them."""
    from brats21_amd import get_model, ops
    from brats21_amd.optim import Ranger2020
    dev = torch.device("cuda:0")
    m = get_model(argparse.Namespace(model="equiunet", width=8, norm="group", act="relu", num_classes=3, dropout=0)).to(dev)
    m.precision = "fp32"
    x = synth.closed_form_image(1, 4, (16, 16, 16), "cachex").to(dev)
    opt = Ranger2020(m.parameters(), lr=1e-2, use_gc=False)
    m.eval()
    with torch.no_grad():
        y0 = m(x)[0].clone() if isinstance(m(x), (tuple, list)) else m(x).clone()
        n_cached = len(ops._PACK_CACHE)
        y0b = m(x)[0] if isinstance(m(x), (tuple, list)) else m(x)
        assert n_cached > 0 and len(ops._PACK_CACHE) == n_cached and torch.equal(y0, y0b)
    m.train()
    out = m(x)
    loss = out[0].float().mean() + sum(o.float().mean() for o in out[1])
    loss.backward()
    opt.step()
    torch.cuda.synchronize()
    m.eval()
    with torch.no_grad():
        r = m(x)
        y1 = r[0] if isinstance(r, (tuple, list)) else r
        assert not torch.equal(y0, y1), "stale packed weights after an optimizer step"
        with torch.no_grad():
            p = next(m.parameters())
            p.mul_(1.5)
        r = m(x)
        y2 = r[0] if isinstance(r, (tuple, list)) else r
        assert not torch.equal(y1, y2), "stale packed weights after an in-place update"


def test_sliding_window_batch_size_does_not_change_results():
    """Windows are independent samples (per-sample norms): batching 3 of them per launch must give the same stitched
    logits as one at a time (the bench's inference leg uses 3 windows per launch)."""
    from brats21_amd import get_model
    from brats21_amd.inferers import sliding_window_inference
    dev = torch.device("cuda:0")
    sd = synth.fill_state_dict(unet.equiunet_state_shapes(8))
    m = get_model(argparse.Namespace(model="equiunet", width=8, norm="group", act="relu", num_classes=3, dropout=0))
    m.load_state_dict(sd)
    m = m.to(dev).eval()
    m.skip_deep_heads_in_eval = True
    x = synth.closed_form_image(1, 4, (24, 32, 24), "swbatch").to(dev)
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        y1 = sliding_window_inference(x, (16, 16, 16), 1, m, overlap=0.5)
        y3 = sliding_window_inference(x, (16, 16, 16), 3, m, overlap=0.5)
    assert torch.equal(y1, y3)


def test_batched_window_accumulate_is_bit_identical_to_one_launch_per_window():
    """brats_sw_accumulate_multi (all windows of a predictor batch in one output-centric launch) against the per-window
    launches it replaces, on overlapping windows over two samples: same additions in the same order -> torch.equal."""
    from brats21_amd import _lib
    lib = _lib.lib()
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(12)
    nb, k, img, roi = 2, 3, (20, 24, 28), (12, 16, 16)
    wins = [(0, 0, 0, 0), (0, 4, 8, 12), (0, 8, 8, 6), (1, 8, 0, 12), (0, 2, 3, 5), (1, 0, 8, 0), (1, 3, 3, 3)]
    prob = torch.randn((len(wins), k) + roi, generator=g).to(dev)
    imp = (torch.rand(roi, generator=g) + 0.1).to(dev)
    base_o = torch.randn((nb, k) + img, generator=g).to(dev)
    base_c = torch.rand((nb, k) + img, generator=g).to(dev)
    st = torch.cuda.current_stream().cuda_stream
    o1, c1 = base_o.clone(), base_c.clone()
    for j, (n, z0, y0, x0) in enumerate(wins):
        _lib.check(lib.brats_sw_accumulate(prob[j].data_ptr(), imp.data_ptr(), o1.data_ptr(), c1.data_ptr(), k, *img, *roi, n, z0, y0, x0, st),
                   "sw_accumulate")
    o2, c2 = base_o.clone(), base_c.clone()
    tbl = torch.tensor(wins, dtype=torch.int32, device=dev)
    _lib.check(lib.brats_sw_accumulate_multi(prob.data_ptr(), imp.data_ptr(), o2.data_ptr(), c2.data_ptr(), tbl.data_ptr(), len(wins), nb, k,
                                             *img, *roi, st), "sw_accumulate_multi")
    torch.cuda.synchronize()
    assert torch.equal(o1, o2) and torch.equal(c1, c2)
    assert not torch.equal(o1, base_o)

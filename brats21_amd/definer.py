"""The reference's ``--model`` factory for the accelerated models (src/definer.py:37-174):
``get_model(args) -> torch.nn.Module`` with the same Namespace fields (model, width, norm, act,
num_classes, dropout) and the same error behaviour (NameError for an unknown model)."""
import argparse

import torch


def get_model(args: argparse.Namespace) -> torch.nn.Module:
    from .networks import EquiUnet

    kwargs = {
        "inplanes": 4,
        "num_classes": args.num_classes,
        "features": [args.width * 2 ** i for i in range(4)],
        "norm_layer": args.norm,
        "act": args.act,
        "deep_supervision": True,  # hard-wired in the reference factory, src/definer.py:140
        "dropout": args.dropout,
    }
    if args.model == "equiunet":
        return EquiUnet(**kwargs)
    if args.model in ("equiunet_assp_evo", "equiunet_assp_evocor"):
        from .networks.equiunet_assp import EquiUnetASSPEvo

        return EquiUnetASSPEvo(**kwargs)
    raise NameError("Not Supported Model")

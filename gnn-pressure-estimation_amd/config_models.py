"""Model registry for the GATRes family -- mirrors gnn_pressure_estimation/ConfigModels.py:22-42,133-178 for the
two models the north-star path names.  Only the fields ``select_model`` overwrites for these models are touched."""
from __future__ import annotations

import argparse
from typing import Optional, Tuple

import torch

from .graph_models import GATResMeanConv


def config_gatres_large(args: argparse.Namespace, test_model_variant_name: Optional[str] = None
                        ) -> Tuple[argparse.Namespace, torch.nn.Module]:
    """ConfigModels.py:22-32: 25 blocks, 128 channels."""
    args.criterion = "mse"
    args.use_data_edge_attrs = None
    args.norm_type = "znorm"
    name = "GATRes_Large_znorm_25b_128c" if test_model_variant_name is None else test_model_variant_name
    return args, GATResMeanConv(name=name, num_blocks=25, nc=128)


def config_gatres_small(args: argparse.Namespace, test_model_variant_name: Optional[str] = None
                        ) -> Tuple[argparse.Namespace, torch.nn.Module]:
    """ConfigModels.py:35-42: 15 blocks, 32 channels."""
    args.criterion = "mse"
    args.use_data_edge_attrs = None
    args.norm_type = "znorm"
    name = "GATResMeanConv_small_znorm_15b_32c" if test_model_variant_name is None else test_model_variant_name
    return args, GATResMeanConv(name=name, num_blocks=15, nc=32)


def select_model(args: argparse.Namespace, test_model_variant_name: Optional[str] = None,
                 reset_model_path: bool = False) -> Tuple[argparse.Namespace, torch.nn.Module]:
    """ConfigModels.py:133-178 restricted to the GATRes models (the baselines are out of scope)."""
    model = getattr(args, "model", "gatres_small")
    if model == "gatres_small":
        return config_gatres_small(args, test_model_variant_name)
    if model == "gatres_large":
        return config_gatres_large(args, test_model_variant_name)
    raise NotImplementedError(f"Unknown model! Got {model}! (this engine provides gatres_small / gatres_large)")

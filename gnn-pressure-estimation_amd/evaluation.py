"""Evaluation path for GATRes: forward-only passes under fresh masks, the reference's seven metrics on de-normalised
masked predictions, and event-timed latency / throughput.

Mirrors gnn_pressure_estimation/evaluation.py:240-351 (``test_one_epoch``: the all-nodes pass and the sensor pass, whose
masks always contain ``required_idx``) and :355-403 (the trial loop that runs both passes per trial) for the configuration
the GATRes models run with (``use_data_batch=False``, no edge attributes), utils/auxil.py:101-140,185-203 (metric functions
and their registry) and utils/timer.py:12-66 (``Timer``: warm-up, then one event pair per inference call).  The sensor
indices themselves come from a secrets file the reference does not ship (evaluation.py:27-66: without it ``get_sensors``
returns empty lists and the second pass equals the first); here they are an argument.
The model call is the fused forward kernel (no activations kept under ``torch.no_grad()``); everything else here is
caller-side tensor arithmetic, exactly as in the reference.
"""
from __future__ import annotations

from functools import partial
from typing import Callable, Dict, Iterable, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

from .wdn_synth import generate_batch_mask


# ---- metrics (auxil.py:101-140) -----------------------------------------------------------------------------------
def calculate_nse(y_pred, y_true, exponent=2):
    p, t = y_pred.reshape(-1), y_true.reshape(-1)
    return 1.0 - (p - t).pow(exponent).sum() / ((t - t.mean()).pow(exponent).sum() + 1e-12)


def calculate_rmse(y_pred, y_true):
    return ((y_pred - y_true) ** 2).mean().sqrt()


def calculate_rel_error(y_pred, y_true):
    keep = y_true.abs() > 0.01
    return ((y_true - y_pred).abs()[keep] / y_true[keep]).abs().mean()


def calculate_accuracy(y_pred, y_true, threshold=0.2):
    return ((y_true - y_pred).abs() <= y_true * threshold).float().mean()


def calculate_correlation_coefficient(y_pred, y_true):
    vx, vy = y_pred - y_pred.mean(), y_true - y_true.mean()
    return ((vx * vy).sum() / (vx.pow(2).sum().sqrt() * vy.pow(2).sum().sqrt())).clamp(-1.0, 1.0)


def calculate_r2(y_pred, y_true):
    return calculate_correlation_coefficient(y_pred, y_true) ** 2


def get_metric_fn_collection(prefix: str) -> Dict[str, Callable]:
    """auxil.py:185-203: same keys, same order."""
    return {f"{prefix}_error": calculate_rel_error, f"{prefix}_0.1": partial(calculate_accuracy, threshold=0.1),
            f"{prefix}_corr": calculate_correlation_coefficient, f"{prefix}_r2": calculate_r2,
            f"{prefix}_mae": F.l1_loss, f"{prefix}_rmse": calculate_rmse,
            f"{prefix}_mynse": partial(calculate_nse, exponent=2)}


def descale(scaled_data, norm_type="minmax", mean=None, std=None, min=None, max=None):
    """auxil.py:42-64."""
    if norm_type == "minmax":
        return scaled_data * (max - min) + min
    if norm_type == "znorm":
        return scaled_data * std + mean
    return scaled_data


# ---- timing (timer.py:12-66) --------------------------------------------------------------------------------------
class Timer:
    """One HIP event pair per inference call after ``gpu_warmup_times`` untimed calls (timer.py:22-41)."""

    def __init__(self) -> None:
        self.reset()

    def reset(self) -> None:
        self.timings, self.num_graphs, self.finished_warmup = [], [], False

    def auto_measure(self, inference_func: Callable, num_graphs_per_batch: int, gpu_warmup_times: int = 10) -> Callable:
        def inference(*args, **kwargs):
            if gpu_warmup_times > 0 and not self.finished_warmup:
                for _ in range(gpu_warmup_times):
                    inference_func(*args, **kwargs)
                self.finished_warmup = True
            start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            start.record()
            out = inference_func(*args, **kwargs)
            end.record()
            torch.cuda.synchronize()
            self.timings.append(start.elapsed_time(end))
            self.num_graphs.append(num_graphs_per_batch)
            return out
        return inference

    def compute_time(self, len_dataset: int) -> float:
        """mean milliseconds per graph, timer.py:43-51 (batch time weighted by its graph count)."""
        return float(np.dot(np.array(self.timings), np.array(self.num_graphs)) / len_dataset)

    def compute_throughput(self, len_dataset: int) -> float:
        """graphs per second as the reference defines it, timer.py:53-66."""
        total_s = float(np.sum(np.array(self.timings) * np.array(self.num_graphs) / len_dataset / 1000.0))
        return float(len(self.timings) * max(self.num_graphs)) / total_s


# ---- the loop (evaluation.py:240-351) -----------------------------------------------------------------------------
def test_one_epoch(model: torch.nn.Module, loader: Iterable[Tuple[torch.Tensor, torch.Tensor, int]], mask_rate: float,
                   mean=None, std=None, min_val=None, max_val=None, norm_type: str = "znorm",
                   criterion: Optional[Callable] = None, metric_fn_dict: Optional[Dict[str, Callable]] = None,
                   nodes_per_graph: Optional[int] = None, gpu_warmup_times: int = 10, use_same_mask: bool = False,
                   rng: Optional[np.random.RandomState] = None, do_test_on_sensors: bool = False,
                   required_idx: Sequence[int] = ()) -> Tuple[float, Dict[str, float]]:
    """``loader`` yields ``(x, edge_index, num_graphs)`` with x == y (e.g. ``SnapshotStore.batches``).  Returns
    ``(loss, metrics)`` with the reference's keys plus ``{prefix}_time`` (ms/graph) and ``{prefix}_throughput``.
    ``do_test_on_sensors`` (evaluation.py:288-294): every mask contains ``required_idx`` (node indices inside a graph) and
    the metric keys carry the ``_sensor`` postfix; with an empty ``required_idx`` it is the plain pass under other keys."""
    model.eval()
    criterion = criterion or torch.nn.MSELoss()
    metric_fn_dict = metric_fn_dict or get_metric_fn_collection("test")
    rng = rng or np.random
    required = list(required_idx) if do_test_on_sensors else []
    postfix = "_sensor" if do_test_on_sensors else ""
    total_loss, total = 0.0, {k: 0.0 for k in metric_fn_dict}
    timer, all_mask, n_graphs = Timer(), None, 0
    with torch.no_grad():
        for x, edge_index, num_graphs in loader:
            y = x
            npg = nodes_per_graph or x.shape[0] // num_graphs
            if all_mask is None or not use_same_mask or all_mask.shape[0] != x.shape[0]:
                all_mask = generate_batch_mask([npg] * num_graphs, mask_rate, rng, required)
            x1 = x.clone()
            x1[all_mask] = 0
            out = timer.auto_measure(model, num_graphs, gpu_warmup_times)(x1, edge_index, None, None)
            y_pred, y_true = out[all_mask], y[all_mask]
            p_r = descale(y_pred, norm_type, mean, std, min_val, max_val)
            t_r = descale(y_true, norm_type, mean, std, min_val, max_val)
            total_loss += float(criterion(y_pred, y_true)) * num_graphs
            for k, fn in metric_fn_dict.items():
                total[k] += float(fn(p_r, t_r)) * num_graphs
            n_graphs += num_graphs
    metrics = {k: v / n_graphs for k, v in total.items()}
    prefix = list(metric_fn_dict.keys())[0].split("_")[0]
    metrics[prefix + "_time"] = timer.compute_time(n_graphs)
    metrics[prefix + "_throughput"] = timer.compute_throughput(n_graphs)
    return total_loss / n_graphs, {k + postfix: v for k, v in metrics.items()}


def test_trials(model: torch.nn.Module, make_loader: Callable[[], Iterable], num_test_trials: int, mask_rate: float,
                required_idx: Sequence[int] = (), **kw):
    """The trial loop of evaluation.py:355-403: ``num_test_trials`` times an all-nodes pass and a sensor pass over a fresh
    loader, results collected per trial.  Returns ``(test_losses, test_metrics, sensor_losses, sensor_metrics)`` with the
    metric dicts mapping a key to the list of its per-trial values (``defaultdict(list)`` in the reference)."""
    from collections import defaultdict
    losses, metrics, s_losses, s_metrics = [], defaultdict(list), [], defaultdict(list)
    for _ in range(int(num_test_trials)):
        loss, m = test_one_epoch(model, make_loader(), mask_rate, do_test_on_sensors=False, **kw)
        s_loss, s_m = test_one_epoch(model, make_loader(), mask_rate, do_test_on_sensors=True, required_idx=required_idx, **kw)
        losses.append(loss); s_losses.append(s_loss)
        for k, v in m.items():
            metrics[k].append(v)
        for k, v in s_m.items():
            s_metrics[k].append(v)
    return losses, dict(metrics), s_losses, dict(s_metrics)

"""MI355X-native GATRes message-passing engine (drop-in for the GATResMeanConv hot path of
DiTEC-project/gnn-pressure-estimation).  See DESIGN.md / INTEGRATION.md at the repository root."""
from . import _native, dp, evaluation, wdn_io, wdn_synth
from ._build import build_native
from .config_models import config_gatres_large, config_gatres_small, select_model
from .graph_models import GATConv, GATResMeanConv, GResBlockMeanConv, Linear, SimpleConv
from .fused_adam import FusedAdam
from .graph_plan import GraphPlan, PlanCache
from .snapshot_store import SnapshotStore
from .train_step import GATResTrainer

__all__ = ["FusedAdam", "GATResMeanConv", "GResBlockMeanConv", "GATConv", "SimpleConv", "Linear", "GraphPlan", "PlanCache",
           "GATResTrainer", "select_model", "config_gatres_small", "config_gatres_large", "build_native", "wdn_synth", "dp", "evaluation", "wdn_io", "SnapshotStore"]

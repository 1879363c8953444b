"""Model registry -- the plugin API of the path (reference buglab/models/modelregistry.py):
`load_model`, `construct_model_dict`, `gnn`, buggy-sample weight schedules.  Same names, kwargs,
defaults and error behaviour; all six names -- `gnn-mlp`, `ggnn`, `seq-great`, `seq-rat`,
`seq-transformer`, `seq-gru` -- are built on the HIP path."""
import logging
import re
from functools import partial
from pathlib import Path
from typing import Any, Callable, Dict, Optional, Tuple, Union

from buglab.models.gnn import GnnBugLabModel
from buglab.models.gnnlayerdefs import create_ggnn_mp_layers, create_mlp_mp_layers
from buglab.models.graphmodel import GraphNeuralNetworkModel, StrElementRepresentationModel
from buglab.runtime.neuralmodel import AbstractNeuralModel

LOGGER = logging.getLogger(__name__)


def const_schedule(epoch_idx: int, const_weight: float) -> float:
    return const_weight


WARMDOWN_WEIGHT_REGEX = re.compile("warmdown\\(([0-9]+),\\s?([0-9]*\\.[0-9]+)\\)")


def linear_warmdown(epoch_idx: int, num_warmdown_epochs: int, target_weight: float) -> float:
    return max(target_weight, epoch_idx * (target_weight - 1) / num_warmdown_epochs + 1)


def buggy_sample_weight_schedule(weight_spec: Union[str, int, float]) -> Callable[[int], float]:
    """Return a (serializable) function with the appropriate schedule (reference :29-41)."""
    if isinstance(weight_spec, (int, float)):
        return partial(const_schedule, const_weight=weight_spec)
    warmdown = WARMDOWN_WEIGHT_REGEX.match(weight_spec)
    if warmdown:
        return partial(linear_warmdown, num_warmdown_epochs=int(warmdown.group(1)), target_weight=float(warmdown.group(2)))
    raise Exception(f"Unrecognized buggy sample weighting `{weight_spec}`")


def _mp_layers(mp_layer, hidden_state_size, dropout_rate, edge_feature_size, extra, n_edges):
    if mp_layer is create_ggnn_mp_layers:
        extra = {}  # the ggnn recipe has no layer-count / activation knobs (reference gnnlayerdefs.py:42-68)
    return mp_layer(hidden_state_size, dropout_rate, n_edges, features_dimension=edge_feature_size, **extra)


def gnn(*, mp_layer, add_self_edge: bool, use_all_gnn_layer_outputs: bool = False, hidden_state_size: int = 128,
        dropout_rate: float = 0.2, node_representations: Optional[Dict[str, Any]] = None,
        selector_loss_type="classify-max-loss", stop_extending_minibatch_after_num_nodes: int = 30000,
        max_nodes_per_graph: int = 35000, buggy_samples_weight_spec: Union[str, int, float] = 1.0,
        edge_feature_size: int = 0, num_layers: Optional[int] = None, message_activation: Optional[str] = None,
        message_activation_placement: Optional[str] = None, add_backwards_edges: bool = True,
        message_aggregation_function: Optional[str] = None, **kwargs):
    """reference :44-94.  Extra knobs entering through the same kwargs dict:
    `num_layers` (multiple of 4, default the reference's 8), `message_activation`, `message_activation_placement`
    ("aggregated" -- default, ptgnn's order as recollected -- or "message"), `add_backwards_edges`,
    `message_aggregation_function` ("max" -- what the reference's recipe passes, gnnlayerdefs.py:11,21 -- or ptgnn's "sum" / "mean";
    gnn-mlp only)."""
    node_representations = dict(node_representations or {})
    node_representations.setdefault("token_splitting", "subtoken")
    node_representations.setdefault("max_num_subtokens", 6)
    node_representations.setdefault("subtoken_combination", "max")
    node_representations.setdefault("vocabulary_size", 15000)
    edge_representation_model = None
    if edge_feature_size > 0:  # reference :70-76: a token-level embedding of the edges' third elements
        if mp_layer is not create_mlp_mp_layers:
            raise NotImplementedError("edge features are implemented for the gnn-mlp layers (MlpMessagePassingLayer)")
        edge_representation_model = StrElementRepresentationModel(token_splitting="token", embedding_size=edge_feature_size)
    extra = {}
    if num_layers is not None:
        extra["num_layers"] = num_layers
    if message_activation is not None:
        extra["message_activation"] = message_activation
    if message_activation_placement is not None:
        extra["message_activation_placement"] = message_activation_placement
    if message_aggregation_function is not None and message_aggregation_function != "max":
        if mp_layer is not create_mlp_mp_layers:
            raise NotImplementedError("sum / mean aggregation is implemented for the gnn-mlp layers (MlpMessagePassingLayer)")
        extra["message_aggregation_function"] = message_aggregation_function
    return GnnBugLabModel(
        GraphNeuralNetworkModel(
            # reference :79-82: the node embedder gets NO dropout_rate from gnn() -- only what `node_representations` carries,
            # else the representation model's own default (0.0 here; ptgnn's constructor default, which is not in the tree).
            # The sequence models do pass theirs (seqmodel.py:425-431).
            node_representation_model=StrElementRepresentationModel(embedding_size=hidden_state_size, **node_representations),
            edge_representation_model=edge_representation_model,
            add_self_edges=add_self_edge,
            add_backwards_edges=add_backwards_edges,
            message_passing_layer_creator=partial(_mp_layers, mp_layer, hidden_state_size, dropout_rate, edge_feature_size, extra),
            stop_extending_minibatch_after_num_nodes=stop_extending_minibatch_after_num_nodes,
            max_nodes_per_graph=max_nodes_per_graph,
        ),
        use_all_gnn_layer_outputs=use_all_gnn_layer_outputs,
        generator_loss_type=selector_loss_type,
        buggy_samples_weight_schedule=buggy_sample_weight_schedule(buggy_samples_weight_spec),
    )


def seq_transformer(*, layer_type, hidden_state_size: int = 256, dropout_rate: float = 0.1, vocab_size: int = 15000,
                    selector_loss_type: str = "classify-max-loss", num_layers: int = 5, num_heads: int = 8, max_seq_size: int = 400,
                    intermediate_dimension_size: int = 1024, buggy_samples_weight_spec: Union[str, int, float] = 1.0,
                    rezero_mode: str = "off", normalisation_mode: str = "postnorm", **__):
    """reference :97-126 (same kwargs and defaults).  All four layer types run on the HIP path: `great` / `rat` (relational
    transformer), `transformer` (torch.nn.TransformerEncoderLayer's arithmetic, layers/transformer.py) and `gru` (torch.nn.GRU's,
    bidirectional over packed sequences: layers/gru.py, csrc/bl_gru_scan.hip)."""
    from buglab.models.seqmodel import SeqBugLabModel

    return SeqBugLabModel(hidden_state_size, max_subtoken_vocab_size=vocab_size, dropout_rate=dropout_rate, layer_type=layer_type,
                          generator_loss_type=selector_loss_type, intermediate_dimension_size=intermediate_dimension_size,
                          buggy_samples_weight_schedule=buggy_sample_weight_schedule(buggy_samples_weight_spec),
                          max_seq_size=max_seq_size, num_heads=num_heads, num_layers=num_layers, rezero_mode=rezero_mode,
                          normalisation_mode=normalisation_mode)


def construct_model_dict(gnn_constructor: Callable, seq_constructor: Callable) -> Dict[str, Callable]:
    """reference :129-137 -- same six names."""
    return {
        "gnn-mlp": lambda kwargs: gnn_constructor(mp_layer=create_mlp_mp_layers, add_self_edge=True, **kwargs),
        "ggnn": lambda kwargs: gnn_constructor(mp_layer=create_ggnn_mp_layers, add_self_edge=False, **kwargs),
        "seq-great": lambda kwargs: seq_constructor(layer_type="great", **kwargs),
        "seq-rat": lambda kwargs: seq_constructor(layer_type="rat", **kwargs),
        "seq-transformer": lambda kwargs: seq_constructor(layer_type="transformer", **kwargs),
        "seq-gru": lambda kwargs: seq_constructor(layer_type="gru", **kwargs),
    }


def load_model(model_spec: Dict[str, Any], model_path: Path, restore_path: Optional[str] = None,
               restore_if_model_exists: bool = False, type_model: bool = False) -> Tuple[AbstractNeuralModel, Any, bool]:
    """reference :140-166 -> (model, nn | None, initialize_metadata)."""
    assert model_path.name.endswith(".pkl.gz"), "MODEL_FILENAME must have a `.pkl.gz` suffix."
    initialize_metadata = True
    if restore_path is not None or (restore_if_model_exists and model_path.exists()):
        import torch

        LOGGER.info("Resuming training from %s." % model_path)
        initialize_metadata = False
        # each rank restores onto ITS device (the reference is single-GPU and hard-wires cuda:0, modelregistry.py:155)
        device = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")
        model, nn = AbstractNeuralModel.restore_model(Path(restore_path) if restore_path is not None else model_path, device)
    else:
        nn = None
        models = construct_model_dict(gnn, seq_transformer)
        if model_spec["modelName"] not in models:
            raise ValueError("Unknown model `%s`. Known models: %s", model_spec["modelName"], models.keys())
        spec = dict(model_spec)
        del spec["modelName"]
        model = models[model_spec["modelName"]](spec)
    return model, nn, initialize_metadata

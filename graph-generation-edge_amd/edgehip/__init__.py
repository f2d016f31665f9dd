"""edgehip: MI355X-native hot path of EDGE (degree-guided binomial graph diffusion).

Importing the package loads libedgehip.so; it fails loudly when the library is missing."""
from . import capi  # noqa: F401  (loads the shared library)
from .batch import GraphBatch, to_graph_batch  # noqa: F401
from .model import (BinomialDiffusionActive, TGNN_degree_guided, add_model_args, elbo_bpd, get_model,  # noqa: F401
                    get_model_id)
from .sampler import EmpiricalEmptyGraphGenerator  # noqa: F401
from . import shard  # noqa: F401
from .trainer import make_optimizer, train_step  # noqa: F401
from . import stats  # noqa: F401
from .stats import NetworkEvaluator, edge_overlap, graph_statistics  # noqa: F401

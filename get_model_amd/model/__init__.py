"""Host-side mirror of the reference's `get_model.model` package for the region models (hot path only)."""
from .model import (GETLoss, GETRegionFinetune, GETRegionFinetuneATAC, GETRegionPretrain,  # noqa: F401
                    RegressionMetrics)
from .modules import ExpressionHead, RegionEmbed  # noqa: F401
from .transformer import Attention, Block, DropPath, GETTransformer, Mlp  # noqa: F401

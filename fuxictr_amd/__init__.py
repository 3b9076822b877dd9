"""fuxictr_amd — MI355X-native hot path (FeatureEmbedding lookup + sparse-row update + FM /
CrossNet / MLP towers) behind the fuxictr.pytorch class API.  See DESIGN.md / INTEGRATION.md."""
from .features import FeatureMap  # noqa: F401

__all__ = ["FeatureMap"]

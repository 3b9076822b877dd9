"""Install the native layers into an importable reference checkout, so that the reference's OWN
`model_zoo` classes and `run_expid.py` run unmodified on the MI355X path.

    import fuxictr_amd.patch as patch
    patch.install()                      # BEFORE `import model_zoo` / `from fuxictr.pytorch...`
    from model_zoo import DeepFM, DCNv2  # the reference's classes, now built from native layers

The reference binds layers by name at import time (`from fuxictr.pytorch.layers import
FeatureEmbedding, MLP_Block, ...`, model_zoo/DeepFM/DeepFM_torch/src/DeepFM.py:21) and
`regularization_loss` tests `type(module) == FeatureEmbeddingDict` by identity
(fuxictr/pytorch/models/rank_model.py:107), so the names themselves are re-bound, in the package
and in every submodule that already imported them.
"""
import importlib
import sys

LAYER_NAMES = ["FeatureEmbedding", "FeatureEmbeddingDict", "LogisticRegression",
               "FactorizationMachine", "InnerProductInteraction", "MLP_Block", "CrossNetV2",
               "MaskedAveragePooling", "MaskedSumPooling", "DIN_Attention", "Dice", "CompressedInteractionNet"]


def install():
    from . import layers as nat_layers
    from . import rank_model as nat_rm
    from . import features as nat_feat
    ref_layers = importlib.import_module("fuxictr.pytorch.layers")
    ref_models = importlib.import_module("fuxictr.pytorch.models")
    ref_rm = importlib.import_module("fuxictr.pytorch.models.rank_model")
    ref_tu = importlib.import_module("fuxictr.pytorch.torch_utils")
    for name in LAYER_NAMES:
        native = getattr(nat_layers, name)
        setattr(ref_layers, name, native)
        for modname, mod in list(sys.modules.items()):
            if modname.startswith("fuxictr.pytorch.layers") and hasattr(mod, name):
                setattr(mod, name, native)
    ref_rm.BaseModel = nat_rm.BaseModel
    ref_rm.FeatureEmbeddingDict = nat_layers.FeatureEmbeddingDict
    ref_models.BaseModel = nat_rm.BaseModel
    ref_tu.get_device = nat_rm.get_device
    feats = importlib.import_module("fuxictr.features")
    feats.FeatureMap = nat_feat.FeatureMap
    return ref_layers

"""poet_amd -- MI355X-native (gfx950) PoET encoder-decoder hot path.

Hand-written HIP kernels behind a C ABI (include/poet_hip.h, poet_amd/csrc/libpoet_hip.so), loaded with
ctypes; this package is the Python host side mirroring the reference's operator/module interface
(`deformable_attention.MSDeformAttn`, DeformableTransformer, PoET) with identical state_dict keys.
There is NO CPU / eager-PyTorch fallback: without the built library or without a GPU tensor the ops raise.
"""
from ._lib import PoetHipError, load as load_library  # noqa: F401
from .modules import (BoundingBoxEmbeddingSine, DeformableTransformer, DeformableTransformerDecoder,  # noqa: F401
                      DeformableTransformerDecoderLayer, DeformableTransformerEncoder,
                      DeformableTransformerEncoderLayer, MLP, MSDeformAttn, NestedTensor, PoET,
                      PositionEmbeddingLearned, PositionEmbeddingSine)
from .engine import (BucketReducer, losses_for, GraphedTrainer, GraphedInference, ParamArena, PoseMatcher, SetCriterion, Trainer,  # noqa: F401
                     build_weight_dict, reduce_dict)
from .blocks import manual_seed  # noqa: F401

__version__ = "0.1.0"
from .data import DataPrefetcher, nested_tensor_from_tensor_list  # noqa: F401,E402

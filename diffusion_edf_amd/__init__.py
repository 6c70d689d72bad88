"""MI355X-native implementation of the Diffusion-EDF SE(3) score-head hot path.

    gnn_data            FeaturedPoints, GraphEdge and their helpers              (reference diffusion_edf/gnn_data.py)
    score_head          ScoreModelHead, EbmScoreModelHead                        (score_head.py, score_head_ebm.py)
    score_model_base    ScoreModelBase.sample / forward                          (score_model_base.py)
    agent               MultiscaleScoreModel, PointAttentiveScoreModel, get_models, DiffusionEdfAgent
                                                                                 (multiscale_score_model.py, point_attentive_score_model.py, agent.py)
    unet                UnetFeatureExtractor, ForwardOnlyFeatureExtractor, UnetLayer      (unet_feature_extractor.py, forward_only_feature_extractor.py, block.py)
    keypoint_extractor  KeypointExtractor, MultiscaleTensorField (context-free), StaticKeypointModel   (keypoint_extractor.py, multiscale_tensor_field.py)
    connectivity        fps, radius, radius_graph, RadiusGraph, RadiusConnect, FpsPool   (connectivity.py)
    configs, preprocess task front-end: agent.yaml / server.yaml / preprocess.yaml     (agent_server.py:48-86, train_utils.py:24-31)
    dist                pose sharding over ranks + the closing RCCL all-gather

Everything computes through ``csrc/libdedf.so`` (HIP, gfx950); importing a submodule that needs it fails loudly when the library
is missing — there is no CPU path."""
from .gnn_data import FeaturedPoints, GraphEdge  # noqa: F401

__version__ = "0.1.0"

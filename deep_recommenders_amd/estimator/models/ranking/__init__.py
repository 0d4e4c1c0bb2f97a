from deep_recommenders_amd.estimator.models.ranking.fnn import FNN
from deep_recommenders_amd.estimator.models.ranking.wide_and_deep import WDL
from deep_recommenders_amd.estimator.models.ranking.deepfm import DeepFM

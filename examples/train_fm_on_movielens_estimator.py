#!/usr/bin/env python
"""FM on MovieLens, estimator style -- BASELINE.json configs[0] (the reference's own CPU-runnable case), the runnable
equivalent of the reference's examples/train_fm_on_movielens_estimator.py on the MI355X hot path:

    build_columns()  the six feature columns of :10-34 (hash buckets for user_id / movie_id, vocabulary lists for the rest --
                     including the reference's quirk of building `movie_genres` with gender_vocab, :22-23, so every genre is
                     out of vocabulary), an indicator and a 16-d embedding column each
    model_fn()       FM(indicator_columns, embedding_columns)(features) -> logits; tf.losses.sigmoid_cross_entropy; AUC of
                     sigmoid(logits); AdamOptimizer(0.01)                                                       (:37-54)
    train_and_evaluate()  the Estimator loop reduced to what the example uses: train on `training_input_fn`, evaluate on
                     `testing_input_fn`, log every 100 steps, stop when the loss has not decreased for 1000 steps (:83-96)

Data: `--data movielens.tfrecords` reads the reference's TFRecord file through the native reader (datasets.MovielensRanking,
batch 256 as in BASELINE.json; the reference's class default is 1024).  Without it, MovieLens-shaped synthetic batches are
generated in memory (there is no network to fetch MovieLens-1M): the same feature dict, label = Bernoulli(0.575) tied weakly
to the user / movie ids so that the model has something to learn.

    python examples/train_fm_on_movielens_estimator.py --steps 300
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from deep_recommenders_amd import feature_column as fc          # noqa: E402
from deep_recommenders_amd import losses, optim                  # noqa: E402
from deep_recommenders_amd.datasets import MovielensRanking      # noqa: E402
from deep_recommenders_amd.estimator.models.feature_interaction import FM   # noqa: E402

TRAIN, EVAL, PREDICT = "train", "eval", "infer"                  # tf.estimator.ModeKeys


def build_columns():
    movielens = MovielensRanking()
    user_id = fc.categorical_column_with_hash_bucket("user_id", movielens.num_users)
    user_gender = fc.categorical_column_with_vocabulary_list("user_gender", movielens.gender_vocab)
    user_age = fc.categorical_column_with_vocabulary_list("user_age", movielens.age_vocab)
    user_occupation = fc.categorical_column_with_vocabulary_list("user_occupation", movielens.occupation_vocab)
    movie_id = fc.categorical_column_with_hash_bucket("movie_id", movielens.num_movies)
    movie_genres = fc.categorical_column_with_vocabulary_list("movie_genres", movielens.gender_vocab)   # sic (:22-23)
    base_columns = [user_id, user_gender, user_age, user_occupation, movie_id, movie_genres]
    indicator_columns = [fc.indicator_column(c) for c in base_columns]
    embedding_columns = [fc.embedding_column(c, dimension=16) for c in base_columns]
    return indicator_columns, embedding_columns


def auc(labels, probs):
    """Area under the ROC curve (rank statistic with tie correction; tf.metrics.auc approximates the same area with 200
    thresholds)."""
    y = np.asarray(labels).reshape(-1)
    p = np.asarray(probs).reshape(-1)
    n1, n0 = int((y > 0.5).sum()), int((y <= 0.5).sum())
    if n1 == 0 or n0 == 0:
        return float("nan")
    order = np.argsort(p, kind="mergesort")
    ranks = np.empty(len(p), dtype=np.float64)
    sp = p[order]
    i = 0
    while i < len(sp):                       # average ranks over ties
        j = i
        while j + 1 < len(sp) and sp[j + 1] == sp[i]:
            j += 1
        ranks[order[i:j + 1]] = 0.5 * (i + j) + 1.0
        i = j + 1
    return float((ranks[y > 0.5].sum() - n1 * (n1 + 1) / 2.0) / (n1 * n0))


class Estimator:
    """model_fn(features, labels, mode) evaluated eagerly: the model object and its optimizer are created on the first call and
    kept (tf.estimator re-builds the graph per mode and restores the variables from the checkpoint; here they simply persist)."""

    def __init__(self):
        torch.manual_seed(42)                # tf_random_seed=42 (:63)
        indicator_columns, embedding_columns = build_columns()
        self.model = FM(indicator_columns, embedding_columns)
        self.optimizer = None
        self.global_step = 0

    def model_fn(self, features, labels, mode):
        outputs = self.model(features)                                        # logits [B, 1]  (:39)
        predictions = {"predictions": outputs}
        if mode == PREDICT:
            return {"predictions": predictions}
        y = torch.as_tensor(labels, dtype=torch.float32).to(outputs.device)
        loss = losses.sigmoid_cross_entropy(y, outputs)                       # :46
        if mode == EVAL:
            return {"loss": loss, "labels": y, "probs": losses.sigmoid(outputs.detach())}      # :47-49
        if self.optimizer is None:
            self.optimizer = optim.Adam(list(self.model.parameters()), 0.01, epsilon=1e-8)   # tf.train.AdamOptimizer(0.01)  (:51)
        self.optimizer.zero_grad()
        loss.backward()
        self.optimizer.step()                                                 # :52
        self.global_step += 1
        return {"loss": loss}

    def train(self, input_fn, max_steps=None, log_every=100, patience=1000):
        best, best_step = float("inf"), 0
        for features, labels in input_fn:
            loss = float(self.model_fn(features, labels, TRAIN)["loss"])
            if self.global_step % log_every == 0:                             # log_step_count_steps=100 (:66)
                print("step %d  loss %.6f" % (self.global_step, loss), flush=True)
            if loss < best:
                best, best_step = loss, self.global_step
            elif self.global_step - best_step >= patience:                    # stop_if_no_decrease_hook(.., "loss", 1000) (:85)
                print("no decrease of the loss for %d steps: stopping at step %d" % (patience, self.global_step))
                break
            if max_steps is not None and self.global_step >= max_steps:
                break
        return best

    def evaluate(self, input_fn, steps=None):
        tot, n, ys, ps = 0.0, 0, [], []
        with torch.no_grad():
            for i, (features, labels) in enumerate(input_fn):
                if steps is not None and i >= steps:
                    break
                out = self.model_fn(features, labels, EVAL)
                b = out["labels"].shape[0]
                tot += float(out["loss"]) * b
                n += b
                ys.append(out["labels"].cpu().numpy())
                ps.append(out["probs"].cpu().numpy())
        if n == 0:
            return {"loss": float("nan"), "auc": float("nan"), "examples": 0}
        return {"loss": tot / n, "auc": auc(np.concatenate(ys), np.concatenate(ps)), "examples": n, "global_step": self.global_step}


def synthetic_input_fn(steps, batch_size, seed):
    """MovieLens-shaped batches in the layout of MovielensRanking.input_fn (datasets/movielens.py:170-186 of the reference)."""
    ml = MovielensRanking()
    world = np.random.default_rng(12345)          # the "true" effects: the same for the training and the evaluation stream
    user_bias = world.normal(0, 1.0, ml.num_users + 1)
    movie_bias = world.normal(0, 1.0, ml.num_movies + 1)
    age_effect = world.normal(0, 0.8, len(ml.age_vocab))
    occ_effect = world.normal(0, 0.8, 21)
    rng = np.random.default_rng(seed)             # the examples drawn from it
    for _ in range(steps):
        u = rng.integers(1, ml.num_users + 1, batch_size)
        m = rng.integers(1, ml.num_movies + 1, batch_size)
        gender = rng.integers(0, 2, batch_size)
        age = rng.integers(0, len(ml.age_vocab), batch_size)
        occ = rng.integers(0, 21, batch_size)
        features = {
            "user_id": [str(v) for v in u],
            "user_gender": [ml.gender_vocab[i] for i in gender],
            "user_age": np.asarray(ml.age_vocab)[age].astype(np.int64),
            "user_occupation": occ.astype(np.int64),
            "movie_id": [str(v) for v in m],
            "movie_genres": [[ml.genres_vocab[g] for g in rng.choice(18, rng.integers(1, 4), replace=False)]
                             for _ in range(batch_size)],
        }
        # P(rating > 3) ~ 0.575 on average; the small-vocabulary features carry most of the signal (learnable in a few hundred steps)
        z = 0.3 + 1.0 * (gender - 0.5) + age_effect[age] + occ_effect[occ] + 0.4 * user_bias[u] + 0.4 * movie_bias[m]
        labels = (rng.random(batch_size) < 1.0 / (1.0 + np.exp(-z))).astype(np.float32)[:, None]
        yield features, labels


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--data", default=None, help="movielens.tfrecords written by the reference's datasets/movielens.py")
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--steps", type=int, default=300, help="training steps (None in the reference: until early stopping)")
    ap.add_argument("--eval-steps", type=int, default=20)
    ap.add_argument("--seed", type=int, default=42)
    a = ap.parse_args(argv)
    estimator = Estimator()
    if a.data:
        movielens = MovielensRanking(batch_size=a.batch, filename=a.data)
        train_fn, test_fn = movielens.training_input_fn, movielens.testing_input_fn
    else:
        train_fn = synthetic_input_fn(a.steps, a.batch, a.seed)
        test_fn = synthetic_input_fn(a.eval_steps, a.batch, a.seed + 1)
    before = None
    if not a.data:
        before = estimator.evaluate(synthetic_input_fn(a.eval_steps, a.batch, a.seed + 1))
        print("before training:", before)
    estimator.train(train_fn, max_steps=a.steps)
    result = estimator.evaluate(test_fn, steps=a.eval_steps)
    print("evaluation:", result)
    return before, result


if __name__ == "__main__":
    main()
